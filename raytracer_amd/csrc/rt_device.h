// Shared declarations between the host side of librtmi.so (rt_api.hip) and its three
// kernel translation units (rt_trace_exact.hip, rt_trace_strict.hip, rt_trace_fast.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef RT_MAT_THREADS
#define RT_MAT_THREADS 1024
#endif

namespace rtmi {

// Device-resident world in structure-of-arrays form, in DEVICE ORDER (rt_set_world splits the spheres in space into
// tiles of 32 and sorts each tile along a space-filling curve; orig[] maps back to the caller's indices).  One float4 per sphere and array:
//   geom[i] = { cx, cy, cz, r*r }            -- all hit_sphere (Main.c:625-653) needs to reject
//   matA[i] = { albedo.r, albedo.g, albedo.b, fuzz }; glass, whose albedo nothing reads: { r0 entering, r0 leaving, -, - }
//             with r0 = ((1 - ri) / (1 + ri))^2 of reflectance() (Main.c:553-554), ri = 1 / index entering, index leaving
//   matB[i] = { refraction_index, bits(mat_type), 1.0f/r, 1.0f/refraction_index }
// geom is padded to a multiple of kGeomPad with never-hit entries (r*r = -inf).
constexpr int kGeomPad = 8;
// Both builds: the scan over the sphere list is a conservative filter, two spheres per packed-FP32
// instruction.  pairs[k] holds spheres 2k and 2k+1 as { cx0, cx1, cy0, cy1, cz0, cz1, k0, k1 } with
// k = |C|^2 - r^2 (read through the scalar cache: the index is wave-uniform).  With a unit
// direction d, the discriminant h^2 - c >= 0 is (d.C - d.O)^2 + 2 O.C - |O|^2 >= |C|^2 - r^2:
// seven packed FMAs per pair.  The expanded form cancels badly, so it is compared against a
// threshold lowered by a bound of its rounding error (it may pass a sphere the ray misses, never
// the reverse); lanes queue the indices that pass in LDS (kQueueSlots per lane) and the exact
// test, in the reference's (C - O) form, runs on the queued spheres after the scan.  Spheres far
// larger than the rest (the r = 1000 ground) would inflate that bound for every sphere, so they
// are kept out of the filter (k = +inf) and tested exactly for every ray: big[] lists them.
constexpr int kMaxBig = 16;
// Workgroups of 4 waves, one per SIMD.  The fast build fits 72 VGPRs, i.e. 7 waves per SIMD (8
// would need 64 VGPRs and spill): 7 workgroups per CU, whose LDS (geometry image + queues) fits
// 160 KB with 14 queue slots per lane.  The strict build needs more registers (6 waves per SIMD).
constexpr int kBlockThreads = 256;
constexpr int kQueueSlotsFast = 14;        // candidate-queue slots per lane
constexpr int kQueueSlotsStrict = 16;
constexpr int queue_lds_bytes(int block_threads, int slots) { return (block_threads / 64) * slots * 64 * 4; }   // waves x slots x 64 lanes x u32
// The sphere scan comes in four variants.
//   kScanMatrix       scenes whose matrix image fits the LDS: the conservative filter runs on the matrix
//                     cores.  The discriminant of (ray, sphere) with a unit direction d and the origin slid
//                     along the ray to the point of the line nearest to the frame's origin,
//                     P = O - (d.O) d (so d.P = 0), is
//                         disc = sum_ij (d_i d_j)(c_i c_j) + 2 P.C - |P|^2 - (|C|^2 - r^2),
//                     one inner product of ten ray features with ten sphere features.  In a frame scaled
//                     by a power of two (spheres within 120 of the origin) each f32 feature is split into
//                     two f16 pieces x = x1 + x2 and the products x1 y1 + x1 y2 + x2 y1 are summed by
//                     v_mfma_f32_32x32x16_f16 in f32 (K = 9 x 3 + 2 + 2 = 31 of 32): 32 spheres x 32 rays
//                     per 2 instructions; the vector ALU only compares the accumulators with zero and writes what
//                     passes into the wave's pooled candidate list (exact tests 64 candidates at a time).
//                     aimg[tile][s][lane] holds the sphere side as MFMA A fragments (lane l: sphere
//                     32 tile + (l & 31), K entries 16 s + 8 (l >> 5) ..+7).
//   kScanVectorLds    packed-FP32 vector scan, per-sphere records in LDS (selectable for comparison)
//   kScanVectorGlobal packed-FP32 vector scan, per-sphere records from global memory (larger scenes)
//   kScanMatrixGlobal the matrix scan for scenes whose image does not fit the LDS: the fragments come from
//                     global memory (L1 / L2: every wave of the chip reads the same tiles), one tile ahead of
//                     their use, and so do the per-sphere records of the exact test; the LDS holds the wave regions
enum { kScanVectorLds = 0, kScanVectorGlobal = 1, kScanMatrix = 2, kScanMatrixGlobal = 3, kScanVariants = 4 };
constexpr bool is_matrix(int variant) { return variant == kScanMatrix || variant == kScanMatrixGlobal; }
constexpr int kTileSpheres = 32;
constexpr int kBoxGroupTiles = 8;                  // tiles per group of the box hierarchy of large scenes (64 tiles, one mask, are eight groups)
constexpr int kMaxSpheresMatrixGlobal = 65504;   // a candidate entry and a result key hold sphere indices in 16 bits
constexpr int kMinSpheresMatrix = 32;      // smaller scenes: the vector scan (measured break-even 20-40 spheres)
constexpr int kMatBlockThreads = RT_MAT_THREADS;   // one workgroup per CU: 4 waves per SIMD at 128 VGPRs
constexpr int kLdsBytes = 160 * 1024;
// Per-wave LDS region of the matrix scan: one result key per ray and the wave's pooled candidate list.
// Candidates are (ray, sphere) pairs appended by whichever lane saw them pass and tested 64 at a time, one per
// lane, as soon as 64 are waiting: a half tile can add 32 x 32 = 1024 entries to the at most 63 left over.
constexpr int kMatKeyBytes = 64 * 8;
constexpr int kMatListCap = 63 + 1024 + 1;
constexpr int kMatWaveBytes = kMatKeyBytes + kMatListCap * 4;
static_assert(kMatWaveBytes % 16 == 0, "the wave regions stay 16-byte aligned");
// LDS of the matrix scan with its images staged: [sphere image 2 KB per tile][geometry 16 B per sphere]
// [caller's index 2 B per sphere][tile boxes 32 B per tile][wave regions]
inline size_t matrix_lds_bytes(int n_pad, int ntiles)
{
    return (size_t)ntiles * 128 * 16 + (size_t)n_pad * 16 + (size_t)n_pad * 2 + (size_t)((ntiles + 1) / 2) * 48 + (size_t)(kMatBlockThreads / 64) * kMatWaveBytes;
}
constexpr int kMaxTilesMatrix = 32;        // what fits beside the wave regions of 16 waves (the host checks matrix_lds_bytes)
constexpr int kMaxSpheresMatrix = kMaxTilesMatrix * kTileSpheres;
constexpr int block_threads_of(int variant) { return is_matrix(variant) ? kMatBlockThreads : kBlockThreads; }
constexpr int kMaxSpheresLds = 560;        // up to here the per-sphere records are staged in LDS: 7 workgroups of (9 KB + 14 KB of queues) still fit a CU
constexpr int kMaxSpheres = 1 << 20;       // beyond kMaxSpheresLds they are read from global memory

struct Camera {                            // rt_camera, flattened
    float pos[3], p00[3], du[3], dv[3], ddu[3], ddv[3];
    float defocus_angle;
};

struct TraceParams {
    Camera cam;
    const float4 *geom;                    // [n_pad]
    const float4 *matA;                    // [n]
    const float4 *matB;                    // [n]
    const int32_t *orig;                   // [n_pad] device order -> index in the caller's list (hit indices, ties)
    const float *pairs;                    // [n_pad / 2][8]
    const uint4 *aimg;                     // [ntiles][2][64] f16 fragments of the matrix scan, or nullptr
    const float4 *tile_bounds;             // [ntiles][5] bounding spheres {centre, radius} of a tile's filtered spheres and of its two
                                           // 16-sphere halves (radius < 0: empty); read by the diagnostic build only
    const float4 *tile_boxes;              // [pairs of tiles][3] axis-aligned boxes of the tiles' filtered spheres, centres and half extents of two tiles interleaved (the matrix scan
                                           // skips the tiles whose box no live ray of the wave meets), or nullptr
    const float4 *group_boxes;             // [pairs of groups of kBoxGroupTiles tiles][3] the same for groups of consecutive tiles (scenes of more than 64 tiles ask
                                           // them first), or nullptr: every tile is asked
    const float4 *mask_group_boxes;        // [pairs of groups of 64 tiles][3] and for the groups of 64 tiles that share a mask, or nullptr
    float box_margin;                      // by how much a ray widens every box, per unit of |O| + max |C|
    float mat_scale;                       // power-of-two scale of the matrix scan's frame
    float mat_c_near;                      // max |C| over the filtered spheres, in that frame
    int ntiles;
    int n, n_pad;
    int nbig;                              // spheres tested exactly for every ray
    int big[kMaxBig];
    float c_near;                          // max |C| over the filtered spheres
    int coop_max;                          // a wave with at most this many live rays tests each ray with all 64 lanes
    int core_div_ok;                       // every sphere lies within 2^24 of the frame's origin (|C| + r): the exact test may divide by the
                                           // division's core (rt_trace_impl.h)
    uint32_t width, height;                // full image
    uint32_t npix;                         // owned pixels = owned_rows * width
    int tile_rows, part_index, part_count; // owned row k -> image row, see part_row_at()
    const uint32_t *rows;                  // or, if not null, rows[k] (explicit partition)
    int spp, chunk_spp, nchunks, max_depth;
    int pool_units;                        // work units a wave takes from the global counter at a time
    int mode;                              // RT_MODE_*
    uint32_t seed, sample_base;
    float4 *partial;                       // [nchunks][npix] per-chunk colour sums
    int32_t *hit_index;                    // [npix] compact, or nullptr
    unsigned long long *counters;          // [0] next work unit, [1] rays cast, [19] sphere scans run by the waves, [22] tiles they multiplied
                                           // ([2..18], [20], [21]: diagnostic builds)
    unsigned long long total_units;        // npix * nchunks
};

struct ResolveParams {
    const float4 *partial;
    int nchunks;
    uint32_t npix;
    float4 *running;                       // progressive frames: per-pixel sum carried between calls, or nullptr
    float inv_spp;                         // (f32)1.0f/(f32)spp, Main.c:977 (spp = all samples summed so far)
    uint8_t *rgba;                         // compact [npix][4]
    float *accum;                          // compact [npix][3] or nullptr
};

// Image row of the k-th owned row: row tiles of tile_rows rows are dealt round-robin.
__host__ __device__ inline uint32_t part_row_at(uint32_t k, int tile_rows, int part_index, int part_count)
{
    if (part_count <= 1) return k;
    uint32_t tr = (uint32_t)tile_rows;
    uint32_t t = k / tr, r = k % tr;
    return (t * (uint32_t)part_count + (uint32_t)part_index) * tr + r;
}

// The three builds of the path-tracing kernel (rt_opts.build): each is a translation unit of its own.
enum { kBuildExact = 0, kBuildStrict = 1, kBuildFast = 2, kBuilds = 3 };
constexpr bool reference_arithmetic(int build) { return build != kBuildFast; }

// The dynamic-LDS attribute and the occupancy answer depend only on (device, kernel, LDS bytes):
// each device context looks them up once per change of scene size, not once per frame.
struct LaunchCache {
    size_t attr_lds[kBuilds][2][kScanVariants];     // [build][mode][variant]: LDS size the attribute was last set to
    size_t occ_lds[kBuilds][kScanVariants];         // [build][variant]
    int occ_blocks[kBuilds][kScanVariants];
    LaunchCache() { reset(); }
    void reset()
    {
        for (int b = 0; b < kBuilds; b++)
            for (int g = 0; g < kScanVariants; g++) {
                occ_lds[b][g] = (size_t)-1; occ_blocks[b][g] = 1;
                for (int m = 0; m < 2; m++) attr_lds[b][m][g] = (size_t)-1;
            }
    }
};

// The kernel instantiation of a build for (mode, variant): one function per translation unit.
typedef void (*trace_fn)(const TraceParams);
trace_fn trace_kernel_exact(int mode, int variant);
trace_fn trace_kernel_strict(int mode, int variant);
trace_fn trace_kernel_fast(int mode, int variant);
hipError_t launch_resolve(const ResolveParams &p, hipStream_t s);

// Dynamic LDS of a trace kernel: [matrix image][geometry image][caller's indices][wave regions] (matrix scan),
// [geometry image][candidate queues] (vector scan).
inline size_t trace_lds_bytes(int variant, bool strict, int n_pad, int ntiles)     // strict: reference_arithmetic(build)
{
    if (variant == kScanMatrixGlobal) return (size_t)(kMatBlockThreads / 64) * kMatWaveBytes;
    if (variant == kScanMatrix) return matrix_lds_bytes(n_pad, ntiles);
    return (variant == kScanVectorGlobal ? 0 : (size_t)n_pad * sizeof(float4)) +
           (size_t)queue_lds_bytes(kBlockThreads, strict ? kQueueSlotsStrict : kQueueSlotsFast);
}

}  // namespace rtmi
