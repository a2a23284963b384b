// Host side of librtmi.so: the extern "C" entry points declared in include/rt.h.
// Owns one context per device given to rt_init / rt_init_devices (stream, events, the
// structure-of-arrays world, scratch: per-chunk partial sums, work/ray counters, staging for
// host-pointer callers).  With several contexts rt_render deals the image rows to them and runs
// one persistent host thread per context.
#include "../../include/rt.h"
#include "rt_device.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#pragma clang fp contract(off)

using namespace rtmi;

// Scratch of rt_render_device_noalloc: the work/ray counters (first 256 bytes) and the per-chunk
// partial sums of one call.  Every device the library runs on holds its own copy until the process
// ends; host code reaches it only through the per-device address looked up in init_ctx().
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];

namespace {

constexpr int kCounters = 24;                    // 64-bit words: [0] next work unit, [1] rays cast, [19] wave scans, the rest diagnostics
constexpr size_t kScratchCountersBytes = 256;
constexpr size_t kScratchPartialBytes = (size_t)RT_NOALLOC_SCRATCH_BYTES - kScratchCountersBytes;
constexpr uint32_t kProgBroken = 0xffffffffu;    // prog_frames of a sequence that must restart at frame 0

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 4;          // headroom: a slightly larger partition must not reallocate
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) { cap = want; return e; }
        (void)hipGetLastError();
        e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

constexpr int kMaxContexts = 16;

struct Ctx {
    bool inited = false;
    int device = -1;
    int num_cu = 0;
    LaunchCache launch;
    double speed = 1.0;          // rows per kernel-millisecond of the last multi-context frame (weight of the next deal)
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    unsigned char *scratch_ptr = nullptr;   // this device's copy of `scratch`
    // world
    int n = 0, n_pad = 0;
    int nbig = 0;
    int big[kMaxBig] = {0};
    float c_near = 0.0f;
    int core_div_ok = 0;         // every sphere within 2^24 of the origin (TraceParams::core_div_ok)
    DevBuf geom, matA, matB, pairs, aimg, orig, tbounds, tboxes;
    int ntiles = 0;              // 32-sphere tiles of the matrix scan's image (0: the scene is too large for it)
    float mat_scale = 1.0f;      // power-of-two scale of the matrix scan's frame
    // scratch
    DevBuf partial, counters, rgba, accum, hit, rows;
    // progressive accumulation (rt_render_progressive): running per-pixel sums and what they belong to
    DevBuf running;
    uint32_t prog_w = 0, prog_h = 0, prog_npix = 0, prog_frames = kProgBroken;
    int prog_spp = 0, prog_chunk = 0;
    std::vector<uint32_t> prog_rows;        // image row of every owned row of the sequence, in buffer order
    unsigned long long last_counters[kCounters] = {0};
};

Ctx g_ctx[kMaxContexts];
int g_nctx = 0;
bool g_vector_scan = false;      // RTMI_SCAN=vector in the environment at rt_init: packed-FP32 scan for every scene size
bool g_caller_order = false;     // RTMI_ORDER=caller in the environment at rt_init: spheres stay in the caller's order on the device (comparison runs)
bool g_curve_order = false;      // RTMI_ORDER=curve: along the Z-order curve only, tiles not split in space (comparison runs)

thread_local char g_err[512] = "";      // per thread: the worker threads of a multi-device frame report through it

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            (void)hipGetLastError();                                                      \
            return fail(e_ == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_DEVICE,         \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                   \
        }                                                                                 \
    } while (0)

// The entry points select devices on the caller's thread; a host that also drives HIP or PyTorch
// itself must find its own current device unchanged afterwards.
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; } }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// Nothing may throw across the C boundary: the library's own containers and threads can
// (std::bad_alloc, std::system_error).
template <class F>
int guarded(const char *what, F &&body)
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(RT_ERR_NOMEM, "%s: out of host memory", what);
    } catch (const std::exception &e) {
        return fail(RT_ERR_DEVICE, "%s: %s", what, e.what());
    } catch (...) {
        return fail(RT_ERR_DEVICE, "%s: unknown failure", what);
    }
}

// One persistent host thread per context beyond the first (the caller's thread drives context 0):
// a frame hands each a job and waits; no thread is created per frame.
class Worker {
public:
    Worker() : th_([this] { loop(); }) {}
    ~Worker()
    {
        { std::lock_guard<std::mutex> l(mu_); quit_ = true; }
        cv_.notify_all();
        th_.join();
    }
    void start(std::function<void()> job)
    {
        { std::lock_guard<std::mutex> l(mu_); job_ = std::move(job); busy_ = true; }
        cv_.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> l(mu_);
        cv_.wait(l, [this] { return !busy_; });
    }
private:
    void loop()
    {
        std::unique_lock<std::mutex> l(mu_);
        for (;;) {
            cv_.wait(l, [this] { return quit_ || busy_; });
            if (quit_) return;
            std::function<void()> job = std::move(job_);
            l.unlock();
            job();                         // jobs catch their own exceptions
            l.lock();
            busy_ = false;
            cv_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::function<void()> job_;
    bool busy_ = false, quit_ = false;
    std::thread th_;                       // last: the thread starts when every other member exists
};

std::vector<std::unique_ptr<Worker>> g_workers;     // g_workers[i - 1] serves context i

inline void put3(float *d, rt_vec3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

struct V3 { float x, y, z; };
inline V3 V(rt_vec3 a) { return V3{a.x, a.y, a.z}; }
inline V3 vsub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 vadd(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 vscale(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 vcross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 vunit(V3 v)
{
    float len = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    if (len > 0.0f) { float inv = 1.0f / len; return V3{v.x * inv, v.y * inv, v.z * inv}; }
    return V3{0.0f, 0.0f, 0.0f};
}
inline rt_vec3 R(V3 a) { rt_vec3 r; r.x = a.x; r.y = a.y; r.z = a.z; return r; }

// radians_from_turns_f32 / radians_from_degrees_f32, include/util.h:134-139
inline float turns_to_rad(float v) { return v * (2 * 3.1415926535897f); }
inline float deg_to_rad(float v) { return turns_to_rad(v / 360.f); }

// Samples per work unit.  RT_CHUNK_AUTO is a function of the FULL image size and spp only, never of
// the rows a device or a rank owns or of the device: a pixel's sum is the in-order sum of its
// chunks' partial sums, so every part of a split frame must use the same unit size for the
// assembled image to equal the single-device image.
// Short units keep the end of a launch, when units run out and lanes fall idle, short; each halving
// of the unit below 8 samples costs about 0.7 % in unit hand-outs, so only images that give a lane of
// a full MI355X (256 CUs x 2048 lanes) fewer than 128 eight-sample units get shorter ones.  Every unit
// writes one 16-byte partial sum that the resolve kernel reads back, so the unit grows with the work:
// the per-chunk sums of the full image stay below 4 GB (3840x2160 at 1000 spp: 34 samples per unit,
// 30 chunks, 3.98 GB; with 8 samples per unit it was 16.6 GB of HBM held and written for nothing).
// Measured on that frame, 8 / 16 / 32 / 63 samples per unit: the whole frame 995 / 991 / 987 / 986 ms, and
// one eighth of it (what a rank of eight renders) at 0.976 / 0.978 / 0.973 / 0.965 of an eighth of the whole:
// beyond about 32 samples the end of a part's launch starts to show, hence 4 GB and not less.
int resolve_chunk(int chunk_opt, uint32_t width, uint32_t height, int spp, bool progressive)
{
    if (chunk_opt != RT_CHUNK_AUTO) return (chunk_opt > 0 && chunk_opt < spp) ? chunk_opt : spp;
    const uint64_t full = (uint64_t)width * (uint64_t)height;
    const uint64_t max_chunks = std::max<uint64_t>(1, 4000000000ULL / (std::max<uint64_t>(full, 1) * sizeof(float4)));
    const uint64_t least = ((uint64_t)spp + max_chunks - 1) / max_chunks;
    const uint64_t lanes = 256ULL * 2048ULL;
    const uint64_t small = std::max<uint64_t>(1, full * (uint64_t)spp / (128 * lanes));
    int chunk = (int)std::min<uint64_t>((uint64_t)spp, std::max<uint64_t>(std::min<uint64_t>(8, small), least));
    if (progressive) while (spp % chunk != 0) chunk++;          // frames add whole chunks
    return chunk;
}

int scan_variant(int n, int n_pad, int ntiles)
{
    if (ntiles > 0 && !g_vector_scan)
        return (ntiles <= kMaxTilesMatrix && matrix_lds_bytes(n_pad, ntiles) <= (size_t)kLdsBytes) ? kScanMatrix : kScanMatrixGlobal;
    return n > kMaxSpheresLds ? kScanVectorGlobal : kScanVectorLds;
}
int scan_variant(const Ctx &g) { return scan_variant(g.n, g.n_pad, g.ntiles); }

trace_fn kernel_of(int build, int mode, int variant)
{
    return build == kBuildFast ? trace_kernel_fast(mode, variant)
         : build == kBuildStrict ? trace_kernel_strict(mode, variant) : trace_kernel_exact(mode, variant);
}

// Sets the dynamic-LDS attribute of the kernel instantiation (build, mode, variant) if it differs from
// the cached one; the first call per instantiation also makes the runtime load its code object.
hipError_t prepare_trace(LaunchCache &c, int build, int mode, int variant, size_t lds_bytes)
{
    mode = mode == 1 ? 1 : 0;
    if (c.attr_lds[build][mode][variant] == lds_bytes) return hipSuccess;
    hipError_t e = hipFuncSetAttribute((const void *)kernel_of(build, mode, variant),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess) c.attr_lds[build][mode][variant] = lds_bytes;
    return e;
}

hipError_t launch_trace(LaunchCache &c, int build, const TraceParams &p, int variant, int blocks, size_t lds_bytes, hipStream_t s)
{
    const int mode = p.mode == 1 ? 1 : 0;
    hipError_t e = prepare_trace(c, build, mode, variant, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel_of(build, mode, variant), dim3(blocks), dim3(block_threads_of(variant)), lds_bytes, s, p);
    return hipGetLastError();
}

int trace_blocks_per_cu(LaunchCache &c, int build, int variant, size_t lds_bytes)
{
    if (c.occ_lds[build][variant] == lds_bytes) return c.occ_blocks[build][variant];
    int n = 0;
    if (prepare_trace(c, build, 0, variant, lds_bytes) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel_of(build, 0, variant), block_threads_of(variant), lds_bytes) != hipSuccess)
        return 1;
    c.occ_lds[build][variant] = lds_bytes;
    c.occ_blocks[build][variant] = n > 0 ? n : 1;
    return c.occ_blocks[build][variant];
}

// f16 pieces of the matrix scan's sphere image: x = p[0] + p[1] + e, |e| <= 2^-24 |x| (round to nearest even)
uint16_t f16_bits(float x)
{
    const _Float16 h = (_Float16)x;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
void split2(float x, uint16_t p[2])
{
    const _Float16 h = (_Float16)x;
    p[0] = f16_bits(x);
    p[1] = f16_bits(x - (float)h);
}

struct Call {
    bool device_ptrs = false;     // outputs are device pointers (compact over the owned rows)
    bool no_alloc = false;        // partial sums and counters live in the static scratch array; nothing is allocated
    int64_t frame_index = -1;     // >= 0: progressive frame
    int chunk = 0;                // resolved samples per work unit (resolve_chunk)
};

int render_common(Ctx &g, const rt_camera *cam, uint8_t *rgba, uint32_t width, uint32_t height, int spp,
                  int max_depth, const rt_opts *opts, const Call &call)
{
    using clk = std::chrono::steady_clock;
    const auto t_start = clk::now();
    rt_opts dflt;
    memset(&dflt, 0, sizeof dflt);
    const rt_opts &o = opts ? *opts : dflt;
    const bool device_ptrs = call.device_ptrs;
    const int64_t frame_index = call.frame_index;

    if (!g.inited) return fail(RT_ERR_STATE, "rt_render: rt_init has not been called");
    if (g.n_pad == 0) return fail(RT_ERR_STATE, "rt_render: rt_set_world has not been called");
    if (!cam || !rgba) return fail(RT_ERR_ARG, "rt_render: null camera or framebuffer");
    if (width == 0 || height == 0 || (uint64_t)width * height > 0x7fffffffULL)
        return fail(RT_ERR_ARG, "rt_render: bad image size %ux%u", width, height);
    if (spp < 1) return fail(RT_ERR_ARG, "rt_render: spp must be >= 1 (got %d)", spp);
    if (max_depth < 0) return fail(RT_ERR_ARG, "rt_render: max_depth must be >= 0 (got %d)", max_depth);
    if (o.mode != RT_MODE_MC && o.mode != RT_MODE_CONST_U) return fail(RT_ERR_ARG, "rt_render: unknown mode %d", o.mode);
    if (o.build < 0 || o.build >= kBuilds) return fail(RT_ERR_ARG, "rt_render: unknown build %d", o.build);
    const int build = o.build;
    if (o.tile_rows < 0) return fail(RT_ERR_ARG, "rt_render: negative tile_rows");
    const int chunk = call.chunk;
    if (chunk < 1 || chunk > spp) return fail(RT_ERR_ARG, "rt_render: bad chunk_spp %d for %d spp", chunk, spp);
    const int part_count = o.part_count > 1 ? o.part_count : 1;
    const int part_index = part_count > 1 ? o.part_index : 0;
    if (part_index < 0 || part_index >= part_count) return fail(RT_ERR_ARG, "rt_render: part_index %d outside [0,%d)", o.part_index, part_count);
    const int tile_rows = o.tile_rows > 0 ? o.tile_rows : 16;

    HIP_TRY(hipSetDevice(g.device));

    const bool explicit_rows = o.rows != nullptr;
    if (explicit_rows) {
        if (call.no_alloc) return fail(RT_ERR_ARG, "rt_render_device_noalloc: opts->rows is not supported; use part_index / part_count");
        if (o.row_count > height) return fail(RT_ERR_ARG, "rt_render: row_count %u exceeds the image height %u", o.row_count, height);
        for (uint32_t k = 0; k < o.row_count; k++)
            if (o.rows[k] >= height || (k > 0 && o.rows[k] <= o.rows[k - 1]))
                return fail(RT_ERR_ARG, "rt_render: rows[] must be strictly ascending and below the image height (entry %u)", k);
    }
    const uint32_t rows = explicit_rows ? o.row_count : rt_part_rows(height, tile_rows, part_index, part_count);
    const uint32_t npix = rows * width;
    const bool progressive = frame_index >= 0;
    const int nchunks = (spp + chunk - 1) / chunk;
    const uint64_t units = (uint64_t)npix * (uint64_t)nchunks;
    if (units >= 0xffffffffULL) return fail(RT_ERR_ARG, "rt_render: %llu work units; raise chunk_spp", (unsigned long long)units);
    if (call.no_alloc && units * sizeof(float4) > kScratchPartialBytes)
        return fail(RT_ERR_ARG, "rt_render_device_noalloc: %u owned pixels x %d chunks need %llu bytes of per-chunk sums; "
                    "the scratch array holds %llu (raise chunk_spp, or 0 for one chunk, or render fewer rows per call)",
                    npix, nchunks, (unsigned long long)(units * sizeof(float4)), (unsigned long long)kScratchPartialBytes);

    uint32_t sample_base = o.sample_base;
    uint32_t frames_after = 0;
    if (progressive) {
        if (max_depth < 1) return fail(RT_ERR_ARG, "rt_render_progressive: max_depth must be >= 1");
        if (spp % chunk != 0) return fail(RT_ERR_ARG, "rt_render_progressive: chunk_spp (%d) must divide spp (%d)", chunk, spp);
        if ((uint64_t)spp * (uint64_t)(frame_index + 1) > 0x7fffffffULL) return fail(RT_ERR_ARG, "rt_render_progressive: too many samples");
        std::vector<uint32_t> owned(rows);
        for (uint32_t k = 0; k < rows; k++) owned[k] = explicit_rows ? o.rows[k] : rt_part_row_at(k, tile_rows, part_index, part_count);
        if (frame_index == 0) {
            g.prog_frames = kProgBroken;
            HIP_TRY(g.running.reserve((size_t)npix * sizeof(float4) + 16));
            HIP_TRY(hipMemsetAsync(g.running.p, 0, (size_t)npix * sizeof(float4), g.stream));
            g.prog_w = width; g.prog_h = height; g.prog_npix = npix; g.prog_spp = spp; g.prog_chunk = chunk;
            g.prog_rows = owned;
            g.prog_frames = 0;
        }
        if (g.prog_frames == kProgBroken)
            return fail(RT_ERR_STATE, "rt_render_progressive: frame %lld does not continue a sequence (none started, or an earlier frame failed); "
                        "start again with frame_index 0", (long long)frame_index);
        if (g.prog_w != width || g.prog_h != height || g.prog_npix != npix || g.prog_spp != spp || g.prog_chunk != chunk ||
            (int64_t)g.prog_frames != frame_index || g.prog_rows != owned)
            return fail(RT_ERR_STATE, "rt_render_progressive: frame %lld does not continue the sequence "
                        "(%u frames of %ux%u at %d spp in units of %d accumulated, %u rows owned); start again with frame_index 0",
                        (long long)frame_index, g.prog_frames, g.prog_w, g.prog_h, g.prog_spp, g.prog_chunk, (unsigned)g.prog_rows.size());
        sample_base += (uint32_t)frame_index * (uint32_t)spp;
        frames_after = g.prog_frames + 1;
        g.prog_frames = kProgBroken;        // until this frame is complete: a failure below ends the sequence
    }

    rt_stats st;
    memset(&st, 0, sizeof st);

    uint8_t *d_rgba = rgba;
    float *d_accum = o.accum_out;
    int32_t *d_hit = o.hit_index_out;
    if (!device_ptrs) {
        HIP_TRY(g.rgba.reserve((size_t)npix * 4 + 4));
        d_rgba = (uint8_t *)g.rgba.p;
        if (o.accum_out) { HIP_TRY(g.accum.reserve((size_t)npix * 12 + 4)); d_accum = (float *)g.accum.p; }
        if (o.hit_index_out) { HIP_TRY(g.hit.reserve((size_t)npix * 4 + 4)); d_hit = (int32_t *)g.hit.p; }
    }
    unsigned long long *d_counters = call.no_alloc ? (unsigned long long *)g.scratch_ptr : (unsigned long long *)g.counters.p;

    if (npix > 0) {
        if (max_depth == 0) {
            // ray_color's loop does not run: every sample is (0,0,0) (Main.c:730,761).
            HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)d_rgba, (int)0xFF000000u, npix, g.stream));   // r=g=b=0, a=255
            if (d_accum) HIP_TRY(hipMemsetAsync(d_accum, 0, (size_t)npix * 12, g.stream));
            if (d_hit) HIP_TRY(hipMemsetAsync(d_hit, 0xff, (size_t)npix * 4, g.stream));
        } else {
            float4 *d_partial;
            if (call.no_alloc) {
                d_partial = (float4 *)(g.scratch_ptr + kScratchCountersBytes);
            } else {
                HIP_TRY(g.partial.reserve((size_t)units * sizeof(float4)));
                d_partial = (float4 *)g.partial.p;
            }
            // the counters start at zero; the per-chunk sums need no clearing (every unit writes its sum)
            HIP_TRY(hipMemsetAsync(d_counters, 0, kScratchCountersBytes, g.stream));

            TraceParams tp;
            memset(&tp, 0, sizeof tp);
            put3(tp.cam.pos, cam->pos); put3(tp.cam.p00, cam->pixel00_loc);
            put3(tp.cam.du, cam->pixel_delta_u); put3(tp.cam.dv, cam->pixel_delta_v);
            put3(tp.cam.ddu, cam->defocus_disk_u); put3(tp.cam.ddv, cam->defocus_disk_v);
            tp.cam.defocus_angle = cam->defocus_angle;
            tp.geom = (const float4 *)g.geom.p; tp.matA = (const float4 *)g.matA.p; tp.matB = (const float4 *)g.matB.p;
            tp.orig = (const int32_t *)g.orig.p;
            tp.pairs = (const float *)g.pairs.p;
            tp.aimg = (const uint4 *)g.aimg.p; tp.ntiles = g.ntiles;
            tp.tile_bounds = g.ntiles > 0 ? (const float4 *)g.tbounds.p : nullptr;
            tp.tile_boxes = (g.ntiles > 0 && !getenv("RTMI_NO_CULL")) ? (const float4 *)g.tboxes.p : nullptr;      // (RTMI_NO_CULL: comparison runs)
            {
                const size_t ngroups = (size_t)(g.ntiles + kBoxGroupTiles - 1) / kBoxGroupTiles;
                const bool groups = tp.tile_boxes != nullptr && !getenv("RTMI_NO_BOX_GROUPS");
                tp.group_boxes = groups ? tp.tile_boxes + ngroups * (kBoxGroupTiles / 2) * 3 : nullptr;
                tp.mask_group_boxes = groups ? tp.group_boxes + ((ngroups + 1) / 2) * 3 : nullptr;
            }
            tp.box_margin = 1.0f / 512.0f;
            tp.mat_scale = g.mat_scale; tp.mat_c_near = g.c_near * g.mat_scale;
            tp.n = g.n; tp.n_pad = g.n_pad;
            tp.nbig = g.nbig; memcpy(tp.big, g.big, sizeof tp.big); tp.c_near = g.c_near; tp.core_div_ok = g.core_div_ok;
            // one ray over all spheres costs about 9 vector instructions per sphere pair in the packed-FP32
            // scan (for all 64 lanes at once) and about 26 per 64 spheres plus a 6-step reduction when the
            // lanes share one ray: the shared form wins while few rays are left in the wave.  The matrix scan
            // costs a wave about 54 instruction times per 32-sphere tile (its matrix instructions included) plus
            // 150 for the ray features: break-even at 4 rays whatever the scene size (measured on the 488-sphere
            // scene, 1920x1080 at 1 spp per frame: 800 / 809 / 783 / 757 frames/s at 2 / 4 / 6 / 8 rays).
            // With the box test a wave of few rays multiplies few tiles, whatever the scene holds: about 150 + 12 per
            // tile box asked (at most the 16 of a small scene, or of the levels of a large one) + 54 for each of some three
            // tiles, while the shared form still walks all spheres: it keeps its place in small scenes (2 rays at 488
            // spheres; 2.64 to 2.68 ms for 1080p x 8 spp at every threshold from 0 to 4) and is off in large ones (4098
            // spheres, 1920x1080 x 20 spp: 15.2 ms with the old threshold of 4 rays, 13.9 ms without the shared form).
            const int shared_cost = ((g.n + 63) / 64) * 26 + 60;
            if (is_matrix(scan_variant(g)) && tp.tile_boxes != nullptr)
                tp.coop_max = std::min(16, (150 + std::min(g.ntiles, 16) * 12 + 3 * 54 + shared_cost / 2) / shared_cost);
            else if (is_matrix(scan_variant(g))) tp.coop_max = std::max(2, std::min(16, (150 + g.ntiles * 54 + shared_cost / 2) / shared_cost));
            else tp.coop_max = std::min(16, (g.n_pad / 2 * 9 + 200) / shared_cost);
            if (const char *cm = getenv("RTMI_COOP")) tp.coop_max = std::max(0, atoi(cm));       // comparison runs
            tp.width = width; tp.height = height; tp.npix = npix;
            tp.tile_rows = tile_rows; tp.part_index = part_index; tp.part_count = part_count;
            tp.rows = nullptr;
            if (explicit_rows) {
                HIP_TRY(g.rows.reserve((size_t)rows * sizeof(uint32_t)));
                HIP_TRY(hipMemcpyAsync(g.rows.p, o.rows, (size_t)rows * sizeof(uint32_t), hipMemcpyHostToDevice, g.stream));
                tp.rows = (const uint32_t *)g.rows.p;
            }
            tp.spp = spp; tp.chunk_spp = chunk; tp.nchunks = nchunks; tp.max_depth = max_depth;
            tp.mode = o.mode; tp.seed = o.seed; tp.sample_base = sample_base;
            tp.partial = d_partial;
            tp.hit_index = d_hit;
            tp.counters = d_counters;
            tp.total_units = units;

            const int variant = scan_variant(g);
            const size_t lds = trace_lds_bytes(variant, reference_arithmetic(build), g.n_pad, g.ntiles);
            const int per_cu = trace_blocks_per_cu(g.launch, build, variant, lds);
            uint64_t blocks = (uint64_t)g.num_cu * (uint64_t)per_cu;
            const uint64_t need = (units + block_threads_of(variant) - 1) / block_threads_of(variant);
            if (blocks > need) blocks = need;
            // Units a wave takes from the global counter at a time.  A refill is one atomic on one address for the whole
            // chip and stalls the wave for about one and a half of its iterations, so it should be rare; the last pool
            // of a wave ends at a random point of the launch, so it should be short.  With U units per wave and T
            // iterations per unit (3.3 rays per sample) the two losses, 1.5 U / pool and pool T / 128 iterations,
            // balance at pool = sqrt(192 U / T); never more than half of what a wave gets in all.  Measured, 64 against
            // the choice made here: 3840x2160 x 250 spp 246.5 -> 232.9 ms, 1920x1080 x 8 spp 3.80 -> 2.91 ms, the 1-spp
            // 1080p frame 1.00 -> 0.89 ms.
            {
                const double waves = (double)blocks * (block_threads_of(variant) / 64);
                const double per_wave = (double)units / std::max(1.0, waves);
                const double iters = 3.3 * (double)std::max(1, std::min(chunk, spp));
                double pool = std::sqrt(192.0 * per_wave / iters);
                // Short launches end on their longest paths (one sample in two hundred runs all 50 bounces), which a
                // wave's last pool sits in front of: there a wave should get its work in sixteen pools at least, of
                // 128 units or more (measured with the exact build, 128 / 256 / 512 units: 1920x1080 x 8 spp 3.08 / 3.04 /
                // 3.38 ms, 1200x675 x 16 spp 2.60 / 2.60 / 2.81 ms, the 1-spp 1080p frame 1.09 / 1.09 / -; long
                // launches keep 512: 1200x675 x 500 spp 52.8 / 52.3 / 52.0 ms).
                pool = std::min(pool, std::max(128.0, per_wave / 16.0));
                pool = std::min(pool, per_wave / 2.0);
                tp.pool_units = 64 * (int)std::max(1.0, std::min(8.0, std::floor(pool / 64.0 + 0.5)));
                if (const char *pu = getenv("RTMI_POOL")) tp.pool_units = std::max(1, atoi(pu));     // comparison runs
            }

            HIP_TRY(hipEventRecord(g.ev0, g.stream));
            HIP_TRY(launch_trace(g.launch, build, tp, variant, (int)blocks, lds, g.stream));
            HIP_TRY(hipEventRecord(g.ev1, g.stream));
            ResolveParams rp;
            rp.partial = d_partial; rp.nchunks = nchunks; rp.npix = npix;
            rp.running = progressive ? (float4 *)g.running.p : nullptr;
            rp.inv_spp = (float)1.0f / (float)(progressive ? spp * (int)(frame_index + 1) : spp);
            rp.rgba = d_rgba; rp.accum = d_accum;
            HIP_TRY(launch_resolve(rp, g.stream));
            HIP_TRY(hipEventRecord(g.ev2, g.stream));
        }
    }

    // read back (host-pointer entry) and/or wait
    // The owned rows sit compactly in the device buffers; every run of consecutive image rows (the whole
    // image for an undivided frame, one row tile of a part otherwise) is copied straight to its place in
    // the caller's buffers, with no staging copy on the host in between: by the DMA engine if the buffer
    // came from rt_host_alloc, staged by the runtime otherwise.
    if (!device_ptrs && npix > 0) {
        uint32_t k0 = 0;
        while (k0 < rows) {
            const uint32_t y0 = explicit_rows ? o.rows[k0] : rt_part_row_at(k0, tile_rows, part_index, part_count);
            uint32_t k1 = k0 + 1;
            while (k1 < rows && (explicit_rows ? o.rows[k1] : rt_part_row_at(k1, tile_rows, part_index, part_count)) == y0 + (k1 - k0)) k1++;
            const size_t px0 = (size_t)k0 * width, at = (size_t)y0 * width, npx = (size_t)(k1 - k0) * width;
            HIP_TRY(hipMemcpyAsync(rgba + at * 4, d_rgba + px0 * 4, npx * 4, hipMemcpyDeviceToHost, g.stream));
            if (o.accum_out) HIP_TRY(hipMemcpyAsync(o.accum_out + at * 3, d_accum + px0 * 3, npx * 12, hipMemcpyDeviceToHost, g.stream));
            if (o.hit_index_out) HIP_TRY(hipMemcpyAsync(o.hit_index_out + at, d_hit + px0, npx * 4, hipMemcpyDeviceToHost, g.stream));
            k0 = k1;
        }
    }
    HIP_TRY(hipStreamSynchronize(g.stream));

    if (progressive) g.prog_frames = frames_after;

    if (o.stats) {
        st.samples = (uint64_t)npix * (uint64_t)spp;
        st.chunk_spp = (uint32_t)chunk;
        if (npix > 0 && max_depth > 0) {
            unsigned long long *c = g.last_counters;
            HIP_TRY(hipMemcpy(c, d_counters, sizeof g.last_counters, hipMemcpyDeviceToHost));
            st.rays = c[1];
            st.wave_scans = c[19];
            st.scan_tiles = is_matrix(scan_variant(g)) ? (uint32_t)g.ntiles : 0;
            st.tile_products = c[22];
#ifdef RT_DIAG
            fprintf(stderr, "diag: wave scans %llu; per wave scan: pair branches taken %.1f, filter passes %.1f, "
                    "queued candidates %.2f, drain iterations %.2f, live lanes %.1f; rays traced in shared-ray mode %llu of %llu; "
                    "rays whose nearest hit differs from the all-spheres exact test: %llu\n",
                    c[5], (double)c[2] / c[5], (double)c[7] / c[5], (double)c[3] / c[5], (double)c[4] / c[5],
                    (double)(c[6] & ((1ULL << 40) - 1)) / c[5], c[6] >> 40, c[1], c[8]);
#ifndef RT_DIAG_STAMPS
            if (g.ntiles > 0 && c[5] > 0)
                fprintf(stderr, "diag: of %d tiles per wave scan, %.2f have a bounding sphere some live ray passes; of %d (tile, half of the wave's rays) "
                        "pairs %.2f; of %d (16-sphere half tile, half of the rays) pairs %.2f; with the tiles' axis-aligned boxes instead: %.2f tiles, "
                        "%.2f (tile, half) pairs; tiles multiplied %.2f\n", g.ntiles, (double)c[9] / c[5], 2 * g.ntiles,
                        (double)c[10] / c[5], 4 * g.ntiles, (double)c[15] / c[5], (double)c[16] / c[5], (double)c[17] / c[5], (double)c[18] / c[5]);
            if (c[5] > 0)
                fprintf(stderr, "diag: share of the wave scans with 5-8 / 9-16 / 17-32 / 33-48 live rays: %.1f / %.1f / %.1f / %.1f %%\n",
                        100.0 * c[11] / c[5], 100.0 * c[12] / c[5], 100.0 * c[13] / c[5], 100.0 * c[14] / c[5]);
#endif
#ifdef RT_DIAG_STAMPS
            {
                const double tot = (double)(c[9] + c[10] + c[11] + c[12] + c[13] + c[14] + c[15] + c[20] + c[21] + c[23]);
                fprintf(stderr, "diag: wave-cycles per phase (share of the main loop): unit hand-out %.2f %%, new sample %.2f %%, tile boxes %.2f %%, "
                        "ray features %.2f %%, tile loop (products, compares) %.2f %%, appends %.2f %%, drain %.2f %%, big spheres %.2f %%, shade %.2f %%, shared-ray search %.2f %%; "
                        "%.0f cycles per wave scan\n", 100.0 * c[9] / tot, 100.0 * c[10] / tot, 100.0 * c[23] / tot, 100.0 * c[20] / tot, 100.0 * c[11] / tot, 100.0 * c[21] / tot, 100.0 * c[12] / tot,
                        100.0 * c[13] / tot, 100.0 * c[14] / tot, 100.0 * c[15] / tot, tot / (double)c[5]);
                if (c[18] > 0 && c[17] > 0)
                    fprintf(stderr, "diag: %llu waves lived %.1f %% of the longest wave's %.3f M cycles on average "
                            "(the rest is the end of the launch: waves that found no more units)\n",
                            c[18], 100.0 * (double)c[16] / (double)c[18] / (double)c[17], (double)c[17] * 1e-6);
            }
#endif
#endif
            st.sphere_tests = c[1] * (uint64_t)g.n;
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, g.ev0, g.ev1));
            st.kernel_ms = ms;
            HIP_TRY(hipEventElapsedTime(&ms, g.ev1, g.ev2));
            st.resolve_ms = ms;
        }
        st.total_ms = std::chrono::duration<double, std::milli>(clk::now() - t_start).count();
        *o.stats = st;
    }
    return RT_OK;
}

void release_ctx(Ctx &g)
{
    if (!g.inited) return;
    (void)hipSetDevice(g.device);
    if (g.stream) (void)hipStreamSynchronize(g.stream);
    g.geom.release(); g.matA.release(); g.matB.release(); g.pairs.release();
    g.aimg.release(); g.tbounds.release(); g.tboxes.release();
    g.orig.release();
    g.partial.release(); g.running.release(); g.rows.release(); g.counters.release(); g.rgba.release(); g.accum.release(); g.hit.release();
    if (g.ev0) (void)hipEventDestroy(g.ev0);
    if (g.ev1) (void)hipEventDestroy(g.ev1);
    if (g.ev2) (void)hipEventDestroy(g.ev2);
    if (g.stream) (void)hipStreamDestroy(g.stream);
    g = Ctx();
}

int init_ctx(Ctx &g, int device, int count)
{
    if (device < 0 || device >= count) return fail(RT_ERR_ARG, "rt_init: device %d outside [0,%d)", device, count);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RT_ERR_DEVICE, "rt_init: device %d is %s; this library holds gfx950 code only", device, prop.gcnArchName);
    g.device = device;
    g.num_cu = prop.multiProcessorCount;
    g.inited = true;                      // from here on release_ctx cleans up
    HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&g.ev0));
    HIP_TRY(hipEventCreate(&g.ev1));
    HIP_TRY(hipEventCreate(&g.ev2));
    HIP_TRY(g.counters.reserve(kScratchCountersBytes));
    // this device's copy of the static scratch array (each device holds its own, at its own address)
    void *sp = nullptr;
    HIP_TRY(hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch)));
    g.scratch_ptr = (unsigned char *)sp;
    return RT_OK;
}

void shutdown_all()
{
    g_workers.clear();                    // joins the worker threads
    for (int i = 0; i < kMaxContexts; i++) release_ctx(g_ctx[i]);
    g_nctx = 0;
}

// Several contexts: the rows this call owns (all rows, the caller's part, or the caller's row list)
// are dealt in tiles to the contexts in proportion to the speed each showed in the previous frame
// (smooth weighted round robin: every context's tiles stay spread over the image), one host thread
// per context renders its rows into the caller's buffers.  The image does not depend on the deal:
// the unit size is resolved once, from the full image, for all contexts.
// Progressive frames keep the deal of frame 0 (each context carries the running sums of its rows).
std::vector<std::vector<uint32_t>> g_prog_rows;      // the deal of frame 0
std::vector<uint32_t> g_prog_mine;                   // the rows the caller owned in frame 0

int render_on_all(const rt_camera *cam, uint8_t *rgba, uint32_t width, uint32_t height, int spp,
                  int max_depth, const rt_opts *opts, int chunk, int64_t frame_index = -1)
{
    using clk = std::chrono::steady_clock;
    const auto t_start = clk::now();
    rt_opts base;
    memset(&base, 0, sizeof base);
    if (opts) base = *opts;
    if (base.tile_rows < 0) return fail(RT_ERR_ARG, "rt_render: negative tile_rows");
    const int tile_rows = base.tile_rows > 0 ? base.tile_rows : 16;
    const int part_count = base.part_count > 1 ? base.part_count : 1;
    const int part_index = part_count > 1 ? base.part_index : 0;
    if (part_index < 0 || part_index >= part_count) return fail(RT_ERR_ARG, "rt_render: part_index %d outside [0,%d)", base.part_index, part_count);
    std::vector<uint32_t> mine;
    if (base.rows) {
        for (uint32_t k = 0; k < base.row_count; k++) {
            if (base.rows[k] >= height || (k > 0 && base.rows[k] <= base.rows[k - 1]))
                return fail(RT_ERR_ARG, "rt_render: rows[] must be strictly ascending and below the image height (entry %u)", k);
            mine.push_back(base.rows[k]);
        }
    } else {
        const uint32_t n = rt_part_rows(height, tile_rows, part_index, part_count);
        for (uint32_t k = 0; k < n; k++) mine.push_back(rt_part_row_at(k, tile_rows, part_index, part_count));
    }
    const int nc = g_nctx;
    const bool progressive = frame_index >= 0;
    auto end_sequence = [&]() {
        for (int i = 0; i < nc; i++) g_ctx[i].prog_frames = kProgBroken;
        g_prog_rows.clear();
        g_prog_mine.clear();
    };
    std::vector<std::vector<uint32_t>> rows(nc);
    if (progressive && frame_index > 0) {
        if ((int)g_prog_rows.size() != nc || g_prog_mine != mine) {
            end_sequence();
            return fail(RT_ERR_STATE, "rt_render_progressive: frame %lld does not own the rows frame 0 owned; start again with frame_index 0",
                        (long long)frame_index);
        }
        rows = g_prog_rows;
    } else {
        double wsum = 0.0;
        for (int i = 0; i < nc; i++) wsum += g_ctx[i].speed;
        std::vector<double> credit(nc, 0.0);
        const size_t deal_rows = 2;                           // fine tiles: the split follows the weights closely
        for (size_t t = 0; t < mine.size(); t += deal_rows) {
            int best = 0;
            for (int i = 0; i < nc; i++) {
                credit[i] += g_ctx[i].speed / wsum;
                if (credit[i] > credit[best] + 1e-12) best = i;
            }
            credit[best] -= 1.0;
            for (size_t k = t; k < std::min(mine.size(), t + deal_rows); k++) rows[best].push_back(mine[k]);
        }
        if (progressive) { g_prog_rows = rows; g_prog_mine = mine; }
    }

    std::vector<rt_stats> st(nc);
    std::vector<int> rc(nc, RT_OK);
    std::vector<std::string> msg(nc);
    for (int i = 0; i < nc; i++) memset(&st[i], 0, sizeof(rt_stats));
    Call call;
    call.chunk = chunk;
    call.frame_index = frame_index;
    auto work = [&](int i) {
        if (rows[i].empty()) return;                     // fewer tiles than contexts
        rc[i] = guarded("rt_render", [&]() {
            rt_opts o = base;
            o.rows = rows[i].data();
            o.row_count = (uint32_t)rows[i].size();
            o.stats = &st[i];
            return render_common(g_ctx[i], cam, rgba, width, height, spp, max_depth, &o, call);
        });
        if (rc[i] != RT_OK) { try { msg[i] = g_err; } catch (...) {} }
    };
    {
        // whatever happens below, every started worker has finished before the locals it uses go away
        struct Joiner {
            int started = 0;
            ~Joiner() { for (int i = 0; i < started; i++) g_workers[i]->wait(); }
        } joiner;
        for (int i = 1; i < nc; i++) {
            g_workers[i - 1]->start([&work, i] { work(i); });
            joiner.started = i;
        }
        work(0);
    }
    for (int i = 0; i < nc; i++)
        if (rc[i] != RT_OK) {
            if (progressive) end_sequence();             // the other contexts have already added this frame
            return fail(rc[i], "device %d: %s", g_ctx[i].device, msg[i].c_str());
        }

    rt_stats total;
    memset(&total, 0, sizeof total);
    total.chunk_spp = (uint32_t)chunk;
    for (int i = 0; i < nc; i++) {
        total.samples += st[i].samples; total.rays += st[i].rays; total.sphere_tests += st[i].sphere_tests;
        total.wave_scans += st[i].wave_scans; total.scan_tiles = std::max(total.scan_tiles, st[i].scan_tiles);
        total.kernel_ms = std::max(total.kernel_ms, st[i].kernel_ms);
        total.resolve_ms = std::max(total.resolve_ms, st[i].resolve_ms);
        if (st[i].kernel_ms > 0.0 && !rows[i].empty())
            g_ctx[i].speed = (double)rows[i].size() / st[i].kernel_ms;
    }
    // keep the weights comparable and no context starved by one odd timing
    double mean = 0.0;
    for (int i = 0; i < nc; i++) mean += g_ctx[i].speed / nc;
    for (int i = 0; i < nc; i++) g_ctx[i].speed = std::min(1.33, std::max(0.75, g_ctx[i].speed / mean));
    total.total_ms = std::chrono::duration<double, std::milli>(clk::now() - t_start).count();
    if (base.stats) *base.stats = total;
    return RT_OK;
}

// Common front of the render entry points: state and argument checks that must precede the choice of
// the unit size, then the single- or multi-context path.
int render_entry(const char *name, const rt_camera *cam, uint8_t *rgba, uint32_t width, uint32_t height, int spp,
                 int max_depth, const rt_opts *opts, bool device_ptrs, bool no_alloc, int64_t frame_index)
{
    if (g_nctx == 0) return fail(RT_ERR_STATE, "%s: rt_init has not been called", name);
    if (device_ptrs && g_nctx > 1)
        return fail(RT_ERR_STATE, "%s: device buffers belong to one device; the library was opened on %d", name, g_nctx);
    if (width == 0 || height == 0 || (uint64_t)width * height > 0x7fffffffULL)
        return fail(RT_ERR_ARG, "%s: bad image size %ux%u", name, width, height);
    if (spp < 1) return fail(RT_ERR_ARG, "%s: spp must be >= 1 (got %d)", name, spp);
    const int chunk_opt = opts ? opts->chunk_spp : RT_CHUNK_AUTO;
    if (chunk_opt < 0 && chunk_opt != RT_CHUNK_AUTO) return fail(RT_ERR_ARG, "%s: negative chunk_spp", name);
    DeviceGuard keep;
    const int chunk = resolve_chunk(chunk_opt, width, height, spp, frame_index >= 0);
    if (g_nctx > 1) return render_on_all(cam, rgba, width, height, spp, max_depth, opts, chunk, frame_index);
    Call call;
    call.device_ptrs = device_ptrs;
    call.no_alloc = no_alloc;
    call.frame_index = frame_index;
    call.chunk = chunk;
    return render_common(g_ctx[0], cam, rgba, width, height, spp, max_depth, opts, call);
}

}  // namespace

extern "C" {

const char *rt_last_error(void) { return g_err; }

int rt_init_devices(const int *devices, int count)
{
    return guarded("rt_init", [&]() -> int {
        DeviceGuard keep;
        shutdown_all();
        const char *scan = getenv("RTMI_SCAN");
        g_vector_scan = scan && strcmp(scan, "vector") == 0;
        const char *ord = getenv("RTMI_ORDER");
        g_caller_order = ord && strcmp(ord, "caller") == 0;
        g_curve_order = ord && strcmp(ord, "curve") == 0;
        if (!devices || count < 1 || count > kMaxContexts)
            return fail(RT_ERR_ARG, "rt_init_devices: between 1 and %d devices (got %d)", kMaxContexts, count);
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0) {
            (void)hipGetLastError();
            return fail(RT_ERR_DEVICE, "rt_init: no HIP device (%s)", hipGetErrorString(e));
        }
        for (int i = 0; i < count; i++) {
            const int rc = init_ctx(g_ctx[i], devices[i], ndev);
            if (rc != RT_OK) {
                for (int k = 0; k <= i; k++) release_ctx(g_ctx[k]);
                return rc;
            }
        }
        try {
            for (int i = 1; i < count; i++) g_workers.emplace_back(new Worker());
        } catch (...) {
            g_workers.clear();
            for (int k = 0; k < count; k++) release_ctx(g_ctx[k]);
            return fail(RT_ERR_NOMEM, "rt_init_devices: cannot start the per-device host threads");
        }
        g_nctx = count;
        g_err[0] = 0;
        return RT_OK;
    });
}

int rt_init(int device) { return rt_init_devices(&device, 1); }

// Interleaves the low 10 bits of x, y, z: the position of a cell of a 1024^3 grid along the Z-order curve.
static uint32_t morton3(uint32_t x, uint32_t y, uint32_t z)
{
    auto spread = [](uint32_t v) {
        v &= 0x3ffU;
        v = (v | (v << 16)) & 0x030000ffU;
        v = (v | (v << 8)) & 0x0300f00fU;
        v = (v | (v << 4)) & 0x030c30c3U;
        v = (v | (v << 2)) & 0x09249249U;
        return v;
    };
    return spread(x) | (spread(y) << 1) | (spread(z) << 2);
}

// Device order of the spheres: the 32 spheres of a tile of the matrix scan are neighbours in space.  First along the
// Z-order curve through their centres (cubic cells over the bulk of the finite centres; equal cells keep the caller's
// order): the rays of a wave start near each other, and what they pass then sits in few tiles, so the appends, which
// cost per half tile that holds a candidate, fell by a quarter on the 488-sphere scene (4 % of the kernel time; a
// shuffled list ran 6 % slower than the caller's order, which was sorted by one coordinate).  Then (split_tiles) the
// list is split in space so that the tiles' bounding boxes are small (below).
// Nothing a caller sees depends on the order: hits are reported by the caller's index and equal distances go to
// the lower caller's index (order[] travels to the device for that).
static std::vector<int32_t> device_order(const rt_sphere *spheres, int n, bool split_tiles)
{
    std::vector<int32_t> order(n);
    for (int i = 0; i < n; i++) order[i] = i;
    // the grid spans the 2 % .. 98 % quantiles of each coordinate (cells beyond are clamped): one far-away or huge
    // sphere (a ground of radius 1000 centred 1000 below) must not blow the cells up until the rest shares one
    auto coord = [&](int i, int a) { return (double)(a == 0 ? spheres[i].center.x : a == 1 ? spheres[i].center.y : spheres[i].center.z); };
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    bool any = false;
    std::vector<double> v;
    v.reserve(n);
    for (int a = 0; a < 3; a++) {
        v.clear();
        for (int i = 0; i < n; i++)
            if (std::isfinite(coord(i, 0)) && std::isfinite(coord(i, 1)) && std::isfinite(coord(i, 2))) v.push_back(coord(i, a));
        if (v.empty()) break;
        any = true;
        std::sort(v.begin(), v.end());
        lo[a] = v[(size_t)(0.02 * (double)(v.size() - 1))];
        hi[a] = v[(size_t)(0.98 * (double)(v.size() - 1) + 0.999)];
        if (!(hi[a] > lo[a])) { lo[a] = v.front(); hi[a] = v.back(); }
    }
    const double span = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
    if (!any || !(span > 0.0) || !std::isfinite(span)) return order;
    // fine cells: with 8 to 32 cells per axis the spheres of a cell stay in the caller's order and the 488-sphere
    // scene ran 4 % slower (239.5 against 230.4 ms); 64 cells: 234.2 ms
    const double cells = 1024.0;
    std::vector<uint32_t> key(n, 0xffffffffU);
    std::vector<uint32_t> cell((size_t)n * 3, 1023U);
    for (int i = 0; i < n; i++) {
        bool ok = true;
        for (int a = 0; a < 3; a++) {
            const double c = coord(i, a);
            if (!std::isfinite(c)) { ok = false; break; }
            const double q = (c - lo[a]) / span * cells;
            cell[(size_t)i * 3 + a] = (uint32_t)std::min(cells - 1.0, std::max(0.0, q));
        }
        if (ok) key[i] = morton3(cell[(size_t)i * 3], cell[(size_t)i * 3 + 1], cell[(size_t)i * 3 + 2]);
        else cell[(size_t)i * 3] = cell[(size_t)i * 3 + 1] = cell[(size_t)i * 3 + 2] = 1023U;
    }
    auto along_curve = [&](int32_t a, int32_t b) { return key[a] < key[b]; };
    std::stable_sort(order.begin(), order.end(), along_curve);
    if (!split_tiles || n <= kTileSpheres) return order;
    // Tiles as boxes: the matrix scan skips a tile whose bounding box no ray of the wave meets (rt_trace_impl.h), and
    // 32 consecutive spheres along the curve straddle its jumps.  So the list is split in space instead, as a
    // bounding-volume hierarchy is built: a part goes in two across the axis and at the multiple of 32 spheres where
    // area(left box) * spheres left + area(right box) * spheres right is smallest (a box is met by rays in
    // proportion to its area), until a part is one tile; within a tile the spheres stay along the curve (the compare
    // tree's groups hold neighbours).  A sphere far larger or farther out than the rest (the filter leaves those to a
    // test of their own) counts as the point of its clamped cell.
    std::vector<double> ext(n);
    for (int i = 0; i < n; i++)
        ext[i] = std::sqrt(coord(i, 0) * coord(i, 0) + coord(i, 1) * coord(i, 1) + coord(i, 2) * coord(i, 2)) + std::fabs((double)spheres[i].radius);
    std::vector<double> by_size(ext);
    std::sort(by_size.begin(), by_size.end());          // NaN extents: any order will do, they compare false below
    const double outlier_from = 4.0 * by_size[n / 2];
    std::vector<double> blo((size_t)n * 3), bhi((size_t)n * 3);
    for (int i = 0; i < n; i++) {
        const bool plain = ext[i] <= outlier_from;
        const double r = plain ? std::fabs((double)spheres[i].radius) : 0.0;
        for (int a = 0; a < 3; a++) {
            const double c = plain ? coord(i, a) : lo[a] + ((double)cell[(size_t)i * 3 + a] + 0.5) * span / cells;
            blo[(size_t)i * 3 + a] = c - r;
            bhi[(size_t)i * 3 + a] = c + r;
        }
    }
    struct Box {
        double l[3] = {1e300, 1e300, 1e300}, h[3] = {-1e300, -1e300, -1e300};
        void add(const double *bl, const double *bh) { for (int a = 0; a < 3; a++) { l[a] = std::min(l[a], bl[a]); h[a] = std::max(h[a], bh[a]); } }
        double area() const {
            const double x = h[0] - l[0], y = h[1] - l[1], z = h[2] - l[2];
            return x < 0.0 ? 0.0 : x * y + y * z + z * x;
        }
    };
    std::vector<int32_t> trial;
    std::vector<double> left_area;
    std::function<void(int, int)> split = [&](int begin, int end) {
        const int m = end - begin, tiles = (m + kTileSpheres - 1) / kTileSpheres;
        if (tiles <= 1) return;
        double best_cost = std::numeric_limits<double>::infinity();
        int best_left = 0, best_axis = 0;
        for (int axis = 0; axis < 3; axis++) {
            // [begin, end) is in curve order on entry: the stable sort keeps it within equal coordinates, so the
            // order depends on nothing but the scene
            trial.assign(order.begin() + begin, order.begin() + end);
            std::stable_sort(trial.begin(), trial.end(), [&](int32_t a, int32_t b) {
                return blo[(size_t)a * 3 + axis] + bhi[(size_t)a * 3 + axis] < blo[(size_t)b * 3 + axis] + bhi[(size_t)b * 3 + axis];
            });
            left_area.assign(tiles, 0.0);
            Box box;
            for (int k = 0; k < m; k++) {
                if (k % kTileSpheres == 0) left_area[k / kTileSpheres] = box.area();
                box.add(&blo[(size_t)trial[k] * 3], &bhi[(size_t)trial[k] * 3]);
            }
            box = Box();
            for (int k = m - 1; k >= kTileSpheres; k--) {
                box.add(&blo[(size_t)trial[k] * 3], &bhi[(size_t)trial[k] * 3]);
                if (k % kTileSpheres != 0) continue;
                const double cost = left_area[k / kTileSpheres] * (double)k + box.area() * (double)(m - k);
                // (a part of more than eight tiles is cut in its middle half: the depth, and with it the time, stay bounded)
                if (tiles > 8 && (k < m / 4 || k > m - m / 4)) continue;
                if (cost < best_cost) { best_cost = cost; best_left = k; best_axis = axis; }
            }
        }
        if (best_left == 0) return;                      // nothing finite to go by: the part stays along the curve
        std::stable_sort(order.begin() + begin, order.begin() + end, [&](int32_t a, int32_t b) {
            return blo[(size_t)a * 3 + best_axis] + bhi[(size_t)a * 3 + best_axis] < blo[(size_t)b * 3 + best_axis] + bhi[(size_t)b * 3 + best_axis];
        });
        const int mid = begin + best_left;
        std::stable_sort(order.begin() + begin, order.begin() + mid, along_curve);
        std::stable_sort(order.begin() + mid, order.begin() + end, along_curve);
        split(begin, mid);
        split(mid, end);
    };
    split(0, n);
    return order;
}

static int set_world_impl(const rt_sphere *caller_spheres, size_t count)
{
    if (g_nctx == 0) return fail(RT_ERR_STATE, "rt_set_world: rt_init has not been called");
    if (!caller_spheres && count) return fail(RT_ERR_ARG, "rt_set_world: null sphere array");
    if (count > (size_t)kMaxSpheres)
        return fail(RT_ERR_ARG, "rt_set_world: %zu spheres; at most %d are supported", count, kMaxSpheres);
    for (size_t i = 0; i < count; i++)
        if (caller_spheres[i].mat_type < RT_MAT_LAMBERTIAN || caller_spheres[i].mat_type > RT_MAT_DIELECTRIC)
            return fail(RT_ERR_ARG, "rt_set_world: sphere %zu has unsupported material type %d", i, caller_spheres[i].mat_type);
    DeviceGuard keep;
    const int n = (int)count;
    const int n_pad = n == 0 ? kGeomPad : ((n + kGeomPad - 1) / kGeomPad) * kGeomPad;
    // From here on "sphere i" is the i-th sphere in device order; orig[i] is its index in the caller's list.
    std::vector<int32_t> orig = g_caller_order ? std::vector<int32_t>() : device_order(caller_spheres, n, !g_curve_order);
    if (g_caller_order) { orig.resize(n); for (int i = 0; i < n; i++) orig[i] = i; }
    std::vector<rt_sphere> in_order(n > 0 ? n : 1);
    for (int i = 0; i < n; i++) in_order[i] = caller_spheres[orig[i]];
    const rt_sphere *spheres = in_order.data();
    orig.resize(n_pad);
    for (int i = n; i < n_pad; i++) orig[i] = i;
    std::vector<float4> geom(n_pad), matA(n > 0 ? n : 1), matB(n > 0 ? n : 1);
    for (int i = 0; i < n_pad; i++) {
        if (i < n) {
            const rt_sphere &s = spheres[i];
            geom[i] = make_float4(s.center.x, s.center.y, s.center.z, s.radius * s.radius);   // Main.c:650
            matA[i] = make_float4(s.albedo.x, s.albedo.y, s.albedo.z, s.fuzz);
            if (s.mat_type == RT_MAT_DIELECTRIC) {
                // reflectance()'s r0 (Main.c:553-554) for a ray entering (ri = 1.0f / index, Main.c:589) and leaving (ri = index)
                const float ri_in = 1.0f / s.refraction_index, ri_out = s.refraction_index;
                float r0_in = (1 - ri_in) / (1 + ri_in), r0_out = (1 - ri_out) / (1 + ri_out);
                r0_in = r0_in * r0_in;
                r0_out = r0_out * r0_out;
                matA[i] = make_float4(r0_in, r0_out, 0.0f, s.fuzz);
            }
            int32_t mt = s.mat_type;
            float mtf;
            memcpy(&mtf, &mt, 4);
            matB[i] = make_float4(s.refraction_index, mtf, 1.0f / s.radius, 1.0f / s.refraction_index);   // Main.c:679, :589
        } else {
            geom[i] = make_float4(0.0f, 0.0f, 0.0f, -std::numeric_limits<float>::infinity());
        }
    }
    // Filter data.  A sphere is "big" when its extent |C| + r exceeds four times the
    // median extent (at most kMaxBig of them, the largest first).
    std::vector<double> extent(n > 0 ? n : 1, 0.0);
    for (int i = 0; i < n; i++) {
        const rt_sphere &s = spheres[i];
        extent[i] = std::sqrt((double)s.center.x * s.center.x + (double)s.center.y * s.center.y +
                              (double)s.center.z * s.center.z) + std::fabs((double)s.radius);
    }
    std::vector<double> sorted(extent.begin(), extent.begin() + n);
    std::sort(sorted.begin(), sorted.end());
    double big_from = n > 0 ? 4.0 * sorted[n / 2] : 0.0;
    if (n > kMaxBig && big_from < sorted[n - kMaxBig - 1]) big_from = sorted[n - kMaxBig - 1];
    int nbig = 0;
    int big[kMaxBig] = {0};
    std::vector<char> is_big(n_pad, 0);
    for (int i = 0; i < n && nbig < kMaxBig; i++)
        if (extent[i] > big_from && n > 1) { is_big[i] = 1; big[nbig++] = i; }
    double c_near = 0.0;
    std::vector<float> pairs((size_t)n_pad * 4 + 8, 0.0f);     // one spare record: the scan loads one pair ahead
    for (int i = 0; i < n_pad; i++) {
        float *q = &pairs[(size_t)(i / 2) * 8 + (i & 1)];
        const float inf = std::numeric_limits<float>::infinity();
        if (i < n && !is_big[i]) {
            const rt_sphere &s = spheres[i];
            const double c2 = (double)s.center.x * s.center.x + (double)s.center.y * s.center.y +
                              (double)s.center.z * s.center.z;
            q[0] = s.center.x; q[2] = s.center.y; q[4] = s.center.z;
            q[6] = (float)(c2 - (double)s.radius * s.radius);
            if (std::sqrt(c2) > c_near) c_near = std::sqrt(c2);
        } else {
            q[0] = 0.0f; q[2] = 0.0f; q[4] = 0.0f; q[6] = inf;       // never passes the filter
        }
    }
    // Matrix scan: per 32-sphere tile the MFMA A fragments of the sphere features (rt_device.h), in a
    // frame scaled by a power of two that puts every filtered sphere within 120 of the origin (f16 range:
    // products of coordinates stay below 65504).  K entries of a sphere: feature f of (cx^2, cy^2, cz^2,
    // 2 cx cy, 2 cx cz, 2 cy cz, 2 cx, 2 cy, 2 cz) at 3 f .. 3 f + 2 as (c1, c1, c2); 27, 28: 1, 1; 29, 30: the pieces
    // of k = |C|^2 - r^2; 31: 0.  Rows beyond the scene and the big spheres hold k = 65504 only: the ray's
    // constant is capped at 60000 (rt_trace_impl.h), so their accumulator stays below -5000.
    // Below kMinSpheresMatrix the packed-FP32 scan wins: the matrix scan pays about a hundred instructions per
    // ray for its features whatever the scene holds (the 3-sphere scene ran 27 ms against 37 ms).
    const int ntiles = (n >= kMinSpheresMatrix && n <= kMaxSpheresMatrixGlobal) ? (n + kTileSpheres - 1) / kTileSpheres : 0;
    double reach = 0.0;
    for (int i = 0; i < n; i++) if (!is_big[i]) reach = std::max(reach, extent[i]);
    float mat_scale = 1.0f;
    if (reach > 0.0) mat_scale = (float)std::exp2(std::min(100.0, std::max(-100.0, std::floor(std::log2(120.0 / reach)))));
    std::vector<uint16_t> aimg((size_t)(ntiles + 1) * 2 * 64 * 8, 0);      // one spare tile: the scan reads one tile ahead
    for (int t = 0; t < ntiles; t++)
        for (int row = 0; row < kTileSpheres; row++) {
            // accumulator a of a lane in wave half h is tile row (a & 3) + 8 (a >> 2) + 4 h: that row holds sphere
            // 2 a + h of the tile, so that the three accumulators of a group of the compare tree cover six
            // consecutive spheres (neighbours in space) and a ray's candidates turn up in few groups
            const int i = t * kTileSpheres + 2 * ((row & 3) | ((row >> 3) << 2)) + ((row >> 2) & 1);
            uint16_t ent[32];
            memset(ent, 0, sizeof ent);
            if (i < n && !is_big[i]) {
                const double cx = (double)spheres[i].center.x * mat_scale, cy = (double)spheres[i].center.y * mat_scale,
                             cz = (double)spheres[i].center.z * mat_scale, rr = (double)spheres[i].radius * mat_scale;
                const float feat[9] = {(float)(cx * cx), (float)(cy * cy), (float)(cz * cz), (float)(2.0 * cx * cy), (float)(2.0 * cx * cz),
                                       (float)(2.0 * cy * cz), (float)(2.0 * cx), (float)(2.0 * cy), (float)(2.0 * cz)};
                for (int f = 0; f < 9; f++) {
                    uint16_t pc[2];
                    split2(feat[f], pc);
                    ent[3 * f] = pc[0]; ent[3 * f + 1] = pc[0]; ent[3 * f + 2] = pc[1];
                }
                ent[27] = ent[28] = f16_bits(1.0f);
                uint16_t kp[2];
                split2((float)(cx * cx + cy * cy + cz * cz - rr * rr), kp);
                ent[29] = kp[0]; ent[30] = kp[1];
            } else {
                ent[29] = f16_bits(65504.0f);
            }
            for (int s2 = 0; s2 < 2; s2++)
                for (int hh = 0; hh < 2; hh++)
                    for (int e = 0; e < 8; e++)
                        aimg[(((size_t)t * 2 + s2) * 64 + hh * 32 + row) * 8 + e] = ent[16 * s2 + 8 * hh + e];
        }
    // Until every upload to every context has succeeded no context holds a world: a failure in between (out of
    // memory on one device, say) must not leave new scalars next to old buffers, or some contexts on the new
    // world and others on the old one; rt_render then reports RT_ERR_STATE until rt_set_world succeeds.
    // bounding spheres of the tiles (diagnostic build: how many tiles a coarser test could skip)
    // [tile][5]: bounding sphere of the tile, of its two 16-sphere halves, and the tile's axis-aligned box (lo, hi)
    std::vector<float4> tbounds((size_t)std::max(1, ntiles) * 5, make_float4(0.0f, 0.0f, 0.0f, -1.0f));
    for (int t = 0; t < ntiles; t++) {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int i = t * kTileSpheres; i < std::min(n, (t + 1) * kTileSpheres); i++)
            if (!is_big[i]) {
                const double cc[3] = {spheres[i].center.x, spheres[i].center.y, spheres[i].center.z}, rr = std::fabs((double)spheres[i].radius) * 1.0001 + 1e-4;
                for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], cc[a] - rr); hi[a] = std::max(hi[a], cc[a] + rr); }
            }
        tbounds[(size_t)t * 5 + 3] = make_float4((float)lo[0], (float)lo[1], (float)lo[2], lo[0] <= hi[0] ? 1.0f : -1.0f);
        tbounds[(size_t)t * 5 + 4] = make_float4((float)hi[0], (float)hi[1], (float)hi[2], 0.0f);
        for (int part = 0; part < 3; part++) {
            const int i0 = t * kTileSpheres + (part == 2 ? 16 : 0), i1 = std::min(n, t * kTileSpheres + (part == 1 ? 16 : 32));
            double c[3] = {0, 0, 0};
            int m = 0;
            for (int i = i0; i < i1; i++)
                if (!is_big[i]) { c[0] += spheres[i].center.x; c[1] += spheres[i].center.y; c[2] += spheres[i].center.z; m++; }
            if (m == 0) continue;
            for (double &v : c) v /= m;
            double rad = 0.0;
            for (int i = i0; i < i1; i++)
                if (!is_big[i]) {
                    const double dx = spheres[i].center.x - c[0], dy = spheres[i].center.y - c[1], dz = spheres[i].center.z - c[2];
                    rad = std::max(rad, std::sqrt(dx * dx + dy * dy + dz * dz) + std::fabs((double)spheres[i].radius));
                }
            tbounds[(size_t)t * 5 + part] = make_float4((float)c[0], (float)c[1], (float)c[2], (float)(rad * 1.0001));
        }
    }
    // Axis-aligned boxes of the tiles' filtered spheres as centre and half extents: the matrix scan multiplies a tile only if
    // the box meets a live ray of the wave (rt_trace_impl.h).  Rounded outwards; a tile without filtered spheres gets a
    // box nothing meets.  Two tiles share a record of 12 floats, {cx, cx', cy, cy'}, {cz, cz', hx, hx'}, {hy, hy', hz, hz'}:
    // the test runs on two tiles per packed-FP32 instruction.
    // Behind the tiles' records (padded to whole groups of eight tiles): the boxes of those groups in the same form, two groups
    // per record.  The spheres are in the leaf order of a hierarchy (device_order), so eight consecutive tiles are
    // neighbours too; a scene of more than 64 tiles asks the groups first and the tiles of the groups that are met.
    const int ngroups = (std::max(1, ntiles) + kBoxGroupTiles - 1) / kBoxGroupTiles;
    const int npairs = ngroups * (kBoxGroupTiles / 2), ngpairs = (ngroups + 1) / 2, nmpairs = ((std::max(1, ntiles) + 63) / 64 + 1) / 2;
    std::vector<float> tboxes((size_t)(npairs + ngpairs + nmpairs) * 12, 0.0f);
    auto put_box = [&](float *rec, const double *l, const double *h, bool any) {
        const float ninf = -std::numeric_limits<float>::infinity();
        rec[6] = rec[8] = rec[10] = ninf;
        if (!any) return;
        for (int a = 0; a < 3; a++) {
            const float cf = (float)(0.5 * (l[a] + h[a]));
            const double half = std::max(h[a] - (double)cf, (double)cf - l[a]);      // about the rounded centre
            rec[2 * a] = cf;
            rec[6 + 2 * a] = std::nextafter((float)half, std::numeric_limits<float>::infinity());
        }
    };
    for (int t = 0; t < npairs * 2; t++) {
        const bool any = t < ntiles && tbounds[(size_t)t * 5 + 3].w > 0.0f;
        const float4 lo = any ? tbounds[(size_t)t * 5 + 3] : make_float4(0, 0, 0, 0), hi = any ? tbounds[(size_t)t * 5 + 4] : make_float4(0, 0, 0, 0);
        const double l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
        put_box(&tboxes[(size_t)(t / 2) * 12 + (t & 1)], l, h, any);
    }
    auto put_groups = [&](int first_record, int records, int tiles_per_group) {          // and behind those, the groups of 64 that share a mask
        for (int gidx = 0; gidx < records * 2; gidx++) {
            double l[3] = {1e300, 1e300, 1e300}, h[3] = {-1e300, -1e300, -1e300};
            bool any = false;
            for (int t = gidx * tiles_per_group; t < std::min(ntiles, (gidx + 1) * tiles_per_group); t++) {
                const float4 lo = tbounds[(size_t)t * 5 + 3], hi = tbounds[(size_t)t * 5 + 4];
                if (!(lo.w > 0.0f)) continue;
                any = true;
                const double tl[3] = {lo.x, lo.y, lo.z}, th[3] = {hi.x, hi.y, hi.z};
                for (int a = 0; a < 3; a++) { l[a] = std::min(l[a], tl[a]); h[a] = std::max(h[a], th[a]); }
            }
            put_box(&tboxes[(size_t)(first_record + gidx / 2) * 12 + (gidx & 1)], l, h, any);
        }
    };
    put_groups(npairs, ngpairs, kBoxGroupTiles);
    put_groups(npairs + ngpairs, nmpairs, 64);
    for (int k = 0; k < g_nctx; k++) { g_ctx[k].n_pad = 0; g_ctx[k].n = 0; }
    for (int k = 0; k < g_nctx; k++) {
        Ctx &g = g_ctx[k];
        HIP_TRY(hipSetDevice(g.device));
        HIP_TRY(hipStreamSynchronize(g.stream));
        g.ntiles = ntiles;
        g.mat_scale = mat_scale;
        if (ntiles > 0) {
            HIP_TRY(g.aimg.reserve(aimg.size() * sizeof(uint16_t)));
            HIP_TRY(hipMemcpy(g.aimg.p, aimg.data(), aimg.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        }
        HIP_TRY(g.tboxes.reserve(tboxes.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(g.tboxes.p, tboxes.data(), tboxes.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(g.tbounds.reserve(tbounds.size() * sizeof(float4)));
        HIP_TRY(hipMemcpy(g.tbounds.p, tbounds.data(), tbounds.size() * sizeof(float4), hipMemcpyHostToDevice));
        HIP_TRY(g.pairs.reserve(pairs.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(g.pairs.p, pairs.data(), pairs.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(g.orig.reserve(orig.size() * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(g.orig.p, orig.data(), orig.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(g.geom.reserve(geom.size() * sizeof(float4)));
        HIP_TRY(g.matA.reserve(matA.size() * sizeof(float4)));
        HIP_TRY(g.matB.reserve(matB.size() * sizeof(float4)));
        HIP_TRY(hipMemcpy(g.geom.p, geom.data(), geom.size() * sizeof(float4), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(g.matA.p, matA.data(), matA.size() * sizeof(float4), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(g.matB.p, matB.data(), matB.size() * sizeof(float4), hipMemcpyHostToDevice));
        g.nbig = nbig;
        memcpy(g.big, big, sizeof big);
        g.c_near = (float)c_near;
        g.core_div_ok = (n > 0 && sorted[n - 1] <= 16777216.0) ? 1 : 0;          // NaN extents compare false
        // Look up the launch geometry of every build now (code objects load, LDS attributes are set,
        // occupancy is queried), so that no render call after this one has to.
        const int variant = scan_variant(n, n_pad, ntiles);      // the context itself gets its sizes only at the end
        for (int build = 0; build < kBuilds; build++) {
            const size_t lds = trace_lds_bytes(variant, reference_arithmetic(build), n_pad, ntiles);
            for (int mode = 0; mode < 2; mode++) HIP_TRY(prepare_trace(g.launch, build, mode, variant, lds));
            (void)trace_blocks_per_cu(g.launch, build, variant, lds);
        }
    }
    for (int k = 0; k < g_nctx; k++) { g_ctx[k].n = n; g_ctx[k].n_pad = n_pad; }
    return RT_OK;
}

int rt_world_layout(const rt_sphere *spheres, size_t count, int split_tiles, int32_t *order_out)
{
    return guarded("rt_world_layout", [&]() {
        if ((!spheres || !order_out) && count) return fail(RT_ERR_ARG, "rt_world_layout: null argument");
        if (count > (size_t)kMaxSpheres) return fail(RT_ERR_ARG, "rt_world_layout: %zu spheres; at most %d are supported", count, kMaxSpheres);
        const std::vector<int32_t> order = device_order(spheres, (int)count, split_tiles != 0);
        for (size_t i = 0; i < count; i++) order_out[i] = order[i];
        return (int)RT_OK;
    });
}

int rt_set_world(const rt_sphere *spheres, size_t count)
{
    return guarded("rt_set_world", [&]() { return set_world_impl(spheres, count); });
}

int rt_render(const rt_camera *cam, uint8_t *rgba, uint32_t width, uint32_t height, int spp, int max_depth,
              const rt_opts *opts)
{
    return guarded("rt_render", [&]() {
        return render_entry("rt_render", cam, rgba, width, height, spp, max_depth, opts, false, false, -1);
    });
}

int rt_render_device(const rt_camera *cam, uint8_t *rgba_dev, uint32_t width, uint32_t height, int spp,
                     int max_depth, const rt_opts *opts)
{
    return guarded("rt_render_device", [&]() {
        return render_entry("rt_render_device", cam, rgba_dev, width, height, spp, max_depth, opts, true, false, -1);
    });
}

int rt_render_device_noalloc(const rt_camera *cam, uint8_t *rgba_dev, uint32_t width, uint32_t height, int spp,
                             int max_depth, const rt_opts *opts)
{
    // nothing on this path allocates, so nothing on it throws
    return render_entry("rt_render_device_noalloc", cam, rgba_dev, width, height, spp, max_depth, opts, true, true, -1);
}

int rt_render_progressive(const rt_camera *cam, uint8_t *rgba, uint32_t width, uint32_t height, int spp,
                          int max_depth, const rt_opts *opts, uint32_t frame_index)
{
    return guarded("rt_render_progressive", [&]() {
        return render_entry("rt_render_progressive", cam, rgba, width, height, spp, max_depth, opts, false, false, (int64_t)frame_index);
    });
}

uint32_t rt_part_rows(uint32_t height, int tile_rows, int part_index, int part_count)
{
    if (part_count <= 1) return height;
    if (part_index < 0 || part_index >= part_count) return 0;
    const uint32_t tr = tile_rows > 0 ? (uint32_t)tile_rows : 16u;
    const uint32_t ntiles = (height + tr - 1) / tr;
    uint32_t rows = 0;
    for (uint32_t t = (uint32_t)part_index; t < ntiles; t += (uint32_t)part_count) {
        const uint32_t y0 = t * tr;
        rows += (y0 + tr > height) ? height - y0 : tr;
    }
    return rows;
}

uint32_t rt_part_row_at(uint32_t k, int tile_rows, int part_index, int part_count)
{
    return part_row_at(k, tile_rows > 0 ? tile_rows : 16, part_index, part_count > 1 ? part_count : 1);
}

int rt_chunk_for(uint32_t width, uint32_t height, int spp, int chunk_spp, int progressive)
{
    if (width == 0 || height == 0 || spp < 1 || (chunk_spp < 0 && chunk_spp != RT_CHUNK_AUTO)) return 0;
    return resolve_chunk(chunk_spp, width, height, spp, progressive != 0);
}

void rt_camera_defaults(rt_camera_params *p)
{
    if (!p) return;
    p->pos = rt_vec3{13.0f, 2.0f, 3.0f};        // Main.c:915-922
    p->target = rt_vec3{0.0f, 0.0f, 0.0f};
    p->up = rt_vec3{0.0f, 1.0f, 0.0f};
    p->vfov = 60.0f;
    p->defocus_angle = 0.6f;
    p->focus_dist = 10.0f;
}

uint32_t rt_height_for(uint32_t width, float aspect_ratio)
{
    return (uint32_t)((int)width / aspect_ratio);  // Main.c:908
}

int rt_camera_setup(const rt_camera_params *p, uint32_t width, uint32_t height, rt_camera *out)
{
    if (!p || !out || width == 0 || height == 0) return fail(RT_ERR_ARG, "rt_camera_setup: bad argument");
    const V3 pos = V(p->pos), target = V(p->target), up = V(p->up);
    const float theta = deg_to_rad(p->vfov);                                  // Main.c:866
    const float h = tanf(turns_to_rad(theta / 2.0f));                         // Main.c:867, tan_f32 util.h:142
    const float vp_h = 2.0f * h * p->focus_dist;
    const float vp_w = vp_h * ((float)width / (float)height);
    const V3 w = vunit(vsub(pos, target));
    const V3 u = vunit(vcross(up, w));
    const V3 v = vcross(w, u);
    const V3 vp_u = vscale(u, vp_w);
    const V3 vp_v = vscale(v, -vp_h);
    const V3 du = vscale(vp_u, 1.0f / (float)width);
    const V3 dv = vscale(vp_v, 1.0f / (float)height);
    const V3 centre = vsub(pos, vscale(w, p->focus_dist));
    const V3 ul = vsub(centre, vadd(vscale(vp_u, 0.5f), vscale(vp_v, 0.5f)));
    const V3 p00 = vadd(ul, vadd(vscale(du, 0.5f), vscale(dv, 0.5f)));
    const float radius = p->focus_dist * tanf(turns_to_rad(deg_to_rad(p->defocus_angle / 2)));   // Main.c:900
    out->pos = p->pos;
    out->pixel00_loc = R(p00);
    out->pixel_delta_u = R(du);
    out->pixel_delta_v = R(dv);
    out->defocus_disk_u = R(vscale(u, radius));
    out->defocus_disk_v = R(vscale(v, radius));
    out->defocus_angle = p->defocus_angle;
    return RT_OK;
}

int rt_scene_generate(int half_extent, uint32_t lcg_seed, rt_sphere *out, size_t cap, size_t *count)
{
    if (half_extent < 0 || half_extent > 1000 || (!out && cap) || !count) return fail(RT_ERR_ARG, "rt_scene_generate: bad argument");
    uint32_t state = lcg_seed;
    auto rnd = [&state]() -> float {                                   // RAND_FLOAT, util.h:93; fast_rand, util.c:35-39
        state = 214013u * state + 2531011u;
        return (float)(int)((state >> 16) & 0x7FFF) / ((float)32767 + 1.0f);
    };
    size_t n = 0;
    bool overflow = false;
    auto push = [&](const rt_sphere &s) { if (n < cap) out[n] = s; else overflow = true; n++; };
    auto sphere = [](float x, float y, float z, float r, int32_t type, float ar, float ag, float ab, float fuzz, float ior) {
        rt_sphere s;
        s.center = rt_vec3{x, y, z}; s.radius = r; s.mat_type = type; s.albedo = rt_vec3{ar, ag, ab};
        s.fuzz = fuzz; s.refraction_index = ior;
        return s;
    };
    push(sphere(0.0f, -1000.0f, 0.0f, 1000.0f, RT_MAT_LAMBERTIAN, 0.5f, 0.5f, 0.5f, 0.0f, 0.0f));      // Main.c:1216-1224
    for (int a = -half_extent; a < half_extent; a++) {                                                 // Main.c:1229-1285
        for (int b = -half_extent; b < half_extent; b++) {
            const float choose_mat = rnd();
            const float cx = a + 0.9f * rnd();
            const float cz = b + 0.9f * rnd();
            const float dx = cx - 4.0f, dy = 0.2f - 0.2f, dz = cz - 0.0f;
            if (!(sqrtf(dx * dx + dy * dy + dz * dz) > 0.9f)) continue;
            if (choose_mat < 0.8f) {
                const float a0 = rnd(), a1 = rnd(), a2 = rnd();
                const float b0 = rnd(), b1 = rnd(), b2 = rnd();
                push(sphere(cx, 0.2f, cz, 0.2f, RT_MAT_LAMBERTIAN, a0 * b0, a1 * b1, a2 * b2, 0.0f, 0.0f));
            } else if (choose_mat < 0.95f) {
                const float r0 = 0.5f + (1.0f - 0.5f) * rnd();                                         // RAND_FLOAT_RANGE, util.h:94
                const float r1 = 0.5f + (1.0f - 0.5f) * rnd();
                const float r2 = 0.5f + (1.0f - 0.5f) * rnd();
                const float fuzz = 0.0f + (0.5f - 0.0f) * rnd();
                push(sphere(cx, 0.2f, cz, 0.2f, RT_MAT_METAL, r0, r1, r2, fuzz, 0.0f));
            } else {
                push(sphere(cx, 0.2f, cz, 0.2f, RT_MAT_DIELECTRIC, 0.0f, 0.0f, 0.0f, 0.0f, 1.5f));
            }
        }
    }
    push(sphere(0.0f, 1.0f, 0.0f, 1.0f, RT_MAT_DIELECTRIC, 0.0f, 0.0f, 0.0f, 0.0f, 1.5f));             // Main.c:1288-1296
    push(sphere(-4.0f, 1.0f, 0.0f, 1.0f, RT_MAT_LAMBERTIAN, 0.4f, 0.2f, 0.1f, 0.0f, 0.0f));            // Main.c:1298-1306
    push(sphere(4.0f, 1.0f, 0.0f, 1.0f, RT_MAT_METAL, 0.7f, 0.6f, 0.5f, 0.0f, 0.0f));                  // Main.c:1308-1317
    *count = n;
    if (overflow) return fail(RT_ERR_ARG, "rt_scene_generate: %zu objects do not fit %zu records", n, cap);
    return RT_OK;
}

void *rt_host_alloc(size_t bytes)
{
    if (g_nctx == 0) { fail(RT_ERR_STATE, "rt_host_alloc: rt_init has not been called"); return nullptr; }
    DeviceGuard keep;
    void *p = nullptr;
    if (hipSetDevice(g_ctx[0].device) != hipSuccess || hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        fail(RT_ERR_NOMEM, "rt_host_alloc: %zu bytes of page-locked memory are not available", bytes);
        return nullptr;
    }
    return p;
}

void rt_host_free(void *ptr)
{
    if (ptr) (void)hipHostFree(ptr);
}

void rt_shutdown(void)
{
    (void)guarded("rt_shutdown", [&]() -> int {
        DeviceGuard keep;
        g_prog_rows.clear();
        g_prog_mine.clear();
        shutdown_all();
        return RT_OK;
    });
}

#ifdef RT_DIAG
// Diagnostic library only (not declared in rt.h): the device counters of the last render on
// context 0.  [1] rays, [2] pair branches taken, [3] queued candidates, [4] drain iterations,
// [5] wave scans, [7] filter passes, [8] rays whose nearest hit differs from the exact test over
// all spheres.
void rt_diag_counters(unsigned long long *out, int n)
{
    for (int i = 0; i < n && i < kCounters; i++) out[i] = g_ctx[0].last_counters[i];
}
#endif

}  // extern "C"
