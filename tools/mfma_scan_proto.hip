// Prototype of the conservative sphere filter on the matrix cores (bf16 x 3 split, f32 accumulate)
// against the packed-FP32 vector scan the kernel uses now.  Standalone: hipcc --offload-arch=gfx950 -O3.
//   mfma_scan_proto <scene.bin> [reps]
// Checks, per ray, that the matrix filter's candidate set contains every sphere whose exact
// discriminant (double precision, host) is >= 0, compares its size with the vector filter's, and times both.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) f8 const_f8;

struct Ray { float ox, oy, oz, dx, dy, dz; };     // d is unit length

struct Params {
    const Ray *rays;          // [nbatch][64 * waves_total] laid out batch-major
    int nbatch;
    const float *pairs;       // vector scan: [n_pad/2][8]
    const uint4 *aimg;        // matrix scan: [ntiles][2][64] packed bf16 fragments
    const uint4 *limg;        // lifted matrix scan: [ntiles][4][64]
    const uint4 *himg;        // f16 x 2 lifted scan: [ntiles][2][64]
    float scale;              // f16 scan: power-of-two scene scale
    float *dump;              // calibration: accumulator (discriminant, no slack) per (ray of wave 0.., sphere)
    int reps;                 // timing: the batches are swept this many times
    unsigned long long *out_branches;   // taken append branches (checked runs)
    int n_pad, ntiles;
    float c_near;
    float slack_scale;
    unsigned *out_count;      // [nrays] candidates per ray
    unsigned *out_sum;        // [nrays] sum of candidate indices per ray
    unsigned long long *out_total;
};

constexpr int kSlots = 24;

// ---- the vector scan (as in the library's kernel) ---------------------------------------------
template <bool CHECKED>
__global__ __launch_bounds__(256, 7) void scan_vector(const Params p)
{
    extern __shared__ uint32_t lds[];
    const int lane = (int)__lane_id();
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int nw = gridDim.x * (blockDim.x >> 6);
    uint32_t *queue = lds + (threadIdx.x >> 6) * (kSlots * 64);
    unsigned long long total = 0;
    unsigned taken = 0;
    for (int bb = 0; bb < p.nbatch * p.reps; bb++) {
        const int b = bb % p.nbatch;
        const size_t rid = ((size_t)b * nw + wave) * 64 + lane;
        const Ray r = p.rays[rid];
        const float oo = fmaf(r.oz, r.oz, fmaf(r.oy, r.oy, r.ox * r.ox));
        const float reach = __builtin_amdgcn_sqrtf(oo) * 1.0000005f + p.c_near;
        const float slack = fmaf(reach * reach, p.slack_scale, 1e-30f);
        const float base = slack - oo;
        const float ndo = -fmaf(r.dz, r.oz, fmaf(r.dy, r.oy, r.dx * r.ox));
        const float ox2 = r.ox + r.ox, oy2 = r.oy + r.oy, oz2 = r.oz + r.oz;
        const_f8 *gp = (const_f8 *)p.pairs;
        const int npairs = p.n_pad >> 1;
        int cnt = 0;
        unsigned sum = 0, count = 0;
        f8 ahead = gp[0];
        for (int k0 = 0; k0 < npairs; k0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const f8 s = ahead;
                ahead = gp[k0 + u + 1];
                const f2 cx = {s[0], s[1]}, cy = {s[2], s[3]}, cz = {s[4], s[5]};
                f2 h = __builtin_elementwise_fma(cx, (f2)r.dx, (f2)ndo);
                h = __builtin_elementwise_fma(cy, (f2)r.dy, h);
                h = __builtin_elementwise_fma(cz, (f2)r.dz, h);
                f2 m = __builtin_elementwise_fma(cx, (f2)ox2, (f2)base);
                m = __builtin_elementwise_fma(cy, (f2)oy2, m);
                m = __builtin_elementwise_fma(cz, (f2)oz2, m);
                const f2 e = __builtin_elementwise_fma(h, h, m);
                const bool px = e.x >= s[6], py = e.y >= s[7];
                if (__ballot(px || py) != 0ULL) {
                    if (CHECKED) taken++;
                    if (px) { queue[cnt * 64 + lane] = (uint32_t)(2 * (k0 + u)); cnt++; }
                    if (py) { queue[cnt * 64 + lane] = (uint32_t)(2 * (k0 + u) + 1); cnt++; }
                }
            }
            if (__ballot(cnt > kSlots - 8) != 0ULL) {
                for (int j = 0; j < cnt; j++) { sum += queue[j * 64 + lane]; count++; }
                cnt = 0;
            }
        }
        for (int j = 0; j < cnt; j++) { sum += queue[j * 64 + lane]; count++; }
        total += count;
        if (CHECKED) { p.out_count[rid] = count; p.out_sum[rid] = sum; }
    }
    for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off);
    if (lane == 0) { atomicAdd(p.out_total, total); if (CHECKED) atomicAdd(p.out_branches, (unsigned long long)taken); }
}

// ---- the matrix scan ----------------------------------------------------------------------
// Per sphere: features (cx, cy, cz, 1, k = |C|^2 - r^2); per ray: D1 = d.C - d.O, D2 = 2 O.C - k;
// e = D1^2 + D2 >= |O|^2 - slack  <=>  discriminant >= -slack.  Every f32 factor is split into three
// bf16 pieces x = x1 + x2 + x3 (8 significant bits each) and the six piece products of weight
// >= 2^-16 are summed by the MFMA in f32: K entries per feature (c1 p1, c1 p2, c2 p1, c1 p3, c3 p1, c2 p2).
__device__ __forceinline__ uint32_t top16(float x) { return __float_as_uint(x) & 0xffff0000u; }

struct Split { uint32_t p1, p2, p3; };           // bf16 pieces in the upper halves

__device__ __forceinline__ Split split3(float x)
{
    Split s;
    s.p1 = top16(x);
    const float r1 = x - __uint_as_float(s.p1);
    s.p2 = top16(r1);
    const float r2 = r1 - __uint_as_float(s.p2);
    s.p3 = top16(r2);
    return s;
}

__device__ __forceinline__ uint32_t pack(uint32_t lo_piece, uint32_t hi_piece) { return (lo_piece >> 16) | hi_piece; }

__device__ __forceinline__ bf16x8 frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    uint4 v = make_uint4(a, b, c, d);
    return __builtin_bit_cast(bf16x8, v);
}

template <bool CHECKED>
__global__ __launch_bounds__(512, 2) void scan_matrix(const Params p)
{
    extern __shared__ uint32_t lds[];
    const int lane = (int)__lane_id();
    const int wib = threadIdx.x >> 6;
    const int wave = blockIdx.x * (blockDim.x >> 6) + wib;
    const int nw = gridDim.x * (blockDim.x >> 6);
    uint4 *s_a = reinterpret_cast<uint4 *>(lds);                               // [ntiles][2][64]
    for (int i = threadIdx.x; i < p.ntiles * 128; i += blockDim.x) s_a[i] = p.aimg[i];
    uint32_t *queue = lds + p.ntiles * 128 * 4 + wib * (2 * kSlots * 64);        // [2][kSlots][64]
    __syncthreads();
    const int h = lane >> 5;
    unsigned long long total = 0;
    unsigned taken = 0;
    for (int bb = 0; bb < p.nbatch * p.reps; bb++) {
        const int b = bb % p.nbatch;
        const size_t rid = ((size_t)b * nw + wave) * 64 + lane;
        const Ray r = p.rays[rid];
        const float oo = fmaf(r.oz, r.oz, fmaf(r.oy, r.oy, r.ox * r.ox));
        const float reach = __builtin_amdgcn_sqrtf(oo) * 1.0000005f + p.c_near;
        const float slack = fmaf(reach * reach, p.slack_scale, 1e-30f);
        const float thr_own = oo - slack;
        const float ndo = -fmaf(r.dz, r.oz, fmaf(r.dy, r.oy, r.dx * r.ox));
        // own K entries, packed two bf16 per register: groups (s, h) = (0,0) (0,1) (1,0); (1,1) is zero
        const Split dx = split3(r.dx), dy = split3(r.dy), dz = split3(r.dz), nn = split3(ndo);
        const Split ox = split3(r.ox + r.ox), oy = split3(r.oy + r.oy), oz = split3(r.oz + r.oz);
        const uint32_t m1 = 0xbf800000u;            // -1.0f: its upper half is bf16 -1
        uint32_t g1[3][4], g2[3][4];
        g1[0][0] = pack(dx.p1, dx.p2); g1[0][1] = pack(dx.p1, dx.p3); g1[0][2] = g1[0][0]; g1[0][3] = pack(dy.p1, dy.p2);
        g1[1][0] = pack(dy.p1, dy.p3); g1[1][1] = g1[0][3]; g1[1][2] = pack(dz.p1, dz.p2); g1[1][3] = pack(dz.p1, dz.p3);
        g1[2][0] = g1[1][2]; g1[2][1] = pack(nn.p1, nn.p2); g1[2][2] = pack(nn.p3, 0u); g1[2][3] = 0u;
        g2[0][0] = pack(ox.p1, ox.p2); g2[0][1] = pack(ox.p1, ox.p3); g2[0][2] = g2[0][0]; g2[0][3] = pack(oy.p1, oy.p2);
        g2[1][0] = pack(oy.p1, oy.p3); g2[1][1] = g2[0][3]; g2[1][2] = pack(oz.p1, oz.p2); g2[1][3] = pack(oz.p1, oz.p3);
        g2[2][0] = g2[1][2]; g2[2][1] = 0u; g2[2][2] = pack(0u, m1); g2[2][3] = pack(m1, m1);
        // fragments of the two ray halves: half H holds rays 32H .. 32H+31 in its columns; lanes 0-31
        // supply K entries 0-7 of an instruction, lanes 32-63 entries 8-15, of the COLUMN's ray
        uint32_t B1[2][2][4], B2[2][2][4];
#pragma unroll
        for (int v = 0; v < 4; v++) {
            auto s0 = __builtin_amdgcn_permlane32_swap(g1[0][v], g1[1][v], false, false);
            B1[0][0][v] = s0[0]; B1[1][0][v] = s0[1];
            auto s1 = __builtin_amdgcn_permlane32_swap(g1[2][v], 0u, false, false);
            B1[0][1][v] = s1[0]; B1[1][1][v] = s1[1];
            auto t0 = __builtin_amdgcn_permlane32_swap(g2[0][v], g2[1][v], false, false);
            B2[0][0][v] = t0[0]; B2[1][0][v] = t0[1];
            auto t1 = __builtin_amdgcn_permlane32_swap(g2[2][v], 0u, false, false);
            B2[0][1][v] = t1[0]; B2[1][1][v] = t1[1];
        }
        float thr[2];
        {
            auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(thr_own), __float_as_uint(thr_own), false, false);
            thr[0] = __uint_as_float(s[0]);      // lanes < 32: own; lanes >= 32: thr of lane - 32   (column ray of half 0)
            thr[1] = __uint_as_float(s[1]);      // lanes < 32: thr of lane + 32; lanes >= 32: own    (column ray of half 1)
        }
        int cnt[2] = {0, 0};
        unsigned sum = 0, count = 0;
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = 0; t < p.ntiles; t++) {
            const uint4 a0u = s_a[(t * 2 + 0) * 64 + lane], a1u = s_a[(t * 2 + 1) * 64 + lane];
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, a0u), a1 = __builtin_bit_cast(bf16x8, a1u);
#pragma unroll
            for (int H = 0; H < 2; H++) {
                f32x16 d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, frag(B1[H][0][0], B1[H][0][1], B1[H][0][2], B1[H][0][3]), zero, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, frag(B1[H][1][0], B1[H][1][1], B1[H][1][2], B1[H][1][3]), d1, 0, 0, 0);
                f32x16 d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, frag(B2[H][0][0], B2[H][0][1], B2[H][0][2], B2[H][0][3]), zero, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, frag(B2[H][1][0], B2[H][1][1], B2[H][1][2], B2[H][1][3]), d2, 0, 0, 0);
                const float th = thr[H];
                uint32_t *q = queue + H * (kSlots * 64);
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const float e0 = fmaf(d1[i], d1[i], d2[i]), e1 = fmaf(d1[i + 1], d1[i + 1], d2[i + 1]);
                    const bool p0 = e0 >= th, p1 = e1 >= th;
                    if (__ballot(p0 || p1) != 0ULL) {
                        if (CHECKED) taken++;
                        const int row0 = (i & 3) + 8 * (i >> 2) + 4 * h;
                        if (p0) { q[cnt[H] * 64 + lane] = (uint32_t)(t * 32 + row0); cnt[H]++; }
                        if (p1) { q[cnt[H] * 64 + lane] = (uint32_t)(t * 32 + row0 + 1); cnt[H]++; }
                    }
                }
            }
            if (__ballot(cnt[0] > kSlots - 16 || cnt[1] > kSlots - 16) != 0ULL) {
                // drain (prototype: tally only): a lane's own ray collects from its queue h and its partner's queue h
                auto c = __builtin_amdgcn_permlane32_swap((unsigned)cnt[0], (unsigned)cnt[1], false, false);
                const int ca = h ? (int)c[1] : (int)c[0], cb = h ? (int)c[0] : (int)c[1];
                const uint32_t *qa = queue + h * (kSlots * 64);
                for (int j = 0; j < ca; j++) { sum += qa[j * 64 + lane]; count++; }
                for (int j = 0; j < cb; j++) { sum += qa[j * 64 + (lane ^ 32)]; count++; }
                cnt[0] = 0; cnt[1] = 0;
            }
        }
        {
            auto c = __builtin_amdgcn_permlane32_swap((unsigned)cnt[0], (unsigned)cnt[1], false, false);
            const int ca = h ? (int)c[1] : (int)c[0], cb = h ? (int)c[0] : (int)c[1];
            const uint32_t *qa = queue + h * (kSlots * 64);
            for (int j = 0; j < ca; j++) { sum += qa[j * 64 + lane]; count++; }
            for (int j = 0; j < cb; j++) { sum += qa[j * 64 + (lane ^ 32)]; count++; }
        }
        total += count;
        if (CHECKED) { p.out_count[rid] = count; p.out_sum[rid] = sum; }
    }
    for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off);
    if (lane == 0) { atomicAdd(p.out_total, total); if (CHECKED) atomicAdd(p.out_branches, (unsigned long long)taken); }
}

// ---- the lifted matrix scan -------------------------------------------------------------------
// The whole discriminant as ONE inner product: with g = d.O and w = 2 (O - g d),
//   disc + slack = sum_ij (d_i d_j)(c_i c_j) + w.C + (g^2 - |O|^2 + slack) * 1 + (-1) * k.
// Ten ray features against ten sphere features, each split into three bf16 pieces with the six
// products of weight >= 2^-16 kept: K = 9 x 6 + 3 + 3 = 60 <= 64, four MFMAs per half, and the vector
// ALU only compares the accumulator with zero.  The halves are software-pipelined: the MFMAs of the
// next half are issued before the compares of the current one.
constexpr int kLSlots = 24;

template <bool CHECKED>
__device__ __forceinline__ void lifted_process(const f32x16 &acc, int t, int h, int lane, uint16_t *q, int &cnt, unsigned &taken)
{
    // fast path: the largest of the sixteen values, one compare, one branch
    const float m0 = __builtin_fmaxf(__builtin_fmaxf(acc[0], acc[1]), acc[2]);
    const float m1 = __builtin_fmaxf(__builtin_fmaxf(acc[3], acc[4]), acc[5]);
    const float m2 = __builtin_fmaxf(__builtin_fmaxf(acc[6], acc[7]), acc[8]);
    const float m3 = __builtin_fmaxf(__builtin_fmaxf(acc[9], acc[10]), acc[11]);
    const float m4 = __builtin_fmaxf(__builtin_fmaxf(acc[12], acc[13]), acc[14]);
    const float m5 = __builtin_fmaxf(__builtin_fmaxf(m0, m1), acc[15]);
    const float m6 = __builtin_fmaxf(__builtin_fmaxf(m2, m3), m4);
    const float mx = __builtin_fmaxf(m5, m6);
    if (__ballot(mx >= 0.0f) != 0ULL) {
        if (CHECKED) taken++;
        if (mx >= 0.0f) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (acc[i] >= 0.0f) {
                    q[cnt * 64 + lane] = (uint16_t)(t * 32 + (i & 3) + 8 * (i >> 2) + 4 * h);
                    cnt++;
                }
            }
        }
    }
}

template <bool CHECKED, int THREADS>
__global__ __launch_bounds__(THREADS, THREADS / 256) void scan_lifted(const Params p)
{
    extern __shared__ uint32_t lds[];
    const int lane = (int)__lane_id();
    const int wib = threadIdx.x >> 6;
    const int wave = blockIdx.x * (blockDim.x >> 6) + wib;
    const int nw = gridDim.x * (blockDim.x >> 6);
    uint4 *s_a = reinterpret_cast<uint4 *>(lds);                               // [ntiles][4][64]
    for (int i = threadIdx.x; i < p.ntiles * 256; i += blockDim.x) s_a[i] = p.limg[i];
    uint16_t *queue = reinterpret_cast<uint16_t *>(lds + p.ntiles * 256 * 4) + wib * (2 * kLSlots * 64);   // [2][kLSlots][64] u16
    __syncthreads();
    const int h = lane >> 5;
    unsigned long long total = 0;
    unsigned taken = 0;
    for (int bb = 0; bb < p.nbatch * p.reps; bb++) {
        const int b = bb % p.nbatch;
        const size_t rid = ((size_t)b * nw + wave) * 64 + lane;
        const Ray r = p.rays[rid];
        const float oo = fmaf(r.oz, r.oz, fmaf(r.oy, r.oy, r.ox * r.ox));
        const float reach = __builtin_amdgcn_sqrtf(oo) * 1.0000005f + p.c_near;
        const float slack = fmaf(reach * reach, p.slack_scale, 1e-30f);
        const float g = fmaf(r.dz, r.oz, fmaf(r.dy, r.oy, r.dx * r.ox));
        float feat[10];
        feat[0] = r.dx * r.dx; feat[1] = r.dy * r.dy; feat[2] = r.dz * r.dz;
        feat[3] = 2.0f * r.dx * r.dy; feat[4] = 2.0f * r.dx * r.dz; feat[5] = 2.0f * r.dy * r.dz;
        feat[6] = 2.0f * fmaf(-g, r.dx, r.ox); feat[7] = 2.0f * fmaf(-g, r.dy, r.oy); feat[8] = 2.0f * fmaf(-g, r.dz, r.oz);
        feat[9] = fmaf(g, g, ((CHECKED && p.dump) ? 0.0f : slack) - oo);
        const uint32_t m1 = 0xbf800000u;
        uint32_t R[32];
#pragma unroll
        for (int f = 0; f < 9; f++) {
            const Split sp = split3(feat[f]);
            R[3 * f] = pack(sp.p1, sp.p2); R[3 * f + 1] = pack(sp.p1, sp.p3); R[3 * f + 2] = R[3 * f];
        }
        {
            const Split sp = split3(feat[9]);
            R[27] = pack(sp.p1, sp.p2); R[28] = pack(sp.p3, m1); R[29] = pack(m1, m1); R[30] = 0u; R[31] = 0u;
        }
        uint32_t B[2][4][4];
#pragma unroll
        for (int s2 = 0; s2 < 4; s2++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                auto sw = __builtin_amdgcn_permlane32_swap(R[8 * s2 + v], R[8 * s2 + 4 + v], false, false);
                B[0][s2][v] = sw[0]; B[1][s2][v] = sw[1];
            }
        int cnt[2] = {0, 0};
        unsigned sum = 0, count = 0;
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        auto chain = [&](const bf16x8 (&a)[4], int H) {
            f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], frag(B[H][0][0], B[H][0][1], B[H][0][2], B[H][0][3]), zero, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], frag(B[H][1][0], B[H][1][1], B[H][1][2], B[H][1][3]), d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], frag(B[H][2][0], B[H][2][1], B[H][2][2], B[H][2][3]), d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[3], frag(B[H][3][0], B[H][3][1], B[H][3][2], B[H][3][3]), d, 0, 0, 0);
            return d;
        };
        bf16x8 a[4], an[4];
#pragma unroll
        for (int s2 = 0; s2 < 4; s2++) a[s2] = __builtin_bit_cast(bf16x8, s_a[s2 * 64 + lane]);
        f32x16 acc0 = chain(a, 0);
        for (int t = 0; t < p.ntiles; t++) {
            f32x16 acc1 = chain(a, 1);
            const int tn = t + 1 < p.ntiles ? t + 1 : t;
#pragma unroll
            for (int s2 = 0; s2 < 4; s2++) an[s2] = __builtin_bit_cast(bf16x8, s_a[(tn * 4 + s2) * 64 + lane]);
            if (CHECKED && p.dump) {
                for (int i = 0; i < 16; i++) {
                    const int row = t * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    p.dump[((size_t)wave * 64 + (lane & 31)) * (p.ntiles * 32) + row] = acc0[i];
                    p.dump[((size_t)wave * 64 + 32 + (lane & 31)) * (p.ntiles * 32) + row] = acc1[i];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            lifted_process<CHECKED>(acc0, t, h, lane, queue, cnt[0], taken);
            __builtin_amdgcn_sched_barrier(0);
            acc0 = chain(an, 0);
            __builtin_amdgcn_sched_barrier(0);
            lifted_process<CHECKED>(acc1, t, h, lane, queue + kLSlots * 64, cnt[1], taken);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s2 = 0; s2 < 4; s2++) a[s2] = an[s2];
            if (__ballot(cnt[0] > kLSlots - 16 || cnt[1] > kLSlots - 16) != 0ULL || t + 1 == p.ntiles) {
                auto c = __builtin_amdgcn_permlane32_swap((unsigned)cnt[0], (unsigned)cnt[1], false, false);
                const int ca = h ? (int)c[1] : (int)c[0], cb = h ? (int)c[0] : (int)c[1];
                const uint16_t *qa = queue + h * (kLSlots * 64);
                for (int j = 0; j < ca; j++) { sum += qa[j * 64 + lane]; count++; }
                for (int j = 0; j < cb; j++) { sum += qa[j * 64 + (lane ^ 32)]; count++; }
                cnt[0] = 0; cnt[1] = 0;
            }
        }
        total += count;
        if (CHECKED) { p.out_count[rid] = count; p.out_sum[rid] = sum; }
    }
    for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off);
    if (lane == 0) { atomicAdd(p.out_total, total); if (CHECKED) atomicAdd(p.out_branches, (unsigned long long)taken); }
}

// ---- the lifted scan with f16 x 2 pieces ----------------------------------------------------------
// Same inner product in a frame scaled by a power of two s (spheres within 128 of the origin): every
// feature x = x1 + x2 (+ 2^-24 |x|) as two f16 pieces, products x1 y1 + x1 y2 + x2 y1 kept:
// K = 9 x 3 + 2 + 2 = 31 of 32, two v_mfma_f32_32x32x16_f16 per half.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t packh(_Float16 lo, _Float16 hi) { const h2 v = {lo, hi}; return __builtin_bit_cast(uint32_t, v); }

__device__ __forceinline__ h8 hfrag(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    uint4 v = make_uint4(a, b, c, d);
    return __builtin_bit_cast(h8, v);
}

template <bool CHECKED, int THREADS, bool DUMP>
__global__ __launch_bounds__(THREADS, THREADS / 256) void scan_f16(const Params p)
{
    extern __shared__ uint32_t lds[];
    const int lane = (int)__lane_id();
    const int wib = threadIdx.x >> 6;
    const int wave = blockIdx.x * (blockDim.x >> 6) + wib;
    const int nw = gridDim.x * (blockDim.x >> 6);
    uint4 *s_a = reinterpret_cast<uint4 *>(lds);                               // [ntiles][2][64]
    for (int i = threadIdx.x; i < p.ntiles * 128; i += blockDim.x) s_a[i] = p.himg[i];
    uint16_t *queue = reinterpret_cast<uint16_t *>(lds + p.ntiles * 128 * 4) + wib * (2 * kLSlots * 64);
    __syncthreads();
    const int h = lane >> 5;
    unsigned long long total = 0;
    unsigned taken = 0;
    for (int bb = 0; bb < p.nbatch * p.reps; bb++) {
        const int b = bb % p.nbatch;
        const size_t rid = ((size_t)b * nw + wave) * 64 + lane;
        const Ray r = p.rays[rid];
        const float ox = r.ox * p.scale, oy = r.oy * p.scale, oz = r.oz * p.scale;
        const float oo = fmaf(oz, oz, fmaf(oy, oy, ox * ox));
        const float reach = __builtin_amdgcn_sqrtf(oo) * 1.0000005f + p.c_near * p.scale;
        const float slack = DUMP ? 0.0f : fmaf(reach * reach, p.slack_scale, 0.004f);
        const float g = fmaf(r.dz, oz, fmaf(r.dy, oy, r.dx * ox));
        float feat[10];
        feat[0] = r.dx * r.dx; feat[1] = r.dy * r.dy; feat[2] = r.dz * r.dz;
        feat[3] = 2.0f * r.dx * r.dy; feat[4] = 2.0f * r.dx * r.dz; feat[5] = 2.0f * r.dy * r.dz;
        feat[6] = 2.0f * fmaf(-g, r.dx, ox); feat[7] = 2.0f * fmaf(-g, r.dy, oy); feat[8] = 2.0f * fmaf(-g, r.dz, oz);
        feat[9] = fmaf(g, g, slack - oo);
        _Float16 x1[10], x2[10];
#pragma unroll
        for (int f = 0; f < 10; f++) {
            asm volatile("" : "+v"(feat[f]));      // no fused multiply-convert: x1 must be the rounding of the f32 feature
            x1[f] = (_Float16)feat[f];
            x2[f] = (_Float16)(feat[f] - (float)x1[f]);
        }
        const _Float16 m1 = (_Float16)-1.0f, z0 = (_Float16)0.0f;
        uint32_t R[16];
#pragma unroll
        for (int k = 0; k < 4; k++) {                 // features 2k, 2k+1 fill three registers
            R[3 * k] = packh(x1[2 * k], x2[2 * k]);
            R[3 * k + 1] = packh(x1[2 * k], x1[2 * k + 1]);
            R[3 * k + 2] = packh(x2[2 * k + 1], x1[2 * k + 1]);
        }
        R[12] = packh(x1[8], x2[8]);
        R[13] = packh(x1[8], x1[9]);
        R[14] = packh(x2[9], m1);
        R[15] = packh(m1, z0);
        uint32_t B[2][2][4];
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                auto sw = __builtin_amdgcn_permlane32_swap(R[8 * s2 + v], R[8 * s2 + 4 + v], false, false);
                B[0][s2][v] = sw[0]; B[1][s2][v] = sw[1];
            }
        int cnt[2] = {0, 0};
        unsigned sum = 0, count = 0;
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        auto chain = [&](const h8 (&a)[2], int H) {
            f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], hfrag(B[H][0][0], B[H][0][1], B[H][0][2], B[H][0][3]), zero, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], hfrag(B[H][1][0], B[H][1][1], B[H][1][2], B[H][1][3]), d, 0, 0, 0);
            return d;
        };
        h8 a[2], an[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) a[s2] = __builtin_bit_cast(h8, s_a[s2 * 64 + lane]);
        f32x16 acc0 = chain(a, 0);
        for (int t = 0; t < p.ntiles; t++) {
            f32x16 acc1 = chain(a, 1);
            const int tn = t + 1 < p.ntiles ? t + 1 : t;
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) an[s2] = __builtin_bit_cast(h8, s_a[(tn * 2 + s2) * 64 + lane]);
            if (DUMP) {
                // column = ray 32 H + (lane & 31) of this wave; row = sphere 32 t + (i & 3) + 8 (i >> 2) + 4 h
                for (int i = 0; i < 16; i++) {
                    const int row = t * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    p.dump[((size_t)wave * 64 + (lane & 31)) * (p.ntiles * 32) + row] = acc0[i];
                    p.dump[((size_t)wave * 64 + 32 + (lane & 31)) * (p.ntiles * 32) + row] = acc1[i];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            lifted_process<CHECKED>(acc0, t, h, lane, queue, cnt[0], taken);
            __builtin_amdgcn_sched_barrier(0);
            acc0 = chain(an, 0);
            __builtin_amdgcn_sched_barrier(0);
            lifted_process<CHECKED>(acc1, t, h, lane, queue + kLSlots * 64, cnt[1], taken);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) a[s2] = an[s2];
            if (__ballot(cnt[0] > kLSlots - 16 || cnt[1] > kLSlots - 16) != 0ULL || t + 1 == p.ntiles) {
                auto c = __builtin_amdgcn_permlane32_swap((unsigned)cnt[0], (unsigned)cnt[1], false, false);
                const int ca = h ? (int)c[1] : (int)c[0], cb = h ? (int)c[0] : (int)c[1];
                const uint16_t *qa = queue + h * (kLSlots * 64);
                for (int j = 0; j < ca; j++) { sum += qa[j * 64 + lane]; count++; }
                for (int j = 0; j < cb; j++) { sum += qa[j * 64 + (lane ^ 32)]; count++; }
                cnt[0] = 0; cnt[1] = 0;
            }
        }
        total += count;
        if (CHECKED) { p.out_count[rid] = count; p.out_sum[rid] = sum; }
    }
    for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off);
    if (lane == 0) { atomicAdd(p.out_total, total); if (CHECKED) atomicAdd(p.out_branches, (unsigned long long)taken); }
}

// ---- host -----------------------------------------------------------------------------------
struct Sphere { float c[3], r; int mat; float alb[3], fuzz, ior; };

static uint16_t bf16_trunc(float x) { uint32_t u; memcpy(&u, &x, 4); return (uint16_t)(u >> 16); }
static float bf16_val(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static void split3h(float x, uint16_t out[3])
{
    out[0] = bf16_trunc(x); float r1 = x - bf16_val(out[0]);
    out[1] = bf16_trunc(r1); float r2 = r1 - bf16_val(out[1]);
    out[2] = bf16_trunc(r2);
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s scene.bin [nbatch]\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    std::vector<Sphere> sph;
    Sphere s;
    while (fread(&s, sizeof s, 1, f) == 1) sph.push_back(s);
    fclose(f);
    const int n = (int)sph.size();
    const int nbatch = argc > 2 ? atoi(argv[2]) : 8;
    printf("%d spheres\n", n);

    // big spheres (excluded from the filter): extent > 4 x median
    std::vector<double> ext(n);
    for (int i = 0; i < n; i++) ext[i] = sqrt((double)sph[i].c[0] * sph[i].c[0] + (double)sph[i].c[1] * sph[i].c[1] + (double)sph[i].c[2] * sph[i].c[2]) + sph[i].r;
    std::vector<double> so = ext; std::sort(so.begin(), so.end());
    const double big_from = 4.0 * so[n / 2];
    std::vector<char> big(n, 0);
    double c_near = 0;
    for (int i = 0; i < n; i++) { big[i] = ext[i] > big_from; if (!big[i]) c_near = std::max(c_near, ext[i] - sph[i].r); }

    const int n_pad = (n + 7) / 8 * 8;
    std::vector<float> pairs((size_t)n_pad * 4 + 8, 0.0f);
    for (int i = 0; i < n_pad; i++) {
        float *q = &pairs[(size_t)(i / 2) * 8 + (i & 1)];
        if (i < n && !big[i]) {
            q[0] = sph[i].c[0]; q[2] = sph[i].c[1]; q[4] = sph[i].c[2];
            q[6] = (float)((double)sph[i].c[0] * sph[i].c[0] + (double)sph[i].c[1] * sph[i].c[1] + (double)sph[i].c[2] * sph[i].c[2] - (double)sph[i].r * sph[i].r);
        } else { q[6] = INFINITY; }
    }
    // matrix image: [tile][s][lane] 8 bf16: lane l: sphere tile*32 + (l & 31), K entries 16 s + 8 (l >> 5) + j
    const int ntiles = (n + 31) / 32;
    std::vector<uint16_t> aimg((size_t)ntiles * 2 * 64 * 8, 0);
    for (int t = 0; t < ntiles; t++)
        for (int row = 0; row < 32; row++) {
            const int i = t * 32 + row;
            uint16_t ent[32];
            memset(ent, 0, sizeof ent);
            if (i < n && !big[i]) {
                for (int c = 0; c < 3; c++) {
                    uint16_t p[3];
                    split3h(sph[i].c[c], p);
                    const uint16_t six[6] = {p[0], p[0], p[1], p[0], p[2], p[1]};
                    for (int k = 0; k < 6; k++) ent[6 * c + k] = six[k];
                }
                ent[18] = ent[19] = ent[20] = bf16_trunc(1.0f);
                const float k = (float)((double)sph[i].c[0] * sph[i].c[0] + (double)sph[i].c[1] * sph[i].c[1] + (double)sph[i].c[2] * sph[i].c[2] - (double)sph[i].r * sph[i].r);
                uint16_t kp[3];
                split3h(k, kp);
                ent[21] = kp[0]; ent[22] = kp[1]; ent[23] = kp[2];
            } else {
                ent[21] = bf16_trunc(3.0e38f);       // never passes: D2 = -3e38
            }
            for (int s2 = 0; s2 < 2; s2++)
                for (int hh = 0; hh < 2; hh++)
                    for (int j = 0; j < 8; j++)
                        aimg[(((size_t)t * 2 + s2) * 64 + hh * 32 + row) * 8 + j] = ent[16 * s2 + 8 * hh + j];
        }

    // lifted image: [tile][s][lane] 8 bf16; 60 K entries: 9 features x 6, then 1,1,1, then k1,k2,k3
    std::vector<uint16_t> limg((size_t)ntiles * 4 * 64 * 8, 0);
    for (int t = 0; t < ntiles; t++)
        for (int row = 0; row < 32; row++) {
            const int i = t * 32 + row;
            uint16_t ent[64];
            memset(ent, 0, sizeof ent);
            if (i < n && !big[i]) {
                const double cx = sph[i].c[0], cy = sph[i].c[1], cz = sph[i].c[2];
                const float feat[9] = {(float)(cx * cx), (float)(cy * cy), (float)(cz * cz), (float)(cx * cy), (float)(cx * cz), (float)(cy * cz),
                                       (float)cx, (float)cy, (float)cz};
                for (int c = 0; c < 9; c++) {
                    uint16_t pc[3];
                    split3h(feat[c], pc);
                    const uint16_t six[6] = {pc[0], pc[0], pc[1], pc[0], pc[2], pc[1]};
                    for (int k = 0; k < 6; k++) ent[6 * c + k] = six[k];
                }
                ent[54] = ent[55] = ent[56] = bf16_trunc(1.0f);
                const float k = (float)(cx * cx + cy * cy + cz * cz - (double)sph[i].r * sph[i].r);
                uint16_t kp[3];
                split3h(k, kp);
                ent[57] = kp[0]; ent[58] = kp[1]; ent[59] = kp[2];
            } else {
                ent[57] = bf16_trunc(3.0e38f);
            }
            for (int s2 = 0; s2 < 4; s2++)
                for (int hh = 0; hh < 2; hh++)
                    for (int j = 0; j < 8; j++)
                        limg[(((size_t)t * 4 + s2) * 64 + hh * 32 + row) * 8 + j] = ent[16 * s2 + 8 * hh + j];
        }

    // rays: half camera rays, half rays leaving small spheres
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int waves_v = cus * 28, waves_m = cus * 8;
    const int nw = std::max(waves_v, waves_m);
    const size_t nrays = (size_t)nbatch * nw * 64;
    std::vector<Ray> rays(nrays);
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
    // f16 image: scene scaled by a power of two so that every filtered sphere lies within 128 of the origin
    double max_ext = 0;
    for (int i = 0; i < n; i++) if (!big[i]) max_ext = std::max(max_ext, ext[i]);
    const float scale = max_ext > 0 ? (float)std::exp2(std::floor(std::log2(128.0 / max_ext))) : 1.0f;
    printf("scene scale %g (largest filtered extent %.2f)\n", scale, max_ext);
    std::vector<uint16_t> himg((size_t)ntiles * 2 * 64 * 8, 0);
    auto h16 = [](float x) { _Float16 hx = (_Float16)x; uint16_t u; memcpy(&u, &hx, 2); return u; };
    auto h16v = [](float x) { return (float)(_Float16)x; };
    for (int t = 0; t < ntiles; t++)
        for (int row = 0; row < 32; row++) {
            const int i = t * 32 + row;
            uint16_t ent[32];
            memset(ent, 0, sizeof ent);
            if (i < n && !big[i]) {
                const double cx = (double)sph[i].c[0] * scale, cy = (double)sph[i].c[1] * scale, cz = (double)sph[i].c[2] * scale, rr = (double)sph[i].r * scale;
                const float feat[9] = {(float)(cx * cx), (float)(cy * cy), (float)(cz * cz), (float)(cx * cy), (float)(cx * cz), (float)(cy * cz),
                                       (float)cx, (float)cy, (float)cz};
                for (int c = 0; c < 9; c++) {
                    const uint16_t c1 = h16(feat[c]), c2 = h16(feat[c] - h16v(feat[c]));
                    ent[3 * c] = c1; ent[3 * c + 1] = c1; ent[3 * c + 2] = c2;
                }
                ent[27] = ent[28] = h16(1.0f);
                const float k = (float)(cx * cx + cy * cy + cz * cz - rr * rr);
                ent[29] = h16(k); ent[30] = h16(k - h16v(k));
            } else {
                ent[29] = h16(60000.0f);             // never passes: the discriminant is -60000
            }
            for (int s2 = 0; s2 < 2; s2++)
                for (int hh = 0; hh < 2; hh++)
                    for (int j = 0; j < 8; j++)
                        himg[(((size_t)t * 2 + s2) * 64 + hh * 32 + row) * 8 + j] = ent[16 * s2 + 8 * hh + j];
        }

    // Rays as the renderer's waves hold them: the 64 lanes of a wave own adjacent pixels.  A wave's rays
    // are either camera rays through 64 neighbouring pixels, or bounce rays whose origins are the points
    // those camera rays would hit (close together) with independent random directions.
    auto unit3 = [](float *d) { float l = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]); if (!(l > 1e-6f)) { d[0] = 0; d[1] = 1; d[2] = 0; l = 1; } d[0] /= l; d[1] /= l; d[2] /= l; };
    const float px = 0.00138f;                       // pixel pitch at focus distance 10 of the 3840-wide view
    for (size_t w0 = 0; w0 < nrays; w0 += 64) {
        const float ax = -9.0f + 16.0f * U(rng), az = -5.0f + 10.0f * U(rng), ay = 0.4f * U(rng);     // aim point near the ground
        const bool primary = (rng() % 10) < 3;
        float base_d[3] = {ax - 13.0f, ay - 2.0f, az - 3.0f};
        const float dist = sqrtf(base_d[0] * base_d[0] + base_d[1] * base_d[1] + base_d[2] * base_d[2]);
        for (int l = 0; l < 64; l++) {
            Ray r;
            float d[3];
            if (primary) {
                r.ox = 13.0f + 0.3f * (U(rng) - 0.5f); r.oy = 2.0f + 0.3f * (U(rng) - 0.5f); r.oz = 3.0f + 0.3f * (U(rng) - 0.5f);   // defocus disk
                const float tx = ax + px * dist / 10.0f * l * 0.3f, tz = az - px * dist / 10.0f * l;
                d[0] = tx - r.ox; d[1] = ay - r.oy; d[2] = tz - r.oz;
            } else {
                r.ox = ax + px * dist / 10.0f * l * 0.3f + 0.02f * (U(rng) - 0.5f); r.oy = ay; r.oz = az - px * dist / 10.0f * l + 0.02f * (U(rng) - 0.5f);
                const float z = U(rng), ph = 6.2831853f * U(rng), rad = sqrtf(std::max(0.0f, 1 - z * z));
                d[0] = rad * cosf(ph); d[1] = z + 0.05f; d[2] = rad * sinf(ph);
            }
            unit3(d);
            r.dx = d[0]; r.dy = d[1]; r.dz = d[2];
            rays[w0 + l] = r;
        }
    }

    Params p;
    memset(&p, 0, sizeof p);
    uint4 *d_himg; float *d_dump;
    Ray *d_rays; float *d_pairs; uint4 *d_aimg, *d_limg; unsigned *d_cnt, *d_sum; unsigned long long *d_total, *d_br;
    CHECK(hipMalloc(&d_rays, nrays * sizeof(Ray)));
    CHECK(hipMalloc(&d_pairs, pairs.size() * 4));
    CHECK(hipMalloc(&d_aimg, aimg.size() * 2));
    CHECK(hipMalloc(&d_cnt, nrays * 4));
    CHECK(hipMalloc(&d_sum, nrays * 4));
    CHECK(hipMalloc(&d_total, 8));
    CHECK(hipMalloc(&d_br, 8));
    CHECK(hipMalloc(&d_himg, himg.size() * 2));
    CHECK(hipMemcpy(d_himg, himg.data(), himg.size() * 2, hipMemcpyHostToDevice));
    const int dump_waves = 256;
    CHECK(hipMalloc(&d_dump, (size_t)dump_waves * 64 * ntiles * 32 * 4));
    CHECK(hipMalloc(&d_limg, limg.size() * 2));
    CHECK(hipMemcpy(d_limg, limg.data(), limg.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_rays, rays.data(), nrays * sizeof(Ray), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_pairs, pairs.data(), pairs.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_aimg, aimg.data(), aimg.size() * 2, hipMemcpyHostToDevice));
    p.rays = d_rays; p.nbatch = nbatch; p.pairs = d_pairs; p.aimg = d_aimg; p.n_pad = n_pad; p.ntiles = ntiles;
    p.c_near = (float)c_near; p.out_count = d_cnt; p.out_sum = d_sum; p.out_total = d_total; p.out_branches = d_br; p.limg = d_limg; p.reps = 1; p.himg = d_himg; p.scale = scale; p.dump = nullptr;

    const size_t lds_v = 4 * kSlots * 64 * 4;
    const size_t lds_m = (size_t)ntiles * 128 * 16 + 8 * 2 * kSlots * 64 * 4;
    CHECK(hipFuncSetAttribute((const void *)scan_matrix<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m));
    CHECK(hipFuncSetAttribute((const void *)scan_matrix<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m));
    const size_t lds_l = (size_t)ntiles * 256 * 16 + 8 * 2 * kLSlots * 64 * 2;
    const size_t lds_l4 = (size_t)ntiles * 256 * 16 + 16 * 2 * kLSlots * 64 * 2;
    CHECK(hipFuncSetAttribute((const void *)scan_lifted<true, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_l));
    CHECK(hipFuncSetAttribute((const void *)scan_lifted<false, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_l));
    CHECK(hipFuncSetAttribute((const void *)scan_lifted<false, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_l4));
    const size_t lds_h = (size_t)ntiles * 128 * 16 + 8 * 2 * kLSlots * 64 * 2;
    const size_t lds_h4 = (size_t)ntiles * 128 * 16 + 16 * 2 * kLSlots * 64 * 2;
    CHECK(hipFuncSetAttribute((const void *)scan_f16<true, 512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h));
    CHECK(hipFuncSetAttribute((const void *)scan_f16<true, 512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h));
    CHECK(hipFuncSetAttribute((const void *)scan_f16<false, 512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h));
    CHECK(hipFuncSetAttribute((const void *)scan_f16<false, 1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h4));

    // optional: argv[4] = "ox,oy,oz,dx,dy,dz" replaces ray 0 (direction is normalised here) and its accumulators are printed
    const bool probe = argc > 4;
    if (probe) {
        float v[6];
        if (sscanf(argv[4], "%f,%f,%f,%f,%f,%f", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5]) != 6) { fprintf(stderr, "bad ray\n"); return 2; }
        const float a = v[3] * v[3] + v[4] * v[4] + v[5] * v[5];
        const float inv = 1.0f / sqrtf(a);
        for (int l = 0; l < 64; l++) rays[l] = Ray{v[0], v[1], v[2], v[3] * inv, v[4] * inv, v[5] * inv};
        CHECK(hipMemcpy(d_rays, rays.data(), 64 * sizeof(Ray), hipMemcpyHostToDevice));
    }
    // ---- calibration: the accumulator against the discriminant in double precision, in units of (|O| + max|C|)^2
    for (int which = 0; which < 2; which++) {
        Params q = p; q.nbatch = 1; q.reps = 1; q.dump = d_dump; q.slack_scale = 0.0f;
        CHECK(hipMemset(d_dump, 0, (size_t)dump_waves * 64 * ntiles * 32 * 4));
        if (which == 0) hipLaunchKernelGGL((scan_lifted<true, 512>), dim3(dump_waves / 8), dim3(512), lds_l, 0, q);
        else hipLaunchKernelGGL((scan_f16<true, 512, true>), dim3(dump_waves / 8), dim3(512), lds_h, 0, q);
        CHECK(hipDeviceSynchronize());
        std::vector<float> dump((size_t)dump_waves * 64 * ntiles * 32);
        CHECK(hipMemcpy(dump.data(), d_dump, dump.size() * 4, hipMemcpyDeviceToHost));
        // NOTE: with dump_waves < waves the ray index of wave w, lane l in batch 0 is w * 64 + l only for the
        // launch geometry used here (nw = dump_waves)
        double worst = 0, worst_abs = 0, worst_acc = 0, worst_disc = 0;
        int worst_r = 0, worst_k = 0;
        const double sc = which ? (double)scale : 1.0;
        for (int rI = 0; rI < dump_waves * 64; rI++) {
            const Ray &r = rays[rI];
            const double on = sqrt((double)r.ox * r.ox + (double)r.oy * r.oy + (double)r.oz * r.oz) * sc;
            const double S = on + c_near * sc;
            for (int k = 0; k < n; k++) {
                if (big[k]) continue;
                double ocx = ((double)sph[k].c[0] - r.ox) * sc, ocy = ((double)sph[k].c[1] - r.oy) * sc, ocz = ((double)sph[k].c[2] - r.oz) * sc;
                double hh = r.dx * ocx + r.dy * ocy + r.dz * ocz;
                double dd = (double)r.dx * r.dx + (double)r.dy * r.dy + (double)r.dz * r.dz;      // d is unit to f32 rounding only
                double disc = hh * hh - dd * (ocx * ocx + ocy * ocy + ocz * ocz) + (ocx * ocx + ocy * ocy + ocz * ocz) * (dd - 1.0)
                              - ((ocx * ocx + ocy * ocy + ocz * ocz) - (double)sph[k].r * sph[k].r * sc * sc) + (ocx * ocx + ocy * ocy + ocz * ocz);
                // the kernel's formula assumes |d| = 1: disc = h^2 - (|oc|^2 - r^2)
                disc = hh * hh - (ocx * ocx + ocy * ocy + ocz * ocz - (double)sph[k].r * sph[k].r * sc * sc);
                const double err = fabs((double)dump[(size_t)rI * (ntiles * 32) + k] - disc);
                if (probe && which == 1 && (rI == 0 || rI == 37) && (k % 16 == 2 || err > 0.05)) printf("  ray %d sphere %d: accumulator %.6f discriminant %.6f\n", rI, k, dump[(size_t)rI * (ntiles * 32) + k], disc);
                worst = std::max(worst, err / (S * S));
                if (err > worst_abs) { worst_r = rI; worst_k = k; worst_acc = dump[(size_t)rI * (ntiles * 32) + k]; worst_disc = disc; }
                worst_abs = std::max(worst_abs, err);
            }
        }
        printf("calibration %s: max |accumulator - discriminant| = %.3g = 2^%.2f x (|O| + max|C|)^2 over %d rays x %d spheres\n",
               which ? "f16 x 2" : "bf16 x 3", worst_abs, log2(worst), dump_waves * 64, n);
        printf("  worst: ray %d o %.9g %.9g %.9g d %.9g %.9g %.9g sphere %d c %.9g %.9g %.9g r %.9g accumulator %.6f discriminant %.6f\n", worst_r,
               rays[worst_r].ox, rays[worst_r].oy, rays[worst_r].oz, rays[worst_r].dx, rays[worst_r].dy, rays[worst_r].dz, worst_k,
               sph[worst_k].c[0], sph[worst_k].c[1], sph[worst_k].c[2], sph[worst_k].r, worst_acc, worst_disc);
    }

    // ---- check: candidate sets
    const size_t ncheck = (size_t)waves_m * 64;           // first batch, at the matrix kernels' wave count
    std::vector<unsigned> cnts[4], sums[4];
    unsigned long long branches[4] = {0, 0, 0, 0};
    const char *names[4] = {"vector", "matrix", "lifted", "f16 x 2"};
    const float slacks[4] = {1.9073486e-6f, 3.8146973e-6f, 3.8146973e-6f, 3.8146973e-6f};     // 2^-19, 2^-18, 2^-18, 2^-18
    for (int which = 0; which < 4; which++) {
        cnts[which].resize(ncheck); sums[which].resize(ncheck);
        CHECK(hipMemset(d_cnt, 0xff, nrays * 4));
        CHECK(hipMemset(d_total, 0, 8));
        CHECK(hipMemset(d_br, 0, 8));
        Params q = p; q.nbatch = 1; q.reps = 1; q.slack_scale = slacks[which];
        if (which == 0) hipLaunchKernelGGL(scan_vector<true>, dim3(waves_m / 4), dim3(256), lds_v, 0, q);
        else if (which == 1) hipLaunchKernelGGL(scan_matrix<true>, dim3(waves_m / 8), dim3(512), lds_m, 0, q);
        else if (which == 2) hipLaunchKernelGGL((scan_lifted<true, 512>), dim3(waves_m / 8), dim3(512), lds_l, 0, q);
        else hipLaunchKernelGGL((scan_f16<true, 512, false>), dim3(waves_m / 8), dim3(512), lds_h, 0, q);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(cnts[which].data(), d_cnt, ncheck * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(sums[which].data(), d_sum, ncheck * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(&branches[which], d_br, 8, hipMemcpyDeviceToHost));
    }
    size_t exact_total = 0, tot[4] = {0, 0, 0, 0}, missing[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < ncheck; i++) {
        const Ray &r = rays[i];
        unsigned ec = 0, es = 0;
        for (int k = 0; k < n; k++) {
            if (big[k]) continue;
            double ocx = (double)sph[k].c[0] - r.ox, ocy = (double)sph[k].c[1] - r.oy, ocz = (double)sph[k].c[2] - r.oz;
            double hh = r.dx * ocx + r.dy * ocy + r.dz * ocz;
            double disc = hh * hh - (ocx * ocx + ocy * ocy + ocz * ocz - (double)sph[k].r * sph[k].r);
            if (disc >= 0) { ec++; es += k; }
        }
        exact_total += ec;
        // a superset with equal size and equal index sum is the same set; a smaller set lost a sphere
        for (int w = 0; w < 4; w++) {
            tot[w] += cnts[w][i];
            if (cnts[w][i] < ec || (cnts[w][i] == ec && sums[w][i] != es)) missing[w]++;
        }
    }
    printf("check over %zu rays: exact candidates %.3f per ray\n", ncheck, (double)exact_total / ncheck);
    for (int w = 0; w < 4; w++)
        printf("  %s filter: %.3f candidates per ray, rays with a lost sphere: %zu, append branches taken per wave scan: %.1f\n",
               names[w], (double)tot[w] / ncheck, missing[w], (double)branches[w] / (double)waves_m);

    // ---- timing
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    if (probe) return 0;
    const int reps = argc > 3 ? atoi(argv[3]) : 10;
    const char *tnames[6] = {"vector", "matrix", "lifted 2 waves/SIMD", "lifted 4 waves/SIMD", "f16 x 2, 2 waves/SIMD", "f16 x 2, 4 waves/SIMD"};
    for (int which = 0; which < 6; which++) {
        Params q = p;
        q.slack_scale = 3.8146973e-6f;
        if (which == 0) q.slack_scale = 1.9073486e-6f;
        q.reps = reps;
        const int waves = which == 0 ? waves_v : ((which == 3 || which == 5) ? 2 * waves_m : waves_m);
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            if (which == 0) hipLaunchKernelGGL(scan_vector<false>, dim3(waves / 4), dim3(256), lds_v, 0, q);
            else if (which == 1) hipLaunchKernelGGL(scan_matrix<false>, dim3(waves / 8), dim3(512), lds_m, 0, q);
            else if (which == 2) hipLaunchKernelGGL((scan_lifted<false, 512>), dim3(waves / 8), dim3(512), lds_l, 0, q);
            else if (which == 3) hipLaunchKernelGGL((scan_lifted<false, 1024>), dim3(waves / 16), dim3(1024), lds_l4, 0, q);
            else if (which == 4) hipLaunchKernelGGL((scan_f16<false, 512, false>), dim3(waves / 8), dim3(512), lds_h, 0, q);
            else hipLaunchKernelGGL((scan_f16<false, 1024, false>), dim3(waves / 16), dim3(1024), lds_h4, 0, q);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double tests = (double)nbatch * reps * waves * 64 * n;
            printf("%s scan: %d waves x %d batches x %d: %.3f ms, %.2f G tests/s, algorithmic %.1f TFLOP/s = %.3f of the vector peak\n",
                   tnames[which], waves, nbatch, reps, ms, tests / ms / 1e6, 18.0 * tests / ms / 1e9, 18.0 * tests / ms / 1e9 / 157.3);
        }
    }
    return 0;
}
