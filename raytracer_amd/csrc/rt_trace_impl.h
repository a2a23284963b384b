// Path-tracing kernel body, compiled three times:
//   rt_trace_exact.hip  : RT_STRICT 1, RT_DIRECT 1, -ffp-contract=off, IEEE sqrt and divide.  Every f32
//                         operation of the path is written in the order the reference writes it; the
//                         reference's rejection samplers are replaced by loop-free maps a CPU reproduces
//                         bit for bit.  The default build.
//   rt_trace_strict.hip : RT_STRICT 1, the same arithmetic with the reference's rejection samplers and
//                         the reference's order of random draws.
//   rt_trace_fast.hip   : RT_STRICT 0, FMA contraction, unit-length ray directions, hardware
//                         rsqrt/rcp.  Same estimator, not bit-identical.
//
// Mapping: one lane = one stream of pixel samples.  A lane owns a work unit (pixel, chunk of
// samples), traces its samples one after another, and immediately starts the next sample when a
// path ends, so that every lane of the wave has a live ray whenever the wave enters the sphere
// loop.  Lanes that run out of samples take the next unit from their wave's pool of 64 to 512 units
// (one ballot and a prefix count); a wave refills its pool from a global counter with one atomic.
// A wave with only a few rays left tests each of them with all 64 lanes (shared_ray_hit).
//
// Reference functions restated here (file:line under the reference tree):
//   get_ray Main.c:776-803, sample_square :764, defocus_disk_sample :770,
//   hit_sphere :625-686, hit :688-713, set_face_normal :534, reflectance :551,
//   ray_scatter :558-620, ray_color :725-762, render_tile :959-988,
//   vec3f_* src/util.c:83-220.
#include "rt_device.h"

#if !defined(RT_STRICT) || !defined(RT_NS) || !defined(RT_KERNEL_FOR)
#error "RT_STRICT, RT_NS and RT_KERNEL_FOR must be defined"
#endif
#ifndef RT_DIRECT
#define RT_DIRECT 0         // 1 (with RT_STRICT): loop-free samplers in place of the reference's rejection loops
#endif


namespace rtmi {
namespace RT_NS {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 mul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 scale(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float lensq(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }

#if RT_STRICT
// IEEE square root.  The compiler's expansion of sqrtf() is: scale x by 2^32 if it is below 2^-96, v_sqrt_f32 (one ulp),
// one correction step that tests the two neighbours of the estimate with exact FMA residuals, scale back, and pass
// zeros and infinities through: 16 instructions.  The scaling only matters for 0 < x < 2^-96 and the pass-through
// changes nothing (the correction step leaves +-0, +inf, NaN and the NaN of a negative x as v_sqrt_f32 gives them), so
// for every other x the nine instructions of the core give the same bits.
__device__ __forceinline__ float sqrt_core(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float below = __uint_as_float(__float_as_uint(s) - 1U), above = __uint_as_float(__float_as_uint(s) + 1U);
    const float r_below = __builtin_fmaf(-below, s, x), r_above = __builtin_fmaf(-above, s, x);
    s = r_below <= 0.0f ? below : s;
    s = r_above > 0.0f ? above : s;
    return s;
}
// x is known to be zero or at least 2^-96 (or not finite): arguments of the form 1 - y*y with |y| <= 1, (k + 0.5) / 65536 ...
#ifdef RT_LIBM_SQRT             // comparison builds: the compiler's expansion everywhere
__device__ __forceinline__ float rt_sqrt_not_tiny(float x) { return sqrtf(x); }
#else
__device__ __forceinline__ float rt_sqrt_not_tiny(float x) { return sqrt_core(x); }
#endif
// any x: the wave takes the compiler's expansion if one of its lanes holds a tiny positive x
__device__ __forceinline__ float rt_sqrt(float x)
{
#ifdef RT_LIBM_SQRT
    return sqrtf(x);
#else
    if (__builtin_expect(__ballot(__float_as_uint(x) - 1U < 0x0f800000U - 1U) != 0ULL, 0)) return sqrtf(x);     // 0 < x < 2^-96
    return sqrt_core(x);
#endif
}
__device__ __forceinline__ float rt_div(float a, float b) { return a / b; }
__device__ __forceinline__ float rt_inv_sqrt(float x) { return 1.0f / rt_sqrt_not_tiny(x); }      // callers pass 0 or x >= 1e-15
// IEEE division.  The compiler's expansion of n / d is: scale both operands if the quotient, the reciprocal or a residual
// could leave the normal range (v_div_scale_f32 x 2), v_rcp_f32, one Newton step on the reciprocal, the quotient with two
// residual corrections (the last one v_div_fmas_f32, which undoes the scaling), and v_div_fixup_f32 for zeros,
// infinities and NaNs: 11 instructions, four of them of the slow kind.  For 2^-40 <= |d| <= 2^40 and
// 2^-80 <= |n| <= 2^48 nothing is scaled and nothing is fixed up, and the seven fused multiply-adds of the core give the
// same bits.  The refined reciprocal serves every numerator over the same denominator.
struct Reciprocal { float d, r; };
__device__ __forceinline__ Reciprocal div_core_reciprocal(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    Reciprocal k;
    k.d = d;
    k.r = __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
    return k;
}
__device__ __forceinline__ float div_core(float n, const Reciprocal &k)
{
    float q = n * k.r;
    q = __builtin_fmaf(__builtin_fmaf(-k.d, q, n), k.r, q);
    return __builtin_fmaf(__builtin_fmaf(-k.d, q, n), k.r, q);
}
#else
__device__ __forceinline__ float rt_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float rt_sqrt_not_tiny(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float rt_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float rt_inv_sqrt(float x) { return __builtin_amdgcn_rsqf(x); }
#endif

// vec3f_unit, util.c:132-141
__device__ __forceinline__ V3 unit(V3 v)
{
    float l2 = lensq(v);
#if RT_STRICT
#ifndef RT_LIBM_DIV
    // 2^-96 <= |v|^2 <= 2^96, or zero: the square root's core and the division's core (1 / len with 2^-48 <= len <= 2^48);
    // anything else, in any lane of the wave: the compiler's expansions
    if (__builtin_expect(__ballot(__float_as_uint(l2) - 0x0f800000U > 0x6f800000U - 0x0f800000U && l2 != 0.0f) == 0ULL, 1)) {
        const float len = sqrt_core(l2);
        const float inv = div_core(1.0f, div_core_reciprocal(len));
        return len > 0.0f ? scale(v, inv) : v3(0.0f, 0.0f, 0.0f);
    }
#endif
    float len = rt_sqrt(l2);
    if (len > 0.0f) {
        float inv = 1.0f / len;
        return scale(v, inv);
    }
    return v3(0.0f, 0.0f, 0.0f);
#else
    float inv = l2 > 0.0f ? rt_inv_sqrt(l2) : 0.0f;
    return scale(v, inv);
#endif
}

// vec3f_reflect, util.c:208-211
__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return sub(v, scale(n, 2.0f * dot(v, n))); }

// ---- counter-based random numbers ---------------------------------------------------------
// key  = mix32( mix32(pixel + mix32(seed ^ 0x6A09E667)) ^ sample * 0x9E3779B9 )
// draw = mix32(key + n * 0x9E3779B9), n = 0, 1, 2, ... per sample;  u = (draw >> 8) * 2^-24
// Replaces fast_rand / RAND_FLOAT (util.c:27-39, util.h:93).
__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

struct Rng { uint32_t ctr; };          // key + n * 0x9E3779B9 of the next draw

template <int MODE>
__device__ __forceinline__ float rng_float(Rng &r)
{
    if (MODE == 1) return 0.5f;                       // RT_MODE_CONST_U: fast_rand() == 16384
    uint32_t x = mix32(r.ctr);
    r.ctr += 0x9E3779B9U;
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}

// RAND_FLOAT_RANGE(-1, 1), util.h:94
template <int MODE>
__device__ __forceinline__ float rng_pm1(Rng &r) { return -1.0f + 2.0f * rng_float<MODE>(r); }

// The reference accepts a rejection sample when 1e-15 < (double)l2 && l2 <= 1 (util.c:176, 192).  The nearest f32
// to 1e-15 lies above it and the f32 below that lies below it, so for an f32 l2 the first test is l2 >= 1e-15f:
// the same decision without the conversion and the f64 compare.
static_assert(1e-15f == 0x1.203afap-50f && (double)0x1.203afap-50f > 1e-15 && (double)0x1.203af8p-50f < 1e-15, "f32 form of the 1e-15 threshold");
__device__ __forceinline__ bool accepts(float l2) { return l2 >= 1e-15f && l2 <= 1.0f; }

// vec3f_random_direction, util.c:170-184.  The loop only looks for the sample; the normalisation (an IEEE square
// root and divide in the strict build) runs once after it, for all lanes together, instead of in whichever
// iteration a lane happens to leave the loop.
template <int MODE>
__device__ __forceinline__ V3 random_direction(Rng &r)
{
    V3 p = v3(0.0f, 0.0f, 0.0f);
    float l2 = 0.0f;
    bool found = false;
    for (int i = 0; i < 100 && !found; i++) {
        V3 c;
        c.x = rng_pm1<MODE>(r);
        c.y = rng_pm1<MODE>(r);
        c.z = rng_pm1<MODE>(r);
        const float cl = lensq(c);
        if (accepts(cl)) { found = true; p = c; l2 = cl; }
    }
    asm volatile("" : "+v"(l2));            // keeps the normalisation behind the loop
    if (!found) return v3(0.0f, 0.0f, 0.0f);
    return scale(p, rt_inv_sqrt(l2));
}

// vec3f_random_direction_2d, util.c:186-200
template <int MODE>
__device__ __forceinline__ V3 random_direction_2d(Rng &r)
{
    for (int i = 0; i < 100; i++) {
        V3 p;
        p.x = rng_pm1<MODE>(r);
        p.y = rng_pm1<MODE>(r);
        p.z = 0.0f;
        float l2 = lensq(p);
        if (accepts(l2)) return p;
    }
    return v3(0.0f, 0.0f, 0.0f);
}

// reflectance, Main.c:551-556; r0 = ((1 - ri) / (1 + ri))^2 comes from the host, which evaluates it with the same two
// IEEE operations for both sides of every glass sphere (rt_set_world)
__device__ __forceinline__ float reflectance(float cosine, float r0)
{
    float m = 1.0f - cosine;
    return r0 + (1.0f - r0) * (m * m * m * m * m);
}

// vec3f_refract, util.c:214-220
__device__ __forceinline__ V3 refract(V3 v, V3 n, float e)
{
    float cos_theta = fminf(dot(scale(v, -1.0f), n), 1.0f);
    V3 perp = scale(add(v, scale(n, cos_theta)), e);
    V3 par = scale(n, -1.0f * rt_sqrt_not_tiny(fabsf(1.0f - lensq(perp))));      // zero, or at least 2^-24
    return add(perp, par);
}

#if RT_STRICT && RT_DIRECT
// Exact build: the reference's rejection samplers (util.c:170-200) are replaced by loop-free maps with the same
// distributions, written so that a CPU evaluates them to the same bits: integer arithmetic on the 32-bit draw,
// conversions and scalings that are exact in f32, IEEE square root, and polynomials whose every step is either a
// single rounded operation or an explicit fused multiply-add (fmaf on the CPU side).  One draw gives two 16-bit
// fields (hi, lo); a field k stands for the centre of its stratum, (k + 0.5) / 65536.
// In constant-u mode the reference's loops reject u = 0.5 a hundred times and return the zero vector.

// cos and sin of 2 pi i / 65536.  The nearest quarter turn q is taken off in integers, which leaves an angle
// within +-pi/4 for the two polynomials (the single-precision kernels of the Cephes library, about one ulp), and
// the quarter turns are put back by swapping and negating.
__device__ __forceinline__ void sincos_turns16(uint32_t i, float &c, float &s)
{
    const uint32_t q = (i + 8192U) >> 14;                                  // 0 .. 4
    const float th = (float)((int)i - (int)(q << 14)) * 9.58737992428525768e-5f;   // * 2 pi / 65536
    const float z = th * th;
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    const float s0 = fmaf(ps * z, th, th);
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    const float c0 = fmaf(pc, z * z, fmaf(-0.5f, z, 1.0f));
    const bool swap = (q & 1U) != 0U;
    const float c1 = swap ? s0 : c0, s1 = swap ? c0 : s0;
    c = __uint_as_float(__float_as_uint(c1) ^ (((q + 1U) & 2U) << 30));    // quarter turns 1 and 2: cos changes sign
    s = __uint_as_float(__float_as_uint(s1) ^ ((q & 2U) << 30));           // quarter turns 2 and 3: sin changes sign
}

// sample_square (Main.c:764-768) from one draw: offsets (k + 0.5) / 65536 - 0.5
template <int MODE>
__device__ __forceinline__ void jitter_direct(Rng &r, float &offx, float &offy)
{
    if (MODE == 1) { offx = 0.0f; offy = 0.0f; return; }
    const uint32_t x = mix32(r.ctr);
    r.ctr += 0x9E3779B9U;
    const float bias = -0.5f + 1.0f / 131072.0f;
    offx = fmaf((float)(x >> 16), 1.0f / 65536.0f, bias);
    offy = fmaf((float)(x & 0xffffU), 1.0f / 65536.0f, bias);
}

// uniform on the unit sphere (replaces vec3f_random_direction, util.c:170-184): z uniform in (-1, 1), azimuth uniform
template <int MODE>
__device__ __forceinline__ V3 sphere_direction_direct(Rng &r)
{
    if (MODE == 1) return v3(0.0f, 0.0f, 0.0f);
    const uint32_t x = mix32(r.ctr);
    r.ctr += 0x9E3779B9U;
    const float z = fmaf((float)(x >> 16), -1.0f / 32768.0f, 65535.0f / 65536.0f);   // (65535 - 2 k) / 65536, exact
    const float rad = rt_sqrt_not_tiny(fmaf(-z, z, 1.0f));             // at least 1 - (65535 / 65536)^2
    float c, s;
    sincos_turns16(x & 0xffffU, c, s);
    return v3(rad * c, rad * s, z);
}

// uniform in the unit disk (replaces vec3f_random_direction_2d, util.c:186-200): radius sqrt(u), azimuth uniform
template <int MODE>
__device__ __forceinline__ void disk_point_direct(Rng &r, float &px, float &py)
{
    if (MODE == 1) { px = 0.0f; py = 0.0f; return; }
    const uint32_t x = mix32(r.ctr);
    r.ctr += 0x9E3779B9U;
    const float rad = rt_sqrt_not_tiny(fmaf((float)(x >> 16), 1.0f / 65536.0f, 1.0f / 131072.0f));    // (k + 0.5) / 65536, exact
    float c, s;
    sincos_turns16(x & 0xffffU, c, s);
    px = rad * c;
    py = rad * s;
}
#endif   // RT_STRICT && RT_DIRECT

#if !RT_STRICT
// Fast build: the reference's rejection samplers (util.c:170-200) are replaced by direct maps
// with the same distributions (uniform on the unit sphere / in the unit disk); no loop, so the
// lanes of a wave do not wait for the slowest rejection chain.  v_sin/v_cos take revolutions.
// In constant-u mode the reference's loops reject u = 0.5 a hundred times and return zero.
// Two uniforms in [0,1) from one 32-bit draw, 16 bits each (the reference's RAND_FLOAT has 15).
template <int MODE>
__device__ __forceinline__ void rng_float2(Rng &r, float &u0, float &u1)
{
    if (MODE == 1) { u0 = 0.5f; u1 = 0.5f; return; }
    const uint32_t x = mix32(r.ctr);
    r.ctr += 0x9E3779B9U;
    u0 = (float)(x >> 16) * (1.0f / 65536.0f);
    u1 = (float)(x & 0xffffu) * (1.0f / 65536.0f);
}

template <int MODE>
__device__ __forceinline__ V3 sphere_direction(Rng &r)
{
    if (MODE == 1) return v3(0.0f, 0.0f, 0.0f);
    float u, phi;
    rng_float2<MODE>(r, u, phi);
    const float z = fmaf(-2.0f, u, 1.0f);
    const float rad = rt_sqrt(fmaxf(0.0f, fmaf(-z, z, 1.0f)));
    return v3(rad * __builtin_amdgcn_cosf(phi), rad * __builtin_amdgcn_sinf(phi), z);
}

template <int MODE>
__device__ __forceinline__ void disk_point(Rng &r, float &x, float &y)
{
    if (MODE == 1) { x = 0.0f; y = 0.0f; return; }
    float u, phi;
    rng_float2<MODE>(r, u, phi);
    const float rad = rt_sqrt(u);
    x = rad * __builtin_amdgcn_cosf(phi);
    y = rad * __builtin_amdgcn_sinf(phi);
}

#endif   // !RT_STRICT

// q = u / n, r = u % n for u < 2^32 without the 32-bit integer divide sequence: the f64
// product is at most one below the quotient.
__device__ __forceinline__ void divmod(uint32_t u, uint32_t n, double inv_n, uint32_t &q, uint32_t &r)
{
    q = (uint32_t)((double)u * inv_n);
    r = u - q * n;
    if (r >= n) { q++; r -= n; }
}

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) f8 const_f8;      // scalar-cache loads

// The ray a sphere is tested against, and the nearest hit found so far.  hit() keeps the first
// of equal distances (strict '<' while scanning in index order, Main.c:699-704); candidates are
// resolved out of order here, so the rule is written out: nearer wins, then the lower index.
struct Probe {
    float ox, oy, oz, dx, dy, dz, tmin;
#if RT_STRICT
    float a;            // |d|^2, Main.c:648
#endif
    float closest;
    int hit;                    // in device order
    const int32_t *orig;        // device order -> the caller's index (wave-uniform; read only to settle equal distances)
};

// Spheres are resolved out of order and the device order is not the caller's: of two hits at exactly the same
// distance the one with the lower index in the CALLER's list wins, as in hit()'s in-order strict '<'.
__device__ __forceinline__ bool wins_tie(const Probe &q, int i)
{
    return q.hit >= 0 && q.orig[i] < q.orig[q.hit];
}

// hit_sphere (Main.c:625-672) for the sphere record g = {C, r^2}: the root the reference would accept if nothing
// nearer had been found yet.  Surrounds(root, tmin, closest) fails for the near root either because it is <= tmin,
// and then the far root is tried, or because it is >= closest, and then the far root, which is not smaller, fails
// too: the choice of root does not depend on closest.  a = |d|^2 (strict build; the fast build's d is a unit vector).
// CORE (strict builds): the two quotients by the division's core.  The caller guarantees 2^-40 <= a <= 2^40, an origin within
// 2^24 of the frame's origin in every coordinate and spheres within 2^24 (rays for which it cannot are tested again with
// CORE = false): then |h +- sq| <= 2^48, and a numerator below 2^-80 gives a quotient below 2^-40 either way, which
// tmin = 0.001 rejects either way.
template <bool CORE = true>
__device__ __forceinline__ bool sphere_root(const float4 g, float ox, float oy, float oz, float dx, float dy, float dz,
                                            float a, float tmin, float &t)
{
    const float ocx = g.x - ox, ocy = g.y - oy, ocz = g.z - oz;
#if RT_STRICT
    const float h = dx * ocx + dy * ocy + dz * ocz;
    const float c = (ocx * ocx + ocy * ocy + ocz * ocz) - g.w;
    const float disc = h * h - a * c;
    if (disc < 0.0f) return false;
    const float sq = rt_sqrt(disc);
#ifndef RT_LIBM_DIV
    if (CORE) {
        const Reciprocal k = div_core_reciprocal(a);
        float root = div_core(h - sq, k);
        if (!(tmin < root)) root = div_core(h + sq, k);
        t = root;
        return tmin < root;
    }
#endif
    float root = (h - sq) / a;
    if (!(tmin < root)) root = (h + sq) / a;
    t = root;
    return tmin < root;
#else
    (void)a;
    const float h = fmaf(dz, ocz, fmaf(dy, ocy, dx * ocx));
    const float negc = fmaf(-ocz, ocz, fmaf(-ocy, ocy, fmaf(-ocx, ocx, g.w)));
    const float disc = fmaf(h, h, negc);
    if (!(disc >= 0.0f)) return false;
    const float sq = rt_sqrt(disc);
    const float t0 = h - sq;
    t = (t0 > tmin) ? t0 : (h + sq);
    return t > tmin;
#endif
}

// The accept decision of hit() for sphere i (device order) against the nearest hit found so far.
template <bool CORE = true>
__device__ __forceinline__ void exact_test_record(const float4 g, int i, Probe &q)
{
    float t;
#if RT_STRICT
    const float a = q.a;
#else
    const float a = 1.0f;
#endif
    if (sphere_root<CORE>(g, q.ox, q.oy, q.oz, q.dx, q.dy, q.dz, a, q.tmin, t)) {
        if (t < q.closest) { q.closest = t; q.hit = i; }
        else if (__builtin_expect(t == q.closest, 0)) { if (wins_tie(q, i)) q.hit = i; }
    }
}

template <bool CORE = true>
__device__ __forceinline__ void exact_test(const float4 *s_geom, int i, Probe &q)
{
    exact_test_record<CORE>(s_geom[i], i, q);
}

#ifdef RT_DIAG
// Diagnostic builds only: where the non-scan instructions go.  RT_DIAG_BRUTE adds the exact test over
// all spheres per ray (lost_hits); RT_DIAG_STAMPS adds shader-clock stamps at the phase boundaries of
// the main loop (wave-cycles per phase, summed over waves; a stamp drains the scalar-memory queue).
struct Diag {
    uint32_t branches, passes, drain_iters, drain_lanes, scans, scan_lanes, shared_rays, lost_hits, tiles_bound, tiles_halfbound, quads_bound, tiles_box, pairs_box, tiles_run;
    uint32_t scans_by_live[4];      // wave scans with at most 8 / 16 / 32 / 48 live rays
    unsigned long long t_handout, t_regen, t_hit, t_drain, t_big, t_shade, t_shared, t_feat, t_append, t_cull;
};
#define RT_DIAG_ARG , Diag &dg
#define RT_DIAG_PASS , dg
#ifdef RT_DIAG_STAMPS
#define RT_STAMP() __builtin_amdgcn_s_memtime()
#else
#define RT_STAMP() 0ULL
#endif
#else
#define RT_DIAG_ARG
#define RT_DIAG_PASS
#endif

// make isa ISADEF=-DRT_ISA_MARKS: comment lines in the generated assembly at the phase boundaries (tools/isa_cost.py)
#ifdef RT_ISA_MARKS
#define RT_MARK(name) asm volatile("; MARK " name)
#else
#define RT_MARK(name)
#endif

constexpr int kQueueSlots = RT_STRICT ? kQueueSlotsStrict : kQueueSlotsFast;   // per lane
constexpr int kPairBatch = 4;              // pairs between two queue-overflow checks
static_assert(kQueueSlots > 2 * kPairBatch, "a batch of pairs must fit the queue");
static_assert(kGeomPad == 2 * kPairBatch, "the sphere list is padded to whole batches");

// Exact test of the queued candidates of every lane of the wave; queue[j * 64 + lane] is the
// j-th sphere index the lane queued (slot-major: a lane's LDS bank does not depend on j).
__device__ __forceinline__ void drain(const float4 *s_geom, const uint32_t *queue, int cnt, int lane, Probe &q RT_DIAG_ARG)
{
#ifdef RT_DIAG
    const unsigned long long t0 = RT_STAMP();
#endif
    for (int j = 0; __ballot(j < cnt) != 0ULL; j++) {
#ifdef RT_DIAG
        dg.drain_iters++; dg.drain_lanes += (uint32_t)__popcll(__ballot(j < cnt));
#endif
        if (j < cnt) exact_test(s_geom, (int)queue[j * 64 + lane], q);
    }
#ifdef RT_DIAG
    dg.t_drain += RT_STAMP() - t0;
#endif
}

// hit() (Main.c:688-713) as filter + exact test.
//
// The filter is conservative and runs on two spheres per packed-FP32 instruction with the sphere
// data in SGPRs (scalar-cache loads, wave-uniform index).  With a unit direction u the
// discriminant is D = (u.C - u.O)^2 + 2 O.C - |O|^2 - (|C|^2 - r^2); the expanded form needs seven
// packed FMAs per pair but cancels badly, so a sphere passes when D >= -slack, where slack bounds
// the rounding error of the filter AND of the exact test it stands in front of (slack_scale *
// (|O| + max|C|)^2): a sphere the exact test would accept is never dropped.  A pass only queues
// the sphere's index; the exact test runs on the queue afterwards.  Spheres much larger than the
// rest would inflate slack for everybody and are tested exactly for every ray instead (big[]).
__device__ __forceinline__ void scan_spheres(const TraceParams &p, const float4 *s_geom, uint32_t *queue, int lane,
                                             float ux, float uy, float uz, float slack_scale, Probe &q RT_DIAG_ARG)
{
#ifdef RT_DIAG
    const unsigned long long tb0 = RT_STAMP();
#endif
    for (int b = 0; b < p.nbig; b++) exact_test(s_geom, p.big[b], q);
#ifdef RT_DIAG
    dg.t_big += RT_STAMP() - tb0;
#endif

    const float ox = q.ox, oy = q.oy, oz = q.oz;
    const float oo = fmaf(oz, oz, fmaf(oy, oy, ox * ox));
    const float reach = __builtin_amdgcn_sqrtf(oo) * 1.0000005f + p.c_near;   // >= |O| + |C|
    const float slack = fmaf(reach * reach, slack_scale, 1e-30f);
    const float base = slack - oo;
    const float ndo = -fmaf(uz, oz, fmaf(uy, oy, ux * ox));
    const float ox2 = ox + ox, oy2 = oy + oy, oz2 = oz + oz;
    const_f8 *gp = (const_f8 *)p.pairs;
    const int npairs = p.n_pad >> 1;
    int cnt = 0;
    f8 ahead = gp[0];                      // loaded one pair ahead: a wave alone on its SIMD does not wait per pair
    for (int k0 = 0; k0 < npairs; k0 += kPairBatch) {
#pragma unroll
        for (int u = 0; u < kPairBatch; u++) {
            const f8 s = ahead;
            ahead = gp[k0 + u + 1];
            const f2 cx = {s[0], s[1]}, cy = {s[2], s[3]}, cz = {s[4], s[5]};
            f2 h = __builtin_elementwise_fma(cx, (f2)ux, (f2)ndo);
            h = __builtin_elementwise_fma(cy, (f2)uy, h);
            h = __builtin_elementwise_fma(cz, (f2)uz, h);
            f2 m = __builtin_elementwise_fma(cx, (f2)ox2, (f2)base);
            m = __builtin_elementwise_fma(cy, (f2)oy2, m);
            m = __builtin_elementwise_fma(cz, (f2)oz2, m);
            const f2 e = __builtin_elementwise_fma(h, h, m);
            const bool px = e.x >= s[6], py = e.y >= s[7];
            // One wave-uniform branch per pair keeps the common no-pass case at two compares
            // and one scalar branch.
            if (__ballot(px || py) != 0ULL) {
#ifdef RT_DIAG
                dg.branches++;
                dg.passes += (uint32_t)(__popcll(__ballot(px)) + __popcll(__ballot(py)));
#endif
                if (px) { queue[cnt * 64 + lane] = (uint32_t)(2 * (k0 + u)); cnt++; }
                if (py) { queue[cnt * 64 + lane] = (uint32_t)(2 * (k0 + u) + 1); cnt++; }
            }
        }
        if (__ballot(cnt > kQueueSlots - 2 * kPairBatch) != 0ULL) {
            drain(s_geom, queue, cnt, lane, q RT_DIAG_PASS);
            cnt = 0;
        }
    }
#ifdef RT_DIAG
    dg.scans++;
#endif
    drain(s_geom, queue, cnt, lane, q RT_DIAG_PASS);
}

// ---- the filter on the matrix cores ---------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_f16(_Float16 lo, _Float16 hi)
{
    const f16x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ f16x8 as_frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    return __builtin_bit_cast(f16x8, make_uint4(a, b, c, d));
}

// The wave's LDS region of the matrix scan (rt_device.h): [64 result keys][candidate list].
struct WavePool {
    unsigned long long *keys;       // keys[r] = bits(t) << 32 | caller's index << 16 | device index of ray r's nearest hit so far
    uint32_t *list;                 // ray << 16 | sphere (device order)
};

__device__ __forceinline__ WavePool wave_pool(char *base)
{
    WavePool w;
    w.keys = reinterpret_cast<unsigned long long *>(base);
    w.list = reinterpret_cast<uint32_t *>(base + kMatKeyBytes);
    return w;
}

__device__ __forceinline__ float lane_value(int byte_address, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_address, __builtin_bit_cast(int, v)));
}

// One dense step of exact tests: lane l takes entry first + l of the wave's candidate list (count <= 64 entries),
// whichever ray it belongs to, fetches that ray from the registers of the lane that owns it (all lanes take part
// in the exchange: a lane outside the exec mask would supply zeros), reads the sphere and lowers the ray's key if
// the sphere is hit.  t > tmin > 0, so the bit patterns of t order like t; below t the key orders by the caller's
// index, which is hit()'s rule for equal distances (the first sphere in the caller's list keeps the hit).
__device__ __forceinline__ void drain_pooled(const float4 *s_geom, const uint16_t *s_orig16, const int32_t *orig, const WavePool &w,
                                             int first, int count, int lane, const Probe &q RT_DIAG_ARG)
{
#ifdef RT_DIAG
    const unsigned long long t0 = RT_STAMP();
    dg.drain_iters++; dg.drain_lanes += (uint32_t)count;
#endif
    RT_MARK("drain_pooled");
    const bool mine = lane < count;
    const uint32_t e = w.list[first + (mine ? lane : 0)];
    const uint32_t ray = e >> 16, sph = e & 0xffffU;
    const int from = (int)(ray << 2);
    const float ox = lane_value(from, q.ox), oy = lane_value(from, q.oy), oz = lane_value(from, q.oz);
    const float dx = lane_value(from, q.dx), dy = lane_value(from, q.dy), dz = lane_value(from, q.dz);
    const float tmin = lane_value(from, q.tmin);
#if RT_STRICT
    const float a = lane_value(from, q.a);
#else
    const float a = 1.0f;
#endif
    if (mine) {
        const float4 g = s_geom[sph];
        const uint32_t caller = s_orig16 != nullptr ? (uint32_t)s_orig16[sph] : (uint32_t)orig[sph];
        float t;
        if (sphere_root(g, ox, oy, oz, dx, dy, dz, a, tmin, t))
            atomicMin(&w.keys[ray], ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)((caller << 16) | sph));
    }
#ifdef RT_DIAG
    dg.t_drain += RT_STAMP() - t0;
#endif
}

// One half tile: 16 accumulators per lane = 16 spheres of the tile against the column's ray (accumulator i of a
// lane in wave half h is tile row (i & 3) + 8 (i >> 2) + 4 h, where the image keeps sphere 2 i + h of the tile;
// ray 32 H + (lane & 31)).  The common case (no sphere of the tile passes for any ray) costs eight maxima, one
// compare and one branch.  A pass appends (ray, sphere) to the wave's list at a position taken from the ballot of
// the passing lanes; cnt is wave-uniform.  ebase = ray << 16 | (lane >> 5).
struct RowMaxima { float g[5], all; };     // maxima of five groups of three rows (the sixteenth row goes into "all" only)

__device__ __forceinline__ RowMaxima row_maxima(const f32x16 &acc)
{
    RowMaxima r;
    r.g[0] = __builtin_fmaxf(__builtin_fmaxf(acc[0], acc[1]), acc[2]);
    r.g[1] = __builtin_fmaxf(__builtin_fmaxf(acc[3], acc[4]), acc[5]);
    r.g[2] = __builtin_fmaxf(__builtin_fmaxf(acc[6], acc[7]), acc[8]);
    r.g[3] = __builtin_fmaxf(__builtin_fmaxf(acc[9], acc[10]), acc[11]);
    r.g[4] = __builtin_fmaxf(__builtin_fmaxf(acc[12], acc[13]), acc[14]);
    const float m5 = __builtin_fmaxf(__builtin_fmaxf(r.g[0], r.g[1]), acc[15]);
    const float m6 = __builtin_fmaxf(__builtin_fmaxf(r.g[2], r.g[3]), r.g[4]);
    r.all = __builtin_fmaxf(m5, m6);
    return r;
}

__device__ __forceinline__ void queue_passes(const f32x16 &acc, const RowMaxima &rm, int first, uint32_t ebase, uint32_t *list, int &cnt RT_DIAG_ARG)
{
    RT_MARK("queue_passes");
    if (__ballot(rm.all >= 0.0f) != 0ULL) {
#ifdef RT_DIAG
        dg.branches++;
#endif
        // the group maxima say where to look: five groups of three rows and the last row
        auto put = [&](int i) {
            const bool pass = acc[i] >= 0.0f;
            const unsigned long long m = __ballot(pass);
            if (m != 0ULL) {                                 // most rows of a group that holds a pass hold none
                const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, (uint32_t)cnt));
                if (pass) list[at] = ebase + (uint32_t)(first + 2 * i);
                cnt += (int)__popcll(m);
#ifdef RT_DIAG
                dg.passes += (uint32_t)__popcll(m);
#endif
            }
        };
        if (__ballot(rm.g[0] >= 0.0f) != 0ULL) { put(0); put(1); put(2); }
        if (__ballot(rm.g[1] >= 0.0f) != 0ULL) { put(3); put(4); put(5); }
        if (__ballot(rm.g[2] >= 0.0f) != 0ULL) { put(6); put(7); put(8); }
        if (__ballot(rm.g[3] >= 0.0f) != 0ULL) { put(9); put(10); put(11); }
        if (__ballot(rm.g[4] >= 0.0f) != 0ULL) { put(12); put(13); put(14); }
        put(15);
    }
}

// hit() (Main.c:688-713) as matrix filter + exact test; u is the unit direction.
//
// Error budget of the filter, in the scaled frame (S = |O'| + max|C'|; the absolute values of the 31 terms sum
// to at most 2 S^2).  Pieces: a feature is x = x1 + x2 + e with |e| <= 2^-24 |x| (two f16 pieces, round to
// nearest), the product x2 y2 is dropped: 3 * 2^-24 of every term, 2^-21.4 S^2.  The f32 roundings of the
// features: 2^-21 S^2.  Moving the origin to the nearest point of the line: 2^-20.2 S^2 (below).  The MFMA sums
// the 16 exact products of an instruction and its C input in a window of 24 bits below the largest of them and
// truncates what lies below (measured: tools/mfma_prec.hip): 32 terms, each <= S^2, plus the rounding of the two
// results: 2^-19 S^2 + 2^-22 S^2.  Sum 2^-18.0 S^2; the strict build adds 2^-21 S^2 for its hardware rsqrt and
// for the rounding of the reference-order exact test that decides afterwards.  slack = 2^-17 S^2 (+ 0.01 for
// f16 subnormal pieces) is twice the bound; measured over 16384 rays x all spheres of three scenes the
// accumulator is within 2^-21.4 S^2 of the discriminant (tools/mfma_scan_proto.hip).
//
// What passes the filter goes to the wave's pooled candidate list (LDS) and is tested exactly 64 entries at a
// time, one per lane whichever ray an entry belongs to, as soon as 64 are waiting (drain_pooled): a ray's
// candidates are few and unevenly spread over the lanes (about one per ray, up to five or six), so lane-private
// queues kept four lanes in five idle during their exact tests.  A lane fetches the ray of its entry from the
// registers of the lane that owns it and lowers that ray's key in LDS; every lane reads its own key afterwards.
//
// Before a tile is multiplied the wave asks whether any of its live rays can meet the tile at all: a slab test of the ray
// (from its origin to the nearest hit the big spheres already gave) against the axis-aligned box of the tile's spheres.
// The spheres of a tile are neighbours in space (device order: the host splits the list into tiles with small boxes), so most
// tiles lie off the paths of all 64 rays: of the final scene's 16 tiles a wave multiplies 5.1.  Two tiles are tested per
// packed-FP32 instruction; a scene of more than 64 tiles asks the boxes of groups of tiles first (tiles_in_need).
// The test must never drop a sphere the exact test would accept.  That test
// accepts a sphere whose centre lies within sqrt(r^2 + E) of the ray's line, E <= 2^-21 S^2 being the rounding of its
// discriminant per unit |d|^2 (S = |O| + max |C|): every box is widened by 2^-9 S for the ray at hand (eight times E in
// the square), which also covers the slab arithmetic (hardware reciprocal, one FMA per plane: 2^-22 of the coordinates).
// A zero component of d makes the planes of that axis infinite or NaN; maxima and minima drop NaNs and a NaN interval
// counts as a meeting, so that axis constrains nothing.  `live`: the lane carries a ray.
// ONE_GROUP: the scene has at most 64 tiles (always so with the images in LDS), one mask holds them all.
template <bool ONE_GROUP>
__device__ __forceinline__ void scan_spheres_matrix(const TraceParams &p, const float4 *s_geom, const uint4 *s_aimg, const uint16_t *s_orig16,
                                                    const float4 *s_box, char *pool_base, int lane, bool live, float ux, float uy, float uz,
                                                    Probe &q, uint32_t &tiles_done RT_DIAG_ARG)
{
    RT_MARK("scan_begin_big");
    const WavePool w = wave_pool(pool_base);
    w.keys[lane] = 0x7f7fffffffffffffULL;                    // max_f32, no sphere
    // (Measured and dropped: a wave with at most 32 live rays -- a fifth of the scans of a 1-spp frame -- packing them into
    // the columns of one ray half and running two MFMAs and one compare tree per tile.  Bit-identical, but the sixteen
    // cross-lane fetches and the second loop body cost registers everywhere: 4K x 250 spp 247 -> 265 ms, the short
    // launches it was meant for 3.02 -> 3.21 ms and 1.07 -> 1.05 ms.)
#ifdef RT_DIAG
    const unsigned long long tb0 = RT_STAMP(), t_big_before = dg.t_big;
#endif
    for (int b = 0; b < p.nbig; b++) exact_test(s_geom, p.big[b], q);
#ifdef RT_DIAG
    dg.t_big += RT_STAMP() - tb0;
#ifdef RT_DIAG_STAMPS
    asm volatile("" : "+v"(ux), "+v"(uy), "+v"(uz));          // the feature arithmetic stays between its two stamps
#endif
#endif
    // ---- which tiles of a group of 64 does any live ray of the wave meet?  (The first group is settled here, before the ray
    // features and their fragments occupy registers.)
#ifdef RT_NO_BOX_CULL          // comparison build: every tile is multiplied
    const bool cull = false;
#else
    const bool cull = s_box != nullptr;
#endif
    const unsigned long long live_lanes = __ballot(live);
    // body(meets2) with meets2(boxes, pr, first, second) = the lanes whose ray meets boxes 2 pr and 2 pr + 1.  The ray's side of the
    // test is computed anew per call: kept in registers across the tile loop it would cost more than it does to redo.
    auto box_tests = [&](auto &&body) {
        // the ray in the form the slab test wants: t of the plane x = c is c * (1 / dx) - ox * (1 / dx)
        const float ix = __builtin_amdgcn_rcpf(q.dx), iy = __builtin_amdgcn_rcpf(q.dy), iz = __builtin_amdgcn_rcpf(q.dz);
        const float oix = q.ox * ix, oiy = q.oy * iy, oiz = q.oz * iz;
        const float wo = fmaf(q.oz, q.oz, fmaf(q.oy, q.oy, q.ox * q.ox));
        const float widen = fmaf(__builtin_amdgcn_sqrtf(wo), 1.0000005f, p.c_near) * p.box_margin;          // 2^-9 (|O| + max |C|)
        // Two tiles per packed-FP32 instruction: the boxes come as {cx, cx', cy, cy'}, {cz, cz', hx, hx'}, {hy, hy', hz, hz'} per
        // pair of tiles (an odd last tile is paired with a box nothing meets).
        const f2 ix2 = {ix, ix}, iy2 = {iy, iy}, iz2 = {iz, iz}, oix2 = {oix, oix}, oiy2 = {oiy, oiy}, oiz2 = {oiz, oiz};
        const f2 ax2 = {fabsf(ix), fabsf(ix)}, ay2 = {fabsf(iy), fabsf(iy)}, az2 = {fabsf(iz), fabsf(iz)}, widen2 = {widen, widen};
        auto meets2 = [&](const float4 *boxes, int pr, unsigned long long &first, unsigned long long &second) {    // lanes whose ray meets boxes 2 pr, 2 pr + 1
            const float4 ba = boxes[3 * pr], bb = boxes[3 * pr + 1], bc = boxes[3 * pr + 2];
            const f2 cx = {ba.x, ba.y}, cy = {ba.z, ba.w}, cz = {bb.x, bb.y};
            const f2 hx = f2{bb.z, bb.w} + widen2, hy = f2{bc.x, bc.y} + widen2, hz = f2{bc.z, bc.w} + widen2;
            const f2 mx = __builtin_elementwise_fma(cx, ix2, -oix2), my = __builtin_elementwise_fma(cy, iy2, -oiy2), mz = __builtin_elementwise_fma(cz, iz2, -oiz2);
            const f2 nx = __builtin_elementwise_fma(-hx, ax2, mx), ny = __builtin_elementwise_fma(-hy, ay2, my), nz = __builtin_elementwise_fma(-hz, az2, mz);
            const f2 fx = __builtin_elementwise_fma(hx, ax2, mx), fy = __builtin_elementwise_fma(hy, ay2, my), fz = __builtin_elementwise_fma(hz, az2, mz);
            const float in0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(nx.x, ny.x), nz.x), 0.0f);
            const float out0 = __builtin_fminf(__builtin_fminf(__builtin_fminf(fx.x, fy.x), fz.x), q.closest);
            const float in1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(nx.y, ny.y), nz.y), 0.0f);
            const float out1 = __builtin_fminf(__builtin_fminf(__builtin_fminf(fx.y, fy.y), fz.y), q.closest);
            first = __ballot(!(out0 < in0)) & live_lanes;
            second = __ballot(!(out1 < in1)) & live_lanes;
        };
        return body(meets2);
    };
    auto tiles_in_need = [&](int t0) {                     // of tiles t0 .. t0 + 63
        const int group = p.ntiles - t0 < 64 ? p.ntiles - t0 : 64;
        if (!cull) return group == 64 ? ~0ULL : (1ULL << group) - 1ULL;
        return box_tests([&](auto &meets2) {
        // from the last pair of the group to the first, the mask shifted up by two per pair; two pairs per step, so that
        // their box reads are in flight together
        unsigned long long need = 0ULL;
        const int p0 = t0 >> 1;
        if (!ONE_GROUP && p.group_boxes != nullptr) {
            // A large scene: the boxes of the (at most eight) groups of eight consecutive tiles first, then the tiles of the
            // groups some ray meets.  The host pads the tiles' records to whole groups with boxes nothing meets.
            const int groups = (group + kBoxGroupTiles - 1) / kBoxGroupTiles, g0 = t0 / kBoxGroupTiles;
            uint32_t met = 0U;
            for (int k = (groups + 1) >> 1; k >= 1; k--) {
                unsigned long long m1, m0;
                meets2(p.group_boxes, (g0 >> 1) + k - 1, m0, m1);
                met = (met << 2) | (m1 != 0ULL ? 2U : 0U) | (m0 != 0ULL ? 1U : 0U);
            }
            while (met != 0U) {
                const int gi = __builtin_ctz(met);
                met &= met - 1U;
                unsigned long long m7, m6, m5, m4, m3, m2, m1, m0;
                const int pr = p0 + gi * (kBoxGroupTiles / 2);
                meets2(s_box, pr + 3, m6, m7);
                meets2(s_box, pr + 2, m4, m5);
                meets2(s_box, pr + 1, m2, m3);
                meets2(s_box, pr, m0, m1);
                const unsigned long long bits = (m7 != 0ULL ? 128ULL : 0ULL) | (m6 != 0ULL ? 64ULL : 0ULL) | (m5 != 0ULL ? 32ULL : 0ULL) | (m4 != 0ULL ? 16ULL : 0ULL) |
                                                (m3 != 0ULL ? 8ULL : 0ULL) | (m2 != 0ULL ? 4ULL : 0ULL) | (m1 != 0ULL ? 2ULL : 0ULL) | (m0 != 0ULL ? 1ULL : 0ULL);
                need |= bits << (gi * kBoxGroupTiles);
            }
            return need;
        }
        int k = (group + 1) >> 1;
        for (; k >= 2; k -= 2) {
            unsigned long long m3, m2, m1, m0;
            meets2(s_box, p0 + k - 1, m2, m3);
            meets2(s_box, p0 + k - 2, m0, m1);
            need = (need << 4) | (m3 != 0ULL ? 8ULL : 0ULL) | (m2 != 0ULL ? 4ULL : 0ULL) | (m1 != 0ULL ? 2ULL : 0ULL) | (m0 != 0ULL ? 1ULL : 0ULL);
        }
        if (k >= 1) {
            unsigned long long m1, m0;
            meets2(s_box, p0, m0, m1);
            need = (need << 2) | (m1 != 0ULL ? 2ULL : 0ULL) | (m0 != 0ULL ? 1ULL : 0ULL);
        }
        return need;
        });
    };
    // More than four groups of 64 tiles: which of those groups does any ray meet at all?  (A group nobody meets then costs
    // half a packed test here instead of four in the tile loop.)
    auto groups_of_64_met = [&]() {
        if (ONE_GROUP || !cull || p.mask_group_boxes == nullptr || p.ntiles <= 256) return ~0U;
        return box_tests([&](auto &meets2) {
            uint32_t met = 0U;
            for (int k = (((p.ntiles + 63) >> 6) + 1) >> 1; k >= 1; k--) {
                unsigned long long m1, m0;
                meets2(p.mask_group_boxes, k - 1, m0, m1);
                met = (met << 2) | (m1 != 0ULL ? 2U : 0U) | (m0 != 0ULL ? 1U : 0U);
            }
            return met;
        });
    };
#ifdef RT_DIAG_STAMPS
    const unsigned long long tc0 = RT_STAMP();
#endif
    const uint32_t met64 = groups_of_64_met();
    unsigned long long need_first = tiles_in_need(0);
#ifdef RT_DIAG_STAMPS
    {                                                             // the box tests stay between their two stamps
        uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)need_first), hi = __builtin_amdgcn_readfirstlane((uint32_t)(need_first >> 32));
        asm volatile("" : "+s"(lo), "+s"(hi));
        need_first = ((unsigned long long)hi << 32) | lo;
    }
    const unsigned long long tc1 = RT_STAMP();
    dg.t_cull += tc1 - tc0;
#endif
    RT_MARK("features");
    // ray features in the frame scaled by p.mat_scale (a power of two)
    const float ox = q.ox * p.mat_scale, oy = q.oy * p.mat_scale, oz = q.oz * p.mat_scale;
    const float oo = fmaf(oz, oz, fmaf(oy, oy, ox * ox));
    const float reach = __builtin_amdgcn_sqrtf(oo) * 1.0000005f + p.mat_c_near;
#ifndef RT_MAT_SLACK
#define RT_MAT_SLACK 7.62939453125e-6f                                            // 2^-17
#endif
    const float slack = fmaf(reach * reach, RT_MAT_SLACK, 0.01f);
    // The discriminant does not change when the origin slides along the ray: take the point of the line
    // nearest to the frame's origin, P = O' - (u . O') u.  Then u . P = 0 up to rounding (covered by the
    // slack: |dP| <= 7 * 2^-24 |O'|, which moves the discriminant by <= 2^-20 S^2), the features are
    // (d_i d_j), 2 P and slack - |P|^2, and |P| is bounded by what f16 holds whenever the ray can meet a
    // filtered sphere at all (they lie within 120 of the origin): a line farther than 248 away gets
    // features that pass nothing.
    const float g = fmaf(uz, oz, fmaf(uy, oy, ux * ox));
    const float px = fmaf(-g, ux, ox), py = fmaf(-g, uy, oy), pz = fmaf(-g, uz, oz);
    const float pp = fmaf(pz, pz, fmaf(py, py, px * px));
    const bool apart = !(pp <= 61504.0f);                                        // 248^2; also catches NaN
    float feat[10];                                    // the factors of two of the mixed and linear terms are on the sphere side
    feat[0] = ux * ux; feat[1] = uy * uy; feat[2] = uz * uz;
    feat[3] = ux * uy; feat[4] = ux * uz; feat[5] = uy * uz;
    feat[6] = px; feat[7] = py; feat[8] = pz;
    feat[9] = fminf(slack - pp, 60000.0f);                                       // a slack beyond f16: everything passes
    if (apart) {
        // without the linear terms the sum is (u . C)^2 - |C|^2 + r^2 + constant <= r^2 + constant, r <= 120
        feat[6] = 0.0f; feat[7] = 0.0f; feat[8] = 0.0f;
        feat[9] = oo < 1e15f ? -20000.0f : 60000.0f;     // so far out that P itself is uncertain: everything passes
    }
#pragma unroll
    for (int f = 0; f < 10; f++) {
        // The feature must be an f32 VALUE before it is split.  Left visible, the multiply that produced it is
        // fused with the conversion into one mixed-precision FMA (one rounding, product -> f16) in one place and
        // not in another: two x1 that differ by an f16 ulp now and then, and x1 + x2 misses the feature by 2^-11 of it.
        asm volatile("" : "+v"(feat[f]));
    }
    // The two f16 pieces of a feature, x1 = f16(x) and x2 = f16(x - x1): the difference is exact in f32, so the
    // mixed-precision FMA that subtracts the f16 half of a register from x and writes the other half gives it in
    // one instruction next to the x1 it was taken from.
    auto pieces_lo_hi = [](float x) {                 // (x1, x2)
        uint32_t r = pack_f16((_Float16)x, (_Float16)x);
        asm("v_fma_mixhi_f16 %0, %1, 1.0, -%0 op_sel_hi:[0,0,1]" : "+v"(r) : "v"(x));
        return r;
    };
    auto pieces_hi_lo = [](float x) {                 // (x2, x1)
        uint32_t r = pack_f16((_Float16)x, (_Float16)x);
        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(r) : "v"(x));
        return r;
    };
    // K entries of the lane's own ray, two per register.  Feature f (f < 9) at entries 3 f .. 3 f + 2 as
    // (x1, x2, x1) against the sphere side's (c1, c1, c2); entries 27, 28: the constant's pieces against
    // 1, 1; entries 29, 30: -1, -1 against the pieces of k; entry 31: 0.
    uint32_t R[16];
#pragma unroll
    for (int k = 0; k < 4; k++) {                     // features 2 k and 2 k + 1 fill three registers
        R[3 * k] = pieces_lo_hi(feat[2 * k]);
        R[3 * k + 1] = pack_f16((_Float16)feat[2 * k], (_Float16)feat[2 * k + 1]);
        R[3 * k + 2] = pieces_hi_lo(feat[2 * k + 1]);
    }
    R[12] = pieces_lo_hi(feat[8]);
    R[13] = pack_f16((_Float16)feat[8], (_Float16)feat[9]);
    R[14] = 0xbc000000U;                              // (x2 of the constant, -1)
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(R[14]) : "v"(feat[9]), "v"(R[13]));
    R[15] = 0x0000bc00U;                              // (-1, 0)
    // B fragments: half H holds rays 32 H .. 32 H + 31 in its columns; lanes 0-31 supply K entries 0-7
    // of an instruction, lanes 32-63 entries 8-15, of the COLUMN's ray.  One half-swap per register
    // pair gives both halves' fragments.
    uint32_t B[2][2][4];
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const auto sw = __builtin_amdgcn_permlane32_swap(R[8 * s2 + v], R[8 * s2 + 4 + v], false, false);
            B[0][s2][v] = sw[0]; B[1][s2][v] = sw[1];
        }
#ifdef RT_DIAG
#ifdef RT_DIAG_STAMPS
#pragma unroll
    for (int k = 0; k < 16; k++) asm volatile("" : "+v"(B[k >> 3][(k >> 2) & 1][k & 3]));
#endif
#ifdef RT_DIAG_STAMPS
    dg.t_feat += RT_STAMP() - tb0 - (dg.t_big - t_big_before) - (tc1 - tc0);        // ray features and their fragments
#else
    (void)tb0; (void)t_big_before;
#endif
#endif
    RT_MARK("tile_loop_setup");
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int cnt = 0;                                             // entries waiting in the wave's list (wave-uniform)
    const uint32_t ebase0 = ((uint32_t)(lane & 31) << 16) | (uint32_t)(lane >> 5), ebase1 = ebase0 + (32U << 16);
    auto drain_full_steps = [&]() {
        while (cnt >= 64) {
            cnt -= 64;
            drain_pooled(s_geom, s_orig16, p.orig, w, cnt, 64, lane, q RT_DIAG_PASS);
#ifdef RT_X_DOUBLE_DRAIN       // marginal-cost experiment: every dense step of exact tests runs twice (lowering a key twice changes nothing)
            drain_pooled(s_geom, s_orig16, p.orig, w, cnt, 64, lane, q RT_DIAG_PASS);
#endif
        }
    };
    // Per tile: read the NEXT tile's fragments, issue this tile's four MFMAs (the two halves interleaved, so
    // that no instruction waits for the one before it), compare.  The fragment reads are issued BEFORE the MFMAs,
    // into a buffer of their own (two buffers used in turn); every register an MFMA reads stays untouched until
    // its accumulators have been read (the empty asm in compare() makes the register allocator guarantee it, and
    // tools/mfma_operand_lint.py checks the generated code), and no MFMA is in flight when the scan ends.
    // Two other forms of this loop are kept below as compile-time variants because an earlier version of this
    // comment blamed the wrong thing for what went wrong with them.  Measured on MI355X (profiles/r3_hazard/):
    // a load that lands in the A/B registers of MFMAs issued just before it does NOT change their result (the
    // "pipelined" form is bit-identical to this one over 3.3e9 tile products at four waves per SIMD), nor does a
    // change of EXEC behind an MFMA in flight or right ahead of its issue (tools/mfma_exec_probe.hip); what the
    // hardware does NOT wait for is a vector instruction's write directly ahead of the MFMA that reads the register
    // (tools/mfma_valu_raw_probe.hip: one instruction in between suffices; the compiler pads its own code).
    // (the read one tile ahead goes one tile beyond the image at the end of the scan: in LDS that is the geometry
    // and the wave regions behind the image, in global memory a spare tile the host allocates; it is never used)
    auto read_tile = [&](f16x8 (&x)[2], const uint4 *tile) {              // tile: the lane's entry of the tile's first fragment
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) x[s2] = __builtin_bit_cast(f16x8, tile[s2 * 64]);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto multiply = [&](const f16x8 (&x)[2], f32x16 &acc0, f32x16 &acc1) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[0], as_frag(B[0][0][0], B[0][0][1], B[0][0][2], B[0][0][3]), zero, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[0], as_frag(B[1][0][0], B[1][0][1], B[1][0][2], B[1][0][3]), zero, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[1], as_frag(B[0][1][0], B[0][1][1], B[0][1][2], B[0][1][3]), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[1], as_frag(B[1][1][0], B[1][1][1], B[1][1][2], B[1][1][3]), acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // Compares a tile's accumulators with zero and queues what passes.  `x` are the fragments the tile's MFMAs read:
    // the empty asm takes them as inputs and sits on the data path between the first value computed from the
    // accumulators and its use, so the register allocator must keep the fragment registers intact until the
    // accumulators have been read, which the hardware lets happen only when the MFMAs have finished (RT_NO_OPERAND_FENCE
    // removes it, for the record of what goes wrong without it).
    auto compare = [&](const f32x16 &acc0, const f32x16 &acc1, const f16x8 (&x)[2], int t) {
        // a half tile adds at most 1024 entries to the at most 63 that wait: kMatListCap
        const RowMaxima rm0 = row_maxima(acc0), rm1 = row_maxima(acc1);
        float any = __builtin_fmaxf(rm0.all, rm1.all);
#ifndef RT_NO_OPERAND_FENCE
        asm volatile("" : "+v"(any) : "v"(x[0]), "v"(x[1]));
#endif
        if (__ballot(any >= 0.0f) != 0ULL) {       // a tile whose box a ray meets may still hold no candidate at all
#ifdef RT_DIAG
            const unsigned long long ta0 = RT_STAMP(), drain_before = dg.t_drain;
#endif
#ifdef RT_X_DOUBLE_APPEND      // marginal-cost experiment (tools/marginal_costs.sh): the appends run twice, the same entries to the same slots
            { int again = cnt; queue_passes(acc0, rm0, t * kTileSpheres, ebase0, w.list, again RT_DIAG_PASS); }
#endif
            queue_passes(acc0, rm0, t * kTileSpheres, ebase0, w.list, cnt RT_DIAG_PASS);
            drain_full_steps();
#ifdef RT_X_DOUBLE_APPEND
            { int again = cnt; queue_passes(acc1, rm1, t * kTileSpheres, ebase1, w.list, again RT_DIAG_PASS); }
#endif
            queue_passes(acc1, rm1, t * kTileSpheres, ebase1, w.list, cnt RT_DIAG_PASS);
            drain_full_steps();
#ifdef RT_DIAG
            dg.t_append += RT_STAMP() - ta0 - (dg.t_drain - drain_before);
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    const uint4 *tile = s_aimg + lane;                 // one address register stepped per pair of tiles, constant offsets from it
#ifdef RT_TILE_LOOP_PIPELINED
    // The REJECTED form of the loop, kept compilable for the record and for the check of the generated code
    // (make variant NAME=pipelined DEF="-DRT_TILE_LOOP_PIPELINED -DRT_NO_OPERAND_FENCE"): one fragment buffer, the next
    // tile's LDS reads issued right after this tile's MFMAs.  The fragments are dead to the compiler once the MFMAs
    // are issued, so it loads the next tile into the very registers the MFMAs in flight still read.
    f16x8 x[2];
    read_tile(x, tile);
    for (int t = 0; t < p.ntiles; t++, tile += 128) {
        f32x16 acc0, acc1;
        multiply(x, acc0, acc1);
        f16x8 used[2] = {x[0], x[1]};
        read_tile(x, tile + 128);
        compare(acc0, acc1, used, t);
    }
#elif defined(RT_TILE_LOOP_ROLLING)
    // Another REJECTED form, kept compilable for the record (make variant NAME=rolling DEF="-DRT_TILE_LOOP_ROLLING
    // -DRT_NO_OPERAND_FENCE"): the two MFMAs of one ray half are issued before the other half's accumulators are compared
    // and queued.  In the two builds with IEEE arithmetic its images differ from the kept loop's and from run to run
    // (long paths at the end of a launch, runs of neighbouring lanes); the fast build's do not.  What was established
    // about it (profiles/r3_hazard/): the difference does not come from the matrix pipe -- it stays when every product
    // is followed by a wait that drains the pipe (RT_ROLLING_DRAINED) and goes away when the same pieces run in the
    // kept loop's order (RT_ROLLING_SERIAL).
    auto half_product = [&](const f16x8 (&x)[2], int h) {
        f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[0], as_frag(B[h][0][0], B[h][0][1], B[h][0][2], B[h][0][3]), zero, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[1], as_frag(B[h][1][0], B[h][1][1], B[h][1][2], B[h][1][3]), acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#ifdef RT_ROLLING_DRAINED
        asm volatile("s_sleep 8\n\ts_nop 15" : "+v"(acc));
#endif
        return acc;
    };
    auto compare_half = [&](const f32x16 &acc, uint32_t ebase, int t) {
        const RowMaxima rm = row_maxima(acc);
        if (__ballot(rm.all >= 0.0f) != 0ULL) {
            queue_passes(acc, rm, t * kTileSpheres, ebase, w.list, cnt RT_DIAG_PASS);
            drain_full_steps();
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    f16x8 a[2], b[2];
    read_tile(a, tile);
#ifdef RT_ROLLING_SERIAL
    for (int t = 0; t < p.ntiles; t++, tile += 128) {
        read_tile(b, tile + 128);
        const f32x16 h0 = half_product(a, 0);
        const f32x16 h1 = half_product(a, 1);
        compare_half(h0, ebase0, t);
        compare_half(h1, ebase1, t);
        a[0] = b[0]; a[1] = b[1];
    }
#else
    f32x16 acc0 = half_product(a, 0);
    for (int t = 0; t < p.ntiles; t++, tile += 128) {
        const f32x16 acc1 = half_product(a, 1);
        read_tile(b, tile + 128);
        compare_half(acc0, ebase0, t);
        acc0 = half_product(b, 0);
        compare_half(acc1, ebase1, t);
        a[0] = b[0]; a[1] = b[1];
    }
#endif
#else
    auto do_tile = [&](const f16x8 (&x)[2], int t) {
        f32x16 acc0, acc1;
#ifdef RT_X_DOUBLE_TILE        // marginal-cost experiment: every tile's products (1), maxima (2) or both (3) are computed twice
        {
            // the extra products permute B's lane groups (blgp), so that they are not the same expression as the real ones; the extra
            // maxima are taken of them.  Inputs of an empty asm keep them alive without copies.
            f32x16 g0, g1;
#if RT_X_DOUBLE_TILE != 2
            g0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[0], as_frag(B[0][0][0], B[0][0][1], B[0][0][2], B[0][0][3]), zero, 0, 0, 1);
            g1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[0], as_frag(B[1][0][0], B[1][0][1], B[1][0][2], B[1][0][3]), zero, 0, 0, 1);
            g0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[1], as_frag(B[0][1][0], B[0][1][1], B[0][1][2], B[0][1][3]), g0, 0, 0, 1);
            g1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[1], as_frag(B[1][1][0], B[1][1][1], B[1][1][2], B[1][1][3]), g1, 0, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
#endif
#if RT_X_DOUBLE_TILE == 1
            asm volatile("" :: "v"(g0), "v"(g1));
#elif RT_X_DOUBLE_TILE == 3
            const RowMaxima r0 = row_maxima(g0), r1 = row_maxima(g1);
            float sink = __builtin_fmaxf(r0.all, r1.all);
            asm volatile("" : "+v"(sink));
            if (__ballot(sink == 12345.0f) != 0ULL) asm volatile("s_nop 0");
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        multiply(x, acc0, acc1);
#ifdef RT_X_SHADOW_PAD         // marginal-cost experiment: RT_X_SHADOW_PAD independent v_fma_f32 of the same wave behind its four products, i.e. in
                               // the time it would otherwise wait for them
        {
            float d0 = ux, d1 = uy, d2 = uz, d3 = q.tmin;
#pragma unroll
            for (int k = 0; k < RT_X_SHADOW_PAD / 4; k++)
                asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
            asm volatile("" :: "v"(d0), "v"(d1), "v"(d2), "v"(d3));
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
#if defined(RT_X_DOUBLE_TILE) && RT_X_DOUBLE_TILE == 2
        {
            // the same number of maxima over the real accumulators, grouped differently (not the expressions compare() evaluates)
            auto shifted = [](const f32x16 &a) {
                float m[5];
#pragma unroll
                for (int k = 0; k < 5; k++) m[k] = __builtin_fmaxf(__builtin_fmaxf(a[3 * k + 1], a[3 * k + 2]), a[3 * k + 3]);
                return __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(m[0], m[1]), a[0]), __builtin_fmaxf(__builtin_fmaxf(m[2], m[3]), m[4]));
            };
            float sink = __builtin_fmaxf(shifted(acc0), shifted(acc1));
            asm volatile("" : "+v"(sink));
            if (__ballot(sink == 12345.0f) != 0ULL) asm volatile("s_nop 0");
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        compare(acc0, acc1, x, t);
    };
    f16x8 a[2], b[2];
    for (int t0 = 0; t0 < (ONE_GROUP ? 1 : p.ntiles); t0 += 64) {
        unsigned long long need = t0 == 0 ? need_first : ((met64 >> (t0 >> 6)) & 1U) != 0U ? tiles_in_need(t0) : 0ULL;
#ifdef RT_X_CULL_IGNORE        // marginal-cost experiment: the masks are computed and every tile is multiplied all the same
        asm volatile("" : "+s"(need));
        need |= (p.ntiles - t0 < 64) ? (1ULL << (p.ntiles - t0)) - 1ULL : ~0ULL;
#endif
#ifdef RT_DIAG
        dg.tiles_run += (uint32_t)__popcll(need);
#endif
        tiles_done += (uint32_t)__popcll(need);
        if (need == 0ULL) continue;
        // the tiles in need, two fragment buffers used in turn: the next tile's fragments are read before this tile's products are issued
        // (every step reads fragments, the last one the tile it already has: a read inside a branch would make the compiler wait
        // for ALL outstanding LDS reads ahead of the products, the ones just issued for the next tile included)
        auto next_tile = [&](int now) { const int k = need != 0ULL ? t0 + __ffsll((long long)need) - 1 : now; need &= need - 1ULL; return k; };
        int t = next_tile(0);
        read_tile(a, tile + t * 128);
        for (;;) {
            const bool more_b = need != 0ULL;
            const int t2 = next_tile(t);
            read_tile(b, tile + t2 * 128);
            do_tile(a, t);
            if (!more_b) break;
            const bool more_a = need != 0ULL;
            t = next_tile(t2);
            read_tile(a, tile + t * 128);
            do_tile(b, t2);
            if (!more_a) break;
        }
    }
#endif
#ifdef RT_DIAG
    dg.scans++;
#endif
    RT_MARK("scan_tail");
    if (cnt > 0) drain_pooled(s_geom, s_orig16, p.orig, w, 0, cnt, lane, q RT_DIAG_PASS);
#ifdef RT_X_DOUBLE_DRAIN
    if (cnt > 0) drain_pooled(s_geom, s_orig16, p.orig, w, 0, cnt, lane, q RT_DIAG_PASS);
#endif
    // the lane's own ray: what the pooled tests found against what the big spheres gave
    const unsigned long long key = w.keys[lane];
    const uint32_t low = (uint32_t)key;
    if (low != 0xffffffffU) {
        const float t = __uint_as_float((uint32_t)(key >> 32));
        const int sph = (int)(low & 0xffffU);
        if (t < q.closest) { q.closest = t; q.hit = sph; }
        else if (__builtin_expect(t == q.closest, 0)) { if (wins_tie(q, sph)) q.hit = sph; }
    }
}

// hit() for a wave in which only a few lanes still carry a ray (the end of a launch): the 64
// lanes take one ray at a time, test 64 spheres per step with the exact test and reduce to the
// nearest hit.  Same candidates, same test, same tie rule as the scan, hence the same result,
// at 1/64 of the scan's latency per ray.
__device__ __forceinline__ void shared_ray_hit(const TraceParams &p, const float4 *s_geom, int lane,
                                               unsigned long long live, Probe &q)
{
    while (live != 0ULL) {
        const int src = __ffsll((long long)live) - 1;
        live &= live - 1ULL;
        Probe t;
        t.ox = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q.ox), src));
        t.oy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q.oy), src));
        t.oz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q.oz), src));
        t.dx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q.dx), src));
        t.dy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q.dy), src));
        t.dz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q.dz), src));
        t.tmin = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q.tmin), src));
#if RT_STRICT
        t.a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q.a), src));
#endif
        t.closest = 3.402823466e+38f;
        t.hit = -1;
        t.orig = q.orig;
        for (int i = lane; i < p.n; i += 64) exact_test(s_geom, i, t);
        auto merge = [&](float oc, int oh) {
            if (oc < t.closest) { t.closest = oc; t.hit = oh; }
            else if (oc == t.closest && oh >= 0 && wins_tie(t, oh)) t.hit = oh;
        };
        // Butterfly over the wave without per-lane address registers (the kernel keeps none alive for this rare
        // path): lane ^ 32 through the half swap, lane ^ 16 .. 1 through the LDS crossbar's bit-mask mode.
        {
            const auto c = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, t.closest), __builtin_bit_cast(unsigned, t.closest), false, false);
            const auto h = __builtin_amdgcn_permlane32_swap((unsigned)t.hit, (unsigned)t.hit, false, false);
            merge(__builtin_bit_cast(float, lane < 32 ? c[1] : c[0]), (int)(lane < 32 ? h[1] : h[0]));
        }
#define RT_XOR_STEP(OFF) merge(__builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, t.closest), ((OFF) << 10) | 0x1f)), \
                               __builtin_amdgcn_ds_swizzle(t.hit, ((OFF) << 10) | 0x1f))
        RT_XOR_STEP(16); RT_XOR_STEP(8); RT_XOR_STEP(4); RT_XOR_STEP(2); RT_XOR_STEP(1);
#undef RT_XOR_STEP
        if (lane == src) { q.closest = t.closest; q.hit = t.hit; }
    }
}

struct Lane {
    // current ray (fast build: d is unit length and tmin carries the original length)
    V3 o, d;
    float tmin;
    bool zero_dir;
    V3 thr;             // path throughput, "color" in ray_color
    V3 acc;             // partial sum of the current unit
    Rng rng;
    uint32_t pixhash;   // mix32(pixel + mix32(seed ^ c)): the per-pixel part of the sample key
    uint32_t pixc;      // compact pixel index of the unit
    uint32_t x, y;      // image coordinates of that pixel
    int s, s_end;       // current / one-past-last sample index of the unit
    int chunk;
    int bounce;
};

template <int MODE>
__device__ __forceinline__ void set_ray(Lane &L, V3 o, V3 d)
{
    L.o = o;
#if RT_STRICT
    L.d = d;
    L.tmin = 0.001f;
    L.zero_dir = false;
#else
    float l2 = lensq(d);
    L.zero_dir = !(l2 > 0.0f);
    float inv = L.zero_dir ? 0.0f : rt_inv_sqrt(l2);
    L.d = scale(d, inv);
    L.tmin = 0.001f * (l2 * inv);     // 0.001 * |d|
#endif
}

// get_ray, Main.c:776-803
template <int MODE>
__device__ __forceinline__ void camera_ray(const TraceParams &p, Lane &L)
{
    const Camera &c = p.cam;
#if RT_STRICT && RT_DIRECT
    float offx, offy;
    jitter_direct<MODE>(L.rng, offx, offy);
#elif RT_STRICT
    float offx = rng_float<MODE>(L.rng) - 0.5f;
    float offy = rng_float<MODE>(L.rng) - 0.5f;
#else
    float offx, offy;
    rng_float2<MODE>(L.rng, offx, offy);
    offx -= 0.5f; offy -= 0.5f;
#endif
    float fx = (float)(int)L.x + offx;
    float fy = (float)(int)L.y + offy;
    V3 ps;
    ps.x = c.p00[0] + (fx * c.du[0]) + (fy * c.dv[0]);
    ps.y = c.p00[1] + (fx * c.du[1]) + (fy * c.dv[1]);
    ps.z = c.p00[2] + (fx * c.du[2]) + (fy * c.dv[2]);
    V3 org = v3(c.pos[0], c.pos[1], c.pos[2]);
    if (!(c.defocus_angle <= 0.0f)) {
#if RT_STRICT && RT_DIRECT
        V3 q = v3(0.0f, 0.0f, 0.0f);
        disk_point_direct<MODE>(L.rng, q.x, q.y);
#elif RT_STRICT
        V3 q = random_direction_2d<MODE>(L.rng);
#else
        V3 q = v3(0.0f, 0.0f, 0.0f);
        disk_point<MODE>(L.rng, q.x, q.y);
#endif
        V3 a = scale(v3(c.ddu[0], c.ddu[1], c.ddu[2]), q.x);
        V3 b = scale(v3(c.ddv[0], c.ddv[1], c.ddv[2]), q.y);
        org = add(org, add(a, b));
    }
    set_ray<MODE>(L, org, sub(ps, org));
}

// VARIANT (rt_device.h): kScanMatrix runs the filter on the matrix cores in one large workgroup per CU;
// kScanVectorGlobal serves scenes whose geometry image does not fit the LDS next to the queues: the
// per-sphere records (exact test, shading) come from global memory through the vector L1 instead.
template <int MODE, int VARIANT>
__global__ __launch_bounds__(is_matrix(VARIANT) ? kMatBlockThreads : kBlockThreads,
                             is_matrix(VARIANT) ? kMatBlockThreads / 256 : RT_MIN_WAVES) void trace_kernel(const TraceParams p)
{
    extern __shared__ float4 lds[];
    const int lane = (int)__lane_id();
    const float4 *s_geom, *s_matA, *s_matB;
    uint32_t *queue;
    const uint4 *s_aimg = nullptr;
    const uint16_t *s_orig16 = nullptr;
    const float4 *s_box = nullptr;
    char *pool = nullptr;
    if (VARIANT == kScanMatrix) {
        uint4 *img = reinterpret_cast<uint4 *>(lds);                       // [ntiles][2][64]
        for (int i = threadIdx.x; i < p.ntiles * 128; i += blockDim.x) img[i] = p.aimg[i];
        float4 *gm = lds + p.ntiles * 128;                                 // [n_pad]
        for (int i = threadIdx.x; i < p.n_pad; i += blockDim.x) gm[i] = p.geom[i];
        uint16_t *o16 = reinterpret_cast<uint16_t *>(gm + p.n_pad);        // [n_pad] caller's indices (below 65536 here)
        for (int i = threadIdx.x; i < p.n_pad; i += blockDim.x) o16[i] = (uint16_t)p.orig[i];
        float4 *bx = reinterpret_cast<float4 *>(o16 + p.n_pad);            // [pairs of tiles][3] tile boxes (n_pad is a multiple of 8: 16-byte aligned)
        const int nbox = ((p.ntiles + 1) >> 1) * 3;
        if (p.tile_boxes != nullptr)
            for (int i = threadIdx.x; i < nbox; i += blockDim.x) bx[i] = p.tile_boxes[i];
        s_aimg = img; s_geom = gm; s_orig16 = o16;
        s_box = p.tile_boxes != nullptr ? bx : nullptr;
        s_matA = p.matA; s_matB = p.matB;
        pool = reinterpret_cast<char *>(bx + nbox) + (threadIdx.x >> 6) * kMatWaveBytes;
        queue = nullptr;
    } else if (VARIANT == kScanMatrixGlobal) {
        s_aimg = p.aimg; s_geom = p.geom; s_matA = p.matA; s_matB = p.matB;
        s_box = p.tile_boxes;
        pool = reinterpret_cast<char *>(lds) + (threadIdx.x >> 6) * kMatWaveBytes;
        queue = nullptr;
    } else if (VARIANT == kScanVectorGlobal) {
        s_geom = p.geom; s_matA = p.matA; s_matB = p.matB;
        queue = reinterpret_cast<uint32_t *>(lds) + (threadIdx.x >> 6) * (kQueueSlots * 64);
    } else {
        for (int i = threadIdx.x; i < p.n_pad; i += blockDim.x) lds[i] = p.geom[i];
        s_geom = lds;                        // [n_pad]
        // The LDS goes to the geometry image and the candidate queues (more resident waves); the
        // material records are read once per bounce, from global memory through the vector L1.
        s_matA = p.matA; s_matB = p.matB;
        queue = reinterpret_cast<uint32_t *>(lds + p.n_pad) + (threadIdx.x >> 6) * (kQueueSlots * 64);
    }
    __syncthreads();

    const uint32_t total = (uint32_t)p.total_units;
    const double inv_per_block = 1.0 / (64.0 * (double)p.nchunks), inv_width = 1.0 / (double)p.width;
    Lane L;
    L.acc = v3(0.0f, 0.0f, 0.0f);
    L.thr = v3(1.0f, 1.0f, 1.0f);
    L.o = v3(0.0f, 0.0f, 0.0f);
    L.d = v3(0.0f, 0.0f, 1.0f);
    L.tmin = 0.001f;
    L.zero_dir = false;
    L.rng.ctr = 0; L.pixhash = 0;
    L.pixc = 0; L.x = 0; L.y = 0; L.s = 0; L.s_end = 0; L.chunk = 0; L.bounce = 0;
    bool active = true;        // lane still has (or may get) work
    bool need_unit = true;     // lane must pull a new work unit
    bool need_sample = true;   // lane must start a new sample
    uint32_t rays = 0;
    uint32_t wave_scans = 0;   // times this wave ran the sphere scan (wave-uniform)
    uint32_t tiles_done = 0;   // tiles it multiplied in those scans (wave-uniform)
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t pool_units = (uint32_t)p.pool_units;    // work units a wave takes from the global counter at a time
    const unsigned long long static_units = (unsigned long long)gridDim.x * waves_per_block * pool_units;
    const unsigned long long first = (unsigned long long)(blockIdx.x * waves_per_block + (threadIdx.x >> 6)) * pool_units;
    uint32_t pool_next = first < total ? (uint32_t)first : total;
    uint32_t pool_end = first + pool_units < total ? (uint32_t)(first + pool_units) : total;
    pool_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)pool_next);       // wave-uniform by construction
    pool_end = (uint32_t)__builtin_amdgcn_readfirstlane((int)pool_end);
#ifdef RT_DIAG
    Diag dg = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, {0, 0, 0, 0}, 0ULL, 0ULL, 0ULL, 0ULL, 0ULL, 0ULL, 0ULL, 0ULL, 0ULL};
    const unsigned long long t_wave0 = RT_STAMP();
    (void)t_wave0;
#endif

    for (;;) {
        // ---- pull work units ----------------------------------------------------------
        // The wave keeps a pool of pool_units consecutive units (wave-uniform registers) and
        // hands them to the lanes that finished one; the one global counter is touched once
        // per pool, not once per hand-out (thousands of waves on one address saturate the atomic
        // unit).  The first pool of every wave is static (units wave * pool_units ...), so a launch does
        // not open with one atomic per wave either; the counter hands out what lies beyond.
#ifdef RT_DIAG
        const unsigned long long ts0 = RT_STAMP();
#endif
        RT_MARK("handout");
        unsigned long long want = __ballot(active && need_unit);
        while (want != 0ULL) {
            if (pool_next == pool_end) {
                // (pools that shrink towards the end of a launch, half an even share of what is left, were measured and
                // dropped: every refill is a device-wide atomic, and the small launches they were meant for ran 3-4 % slower)
                unsigned long long b64 = 0;
                if (lane == 0) b64 = atomicAdd(&p.counters[0], (unsigned long long)pool_units) + static_units;
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b64);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b64 >> 32));
                if (hi != 0U || lo >= total) {              // no units left: the waiting lanes retire
                    if (need_unit) active = false;
                    need_unit = false;
                    break;
                }
                pool_next = lo;
                pool_end = (total - lo < pool_units) ? total : lo + pool_units;
            }
            const uint32_t avail = pool_end - pool_next;
            const uint32_t asked = (uint32_t)__popcll(want);
            const uint32_t rank = (uint32_t)__popcll(want & ((1ULL << lane) - 1ULL));
            if (active && need_unit && rank < avail) {
                // Unit u -> (pixel, chunk), pixel-block major: 64 consecutive pixels, all their chunks, then the
                // next 64 pixels.  The units in flight at any time (a quarter of a million) then cover a few
                // thousand pixels, so the pools a wave works on one after another lie next to each other in the
                // image and its rays keep meeting the same spheres; chunk-major order spread them over 68 rows
                // of the launch, which for one rank's share of a split frame is a quarter of the image.
                const uint32_t u = pool_next + rank;
                const uint32_t per_block = 64U * (uint32_t)p.nchunks;
                uint32_t blk, rem, rowk;
                divmod(u, per_block, inv_per_block, blk, rem);
                const uint32_t left = p.npix - blk * 64U;              // pixels of this block: 64, fewer in the last one
                if (left >= 64U) {
                    L.chunk = (int)(rem >> 6);
                    L.pixc = blk * 64U + (rem & 63U);
                } else {
                    L.chunk = (int)(rem / left);
                    L.pixc = blk * 64U + (rem - (uint32_t)L.chunk * left);
                }
                divmod(L.pixc, p.width, inv_width, rowk, L.x);
                L.y = p.rows ? p.rows[rowk] : part_row_at(rowk, p.tile_rows, p.part_index, p.part_count);
                L.pixhash = mix32((L.x + L.y * p.width) + mix32(p.seed ^ 0x6A09E667U));
                L.s = L.chunk * p.chunk_spp;
                int e = L.s + p.chunk_spp;
                L.s_end = e < p.spp ? e : p.spp;
                L.acc = v3(0.0f, 0.0f, 0.0f);
                need_unit = false;
            }
            pool_next += asked < avail ? asked : avail;
            want = __ballot(active && need_unit);
        }
        if (__ballot(active) == 0ULL) break;
#ifdef RT_DIAG
        const unsigned long long ts1 = RT_STAMP();
        dg.t_handout += ts1 - ts0;
#endif

        RT_MARK("new_sample");
        // ---- start a new sample where the previous path ended ------------------------
        if (active && need_sample) {
            L.rng.ctr = mix32(L.pixhash ^ ((p.sample_base + (uint32_t)L.s) * 0x9E3779B9U));
            camera_ray<MODE>(p, L);
            L.thr = v3(1.0f, 1.0f, 1.0f);
            L.bounce = 0;
            need_sample = false;
        }
#ifdef RT_DIAG
        const unsigned long long ts2 = RT_STAMP();
        dg.t_regen += ts2 - ts1;
#endif
        RT_MARK("hit");
        // ---- hit(): nearest sphere along the ray, Main.c:688-713 -----------------------
        Probe q;
        q.ox = L.o.x; q.oy = L.o.y; q.oz = L.o.z;
        q.dx = L.d.x; q.dy = L.d.y; q.dz = L.d.z;
        q.tmin = L.tmin;
        q.closest = 3.402823466e+38f;          // max_f32, Main.c:734
        q.hit = -1;
        q.orig = p.orig;
#if RT_STRICT
        {
            // The filter needs a unit direction only to stay conservative, so hardware rsqrt
            // is good enough; the exact test uses d as it is.  slack 2^-17 (|O|+|C|)^2 covers
            // the filter's rounding, the rsqrt error and the rounding of the reference-order
            // discriminant h*h - a*c that decides in exact_test.
            q.a = q.dx * q.dx + q.dy * q.dy + q.dz * q.dz;
            const float inv = (q.a > 0.0f) ? __builtin_amdgcn_rsqf(q.a) : 0.0f;
            const unsigned long long live = __ballot(active);
            if (__popcll(live) <= p.coop_max) shared_ray_hit(p, s_geom, lane, live, q);
            else {
                wave_scans++;
                if (is_matrix(VARIANT)) scan_spheres_matrix<VARIANT == kScanMatrix>(p, s_geom, s_aimg, s_orig16, s_box, pool, lane, active, q.dx * inv, q.dy * inv, q.dz * inv, q, tiles_done RT_DIAG_PASS);
                else scan_spheres(p, s_geom, queue, lane, q.dx * inv, q.dy * inv, q.dz * inv, 7.6293945e-6f, q RT_DIAG_PASS);
            }
            // A zero or NaN direction gives the filter nothing to work with, and the exact tests above divide by the
            // division's core, which is IEEE for 2^-40 <= a <= 2^40, an origin within 2^24 of the frame's origin and spheres
            // within 2^24 (p.core_div_ok, from the host).  A ray outside that is tested again, against everything, with the
            // compiler's division.
            if (!(q.a >= 0x1p-40f && q.a <= 0x1p40f && __builtin_fmaxf(__builtin_fmaxf(fabsf(q.ox), fabsf(q.oy)), fabsf(q.oz)) <= 0x1p24f
                  && p.core_div_ok != 0)) {
                q.closest = 3.402823466e+38f; q.hit = -1;
                for (int i = 0; i < p.n; i++) exact_test<false>(s_geom, i, q);
            }
        }
#else
        const unsigned long long live = __ballot(active);
        if (__popcll(live) <= p.coop_max) shared_ray_hit(p, s_geom, lane, live, q);
        else {
            wave_scans++;
            if (is_matrix(VARIANT)) scan_spheres_matrix<VARIANT == kScanMatrix>(p, s_geom, s_aimg, s_orig16, s_box, pool, lane, active, q.dx, q.dy, q.dz, q, tiles_done RT_DIAG_PASS);
            else scan_spheres(p, s_geom, queue, lane, q.dx, q.dy, q.dz, 1.9073486e-6f, q RT_DIAG_PASS);   // 2^-19
        }
        if (L.zero_dir) q.hit = -1;
#endif
#ifdef RT_DIAG
        {
            const int nlive = (int)__popcll(__ballot(active));
            if (nlive <= p.coop_max) dg.shared_rays += (uint32_t)nlive;
            else {
                dg.scan_lanes += (uint32_t)nlive;
                if (nlive <= 8) dg.scans_by_live[0]++; else if (nlive <= 16) dg.scans_by_live[1]++;
                else if (nlive <= 32) dg.scans_by_live[2]++; else if (nlive <= 48) dg.scans_by_live[3]++;
            }
        }
#endif
#ifdef RT_DIAG
        const unsigned long long ts3 = RT_STAMP();
        if (__popcll(__ballot(active)) <= p.coop_max) dg.t_shared += ts3 - ts2; else dg.t_hit += ts3 - ts2;
#endif
#ifdef RT_DIAG_BRUTE
        if (is_matrix(VARIANT) && p.tile_bounds != nullptr && __popcll(__ballot(active)) > p.coop_max) {
            // how many tiles would a coarser test let the wave skip?  A tile counts when the line of some live ray
            // passes its bounding sphere ahead of the origin (or starts inside it)
            auto passes = [&](const float4 bs) {
                if (bs.w < 0.0f) return false;
                const float ocx = bs.x - q.ox, ocy = bs.y - q.oy, ocz = bs.z - q.oz;
                const float dd = q.dx * q.dx + q.dy * q.dy + q.dz * q.dz;
                const float h = q.dx * ocx + q.dy * ocy + q.dz * ocz;
                const float c = ocx * ocx + ocy * ocy + ocz * ocz - bs.w * bs.w;
                return active && h * h - dd * c >= 0.0f && (h > 0.0f || c < 0.0f);
            };
            for (int t = 0; t < p.ntiles; t++) {
                // ... and per half of the wave's rays (the 32 columns of one matrix product): (tile, ray half) pairs of 2 per tile,
                // (16-sphere half tile, ray half) pairs of 4 per tile
                const unsigned long long m = __ballot(passes(p.tile_bounds[5 * t]));
                if (m != 0ULL) dg.tiles_bound++;
                dg.tiles_halfbound += ((uint32_t)m != 0U) + ((uint32_t)(m >> 32) != 0U);
                const unsigned long long m1 = __ballot(passes(p.tile_bounds[5 * t + 1])), m2 = __ballot(passes(p.tile_bounds[5 * t + 2]));
                dg.quads_bound += ((uint32_t)m1 != 0U) + ((uint32_t)(m1 >> 32) != 0U) + ((uint32_t)m2 != 0U) + ((uint32_t)(m2 >> 32) != 0U);
                // the same with the tile's axis-aligned box (slab test; NaNs count as a pass)
                const float4 blo = p.tile_bounds[5 * t + 3], bhi = p.tile_bounds[5 * t + 4];
                bool in_box = false;
                if (blo.w > 0.0f && active) {
                    const float ax0 = (blo.x - q.ox) / q.dx, ax1 = (bhi.x - q.ox) / q.dx, ay0 = (blo.y - q.oy) / q.dy, ay1 = (bhi.y - q.oy) / q.dy;
                    const float az0 = (blo.z - q.oz) / q.dz, az1 = (bhi.z - q.oz) / q.dz;
                    const float tn = fmaxf(fmaxf(fminf(ax0, ax1), fminf(ay0, ay1)), fmaxf(fminf(az0, az1), 0.0f));
                    const float tf = fminf(fminf(fmaxf(ax0, ax1), fmaxf(ay0, ay1)), fmaxf(az0, az1));
                    in_box = !(tf < tn);
                }
                const unsigned long long mb = __ballot(in_box);
                dg.tiles_box += (mb != 0ULL);
                dg.pairs_box += ((uint32_t)mb != 0U) + ((uint32_t)(mb >> 32) != 0U);
            }
        }
        {
            // Diagnostic build: the exact test over ALL spheres, in index order, must agree with what
            // the filter, the queues and the drain (or the shared-ray search) found for this ray.
            Probe all = q;
            all.closest = 3.402823466e+38f;
            all.hit = -1;
            for (int i = 0; i < p.n; i++) exact_test<false>(s_geom, i, all);     // (with the compiler's division: also checks the short form)
#if !RT_STRICT
            if (L.zero_dir) all.hit = -1;
#endif
            if (active && (all.hit != q.hit || (all.hit >= 0 && all.closest != q.closest))) {
                dg.lost_hits++;
                if (dg.lost_hits <= 2)
                    printf("lost hit: o %.9g %.9g %.9g d %.9g %.9g %.9g: all-spheres test %d at %.9g, scan %d at %.9g; lane %d block %d live %d bounce %d scans %u drains %u\n",
                           (double)q.ox, (double)q.oy, (double)q.oz, (double)q.dx, (double)q.dy, (double)q.dz,
                           all.hit, (double)all.closest, q.hit, (double)q.closest, lane, (int)blockIdx.x,
                           (int)__popcll(__ballot(active)), L.bounce, dg.scans, dg.drain_iters);
            }
        }
#endif
        const float closest = q.closest;
        const int hit = q.hit;

        RT_MARK("shade");
        // ---- shade: ray_color's loop body, Main.c:730-759 ------------------------------
        if (active) {
            rays++;
            if (p.hit_index != nullptr && L.bounce == 0 && L.s == p.spp - 1) p.hit_index[L.pixc] = hit >= 0 ? p.orig[hit] : -1;
            bool done = false;
            V3 add_col = v3(0.0f, 0.0f, 0.0f);
#if RT_STRICT
            // The three materials and the sky each normalise one vector with an IEEE square root and divide (the
            // reflected direction for metal, the incoming direction for glass and for the sky): the lanes of all four
            // cases do it together, once, instead of in four places one after another; likewise the one random
            // direction that Lambertian and metal draw (same draws, same order, same values).
            int mtype = -1;
            float4 mA = make_float4(0.0f, 0.0f, 0.0f, 0.0f), mB = mA;
            V3 P = L.o, n = v3(0.0f, 0.0f, 0.0f);
            bool front = false;
            if (hit >= 0) {
                const float4 g = s_geom[hit];
                mA = s_matA[hit];
                mB = s_matB[hit];
                const V3 ctr = v3(g.x, g.y, g.z);
                P = add(L.o, scale(L.d, closest));                          // RAY_AT
                const V3 outward = scale(sub(P, ctr), mB.z);                // * (1.0f / radius)
                front = dot(L.d, outward) < 0.0f;                           // set_face_normal
                n = front ? outward : scale(outward, -1.0f);
                mtype = __float_as_int(mB.y);
            }
            V3 rd = v3(0.0f, 0.0f, 0.0f);
#if RT_DIRECT
            if (mtype == 0 || mtype == 1) rd = sphere_direction_direct<MODE>(L.rng);
#else
            if (mtype == 0 || mtype == 1) rd = random_direction<MODE>(L.rng);
#endif
            V3 normed = v3(0.0f, 0.0f, 0.0f);
            if (mtype != 0) normed = unit(mtype == 1 ? reflect(L.d, n) : L.d);
            if (hit >= 0) {
                V3 nd;
                bool attenuate = true;
                V3 att = v3(mA.x, mA.y, mA.z);
                if (mtype == 0) {                                           // Lambertian, Main.c:562-573
                    nd = add(n, rd);
                    const float s = 1e-8f;
                    if (fabsf(nd.x) < s && fabsf(nd.y) < s && fabsf(nd.z) < s) nd = n;
                } else if (mtype == 1) {                                    // Metal, Main.c:575-581
                    nd = add(normed, scale(rd, mA.w));
                    attenuate = dot(nd, n) > 0.0f;
                } else {                                                    // Dielectric, Main.c:583-611
                    att = v3(1.0f, 1.0f, 1.0f);
                    const float ri = front ? mB.w : mB.x;                  // 1.0f / ior, divided on the host (the same IEEE quotient)
                    const V3 ud = normed;
                    const float cos_theta = fminf(dot(scale(ud, -1.0f), n), 1.0f);
                    const float sin_theta = rt_sqrt_not_tiny(1.0f - cos_theta * cos_theta);      // zero, or at least 2^-24
                    bool refl = ri * sin_theta > 1.0f;
                    if (!refl) refl = reflectance(cos_theta, front ? mA.x : mA.y) > rng_float<MODE>(L.rng);
                    nd = refl ? reflect(ud, n) : refract(ud, n, ri);
                }
                if (attenuate) L.thr = mul(L.thr, att);                     // Main.c:738-742
                set_ray<MODE>(L, P, nd);
                L.bounce++;
                if (L.bounce >= p.max_depth) done = true;                   // Main.c:761: black
#else
            if (hit >= 0) {
                // Same scatter rules (Main.c:558-620) for a unit-length incoming direction, with
                // the three materials evaluated side by side and selected, so that a wave with
                // mixed materials does not run three code paths one after another.
                const float4 g = s_geom[hit];
                const float4 mA = s_matA[hit];
                const float4 mB = s_matB[hit];
                const int mtype = __float_as_int(mB.y);
                const V3 P = add(L.o, scale(L.d, closest));
                const V3 outward = scale(sub(P, v3(g.x, g.y, g.z)), mB.z);
                const float dn = dot(L.d, outward);
                const bool front = dn < 0.0f;
                const V3 n = front ? outward : scale(outward, -1.0f);
                const float cosi = fabsf(dn);                               // -dot(d, n)
                const V3 refl = add(L.d, scale(n, 2.0f * cosi));            // vec3f_reflect(d, n)
                V3 rnd = v3(0.0f, 0.0f, 0.0f);
                float u = 0.0f;
                if (mtype != 2) rnd = sphere_direction<MODE>(L.rng);
                else u = rng_float<MODE>(L.rng);
                // Lambertian
                V3 nd_l = add(n, rnd);
                const float eps = 1e-8f;
                if (fabsf(nd_l.x) < eps && fabsf(nd_l.y) < eps && fabsf(nd_l.z) < eps) nd_l = n;
                // Metal
                const V3 nd_m = add(refl, scale(rnd, mA.w));                // |refl| = 1: d and n are unit vectors
                // Dielectric
                const float ri = front ? mB.w : mB.x;                       // 1 / ior from the host
                const float cos_theta = fminf(cosi, 1.0f);
                const float sin_theta = rt_sqrt(fmaxf(0.0f, fmaf(-cos_theta, cos_theta, 1.0f)));
                const bool reflects = (ri * sin_theta > 1.0f) || (reflectance(cos_theta, front ? mA.x : mA.y) > u);
                const V3 perp = scale(add(L.d, scale(n, cos_theta)), ri);
                const V3 par = scale(n, -rt_sqrt(fabsf(1.0f - lensq(perp))));
                const V3 nd_d = reflects ? refl : add(perp, par);
                V3 nd = (mtype == 0) ? nd_l : ((mtype == 1) ? nd_m : nd_d);
                const bool attenuate = (mtype != 1) || (dot(nd_m, n) > 0.0f);
                const V3 att = (mtype == 2) ? v3(1.0f, 1.0f, 1.0f) : v3(mA.x, mA.y, mA.z);
                if (attenuate) L.thr = mul(L.thr, att);
                set_ray<MODE>(L, P, nd);
                L.bounce++;
                if (L.bounce >= p.max_depth) done = true;
#endif
            } else {                                                        // sky, Main.c:744-758
#if RT_STRICT
                const V3 ud = normed;
#else
                const V3 ud = L.d;                                          // already unit length (or zero)
#endif
                const float t = 0.5f * (ud.y + 1.0f);
                const float w = 1.0f - t;
                const V3 sky = v3(1.0f * w + 0.5f * t, 1.0f * w + 0.7f * t, 1.0f * w + 1.0f * t);
                add_col = mul(L.thr, sky);
                done = true;
            }
            if (done) {
                L.acc = add(L.acc, add_col);                                // Main.c:975
                L.s++;
                need_sample = true;
                if (L.s >= L.s_end) {
                    p.partial[(size_t)L.chunk * p.npix + L.pixc] = make_float4(L.acc.x, L.acc.y, L.acc.z, 0.0f);
                    need_unit = true;
                }
            }
        }
#ifdef RT_DIAG
        dg.t_shade += RT_STAMP() - ts3;
#endif
#ifdef RT_X_PAD_VALU           // marginal-cost experiment: RT_X_PAD_VALU more vector instructions per iteration (RT_X_PAD_KIND 0: one dependent
                               // chain of v_fma_f32, 1: four independent chains, 2: v_max3_f32, four chains)
        {
            float d0 = L.o.x, d1 = L.o.y, d2 = L.o.z, d3 = L.tmin;
#pragma unroll
            for (int k = 0; k < RT_X_PAD_VALU / 4; k++) {
#if RT_X_PAD_KIND == 0
                asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0" : "+v"(d0));
#elif RT_X_PAD_KIND == 1
                asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
#else
                asm volatile("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %1, %1, %2, %3\n\tv_max3_f32 %2, %2, %3, %0\n\tv_max3_f32 %3, %3, %0, %1" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
#endif
            }
            asm volatile("" :: "v"(d0), "v"(d1), "v"(d2), "v"(d3));
        }
#endif
    }

    RT_MARK("epilogue");
    // ---- statistics: rays cast, sphere scans run by the waves ------------------------------
    unsigned long long r64 = rays;
    for (int off = 32; off > 0; off >>= 1) r64 += __shfl_down(r64, off);
    if (__lane_id() == 0) {
        atomicAdd(&p.counters[1], r64);
        atomicAdd(&p.counters[19], (unsigned long long)wave_scans);
        atomicAdd(&p.counters[22], (unsigned long long)tiles_done);
    }
#ifdef RT_DIAG
    if (__lane_id() == 0) {
        atomicAdd(&p.counters[2], (unsigned long long)dg.branches);
        atomicAdd(&p.counters[3], (unsigned long long)dg.drain_lanes);
        atomicAdd(&p.counters[4], (unsigned long long)dg.drain_iters);
        atomicAdd(&p.counters[5], (unsigned long long)dg.scans);
        atomicAdd(&p.counters[7], (unsigned long long)dg.passes);
        atomicAdd(&p.counters[6], ((unsigned long long)dg.shared_rays << 40) | (unsigned long long)dg.scan_lanes);
#ifndef RT_DIAG_STAMPS
        atomicAdd(&p.counters[9], (unsigned long long)dg.tiles_bound);
        atomicAdd(&p.counters[10], (unsigned long long)dg.tiles_halfbound);
        atomicAdd(&p.counters[15], (unsigned long long)dg.quads_bound);
        atomicAdd(&p.counters[16], (unsigned long long)dg.tiles_box);
        atomicAdd(&p.counters[17], (unsigned long long)dg.pairs_box);
        atomicAdd(&p.counters[18], (unsigned long long)dg.tiles_run);
        for (int k = 0; k < 4; k++) atomicAdd(&p.counters[11 + k], (unsigned long long)dg.scans_by_live[k]);
#endif
    }
    {
        unsigned long long lost = dg.lost_hits;
        for (int off = 32; off > 0; off >>= 1) lost += __shfl_down(lost, off);
        if (__lane_id() == 0 && lost != 0ULL) atomicAdd(&p.counters[8], lost);
    }
#ifdef RT_DIAG_STAMPS
    if (__lane_id() == 0) {
        atomicAdd(&p.counters[9], dg.t_handout);
        atomicAdd(&p.counters[10], dg.t_regen);
        atomicAdd(&p.counters[11], dg.t_hit - dg.t_drain - dg.t_big - dg.t_feat - dg.t_append - dg.t_cull);      // the tile loop: products and compares
        atomicAdd(&p.counters[23], dg.t_cull);
        atomicAdd(&p.counters[20], dg.t_feat);
        atomicAdd(&p.counters[21], dg.t_append);
        atomicAdd(&p.counters[12], dg.t_drain);
        atomicAdd(&p.counters[13], dg.t_big);
        atomicAdd(&p.counters[14], dg.t_shade);
        atomicAdd(&p.counters[15], dg.t_shared);
        const unsigned long long life = RT_STAMP() - t_wave0;
        atomicAdd(&p.counters[16], life);
        atomicMax(&p.counters[17], life);
        atomicAdd(&p.counters[18], 1ULL);
    }
#endif
#endif
}

}  // namespace RT_NS

// The instantiation for (mode, variant); RT_KERNEL_FOR names this build's selector (rt_device.h).
trace_fn RT_KERNEL_FOR(int mode, int variant)
{
    if (variant == kScanMatrix) return mode == 1 ? RT_NS::trace_kernel<1, kScanMatrix> : RT_NS::trace_kernel<0, kScanMatrix>;
    if (variant == kScanMatrixGlobal) return mode == 1 ? RT_NS::trace_kernel<1, kScanMatrixGlobal> : RT_NS::trace_kernel<0, kScanMatrixGlobal>;
    if (variant == kScanVectorGlobal) return mode == 1 ? RT_NS::trace_kernel<1, kScanVectorGlobal> : RT_NS::trace_kernel<0, kScanVectorGlobal>;
    return mode == 1 ? RT_NS::trace_kernel<1, kScanVectorLds> : RT_NS::trace_kernel<0, kScanVectorLds>;
}

}  // namespace rtmi
