/*
 * refcpu.c -- CPU restatement of the reference's path-tracing hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see refcpu.h).  Plain C, one translation unit, every
 * operation in the order the reference writes it, compiled with
 * -O2 -ffp-contract=off so that each f32 operation rounds once, as in the
 * reference built by oracle/build_ref.sh.
 *
 * Parity pin: REFCPU_RNG_LCG_TILES output == the compiled reference's render_tile
 * output, byte for byte (tests/test_oracle_vs_golden.py, fixtures in tests/golden/).
 *
 * Citations are file:line under /root/reference.
 */
#include "refcpu.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

typedef struct { float x, y, z; } v3;

/* ---- vec3 helpers: src/util.c:83-220 ------------------------------------- */
static inline v3 v3_add(v3 a, v3 b) { v3 r = { a.x + b.x, a.y + b.y, a.z + b.z }; return r; }   /* util.c:83  */
static inline v3 v3_sub(v3 a, v3 b) { v3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }   /* util.c:89  */
static inline v3 v3_mul(v3 a, v3 b) { v3 r = { a.x * b.x, a.y * b.y, a.z * b.z }; return r; }   /* util.c:95  */
static inline v3 v3_scale(v3 v, float s) { v3 r = { v.x * s, v.y * s, v.z * s }; return r; }    /* util.c:101 */
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }             /* util.c:107 */
static inline float v3_lensq(v3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }                 /* util.c:122 */
static inline v3 v3_cross(v3 a, v3 b)                                                           /* util.c:112 */
{
    v3 r = { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
    return r;
}
static inline v3 v3_unit(v3 v)                                                                  /* util.c:132 */
{
    float len = sqrtf(v3_lensq(v));
    if (len > 0.0f) {
        float inv = 1.0f / len;
        v3 r = { v.x * inv, v.y * inv, v.z * inv };
        return r;
    }
    v3 z = { 0.0f, 0.0f, 0.0f };
    return z;
}
static inline v3 v3_lerp(v3 a, v3 b, float t)                                                   /* util.c:154 */
{
    return v3_add(v3_scale(a, 1.0f - t), v3_scale(b, t));
}
static inline int v3_near_zero(v3 v)                                                            /* util.c:202 */
{
    float s = (float)1e-8;
    return (fabsf(v.x) < s) && (fabsf(v.y) < s) && (fabsf(v.z) < s);
}
static inline v3 v3_reflect(v3 v, v3 n)                                                         /* util.c:208 */
{
    return v3_sub(v, v3_scale(n, 2.0f * v3_dot(v, n)));
}
static inline v3 v3_refract(v3 v, v3 n, float e)                                                /* util.c:214 */
{
    float cos_theta = (float)fmin((double)v3_dot(v3_scale(v, -1.0f), n), 1.0);
    v3 perp = v3_scale(v3_add(v, v3_scale(n, cos_theta)), e);
    v3 par = v3_scale(n, -1.0f * sqrtf(fabsf(1.0f - v3_lensq(perp))));
    return v3_add(perp, par);
}

/* ---- random numbers ------------------------------------------------------- */
typedef struct {
    int      mode;      /* REFCPU_RNG_* */
    uint32_t lcg;       /* g_seed, util.c:27 */
    uint32_t key;       /* counter mode: per (seed, pixel, sample) */
    uint32_t n;         /* counter mode: draw number */
} rng_t;

static inline uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

uint32_t refcpu_counter_key(uint32_t seed, uint32_t pixel_index, uint32_t sample_index)
{
    uint32_t h1 = mix32(seed ^ 0x6A09E667U);
    uint32_t h2 = mix32(pixel_index + h1);
    return mix32(h2 ^ (sample_index * 0x9E3779B9U));
}

uint32_t refcpu_counter_draw(uint32_t key, uint32_t n)
{
    return mix32(key + n * 0x9E3779B9U);
}

/* RAND_FLOAT, include/util.h:93: (f32)fast_rand() / ((f32)FRAND_MAX + 1.0f) */
static inline float rng_float(rng_t *r)
{
    switch (r->mode) {
    case REFCPU_RNG_COUNTER_DIRECT:
    case REFCPU_RNG_COUNTER: {
        uint32_t x = refcpu_counter_draw(r->key, r->n++);
        return (float)(x >> 8) * (1.0f / 16777216.0f);
    }
    case REFCPU_RNG_CONST_U:
        return (float)16384 / ((float)32767 + 1.0f);
    default:                                                         /* util.c:35-39 */
        r->lcg = 214013u * r->lcg + 2531011u;
        return (float)(int)((r->lcg >> 16) & 0x7FFF) / ((float)32767 + 1.0f);
    }
}

/* RAND_FLOAT_RANGE(min,max), include/util.h:94 */
static inline float rng_range(rng_t *r, float mn, float mx) { return mn + (mx - mn) * rng_float(r); }

/* vec3f_random_direction, util.c:170-184 */
static v3 random_direction(rng_t *r)
{
    for (size_t i = 0; i < 100; i++) {
        v3 p;
        p.x = rng_range(r, -1.0f, 1.0f);
        p.y = rng_range(r, -1.0f, 1.0f);
        p.z = rng_range(r, -1.0f, 1.0f);
        float lensq = v3_lensq(p);
        if (1e-15 < (double)lensq && lensq <= 1)
            return v3_scale(p, 1.0f / sqrtf(lensq));
    }
    v3 z = { 0.0f, 0.0f, 0.0f };
    return z;
}

/* vec3f_random_direction_2d, util.c:186-200 */
static v3 random_direction_2d(rng_t *r)
{
    for (size_t i = 0; i < 100; i++) {
        v3 p;
        p.x = rng_range(r, -1.0f, 1.0f);
        p.y = rng_range(r, -1.0f, 1.0f);
        p.z = 0.0f;
        float lensq = v3_lensq(p);
        if (1e-15 < (double)lensq && lensq <= 1)
            return p;
    }
    v3 z = { 0.0f, 0.0f, 0.0f };
    return z;
}

/* ---- loop-free samplers of the GPU path's exact build (REFCPU_RNG_COUNTER_DIRECT) ------------
 * They replace the two rejection samplers above (util.c:170-200) with maps of the same
 * distributions.  Definition (include/rt.h, RT_BUILD_EXACT): one 32-bit draw gives two 16-bit
 * fields; every conversion and scaling below is exact in f32, sqrtf is IEEE, and each polynomial
 * step is a single rounded operation or an fmaf, so that any IEEE machine gets the same bits. */

/* cos and sin of 2 pi i / 65536: the nearest quarter turn is taken off in integers, the rest lies
 * within +-pi/4 (Cephes single-precision kernels), quarter turns go back by swap and sign. */
static inline void sincos_turns16(uint32_t i, float *c, float *s)
{
    uint32_t q = (i + 8192u) >> 14;
    float th = (float)((int)i - (int)(q << 14)) * 9.58737992428525768e-5f;
    float z = th * th;
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    float s0 = fmaf(ps * z, th, th);
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    float c0 = fmaf(pc, z * z, fmaf(-0.5f, z, 1.0f));
    float c1 = (q & 1u) ? s0 : c0, s1 = (q & 1u) ? c0 : s0;
    *c = ((q + 1u) & 2u) ? -c1 : c1;
    *s = (q & 2u) ? -s1 : s1;
}

static inline uint32_t rng_bits(rng_t *r) { return refcpu_counter_draw(r->key, r->n++); }

static v3 sphere_direction_direct(rng_t *r)
{
    uint32_t x = rng_bits(r);
    float z = fmaf((float)(x >> 16), -1.0f / 32768.0f, 65535.0f / 65536.0f);
    float rad = sqrtf(fmaf(-z, z, 1.0f));
    float c, s;
    sincos_turns16(x & 0xffffu, &c, &s);
    v3 d = { rad * c, rad * s, z };
    return d;
}

static v3 disk_point_direct(rng_t *r)
{
    uint32_t x = rng_bits(r);
    float rad = sqrtf(fmaf((float)(x >> 16), 1.0f / 65536.0f, 1.0f / 131072.0f));
    float c, s;
    sincos_turns16(x & 0xffffu, &c, &s);
    v3 p = { rad * c, rad * s, 0.0f };
    return p;
}

static inline v3 scatter_direction(rng_t *r)
{
    return r->mode == REFCPU_RNG_COUNTER_DIRECT ? sphere_direction_direct(r) : random_direction(r);
}

/* ---- scene / hit ---------------------------------------------------------- */
typedef struct { v3 orig, dir; } ray_t;                              /* Main.c:16-20 */

typedef struct {                                                     /* Main.c:61-68 */
    v3    p, norm;
    float t;
    int   front;
    const rt_sphere *s;
} hit_t;

typedef struct {
    const rt_camera *cam;
    const rt_sphere *sph;
    size_t n;
    uint32_t W, H;
    int spp, depth;
    const refcpu_opts *o;
    uint8_t *rgba;
} job_t;

static inline v3 V(rt_vec3 a) { v3 r = { a.x, a.y, a.z }; return r; }

/* hit_sphere, Main.c:625-686 */
static inline int hit_sphere(const rt_sphere *s, const ray_t *ray, float tmin, float tmax, hit_t *rec)
{
    v3 oc = v3_sub(V(s->center), ray->orig);
    float a = v3_lensq(ray->dir);
    float h = v3_dot(ray->dir, oc);
    float c = v3_lensq(oc) - s->radius * s->radius;
    float disc = h * h - a * c;
    if (disc < 0) return 0;
    float sq = sqrtf(disc);
    float root = (h - sq) / a;
    if (!(tmin < root && root < tmax)) {                             /* Surrounds, util.h:109 */
        root = (h + sq) / a;
        if (!(tmin < root && root < tmax)) return 0;
    }
    rec->t = root;
    rec->p = v3_add(ray->orig, v3_scale(ray->dir, root));            /* RAY_AT, Main.c:22 */
    v3 outward = v3_scale(v3_sub(rec->p, V(s->center)), 1.0f / s->radius);
    rec->front = v3_dot(ray->dir, outward) < 0;                      /* set_face_normal, Main.c:534 */
    rec->norm = rec->front ? outward : v3_scale(outward, -1.0f);
    rec->s = s;
    return 1;
}

/* hit, Main.c:688-713; *index gets the winning object's position in the array */
static inline int hit_world(const job_t *j, const ray_t *ray, hit_t *rec, int *index)
{
    hit_t tmp;
    int any = 0;
    float closest = FLT_MAX;                                          /* max_f32, Main.c:734 */
    for (size_t i = 0; i < j->n; i++) {
        if (hit_sphere(&j->sph[i], ray, 0.001f, closest, &tmp)) {
            any = 1;
            closest = tmp.t;
            *rec = tmp;
            *index = (int)i;
        }
    }
    return any;
}

/* reflectance, Main.c:551-556 */
static inline float reflectance(float cosine, float ri)
{
    float r0 = (1 - ri) / (1 + ri);
    r0 = r0 * r0;
    return r0 + (1 - r0) * ((1 - cosine) * (1 - cosine) * (1 - cosine) * (1 - cosine) * (1 - cosine));
}

/* ray_scatter, Main.c:558-620 */
static inline int scatter(const ray_t *in, const hit_t *rec, v3 *atten, ray_t *out, rng_t *r)
{
    const rt_sphere *s = rec->s;
    switch (s->mat_type) {
    case RT_MAT_LAMBERTIAN: {
        v3 d = v3_add(rec->norm, scatter_direction(r));
        if (v3_near_zero(d)) d = rec->norm;
        out->orig = rec->p; out->dir = d;
        *atten = V(s->albedo);
        return 1;
    }
    case RT_MAT_METAL: {
        v3 refl = v3_reflect(in->dir, rec->norm);
        refl = v3_add(v3_unit(refl), v3_scale(scatter_direction(r), s->fuzz));
        out->orig = rec->p; out->dir = refl;
        *atten = V(s->albedo);
        return v3_dot(out->dir, rec->norm) > 0;
    }
    case RT_MAT_DIELECTRIC: {
        v3 one = { 1.0f, 1.0f, 1.0f };
        *atten = one;
        float ri = rec->front ? (1.0f / s->refraction_index) : s->refraction_index;
        v3 ud = v3_unit(in->dir);
        float cos_theta = (float)fmin((double)v3_dot(v3_scale(ud, -1.0f), rec->norm), 1.0);
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        int cannot = (double)(ri * sin_theta) > 1.0;
        v3 d;
        if (cannot || reflectance(cos_theta, ri) > rng_float(r))
            d = v3_reflect(ud, rec->norm);
        else
            d = v3_refract(ud, rec->norm, ri);
        out->orig = rec->p; out->dir = d;
        return 1;
    }
    default:
        return -1;   /* Main.c:613-619 leaves the scattered ray uninitialised */
    }
}

/* ray_color, Main.c:725-762 */
static v3 ray_color(const job_t *j, ray_t ray, int depth, rng_t *r, uint64_t *rays, int *first_hit)
{
    v3 color = { 1.0f, 1.0f, 1.0f };
    ray_t cur = ray;
    for (int i = 0; i < depth; i++) {
        hit_t rec;
        int idx = -1;
        (*rays)++;
        int got = hit_world(j, &cur, &rec, &idx);
        if (i == 0) *first_hit = got ? idx : -1;
        if (got) {
            ray_t sc;
            v3 att;
            int ok = scatter(&cur, &rec, &att, &sc, r);
            if (ok < 0) break;
            if (ok) color = v3_mul(color, att);
            cur = sc;                                                 /* Main.c:742: also when absorbed */
        } else {
            v3 ud = v3_unit(cur.dir);
            float t = 0.5f * (ud.y + 1.0f);
            v3 white = { 1.0f, 1.0f, 1.0f }, blue = { 0.5f, 0.7f, 1.0f };
            return v3_mul(color, v3_lerp(white, blue, t));
        }
    }
    v3 z = { 0.0f, 0.0f, 0.0f };
    return z;
}

/* get_ray, Main.c:776-803 (sample_square :764, defocus_disk_sample :770) */
static inline ray_t get_ray(const job_t *j, int x, int y, rng_t *r)
{
    const rt_camera *c = j->cam;
    const int direct = r->mode == REFCPU_RNG_COUNTER_DIRECT;
    float ox, oy;
    if (direct) {                                   /* one draw, stratum centres (k + 0.5) / 65536 - 0.5 */
        uint32_t b = rng_bits(r);
        const float bias = -0.5f + 1.0f / 131072.0f;
        ox = fmaf((float)(b >> 16), 1.0f / 65536.0f, bias);
        oy = fmaf((float)(b & 0xffffu), 1.0f / 65536.0f, bias);
    } else {
        ox = rng_float(r) - 0.5f;
        oy = rng_float(r) - 0.5f;
    }
    v3 ps;
    ps.x = c->pixel00_loc.x + ((x + ox) * c->pixel_delta_u.x) + ((y + oy) * c->pixel_delta_v.x);
    ps.y = c->pixel00_loc.y + ((x + ox) * c->pixel_delta_u.y) + ((y + oy) * c->pixel_delta_v.y);
    ps.z = c->pixel00_loc.z + ((x + ox) * c->pixel_delta_u.z) + ((y + oy) * c->pixel_delta_v.z);
    v3 org;
    if (c->defocus_angle <= 0) {
        org = V(c->pos);
    } else {
        v3 p = direct ? disk_point_direct(r) : random_direction_2d(r);
        org = v3_add(V(c->pos), v3_add(v3_scale(V(c->defocus_disk_u), p.x), v3_scale(V(c->defocus_disk_v), p.y)));
    }
    ray_t ray = { org, v3_sub(ps, org) };
    return ray;
}

/* to_color4(linear_to_gamma(c)), src/base_graphics.c:11-30 */
static inline uint8_t to_byte(float c)
{
    float g = 0.0f;
    if (c > 0) g = sqrtf(c);
    float v = g * 255.0f;
    float m = (255.0f < v) ? 255.0f : v;       /* MIN(255.0f, v), util.h:78 */
    float q = (0.0f > m) ? 0.0f : m;           /* MAX(0.0f, m),   util.h:77 */
    return (uint8_t)q;
}

static void finish_pixel(const job_t *j, uint32_t x, uint32_t y, v3 sum, int first_hit)
{
    v3 c = v3_scale(sum, (float)1.0f / (float)j->spp);               /* Main.c:977 */
    size_t p = (size_t)x + (size_t)y * j->W;                          /* BUF_AT, base_graphics.h:13 */
    if (j->o->accum_out) {
        j->o->accum_out[3 * p + 0] = c.x;
        j->o->accum_out[3 * p + 1] = c.y;
        j->o->accum_out[3 * p + 2] = c.z;
    }
    if (j->o->hit_index_out) j->o->hit_index_out[p] = first_hit;
    if (j->rgba) {
        j->rgba[4 * p + 0] = to_byte(c.x);
        j->rgba[4 * p + 1] = to_byte(c.y);
        j->rgba[4 * p + 2] = to_byte(c.z);
        j->rgba[4 * p + 3] = 255;
    }
}

/* One pixel with the LCG / constant streams: render_tile's body, Main.c:971-979 */
static void pixel_stream(const job_t *j, uint32_t x, uint32_t y, rng_t *r, uint64_t *rays)
{
    v3 color = { 0, 0, 0 };
    int fh = -1;
    for (int s = 0; s < j->spp; s++) {
        ray_t ray = get_ray(j, (int)x, (int)y, r);
        color = v3_add(color, ray_color(j, ray, j->depth, r, rays, &fh));
    }
    finish_pixel(j, x, y, color, fh);
}

/* One pixel with the counter generator: the GPU path's definition.  A pixel's sum
 * is the in-order sum of per-chunk partial sums, each an in-order sum of samples. */
static void pixel_counter(const job_t *j, uint32_t x, uint32_t y, uint64_t *rays)
{
    int chunk = j->o->chunk_spp > 0 ? j->o->chunk_spp : j->spp;
    uint32_t pix = x + y * j->W;
    v3 total = { 0, 0, 0 };
    int fh = -1;
    for (int s0 = 0; s0 < j->spp; s0 += chunk) {
        int s1 = s0 + chunk < j->spp ? s0 + chunk : j->spp;
        v3 part = { 0, 0, 0 };
        for (int s = s0; s < s1; s++) {
            rng_t r;
            r.mode = j->o->rng;
            r.lcg = 0;
            r.key = refcpu_counter_key(j->o->seed, pix, j->o->sample_base + (uint32_t)s);
            r.n = 0;
            ray_t ray = get_ray(j, (int)x, (int)y, &r);
            part = v3_add(part, ray_color(j, ray, j->depth, &r, rays, &fh));
        }
        total = v3_add(total, part);
    }
    finish_pixel(j, x, y, total, fh);
}

/* ---- tile scheduling ------------------------------------------------------ */
typedef struct { uint32_t x0, x1, y0, y1; } tile_t;

typedef struct {
    const job_t *j;
    tile_t *tiles;
    int ntiles;
    volatile int next;
    uint64_t rays;
    pthread_mutex_t mu;
} pool_t;

static void render_tile(const job_t *j, const tile_t *t, uint64_t *rays)
{
    int mode = j->o->rng;
    rng_t r;
    memset(&r, 0, sizeof r);
    r.mode = mode;
    if (mode == REFCPU_RNG_LCG_TILES) r.lcg = t->x0 * 1000u + t->y0 + 1u;        /* Main.c:963 */
    for (uint32_t y = t->y0; y < t->y1; ++y)
        for (uint32_t x = t->x0; x < t->x1; ++x) {
            if (mode == REFCPU_RNG_COUNTER || mode == REFCPU_RNG_COUNTER_DIRECT) pixel_counter(j, x, y, rays);
            else pixel_stream(j, x, y, &r, rays);
        }
}

static void *worker(void *arg)
{
    pool_t *p = (pool_t *)arg;
    uint64_t rays = 0;
    for (;;) {
        int i = __sync_fetch_and_add(&p->next, 1);
        if (i >= p->ntiles) break;
        render_tile(p->j, &p->tiles[i], &rays);
    }
    pthread_mutex_lock(&p->mu);
    p->rays += rays;
    pthread_mutex_unlock(&p->mu);
    return NULL;
}

int refcpu_core_count(void)                                            /* get_core_count, util.c:512-520 */
{
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n > 0 ? (int)n : 1;
}

static int row_owned(uint32_t y, const refcpu_opts *o)
{
    if (o->part_count <= 1) return 1;
    int tr = o->tile_rows > 0 ? o->tile_rows : 16;
    return (int)((y / (uint32_t)tr) % (uint32_t)o->part_count) == o->part_index;
}

int refcpu_render(const rt_camera *cam, const rt_sphere *spheres, size_t count,
                  uint8_t *rgba, uint32_t width, uint32_t height,
                  int spp, int max_depth, const refcpu_opts *opts)
{
    refcpu_opts dflt;
    memset(&dflt, 0, sizeof dflt);
    if (!opts) opts = &dflt;
    if (!cam || (!spheres && count) || !width || !height || spp < 1 || max_depth < 0) return RT_ERR_ARG;
    for (size_t i = 0; i < count; i++)
        if (spheres[i].mat_type < RT_MAT_LAMBERTIAN || spheres[i].mat_type > RT_MAT_DIELECTRIC) return RT_ERR_ARG;
    int lcg = opts->rng == REFCPU_RNG_LCG_TILES || opts->rng == REFCPU_RNG_LCG_SERIAL;
    if (lcg && opts->part_count > 1) return RT_ERR_ARG;

    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);

    job_t j = { cam, spheres, count, width, height, spp, max_depth, opts, rgba };
    pool_t p;
    memset(&p, 0, sizeof p);
    p.j = &j;
    pthread_mutex_init(&p.mu, NULL);

    int nthreads = opts->threads > 0 ? opts->threads : 2 * refcpu_core_count();   /* Main.c:1002 */
    if (opts->rng == REFCPU_RNG_LCG_SERIAL) {
        /* one stream over the whole image: a single tile, a single thread */
        tile_t whole = { 0, width, 0, height };
        p.tiles = &whole; p.ntiles = 1;
        nthreads = 1;
        worker(&p);
    } else {
        uint32_t ts = 64;                                                           /* Main.c:1003 */
        uint32_t th = lcg ? ts : 1;                 /* non-LCG modes: one tile = part of one row */
        uint32_t tx_n = (width + ts - 1) / ts, ty_n = (height + th - 1) / th;
        p.tiles = (tile_t *)malloc(sizeof(tile_t) * (size_t)tx_n * ty_n);
        if (!p.tiles) return RT_ERR_NOMEM;
        for (uint32_t ty = 0; ty < ty_n; ty++) {                                    /* Main.c:1014-1028 */
            uint32_t y0 = ty * th, y1 = (y0 + th > height) ? height : y0 + th;
            if (!lcg && !row_owned(y0, opts)) continue;
            for (uint32_t tx = 0; tx < tx_n; tx++) {
                uint32_t x0 = tx * ts, x1 = (x0 + ts > width) ? width : x0 + ts;
                tile_t t = { x0, x1, y0, y1 };
                p.tiles[p.ntiles++] = t;
            }
        }
        if (nthreads > 256) nthreads = 256;
        pthread_t th_[256];
        int started = 0;
        for (int i = 0; i < nthreads - 1; i++)
            if (pthread_create(&th_[started], NULL, worker, &p) == 0) started++;
        worker(&p);
        for (int i = 0; i < started; i++) pthread_join(th_[i], NULL);
        free(p.tiles);
    }
    pthread_mutex_destroy(&p.mu);

    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (opts->stats) {
        uint64_t rows = 0;
        for (uint32_t y = 0; y < height; y++) rows += (lcg || row_owned(y, opts)) ? 1 : 0;
        opts->stats->samples = rows * width * (uint64_t)spp;
        opts->stats->rays = p.rays;
        opts->stats->sphere_tests = p.rays * (uint64_t)count;
        opts->stats->chunk_spp = (uint32_t)(opts->chunk_spp > 0 && opts->chunk_spp < spp ? opts->chunk_spp : spp);
        opts->stats->scan_tiles = 0;
        opts->stats->wave_scans = 0;
        opts->stats->kernel_ms = 0.0;
        opts->stats->resolve_ms = 0.0;
        opts->stats->total_ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
    }
    return RT_OK;
}

/* update_camera_view, Main.c:839-904.  tan_f32(v) is tanf(v * (2*3.1415926535897f))
 * (include/util.h:134,142) and is fed radians, as in the reference. */
#define TURNS_TO_RAD(v) ((v) * (2 * 3.1415926535897f))
#define DEG_TO_RAD(v)   (TURNS_TO_RAD((v) / 360.f))

int refcpu_camera_setup(const rt_camera_params *p, uint32_t width, uint32_t height, rt_camera *out)
{
    if (!p || !out || !width || !height) return RT_ERR_ARG;
    v3 pos = V(p->pos), target = V(p->target), up = V(p->up);
    float theta = DEG_TO_RAD(p->vfov);
    float h = tanf(TURNS_TO_RAD(theta / 2.0f));
    float vp_h = 2.0f * h * p->focus_dist;
    float vp_w = vp_h * ((float)width / (float)height);
    v3 w = v3_unit(v3_sub(pos, target));
    v3 u = v3_unit(v3_cross(up, w));
    v3 v = v3_cross(w, u);
    v3 vp_u = v3_scale(u, vp_w);
    v3 vp_v = v3_scale(v, -vp_h);
    v3 du = v3_scale(vp_u, 1.0f / (float)width);
    v3 dv = v3_scale(vp_v, 1.0f / (float)height);
    v3 centre = v3_sub(pos, v3_scale(w, p->focus_dist));
    v3 ul = v3_sub(centre, v3_add(v3_scale(vp_u, 0.5f), v3_scale(vp_v, 0.5f)));
    v3 p00 = v3_add(ul, v3_add(v3_scale(du, 0.5f), v3_scale(dv, 0.5f)));
    float radius = p->focus_dist * tanf(TURNS_TO_RAD(DEG_TO_RAD(p->defocus_angle / 2)));
    v3 ddu = v3_scale(u, radius), ddv = v3_scale(v, radius);
    out->pos = p->pos;
    out->pixel00_loc = (rt_vec3){ p00.x, p00.y, p00.z };
    out->pixel_delta_u = (rt_vec3){ du.x, du.y, du.z };
    out->pixel_delta_v = (rt_vec3){ dv.x, dv.y, dv.z };
    out->defocus_disk_u = (rt_vec3){ ddu.x, ddu.y, ddu.z };
    out->defocus_disk_v = (rt_vec3){ ddv.x, ddv.y, ddv.z };
    out->defocus_angle = p->defocus_angle;
    return RT_OK;
}
