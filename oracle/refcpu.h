/*
 * refcpu -- CPU restatement of the reference's path-tracing hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP path is compared with;
 * it is never linked into, called by, or shipped with the product library.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Parity pin: in REFCPU_RNG_LCG_TILES mode the output is byte-identical to the
 * reference's own C code compiled by oracle/build_ref.sh (render_tile per 64x64
 * tile); tests/golden/ holds those outputs, made by oracle/make_golden.py.  The
 * reference itself ships no tests or golden vectors for this path.
 */
#ifndef REFCPU_H_
#define REFCPU_H_

#include "../include/rt.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    REFCPU_RNG_LCG_TILES  = 0, /* fast_rand LCG (src/util.c:27-39), reseeded per 64x64
                                  tile with start_x*1000+start_y+1 (Main.c:963)       */
    REFCPU_RNG_LCG_SERIAL = 1, /* same LCG, one stream from seed 0 over the whole image
                                  in render_all's order (Main.c:1099-1113)            */
    REFCPU_RNG_COUNTER    = 2, /* counter-based generator of the GPU path             */
    REFCPU_RNG_CONST_U    = 3, /* every draw is 16384/32768 = 0.5                      */
    REFCPU_RNG_COUNTER_DIRECT = 4 /* counter generator with the loop-free samplers of the
                                  GPU path's exact build (RT_BUILD_EXACT, include/rt.h) in
                                  place of the reference's rejection loops (util.c:170-200);
                                  everything else as REFCPU_RNG_COUNTER                 */
};

typedef struct refcpu_opts {
    int32_t  rng;            /* REFCPU_RNG_*                                          */
    uint32_t seed;           /* COUNTER modes                                         */
    uint32_t sample_base;    /* COUNTER mode                                          */
    int32_t  chunk_spp;      /* COUNTER mode: 0 = spp                                 */
    int32_t  threads;        /* 0 = 2 * online cores (Main.c:1002)                    */
    int32_t  tile_rows;      /* COUNTER / CONST_U modes: render only the row tiles t   */
    int32_t  part_index;     /*   (tile_rows rows each, 0 = 16) with                   */
    int32_t  part_count;     /*   t % part_count == part_index; rows land at their     */
                             /*   place in the full image.  0 or 1 = all rows          */
    int32_t *hit_index_out;  /* optional, width*height                                */
    float   *accum_out;      /* optional, width*height*3                              */
    rt_stats *stats;         /* optional: samples, rays, sphere_tests, total_ms       */
} refcpu_opts;

int refcpu_render(const rt_camera *cam, const rt_sphere *spheres, size_t count,
                  uint8_t *rgba, uint32_t width, uint32_t height,
                  int spp, int max_depth, const refcpu_opts *opts);

/* update_camera_view (Main.c:839-904) restated. */
int refcpu_camera_setup(const rt_camera_params *p, uint32_t width, uint32_t height, rt_camera *out);

/* One draw of the counter generator, exposed so tests can pin it. */
uint32_t refcpu_counter_key(uint32_t seed, uint32_t pixel_index, uint32_t sample_index);
uint32_t refcpu_counter_draw(uint32_t key, uint32_t n);

int refcpu_core_count(void);

#ifdef __cplusplus
}
#endif
#endif
