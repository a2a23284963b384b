/*
 * rt.h -- C ABI of librtmi.so, the MI355X (gfx950) implementation of the
 * path-tracing hot path of Zelmoghazy/RayTracer.
 *
 * The reference has no render(camera, world, framebuffer, spp, max_depth)
 * function: its frame loop calls render_all_parallel() (Main.c:990-1082),
 * which reads the global `gc` (Main.c:97-158).  The entry points below are the
 * boundary carved out at Main.c:1002-1054 (the 64x64-tile / thread dispatch
 * around render_tile, Main.c:959-988): everything before it (framebuffer
 * allocation, Main.c:998) and after it (text overlay, capture, present_frame,
 * Main.c:1056-1081, 1181-1185) keeps running on the caller-owned RGBA8 buffer.
 *
 * Plain C structs and pointers only.  Every function returns 0 on success or a
 * negative rt_status; nothing throws or exits across the boundary; the text of
 * the last error of the calling process is available from rt_last_error().
 * Calls are blocking and not re-entrant (the reference calls its renderer from
 * the one GLFW thread, Main.c:1440-1443).
 */
#ifndef RT_H_
#define RT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_vec3 { float x, y, z; } rt_vec3;          /* vec3f_t, include/util.h:245-250 */

/* enum material_type, Main.c:45-51 */
enum { RT_MAT_LAMBERTIAN = 0, RT_MAT_METAL = 1, RT_MAT_DIELECTRIC = 2 };

/* One scene object: byte-for-byte the reference's sphere_t (Main.c:74-79) with its
 * embedded material_t (Main.c:53-59); 40 bytes.  The library flattens the array
 * into structure-of-arrays form on upload. */
typedef struct rt_sphere {
    rt_vec3 center;
    float   radius;
    int32_t mat_type;
    rt_vec3 albedo;
    float   fuzz;
    float   refraction_index;
} rt_sphere;

/* What the path reads of the camera: gc.camera.pos, gc.pixel00_loc,
 * gc.pixel_delta_u/v, gc.camera.defocus_disk_u/v, gc.camera.defocus_angle
 * (Main.c:107-109, 26, 37, 40-41; consumed by get_ray, Main.c:776-803). */
typedef struct rt_camera {
    rt_vec3 pos;
    rt_vec3 pixel00_loc;
    rt_vec3 pixel_delta_u;
    rt_vec3 pixel_delta_v;
    rt_vec3 defocus_disk_u;
    rt_vec3 defocus_disk_v;
    float   defocus_angle;
} rt_camera;

/* Inputs of update_camera_view (Main.c:839-904); defaults of init_camera (Main.c:915-922). */
typedef struct rt_camera_params {
    rt_vec3 pos, target, up;
    float   vfov;            /* degrees */
    float   defocus_angle;   /* degrees */
    float   focus_dist;
} rt_camera_params;

/* Random numbers.  The reference's thread-local LCG (fast_rand, src/util.c:27-39) is replaced by a
 * counter-based generator so that an image does not depend on tiles, GPUs or scheduling:
 *     mix32(x): x ^= x>>16; x *= 0x7feb352d; x ^= x>>15; x *= 0x846ca68b; x ^= x>>16
 *     key    = mix32( mix32(pixel + mix32(seed ^ 0x6A09E667)) ^ (sample * 0x9E3779B9) )
 *     draw_n = mix32(key + n * 0x9E3779B9),  n = 0,1,2,... within one sample
 *     u      = (draw_n >> 8) * 2^-24                       (replaces RAND_FLOAT, include/util.h:93)
 * pixel = x + y*width in the FULL image, sample = sample_base + index within the call.  What the
 * draws of a sample are used for depends on the build (RT_BUILD_*, below). */
enum {
    RT_MODE_MC      = 0,  /* Monte-Carlo: counter-based RNG keyed (seed, pixel, sample) */
    RT_MODE_CONST_U = 1   /* deterministic: every random draw returns 0.5, i.e. the
                             reference with fast_rand() == 16384: centre sample, ray
                             origin = camera.pos, scatter along the normal */
};

/* Three builds of the path-tracing kernel.
 *   RT_BUILD_EXACT  (default) every f32 operation of get_ray, hit_sphere, hit, ray_scatter, ray_color and the
 *                   pixel finish is evaluated in the reference's order with IEEE-rounded sqrt and divide and no
 *                   FMA contraction.  The reference's two rejection samplers (vec3f_random_direction(_2d),
 *                   src/util.c:170-200) are replaced by loop-free maps with the same distributions, defined so
 *                   that a CPU reproduces them bit for bit: per sample one draw for the pixel jitter (16 bits per
 *                   axis, stratum centres), one for the defocus disk (radius sqrt((k + 0.5) / 65536), azimuth
 *                   2 pi i / 65536), then per bounce one draw for a Lambertian / metal direction
 *                   (z = (65535 - 2 k) / 65536, azimuth as before) or one uniform for the glass decision.
 *                   RT_MODE_CONST_U output equals the reference's, bit for bit.
 *   RT_BUILD_STRICT the same arithmetic with the reference's rejection samplers and the reference's order of
 *                   draws (jitter x, jitter y, defocus disk, then per bounce the scatter draws).
 *   RT_BUILD_FAST   FMA contraction, unit-length directions, hardware rsqrt / rcp / sin / cos: the same estimator,
 *                   statistically equivalent images, not reproducible on a CPU. */
enum { RT_BUILD_EXACT = 0, RT_BUILD_STRICT = 1, RT_BUILD_FAST = 2 };

typedef struct rt_stats {
    uint64_t samples;        /* pixel samples traced by this call                  */
    uint64_t rays;           /* calls of hit() (Main.c:688), i.e. rays cast         */
    uint64_t sphere_tests;   /* rays * number of spheres (hit_sphere calls)         */
    double   kernel_ms;      /* device time of the path-tracing kernel, HIP events  */
    double   total_ms;       /* wall time of the call                               */
    double   resolve_ms;     /* device time of the chunk-sum / gamma / RGBA8 kernel */
    uint32_t chunk_spp;      /* samples per work unit used (the choice made for RT_CHUNK_AUTO) */
    uint32_t scan_tiles;     /* 32-sphere tiles one sphere scan of a wave multiplies on the
                                matrix cores (0: the scene runs the packed-FP32 scan)        */
    uint64_t wave_scans;     /* sphere scans run (64 rays against all spheres each)        */
    uint64_t tile_products;  /* tiles those scans multiplied on the matrix cores: a scan
                                skips the tiles whose bounding box none of its rays meets,
                                so this is at most wave_scans * scan_tiles.  The executed
                                matrix work of the call is tile_products * 4 instructions
                                of 32768 flop                                              */
} rt_stats;

enum { RT_CHUNK_AUTO = -1 };  /* rt_opts.chunk_spp: let the library choose the unit size */

typedef struct rt_opts {
    int32_t  mode;           /* RT_MODE_*                                            */
    int32_t  build;          /* RT_BUILD_*: which build of the kernel runs (0 = exact)  */
    uint32_t seed;           /* RNG seed (MC mode)                                   */
    uint32_t sample_base;    /* index of the first sample (progressive frames pass
                                frame*spp so that frames draw disjoint streams)      */
    int32_t  chunk_spp;      /* samples per work unit; a pixel's sum is the in-order
                                sum of its chunks' in-order partial sums.  0 = spp
                                (one unit per pixel: the reference's summation order,
                                Main.c:973-976); RT_CHUNK_AUTO = chosen for throughput:
                                a function of the FULL image's width, height and spp
                                only (rt_chunk_for), never of the rows a call, a device
                                or a rank owns, so the parts of a split frame agree  */
    int32_t  tile_rows;      /* multi-GPU row-tile height (0 = 16)                   */
    int32_t  part_index;     /* this call renders row tiles t with                   */
    int32_t  part_count;     /*   t % part_count == part_index  (0 or 1 = all rows)  */
    int32_t *hit_index_out;  /* optional, width*height: index of the object hit by
                                the primary ray of the last sample, -1 = sky         */
    float   *accum_out;      /* optional, width*height*3: linear mean colour (before
                                gamma and quantisation)                              */
    rt_stats *stats;         /* optional                                             */
    const uint32_t *rows;    /* optional explicit partition (HOST pointer): the image rows
                                this call renders, strictly ascending; overrides
                                part_index / part_count.  Lets a multi-GPU caller deal row
                                tiles in proportion to measured GPU speed.  Compact device
                                buffers then hold row_count rows in this order           */
    uint32_t row_count;
} rt_opts;

enum rt_status {
    RT_OK = 0,
    RT_ERR_ARG = -1,         /* bad argument                                          */
    RT_ERR_STATE = -2,       /* rt_init / rt_set_world not called                     */
    RT_ERR_DEVICE = -3,      /* HIP error or no gfx950 device                         */
    RT_ERR_NOMEM = -4
};

/* Selects HIP device `device` for this process and creates the stream, events and
 * device buffers the path needs.  No reference counterpart (the reference spawns
 * OS threads per tile, Main.c:1002-1045). */
int rt_init(int device);

/* Same for `count` devices (1..16) in ONE process: the reference is a single-process program
 * (one GLFW thread, Main.c:1440-1443), so this is how its frame loop reaches several GPUs
 * without a launcher.  rt_set_world uploads to every device; rt_render deals the rows it owns to
 * the devices in 2-row tiles, in proportion to the speed each device showed in the previous
 * frame, renders them concurrently (one persistent host thread per device) and writes every row at its
 * place in the caller's buffers; rt_stats then holds sums (samples, rays, sphere_tests) and
 * maxima (kernel_ms, resolve_ms).  The image is the single-device image, byte for byte.
 * rt_render_progressive works the same way and keeps the deal of frame 0 for the sequence (each
 * device carries the running sums of its rows; a frame that owns other rows than frame 0, or that
 * follows a failed frame, is refused with RT_ERR_STATE); rt_render_device needs a single device
 * (RT_ERR_STATE otherwise).  The calling thread's current HIP device is the same after every call
 * of this library as before it.
 * An id may be listed more than once (contexts sharing a device; used by the tests). */
int rt_init_devices(const int *devices, int count);

/* Replaces gc.scene_objects (Main.c:116, filled by init_scene, Main.c:1212-1318).
 * `spheres` is copied; only RT_MAT_LAMBERTIAN/METAL/DIELECTRIC are accepted (the
 * reference's Emissive case leaves the scattered ray uninitialised, Main.c:613-619). */
int rt_set_world(const rt_sphere *spheres, size_t count);

/* Replaces the tile/thread dispatch of render_all_parallel (Main.c:1002-1054) and
 * the per-pixel work of render_tile (Main.c:959-988): get_ray, ray_color, hit,
 * hit_sphere, ray_scatter, 1/spp scale, linear_to_gamma, to_color4, set_pixel.
 * `rgba` is caller-owned HOST memory, width*height*4 bytes, row 0 = top, alpha 255
 * (color4_t, include/util.h:227-233; BUF_AT, include/base_graphics.h:13).  With
 * opts->part_count > 1 only the owned rows are written, at their place in the
 * full image.  `opts` may be NULL (MC, exact build, seed 0, all rows, RT_CHUNK_AUTO). */
int rt_render(const rt_camera *cam, uint8_t *rgba, uint32_t width, uint32_t height,
              int spp, int max_depth, const rt_opts *opts);

/* Same as rt_render, but `rgba_dev` (and opts->hit_index_out / accum_out) are DEVICE
 * pointers on the device given to rt_init, so that a multi-GPU caller can hand the
 * rows to RCCL without a host round trip.  With opts->part_count > 1 the buffers are
 * COMPACT: they hold only the owned rows, in image order, rt_part_rows() of them. */
int rt_render_device(const rt_camera *cam, uint8_t *rgba_dev, uint32_t width, uint32_t height,
                     int spp, int max_depth, const rt_opts *opts);

/* No counterpart in the reference.  For callers that cannot allocate device memory around the
 * call (tight serving loops, harnesses that forbid hipMalloc, hosts whose memory pool is handed out
 * once): the same
 * as rt_render_device, but the per-chunk partial sums and the work/ray counters live in a fixed
 * array of RT_NOALLOC_SCRATCH_BYTES in the library's code object (one copy per GPU, held until the
 * process ends) instead of buffers the library grows on demand; the call runs no hipMalloc, hipFree
 * or other allocating routine.  Call rt_set_world first (it loads the kernels and looks up their
 * launch geometry).  Limits: owned pixels x ceil(spp / chunk_spp) x 16 bytes must fit
 * RT_NOALLOC_SCRATCH_BYTES - 256 (1920x1080 in up to two chunks; chunk_spp = 0 is always one
 * chunk), else RT_ERR_ARG with the sizes in rt_last_error(); opts->rows is not supported
 * (part_index / part_count are).  Results equal rt_render_device's for the same arguments, bit for
 * bit.  The call is synchronous: it launches on the library's own stream and waits for it, so it
 * cannot be recorded into a caller's HIP graph; what it guarantees is the absence of allocation.
 * All calls on one GPU share that GPU's array: calls that overlap in time on one GPU race;
 * the library's other entry points do not touch the array. */
#define RT_NOALLOC_SCRATCH_BYTES (64u << 20)
int rt_render_device_noalloc(const rt_camera *cam, uint8_t *rgba_dev, uint32_t width, uint32_t height,
                             int spp, int max_depth, const rt_opts *opts);

/* Progressive display (no counterpart in the reference, which re-renders identical frames,
 * Main.c:1440-1443): frame `frame_index` adds `spp` new samples per pixel (sample indices
 * sample_base + frame_index*spp ...) to a per-pixel sum kept on the device and writes the
 * image of ALL samples so far.  frame_index 0 starts a sequence; frames must follow in order
 * with the same size, spp and partition.  After F frames the result equals, bit for bit in the
 * exact and strict builds, one rt_render call with F*spp samples and the same chunk_spp (which must
 * divide spp).  opts->accum_out receives the running mean. */
int rt_render_progressive(const rt_camera *cam, uint8_t *rgba, uint32_t width, uint32_t height,
                          int spp, int max_depth, const rt_opts *opts, uint32_t frame_index);

/* Number of image rows owned by part `part_index` of `part_count` (row tiles of
 * `tile_rows` rows dealt round-robin).  Pure host arithmetic. */
uint32_t rt_part_rows(uint32_t height, int tile_rows, int part_index, int part_count);

/* Image row of the k-th owned row of a part (inverse of the compaction above). */
uint32_t rt_part_row_at(uint32_t k, int tile_rows, int part_index, int part_count);

/* The samples per work unit a render call of a width x height image at `spp` uses for
 * opts->chunk_spp = chunk_spp (RT_CHUNK_AUTO, 0 or an explicit size); `progressive` != 0 for
 * rt_render_progressive, whose units must divide spp.  Pure host arithmetic; 0 on bad arguments.
 * No reference counterpart (the reference sums a pixel's samples in one run, Main.c:973-976). */
int rt_chunk_for(uint32_t width, uint32_t height, int spp, int chunk_spp, int progressive);

/* update_camera_view (Main.c:839-904) restated on the host, quirks included
 * (tan_f32 takes turns but is fed radians, include/util.h:134,142).  A host that
 * keeps the reference's own update_camera_view does not need this. */
int rt_camera_setup(const rt_camera_params *p, uint32_t width, uint32_t height, rt_camera *out);

/* init_camera's defaults (Main.c:915-922) and its height rule (Main.c:908). */
void rt_camera_defaults(rt_camera_params *p);
uint32_t rt_height_for(uint32_t width, float aspect_ratio);

/* init_scene (Main.c:1212-1318) restated on the host with a FIXED evaluation order.  The reference
 * draws its RAND_FLOAT() values inside initialiser lists and argument lists (Main.c:1231-1256),
 * whose order C leaves to the compiler; here it is: material choice, centre x, centre z, then the
 * material's own draws in reading order.  Output: the ground sphere, one r = 0.2 sphere per cell
 * (a, b) of [-half_extent, half_extent)^2 that is farther than 0.9 from (4, 0.2, 0), then the
 * three unit spheres.  half_extent = 11 with lcg_seed = 0 reproduces, byte for byte, the scene
 * the reference generates when built with gcc (its random block enabled); half_extent = 0 is
 * the shipped three-sphere scene.  The generator is the reference's LCG (src/util.c:27-39).
 * Writes at most `cap` records and the number of objects to *count; RT_ERR_ARG if cap is too
 * small.  Pure host code. */
int rt_scene_generate(int half_extent, uint32_t lcg_seed, rt_sphere *out, size_t cap, size_t *count);

/* The order rt_set_world gives the spheres on the device (no counterpart in the reference, whose hit() walks the
 * caller's list, Main.c:688-713; nothing a caller sees depends on the order).  order_out[k] = index in `spheres` of the
 * k-th sphere in device order, `count` entries.  Spheres 32 t .. 32 t + 31 in device order are tile t of the matrix scan,
 * which skips a tile whose axis-aligned bounding box no ray of a wave meets: split_tiles != 0 gives the order
 * rt_set_world uses (the list split in space until a part is one tile), 0 the plain Z-order curve it is compared with
 * (RTMI_ORDER=curve in the environment at rt_init).  For tests and scene tools.  Pure host code. */
int rt_world_layout(const rt_sphere *spheres, size_t count, int split_tiles, int32_t *order_out);

/* Optional: page-locked host memory for a framebuffer the caller keeps for many frames (in place
 * of the arena block of Main.c:998, which is also the texture upload source of render_to_screen,
 * Main.c:935): rt_render's read-back of a whole frame into it is one DMA transfer instead of a
 * staged copy.  Any other host memory works too.  NULL if the memory cannot be had (text in
 * rt_last_error).  Thin wrappers over hipHostMalloc / hipHostFree, so that a C host needs no HIP
 * headers.  No reference counterpart. */
void *rt_host_alloc(size_t bytes);
void rt_host_free(void *ptr);

/* Frees everything rt_init / rt_set_world created. */
void rt_shutdown(void);

/* Text of the last error reported to the calling thread ("" if none). */
const char *rt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_H_ */
