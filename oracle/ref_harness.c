/*
 * Headless driver for the COMPILED REFERENCE (test infrastructure only).
 *
 * This file is appended, by oracle/build_ref.sh, to the text of the reference's
 * own Main.c (streamed from /root/reference, never copied into this repo), so
 * every function called below -- init_camera, init_scene, render_tile, get_ray,
 * ray_color, hit_sphere, to_color4, linear_to_gamma -- is the reference's code
 * (Main.c:534-803, 906-928, 959-988, 1212-1318; src/util.c; src/base_graphics.c).
 * Nothing here is shipped or timed as the product; it only produces golden
 * vectors for tests/golden/ and cross-checks oracle/refcpu.c.
 *
 * usage: ref_harness_<s3|sf> <cmd> <width> <spp> <depth> <out-file> [threads]
 *   scene   raw sphere_t records (40 B each, Main.c:74-79)
 *   camera  19 f32 (pos, pixel00, du, dv, disk_u, disk_v, defocus_angle) + 2 u32 (W,H)
 *           (environment REF_CAM_POS="x,y,z" moves the eye point for every command)
 *   mc      RGBA8 image, reference render_tile per 64x64 tile (per-tile LCG seed)
 *   mcref   as mc, with the reference's own threading shape (Main.c:1002-1054): `threads` slots,
 *           one OS thread created per tile, slot i % threads joined before it is reused
 *   mcser   RGBA8 image, render_all's loop order on one thread (seed 0)
 *   d0      int32 primary-hit index per pixel (centre ray from camera.pos)
 *   d1      constant-u render (fast_rand() == 16384): per pixel 3 f32 linear colour,
 *           then W*H RGBA8, then W*H int32 primary-hit index
 */
#include <pthread.h>

int ref_lcg_fast_rand(void);      /* the reference's fast_rand, renamed at build time */
static int g_const_u = 0;
int fast_rand(void) { return g_const_u ? 16384 : ref_lcg_fast_rand(); }

static tile_data_t *g_tiles;
static int g_ntiles;
static volatile int g_next;

static void *tile_worker(void *arg)
{
    (void)arg;
    for (;;) {
        int i = __sync_fetch_and_add(&g_next, 1);
        if (i >= g_ntiles) break;
        render_tile(&g_tiles[i]);
    }
    return NULL;
}

static void *one_tile(void *arg)
{
    render_tile((tile_data_t *)arg);
    return NULL;
}

static int primary_hit_index(ray_t *ray)
{
    /* replay of hit() (Main.c:688-713) that records the winning index */
    hit_record_t rec;
    f32 closest = max_f32;
    int idx = -1;
    for (size_t i = 0; i < gc.scene_objects->count; i++) {
        if (hit_sphere((sphere_t *)gc.scene_objects->objects[i].object, ray, 0.001f, closest, &rec)) {
            closest = rec.hit_dist;
            idx = (int)i;
        }
    }
    return idx;
}

static ray_t centre_ray(int x, int y)
{
    vec3f_t ps = {
        .x = gc.pixel00_loc.x + ((x + 0.0f) * gc.pixel_delta_u.x) + ((y + 0.0f) * gc.pixel_delta_v.x),
        .y = gc.pixel00_loc.y + ((x + 0.0f) * gc.pixel_delta_u.y) + ((y + 0.0f) * gc.pixel_delta_v.y),
        .z = gc.pixel00_loc.z + ((x + 0.0f) * gc.pixel_delta_u.z) + ((y + 0.0f) * gc.pixel_delta_v.z)
    };
    ray_t r = { gc.camera.pos, vec3f_sub(ps, gc.camera.pos) };
    return r;
}

int main(int argc, char **argv)
{
    if (argc < 6) {
        fprintf(stderr, "usage: %s <scene|camera|mc|mcser|d0|d1> <width> <spp> <depth> <out> [threads]\n", argv[0]);
        return 2;
    }
    const char *cmd = argv[1];
    int width = atoi(argv[2]);
    int spp = atoi(argv[3]);
    int depth = atoi(argv[4]);
    const char *out = argv[5];
    int nthreads = argc > 6 ? atoi(argv[6]) : 8;

    fast_srand(0);
    init_camera(width, 16.0f / 9.0f);
    /* REF_CAM_POS="x,y,z": another eye point, derived by the reference's own update_camera_view
       (Main.c:839-904), as its key handler does after moving the camera (Main.c:382-458) */
    const char *cam_pos = getenv("REF_CAM_POS");
    if (cam_pos) {
        float px, py, pz;
        if (sscanf(cam_pos, "%f,%f,%f", &px, &py, &pz) != 3) { fprintf(stderr, "bad REF_CAM_POS\n"); return 2; }
        gc.camera.pos = (vec3f_t){ .x = px, .y = py, .z = pz };
        update_camera_view();
    }
    init_scene();
    gc.samples_per_pixel = spp;
    gc.max_depth = depth;

    u32 W = gc.screen_width, H = gc.screen_height;
    FILE *f = fopen(out, "wb");
    if (!f) { perror(out); return 1; }

    if (!strcmp(cmd, "scene")) {
        for (size_t i = 0; i < gc.scene_objects->count; i++) {
            sphere_t s;
            memset(&s, 0, sizeof s);
            sphere_t *p = (sphere_t *)gc.scene_objects->objects[i].object;
            s.center = p->center; s.radius = p->radius;
            s.mat.mat_type = p->mat.mat_type; s.mat.albedo = p->mat.albedo;
            s.mat.fuzz = p->mat.fuzz; s.mat.refraction_index = p->mat.refraction_index;
            fwrite(&s, sizeof s, 1, f);
        }
        fprintf(stderr, "scene: %zu objects, %zu B each\n", gc.scene_objects->count, sizeof(sphere_t));
    } else if (!strcmp(cmd, "camera")) {
        vec3f_t v[6] = { gc.camera.pos, gc.pixel00_loc, gc.pixel_delta_u, gc.pixel_delta_v,
                         gc.camera.defocus_disk_u, gc.camera.defocus_disk_v };
        fwrite(v, sizeof v, 1, f);
        fwrite(&gc.camera.defocus_angle, 4, 1, f);
        fwrite(&W, 4, 1, f);
        fwrite(&H, 4, 1, f);
    } else if (!strcmp(cmd, "mc") || !strcmp(cmd, "mcser") || !strcmp(cmd, "mcref")) {
        gc.draw_buffer.width = W; gc.draw_buffer.height = H;
        gc.draw_buffer.pixels = calloc((size_t)W * H, sizeof(color4_t));
        if (!strcmp(cmd, "mcref")) {
            /* the dispatch of render_all_parallel (Main.c:1002-1054) around the reference's render_tile */
            const u32 ts = 64;
            u32 tx_n = CEIL_DIV(W, ts), ty_n = CEIL_DIV(H, ts);
            if (nthreads < 1) nthreads = 1;
            g_tiles = calloc((size_t)tx_n * ty_n, sizeof(tile_data_t));
            pthread_t *slot = calloc((size_t)nthreads, sizeof(pthread_t));
            int created = 0;
            for (u32 ty = 0; ty < ty_n; ty++)
                for (u32 tx = 0; tx < tx_n; tx++) {
                    u32 sx = tx * ts, sy = ty * ts;
                    int i = created;
                    g_tiles[i] = (tile_data_t){
                        .start_x = sx, .end_x = (sx + ts > W) ? W : sx + ts,
                        .start_y = sy, .end_y = (sy + ts > H) ? H : sy + ts,
                        .width = W, .height = H };
                    if (i >= nthreads) pthread_join(slot[i % nthreads], NULL);
                    if (pthread_create(&slot[i % nthreads], NULL, one_tile, &g_tiles[i]) != 0) { perror("pthread_create"); return 1; }
                    created++;
                }
            int live = created < nthreads ? created : nthreads;
            for (int i = 0; i < live; i++) pthread_join(slot[(created - live + i) % nthreads], NULL);
        } else if (!strcmp(cmd, "mc")) {
            const u32 ts = 64;   /* Main.c:1003 */
            u32 tx_n = CEIL_DIV(W, ts), ty_n = CEIL_DIV(H, ts);
            g_tiles = calloc((size_t)tx_n * ty_n, sizeof(tile_data_t));
            g_ntiles = 0;
            for (u32 ty = 0; ty < ty_n; ty++)
                for (u32 tx = 0; tx < tx_n; tx++) {
                    u32 sx = tx * ts, sy = ty * ts;
                    g_tiles[g_ntiles++] = (tile_data_t){
                        .start_x = sx, .end_x = (sx + ts > W) ? W : sx + ts,
                        .start_y = sy, .end_y = (sy + ts > H) ? H : sy + ts,
                        .width = W, .height = H };
                }
            pthread_t th[64];
            if (nthreads > 64) nthreads = 64;
            g_next = 0;
            for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, tile_worker, NULL);
            for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
        } else {
            /* render_all's pixel loop (Main.c:1099-1113) on one thread.  The reference
               never reseeds there, so its stream continues from whatever init_scene
               left; this driver pins the start to seed 0 so that the run does not
               depend on how many draws scene generation made. */
            fast_srand(0);
            for (u32 y = 0; y < H; ++y)
                for (u32 x = 0; x < W; ++x) {
                    vec3f_t color = {0, 0, 0};
                    for (int s = 0; s < gc.samples_per_pixel; s++) {
                        ray_t ray = get_ray(x, y);
                        color = vec3f_add(color, ray_color(ray, gc.max_depth));
                    }
                    color = vec3f_scale(color, (f32)1.0f / (f32)gc.samples_per_pixel);
                    color = linear_to_gamma(color);
                    set_pixel(&gc.draw_buffer, x, y, to_color4(color));
                }
        }
        fwrite(gc.draw_buffer.pixels, sizeof(color4_t), (size_t)W * H, f);
    } else if (!strcmp(cmd, "d0")) {
        for (u32 y = 0; y < H; y++)
            for (u32 x = 0; x < W; x++) {
                ray_t r = centre_ray(x, y);
                i32 idx = primary_hit_index(&r);
                fwrite(&idx, 4, 1, f);
            }
    } else if (!strcmp(cmd, "d1")) {
        g_const_u = 1;
        color4_t *px = calloc((size_t)W * H, sizeof(color4_t));
        i32 *hi = calloc((size_t)W * H, sizeof(i32));
        for (u32 y = 0; y < H; y++)
            for (u32 x = 0; x < W; x++) {
                ray_t ray = get_ray(x, y);
                hi[x + y * W] = primary_hit_index(&ray);
                vec3f_t c = ray_color(ray, gc.max_depth);
                fwrite(&c, sizeof c, 1, f);
                px[x + y * W] = to_color4(linear_to_gamma(c));
            }
        fwrite(px, sizeof(color4_t), (size_t)W * H, f);
        fwrite(hi, sizeof(i32), (size_t)W * H, f);
    } else {
        fprintf(stderr, "unknown cmd %s\n", cmd);
        return 2;
    }
    fclose(f);
    return 0;
}
