/*
 * Drop-in demonstration driver (test infrastructure; built by oracle/build_ref.sh).
 *
 * This text is appended to the reference's own Main.c, streamed from /root/reference with ONE
 * edit: the tile/thread dispatch of render_all_parallel (Main.c:1002-1054) is replaced by the
 * RT_DROPIN_CALL macro below, i.e. by a call to rt_render().  Everything else in that function
 * is the reference's code and still runs: framebuffer allocation from the frame arena
 * (Main.c:998), clear_screen (:1000), the frame-time text overlay drawn with the bitmap font
 * (:1056-1068).  The driver then writes the framebuffer the reference would hand to
 * present_frame (Main.c:1445).
 *
 * usage: ref_dropin_<s3|sf> <width> <spp> <depth> <out.rgba> [build] [seed] [chunk]
 */
int ref_lcg_fast_rand(void);      /* the reference's fast_rand, renamed at build time */
int fast_rand(void) { return ref_lcg_fast_rand(); }

int main(int argc, char **argv)
{
    if (argc < 5) {
        fprintf(stderr, "usage: %s <width> <spp> <depth> <out.rgba> [build] [seed] [chunk]\n", argv[0]);
        return 2;
    }
    static rt_opts opts;
    opts.build = argc > 5 ? atoi(argv[5]) : 0;
    opts.seed = argc > 6 ? (uint32_t)strtoul(argv[6], NULL, 0) : 0;
    opts.chunk_spp = argc > 7 ? atoi(argv[7]) : RT_CHUNK_AUTO;
    g_dropin_opts = &opts;

    fast_srand(0);
    init_camera(atoi(argv[1]), 16.0f / 9.0f);                 /* Main.c:1369 */
    init_scene();                                             /* Main.c:1370 */
    gc.samples_per_pixel = atoi(argv[2]);
    gc.max_depth = atoi(argv[3]);
    gc.frame_arena = arena_new();                             /* Main.c:1375 */
    gc.font = init_font((u32 *)font_pixels);                  /* Main.c:1381 */

    /* the binding a maintainer adds after init_scene: flatten gc.scene_objects into rt_sphere records */
    size_t n = gc.scene_objects->count;
    rt_sphere *tmp = calloc(n ? n : 1, sizeof *tmp);
    for (size_t i = 0; i < n; i++) {
        sphere_t *s = (sphere_t *)gc.scene_objects->objects[i].object;
        tmp[i].center = (rt_vec3){ s->center.x, s->center.y, s->center.z };
        tmp[i].radius = s->radius;
        tmp[i].mat_type = s->mat.mat_type;
        tmp[i].albedo = (rt_vec3){ s->mat.albedo.x, s->mat.albedo.y, s->mat.albedo.z };
        tmp[i].fuzz = s->mat.fuzz;
        tmp[i].refraction_index = s->mat.refraction_index;
    }
    if (rt_init(0) || rt_set_world(tmp, n)) {
        fprintf(stderr, "ref_dropin: %s\n", rt_last_error());
        return 1;
    }
    free(tmp);

    render_all_parallel();                                    /* Main.c:1442 */

    FILE *f = fopen(argv[4], "wb");
    if (!f) { perror(argv[4]); return 1; }
    fwrite(gc.draw_buffer.pixels, sizeof(color4_t), (size_t)gc.draw_buffer.width * gc.draw_buffer.height, f);
    fclose(f);
    rt_shutdown();
    return g_dropin_failed;
}
