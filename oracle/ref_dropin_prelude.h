/* Placed in front of the reference's Main.c by oracle/build_ref.sh for the drop-in build. */
#include "rt.h"
static const rt_opts *g_dropin_opts;
static int g_dropin_failed;
/* Replacement for Main.c:1002-1054 inside render_all_parallel(): `width`, `height` and `gc`
 * are that function's own names; num_threads survives because the overlay prints it (:1067). */
#define RT_DROPIN_CALL                                                                        \
    int num_threads = 1;                                                                      \
    {                                                                                         \
        rt_camera cam_ = {                                                                    \
            .pos = { gc.camera.pos.x, gc.camera.pos.y, gc.camera.pos.z },                     \
            .pixel00_loc = { gc.pixel00_loc.x, gc.pixel00_loc.y, gc.pixel00_loc.z },          \
            .pixel_delta_u = { gc.pixel_delta_u.x, gc.pixel_delta_u.y, gc.pixel_delta_u.z },  \
            .pixel_delta_v = { gc.pixel_delta_v.x, gc.pixel_delta_v.y, gc.pixel_delta_v.z },  \
            .defocus_disk_u = { gc.camera.defocus_disk_u.x, gc.camera.defocus_disk_u.y, gc.camera.defocus_disk_u.z }, \
            .defocus_disk_v = { gc.camera.defocus_disk_v.x, gc.camera.defocus_disk_v.y, gc.camera.defocus_disk_v.z }, \
            .defocus_angle = gc.camera.defocus_angle };                                       \
        if (rt_render(&cam_, (uint8_t *)gc.draw_buffer.pixels, width, height,                 \
                      gc.samples_per_pixel, gc.max_depth, g_dropin_opts)) {                   \
            fprintf(stderr, "rt_render: %s\n", rt_last_error());                              \
            g_dropin_failed = 1;                                                              \
        }                                                                                     \
    }
