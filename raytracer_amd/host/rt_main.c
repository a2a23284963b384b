/*
 * rt_main -- headless C host for librtmi.so: the caller the reference's main loop would be with
 * the window taken away.  It does what init_all + one frame of the loop do around the hot path
 * (Main.c:1364-1410, 1440-1443): derive the camera, hand over the scene, call the renderer into
 * a caller-owned RGBA8 buffer, then write the buffer out.
 *
 *   rt_main (--scene spheres.bin | --generate HALF) [--width 400] [--spp 10] [--depth 20] [--out frame.ppm|.tga]
 *           [--strict | --fast] [--const-u] [--seed N] [--chunk N] [--device N | --devices 0,1,...] [--frames N]
 *
 * --generate HALF builds the scene with rt_scene_generate (init_scene with its random block over
 * [-HALF, HALF)^2; 11 = the reference's final scene, 0 = its shipped three-sphere scene).
 * spheres.bin: raw rt_sphere records (40 B each: center[3], radius, mat_type, albedo[3], fuzz,
 * refraction_index -- the reference's sphere_t, Main.c:74-79).
 * .tga output has the layout of the reference's export_image (src/base_graphics.c:583-615):
 * 18-byte header, type 2, 32 bpp, top-left origin flag, BGRA rows.  .ppm is binary P6.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/rt.h"

static int write_ppm(const char *path, const uint8_t *rgba, uint32_t w, uint32_t h)
{
    FILE *f = fopen(path, "wb");
    if (!f) { perror(path); return -1; }
    fprintf(f, "P6\n%u %u\n255\n", w, h);
    for (size_t i = 0; i < (size_t)w * h; i++) fwrite(rgba + 4 * i, 1, 3, f);
    fclose(f);
    return 0;
}

static int write_tga(const char *path, const uint8_t *rgba, uint32_t w, uint32_t h)
{
    FILE *f = fopen(path, "wb");
    if (!f) { perror(path); return -1; }
    uint8_t header[18] = {0};
    header[2] = 2;
    header[12] = w & 0xFF; header[13] = (w >> 8) & 0xFF;
    header[14] = h & 0xFF; header[15] = (h >> 8) & 0xFF;
    header[16] = 32;
    header[17] |= 0x20;
    fwrite(header, 1, 18, f);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        uint8_t bgra[4] = { rgba[4 * i + 2], rgba[4 * i + 1], rgba[4 * i + 0], rgba[4 * i + 3] };
        fwrite(bgra, 1, 4, f);
    }
    fclose(f);
    return 0;
}

int main(int argc, char **argv)
{
    const char *scene = NULL, *out = NULL;
    uint32_t width = 400;
    int spp = 10, depth = 20, device = 0, frames = 1, generate = -1;      /* init_camera's defaults, Main.c:926-927 */
    int devices[16], ndevices = 0;
    rt_opts o;
    rt_stats st;
    memset(&o, 0, sizeof o);
    memset(&st, 0, sizeof st);
    o.stats = &st;
    o.chunk_spp = RT_CHUNK_AUTO;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        const char *v = (i + 1 < argc) ? argv[i + 1] : NULL;
        if (!strcmp(a, "--strict")) o.build = RT_BUILD_STRICT;
        else if (!strcmp(a, "--fast")) o.build = RT_BUILD_FAST;
        else if (!strcmp(a, "--const-u")) o.mode = RT_MODE_CONST_U;
        else if (!v) { fprintf(stderr, "rt_main: %s needs a value\n", a); return 2; }
        else if (!strcmp(a, "--scene")) { scene = v; i++; }
        else if (!strcmp(a, "--generate")) { generate = atoi(v); i++; }
        else if (!strcmp(a, "--out")) { out = v; i++; }
        else if (!strcmp(a, "--width")) { width = (uint32_t)atoi(v); i++; }
        else if (!strcmp(a, "--spp")) { spp = atoi(v); i++; }
        else if (!strcmp(a, "--depth")) { depth = atoi(v); i++; }
        else if (!strcmp(a, "--seed")) { o.seed = (uint32_t)strtoul(v, NULL, 0); i++; }
        else if (!strcmp(a, "--chunk")) { o.chunk_spp = atoi(v); i++; }
        else if (!strcmp(a, "--device")) { device = atoi(v); i++; }
        else if (!strcmp(a, "--devices")) {                /* 0,1,2,...: several GPUs in this process */
            const char *q = v;
            ndevices = 0;
            while (*q && ndevices < 16) { devices[ndevices++] = (int)strtol(q, (char **)&q, 10); if (*q == ',') q++; }
            i++;
        }
        else if (!strcmp(a, "--frames")) { frames = atoi(v); i++; }
        else { fprintf(stderr, "rt_main: unknown option %s\n", a); return 2; }
    }
    if (!scene && generate < 0) { fprintf(stderr, "usage: rt_main (--scene spheres.bin | --generate HALF) [--width W] [--spp N] [--depth D] [--out f.ppm|f.tga]\n"); return 2; }

    size_t n = 0;
    rt_sphere *spheres = NULL;
    if (scene) {
        FILE *f = fopen(scene, "rb");
        if (!f) { perror(scene); return 1; }
        fseek(f, 0, SEEK_END);
        long bytes = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (bytes < 0 || bytes % (long)sizeof(rt_sphere) != 0) { fprintf(stderr, "rt_main: %s is not a whole number of 40-byte spheres\n", scene); return 1; }
        n = (size_t)bytes / sizeof(rt_sphere);
        spheres = (rt_sphere *)malloc(bytes > 0 ? (size_t)bytes : 1);
        if (!spheres || fread(spheres, sizeof(rt_sphere), n, f) != n) { fprintf(stderr, "rt_main: cannot read %s\n", scene); return 1; }
        fclose(f);
    } else {
        size_t cap = 4 * (size_t)generate * (size_t)generate + 4;
        spheres = (rt_sphere *)malloc(cap * sizeof(rt_sphere));
        if (!spheres || rt_scene_generate(generate, 0, spheres, cap, &n)) { fprintf(stderr, "rt_main: %s\n", rt_last_error()); return 1; }
    }

    uint32_t height = rt_height_for(width, 16.0f / 9.0f);             /* init_all, Main.c:1366-1369 */
    rt_camera_params cp;
    rt_camera cam;
    rt_camera_defaults(&cp);
    /* the caller owns the framebuffer; page-locked memory makes the read-back one DMA transfer */
    uint8_t *rgba = NULL;
    int rgba_locked = 0;

    if ((ndevices > 0 ? rt_init_devices(devices, ndevices) : rt_init(device)) || rt_set_world(spheres, n) || rt_camera_setup(&cp, width, height, &cam)) {
        fprintf(stderr, "rt_main: %s\n", rt_last_error());
        return 1;
    }
    rgba = (uint8_t *)rt_host_alloc((size_t)width * height * 4);
    rgba_locked = rgba != NULL;
    if (!rgba) rgba = (uint8_t *)malloc((size_t)width * height * 4);
    if (!rgba) { fprintf(stderr, "rt_main: no memory for the framebuffer\n"); return 1; }
    for (int fr = 0; fr < frames; fr++) {
        if (rt_render(&cam, rgba, width, height, spp, depth, &o)) {
            fprintf(stderr, "rt_main: %s\n", rt_last_error());
            return 1;
        }
        printf("frame %d: %ux%u, %d spp, depth %d, %zu spheres: %.2f ms (kernel %.2f ms), %.1f Msamples/s, %.3f rays/sample\n",
               fr, width, height, spp, depth, n, st.total_ms, st.kernel_ms,
               (double)st.samples / (st.total_ms * 1e3), st.samples ? (double)st.rays / (double)st.samples : 0.0);
    }
    int rc = 0;
    if (out) {
        size_t len = strlen(out);
        rc = (len > 4 && !strcmp(out + len - 4, ".tga")) ? write_tga(out, rgba, width, height)
                                                         : write_ppm(out, rgba, width, height);
    }
    if (rgba_locked) rt_host_free(rgba); else free(rgba);      /* page-locked memory goes back while the library is open */
    rt_shutdown();
    free(spheres);
    return rc ? 1 : 0;
}
