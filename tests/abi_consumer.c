/*
 * abi_consumer.c — a plain-C host of librt_hip.so: the closest thing to the C# host's P/Invoke
 * binding this image can compile.  Built with `gcc -std=c99 -Wall -Werror -pedantic` against
 * include/rt_hip.h only (tests/test_abi_consumer.py), so the header is exercised as C, its struct
 * layouts are pinned at compile time to what `StructLayout(LayoutKind.Sequential)` structs on the
 * C# side would assume, and a frame goes through the same calls that host makes:
 * rt_create, rt_set_scene, rt_render, rt_get_stats, rt_last_error, rt_destroy.
 *
 * usage: abi_consumer scene.bin width height max_depth expected.raw
 *   scene.bin      int32 num_prims, int32 num_lights, then the arrays of rt_scene in field order
 *   expected.raw   width*height int32, the frame the CPU restatement renders
 * exit status 0 = identical frame; 2 = no usable device (reported, not a failure of the ABI).
 */
#include "rt_hip.h"

#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* C99 has no _Static_assert: a negative array size stops the compilation instead. */
#define LAYOUT_CHECK(name, cond) typedef char layout_check_##name[(cond) ? 1 : -1]

LAYOUT_CHECK(camera_size, sizeof(rt_camera) == 40);
LAYOUT_CHECK(camera_direction, offsetof(rt_camera, direction) == 12);
LAYOUT_CHECK(camera_up, offsetof(rt_camera, up) == 24);
LAYOUT_CHECK(camera_fov, offsetof(rt_camera, fov_deg) == 36);
LAYOUT_CHECK(scene_size, sizeof(rt_scene) == 8 + 16 * sizeof(void *));
LAYOUT_CHECK(scene_types, offsetof(rt_scene, prim_types) == 8);
LAYOUT_CHECK(scene_texture, offsetof(rt_scene, prim_texture) == 8 + 13 * sizeof(void *));
LAYOUT_CHECK(scene_light_colors, offsetof(rt_scene, light_colors) == 8 + 15 * sizeof(void *));
LAYOUT_CHECK(stats_size, sizeof(rt_stats) == 13 * 8 + 6 * 8 + 2 * 4);
LAYOUT_CHECK(stats_shadow_candidates, offsetof(rt_stats, shadow_candidates) == 12 * 8);
LAYOUT_CHECK(stats_kernel_ms, offsetof(rt_stats, kernel_ms) == 13 * 8);
LAYOUT_CHECK(stats_copy_ms, offsetof(rt_stats, copy_ms) == 18 * 8);
LAYOUT_CHECK(stats_num_devices, offsetof(rt_stats, num_devices) == 19 * 8);
LAYOUT_CHECK(debug_ray_size, sizeof(rt_debug_ray) == 56);
LAYOUT_CHECK(debug_ray_prim, offsetof(rt_debug_ray, prim) == 12);
LAYOUT_CHECK(debug_ray_reflected, offsetof(rt_debug_ray, reflected) == 44);
LAYOUT_CHECK(status_is_int, sizeof(rt_status) == sizeof(int));

static void *read_array(FILE *f, size_t count, size_t elem)
{
    void *p = malloc(count * elem + 16);
    if (!p || fread(p, elem, count, f) != count) {
        fprintf(stderr, "abi_consumer: short scene file\n");
        exit(1);
    }
    return p;
}

int main(int argc, char **argv)
{
    rt_ctx *ctx = NULL;
    rt_scene s;
    rt_camera cam;
    rt_stats st;
    FILE *f;
    int32_t counts[2], *pixels, *expected;
    int w, h, depth, rc;
    size_t n, nl, i, npx, differ = 0;

    if (argc != 6) {
        fprintf(stderr, "usage: %s scene.bin width height max_depth expected.raw\n", argv[0]);
        return 1;
    }
    w = atoi(argv[2]);
    h = atoi(argv[3]);
    depth = atoi(argv[4]);
    npx = (size_t)w * (size_t)h;
    printf("abi_consumer: library version %d, header version %d\n", rt_api_version(), RT_API_VERSION);
    if (rt_api_version() != RT_API_VERSION) return 1;

    f = fopen(argv[1], "rb");
    if (!f || fread(counts, sizeof counts[0], 2, f) != 2) {
        fprintf(stderr, "abi_consumer: cannot read %s\n", argv[1]);
        return 1;
    }
    n = (size_t)counts[0];
    nl = (size_t)counts[1];
    memset(&s, 0, sizeof s);
    s.num_prims = counts[0];
    s.num_lights = counts[1];
    s.prim_types = read_array(f, n, 4);
    s.prim_positions = read_array(f, 3 * n, 4);
    s.prim_normals = read_array(f, 3 * n, 4);
    s.prim_diffuse = read_array(f, 4 * n, 4);
    s.prim_glossy = read_array(f, 4 * n, 4);
    s.prim_ambient = read_array(f, 4 * n, 4);
    s.prim_specular = read_array(f, n, 4);
    s.prim_radius = read_array(f, n, 4);
    s.prim_d = read_array(f, n, 4);
    s.prim_v0 = read_array(f, 3 * n, 4);
    s.prim_v1 = read_array(f, 3 * n, 4);
    s.prim_v2 = read_array(f, 3 * n, 4);
    s.prim_checkerboard = read_array(f, n, 4);
    s.prim_texture = read_array(f, n, 4);
    s.light_positions = read_array(f, 3 * nl, 4);
    s.light_colors = read_array(f, 4 * nl, 4);
    fclose(f);

    f = fopen(argv[5], "rb");
    expected = malloc(npx * sizeof *expected);
    pixels = malloc(npx * sizeof *pixels);
    if (!f || !expected || !pixels || fread(expected, sizeof *expected, npx, f) != npx) {
        fprintf(stderr, "abi_consumer: cannot read %s\n", argv[5]);
        return 1;
    }
    fclose(f);

    /* The reference's camera defaults, MyApplication.cs:267-275. */
    memset(&cam, 0, sizeof cam);
    cam.direction[2] = 1.0f;
    cam.up[1] = 1.0f;
    cam.fov_deg = 90.0f;

    /* Errors follow the convention: a status and a message, never a crash. */
    rc = rt_render(NULL, &cam, w, h, depth, pixels);
    if (rc != RT_ERR_INVALID_ARG) return 1;
    rc = rt_create(&ctx, NULL, 0);
    if (rc == RT_ERR_NO_DEVICE) {
        printf("abi_consumer: no device (%s)\n", rt_last_error(NULL));
        return 2;
    }
    if (rc != RT_OK) {
        fprintf(stderr, "abi_consumer: rt_create: %s\n", rt_last_error(NULL));
        return 1;
    }
    rc = rt_render(ctx, &cam, w, h, depth, pixels);
    if (rc != RT_ERR_NO_SCENE || strlen(rt_last_error(ctx)) == 0) return 1;
    if ((rc = rt_set_scene(ctx, &s)) != RT_OK || (rc = rt_render(ctx, &cam, w, h, depth, pixels)) != RT_OK ||
        (rc = rt_get_stats(ctx, &st)) != RT_OK || (rc = rt_frame_status(ctx)) != RT_OK) {
        fprintf(stderr, "abi_consumer: status %d: %s\n", rc, rt_last_error(ctx));
        rt_destroy(ctx);
        return 1;
    }
    for (i = 0; i < npx; i++) differ += pixels[i] != expected[i];
    printf("abi_consumer: %dx%d depth %d: %lu of %lu pixels differ; %lu primary, %lu shadow, %lu secondary rays\n", w, h,
           depth, (unsigned long)differ, (unsigned long)npx, (unsigned long)st.rays_primary, (unsigned long)st.rays_shadow,
           (unsigned long)st.rays_secondary);
    rt_destroy(ctx);
    rt_destroy(NULL);
    return differ == 0 && st.rays_primary == npx ? 0 : 1;
}
