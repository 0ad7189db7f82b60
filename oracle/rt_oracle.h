/*
 * rt_oracle.h — CPU restatement of the reference render loop.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle: a scalar, strict-FP32 (no FMA contraction, doubles only
 * where the reference calls System.Math) restatement in plain C of
 *     /root/reference/MyApplication.cs:352-452   and
 *     /root/reference/objects.cs:35-273,348-353.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it;
 * the shipped library never links or calls it.
 *
 * PARITY UNPINNED: the reference is C#/.NET 6 + OpenTK 4.8.2 and cannot be built or
 * run here, and it ships no tests, golden images or known-answer vectors.  The
 * oracle is therefore pinned only by hand-derived known answers taken from the
 * reference source (tests/test_oracle_kat.py); OpenTK.Mathematics' operator
 * definitions (Dot/Cross/Normalized/Clamp, Vector4 / float) are restated from its
 * published source, not verified against a build.
 *
 * The scene and camera types are the public ABI structs of include/rt_hip.h so the
 * oracle and the library consume byte-identical inputs.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>
#include "../include/rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_texture {
    const uint8_t *rgba8;   /* row-major, row 0 first, R,G,B,A bytes */
    int32_t width, height;
} orc_texture;

/* Work counters.  In memoised mode they count the de-duplicated algorithm (one
 * reflection trace per level); in literal mode everything the reference executes. */
typedef struct orc_counters {
    uint64_t rays_primary;
    uint64_t rays_shadow;
    uint64_t rays_secondary;       /* reflection traces whose result is used (depth < max_depth) */
    uint64_t rays_secondary_dead;  /* reflection traces at depth == max_depth: traced by the
                                      reference (objects.cs:83), result discarded (:84)          */
    uint64_t shaded_hits;          /* Shade() invocations                                        */
    uint64_t sphere_tests;
    uint64_t plane_tests;
    uint64_t triangle_tests;
} orc_counters;

#define ORC_MODE_MEMOISED 0  /* reflected colour computed once per level (bit-identical) */
#define ORC_MODE_LITERAL  1  /* reflected colour re-traced once per light, objects.cs:80-88 */

/* GetRayDirection, MyApplication.cs:428-452. */
void orc_ray_direction(const rt_camera *cam, int x, int y, int w, int h, float out_dir[3]);

/* Scene.Intersect, MyApplication.cs:352-368.  Returns primitive index or -1. */
int orc_intersect(const rt_scene *s, const float o[3], const float d[3],
                  float *out_t, float out_hit[3]);

/* Scene.IsInShadow, MyApplication.cs:370-380. */
int orc_in_shadow(const rt_scene *s, const float o[3], const float d[3], float eps, float dist);

/* Single-primitive Intersect (objects.cs:109-126,155-169,236-268): 1 = hit. */
int orc_prim_intersect(const rt_scene *s, int prim, const float o[3], const float d[3],
                       float *out_t, float out_hit[3]);

/* Primitive.Shade, objects.cs:44-96 (depth bound parametrised: reference = 5). */
void orc_shade(const rt_scene *s, const orc_texture *tex, int ntex, int prim,
               const float hit[3], const float view[3], int depth, int max_depth,
               int mode, float out_rgba[4]);

/* Primitive.MixColor, objects.cs:35-41. */
int32_t orc_mix_color(const float rgba[4]);

/* Plane.GetTextureCoordinates objects.cs:190-204 and the texel address of
 * Texture.GetColor objects.cs:348-353. */
void orc_texture_coords(const float hit[3], float *u, float *v);
void orc_texel(const float hit[3], int tex_w, int tex_h, int *tx, int *ty);

/* Plane.GetCheckerboardColor objects.cs:176-188: 1 = white, 0 = black. */
int orc_checker_white(const float hit[3]);

/*
 * Raytracer.Render, MyApplication.cs:402-426, for rows row_begin, row_begin+row_step,
 * ... < row_end of a w x h frame.  out receives full-frame addressing: out[y*w+x]
 * (rows that are not rendered are left untouched).  nthreads <= 0: all cores.
 * Returns 0, or -1 on invalid arguments.
 */
int orc_render(const rt_scene *s, const orc_texture *tex, int ntex, const rt_camera *cam,
               int w, int h, int row_begin, int row_end, int row_step, int max_depth,
               int mode, int nthreads, int32_t *out, orc_counters *counters);

int orc_num_threads(void);

/* The loop body of Raytracer.Debug, MyApplication.cs:461-472, for pixel (x, y): ray direction,
 * closest hit (returns the primitive index or -1), normal there, mirror direction. */
int orc_debug_ray(const rt_scene *s, const rt_camera *cam, int x, int y, int w, int h, float out_dir[3],
                  float *out_t, float out_hit[3], float out_normal[3], float out_reflected[3]);

/*
 * The present path's anti-aliasing filter, /root/reference/shaders/screen_fxaa_fs.glsl:80-140
 * as main() applies it to the CPU-rendered texture (:426-431), restated over the 0x00RRGGBB
 * frame: pixel (x, y) is the fragment with uv * resolution = (x + 0.5, y + 0.5); texture2D is
 * the sampler surface.cs:36-38 sets up (LINEAR filter, the default REPEAT wrap, texel =
 * byte / 255) evaluated in FP32; the colour goes back to bytes by round-to-nearest.  What a GL
 * implementation computes differs in the precision of its filter weights and its interpolation of
 * uv, so this pins the arithmetic of the library's post-pass, not a particular GPU's output.
 */
void orc_fxaa(const int32_t *in, int w, int h, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif
