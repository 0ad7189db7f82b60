/*
 * rt_oracle.c — CPU restatement of the reference render loop.  TEST INFRASTRUCTURE ONLY
 * (see rt_oracle.h: parity unpinned; never linked into the shipped library).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC (oracle/Makefile).
 * Every float expression below is written operation by operation in the order the C#
 * evaluates it (left to right, .NET 6 RyuJIT x64 scalar SSE: IEEE binary32, no
 * contraction); `double` appears exactly where the reference calls System.Math.
 */
#include "rt_oracle.h"

#include <limits.h>
#include <math.h>
#include <stddef.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

/* ---- OpenTK.Mathematics 4.8.2 Vector3 / Vector4 operators (restated) ------------- */

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_load(const float *p) { v3 r = {p[0], p[1], p[2]}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
/* Vector3.Dot: (l.X*r.X) + (l.Y*r.Y) + (l.Z*r.Z), left to right. */
static inline float v3_dot(v3 a, v3 b) { return ((a.x * b.x) + (a.y * b.y)) + (a.z * b.z); }
/* Vector3.Cross. */
static inline v3 v3_cross(v3 a, v3 b)
{
    return v3_make((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x));
}
/* Vector3.Length = MathF.Sqrt(X*X + Y*Y + Z*Z). */
static inline float v3_length(v3 a) { return sqrtf(((a.x * a.x) + (a.y * a.y)) + (a.z * a.z)); }
/* Vector3.Normalize(): scale = 1.0f / Length; each component *= scale. */
static inline v3 v3_normalized(v3 a)
{
    float scale = 1.0f / v3_length(a);
    return v3_make(a.x * scale, a.y * scale, a.z * scale);
}

static inline v4 v4_make(float x, float y, float z, float w) { v4 r = {x, y, z, w}; return r; }
static inline v4 v4_load(const float *p) { v4 r = {p[0], p[1], p[2], p[3]}; return r; }
static inline v4 v4_add(v4 a, v4 b) { return v4_make(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline v4 v4_mul(v4 a, v4 b) { return v4_make(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
static inline v4 v4_scale(v4 a, float s) { return v4_make(a.x * s, a.y * s, a.z * s, a.w * s); }
/* Vector4 / float: per-component divide. */
static inline v4 v4_div(v4 a, float s) { return v4_make(a.x / s, a.y / s, a.z / s, a.w / s); }
static inline float clamp1(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* Vector4.Clamp: v < min ? min : v > max ? max : v per component (NaN passes through). */
static inline v4 v4_clamp01(v4 a)
{
    return v4_make(clamp1(a.x, 0.0f, 1.0f), clamp1(a.y, 0.0f, 1.0f), clamp1(a.z, 0.0f, 1.0f),
                   clamp1(a.w, 0.0f, 1.0f));
}

/* System.Math.Max(float, float), .NET Core 3.0+: IEEE 754-2019 `maximum`
 * (NaN propagates, +0 > -0). */
static inline float net_maxf(float a, float b)
{
    if (a != b) {
        if (!isnan(a)) return b < a ? a : b;
        return a;
    }
    return signbit(b) ? a : b;
}

/* C# (int)float on x64 = cvttss2si: truncation; NaN and out-of-range give INT_MIN. */
static inline int32_t net_f2i(float f)
{
    if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT_MIN;
    return (int32_t)f;
}

/* ---- per-thread counters ----------------------------------------------------------- */

typedef struct {
    const rt_scene *s;
    const orc_texture *tex;
    int ntex;
    int max_depth;
    int mode;
    orc_counters c;
} ctx_t;

/* ---- Sphere / Plane / Triangle Intersect ------------------------------------------- */

/* Sphere.Intersect, objects.cs:109-126. */
static int sphere_intersect(const rt_scene *s, int i, v3 o, v3 d, float *t_out, v3 *hit_out)
{
    v3 p = v3_load(s->prim_positions + 3 * i);
    float radius = s->prim_radius[i];
    float a = v3_dot(d, d);
    float b = 2.0f * v3_dot(d, v3_sub(o, p));
    float c = v3_dot(v3_sub(o, p), v3_sub(o, p)) - (radius * radius);
    float disc = (b * b) - ((4.0f * a) * c);
    if (disc < 0.0f) return 0;
    float dist = ((-b) - sqrtf(disc)) / (2.0f * a);
    *t_out = dist;
    *hit_out = v3_add(o, v3_scale(d, dist));
    return 1;
}

/* Plane.Intersect, objects.cs:155-169. */
static int plane_intersect(const rt_scene *s, int i, v3 o, v3 d, float *t_out, v3 *hit_out)
{
    v3 n = v3_load(s->prim_normals + 3 * i);
    float denom = v3_dot(n, d);
    if (fabsf(denom) <= 0.0001f) return 0;
    float t = ((-v3_dot(n, o)) - s->prim_d[i]) / denom;
    if (t <= 0.0001f) return 0;
    *t_out = t;
    *hit_out = v3_add(o, v3_scale(d, t));
    return 1;
}

/* Triangle.Intersect, objects.cs:236-268. */
static int triangle_intersect(const rt_scene *s, int i, v3 o, v3 d, float *t_out, v3 *hit_out)
{
    v3 v0 = v3_load(s->prim_v0 + 3 * i);
    v3 v1 = v3_load(s->prim_v1 + 3 * i);
    v3 v2 = v3_load(s->prim_v2 + 3 * i);
    v3 e1 = v3_sub(v1, v0);
    v3 e2 = v3_sub(v2, v0);
    v3 h = v3_cross(d, e2);
    float a = v3_dot(e1, h);
    if (a > -0.0001f && a < 0.0001f) return 0;
    float f = 1.0f / a;
    v3 sv = v3_sub(o, v0);
    float u = f * v3_dot(sv, h);
    if (u < 0.0f || u > 1.0f) return 0;
    v3 q = v3_cross(sv, e1);
    float v = f * v3_dot(d, q);
    if (v < 0.0f || (u + v) > 1.0f) return 0;
    float t = f * v3_dot(e2, q);
    if (t > 0.0001f) {
        *t_out = t;
        *hit_out = v3_add(o, v3_scale(d, t));
        return 1;
    }
    return 0;
}

static int prim_intersect(ctx_t *cx, const rt_scene *s, int i, v3 o, v3 d, float *t, v3 *hit)
{
    switch (s->prim_types[i]) {
    case RT_PRIM_SPHERE:
        if (cx) cx->c.sphere_tests++;
        return sphere_intersect(s, i, o, d, t, hit);
    case RT_PRIM_PLANE:
        if (cx) cx->c.plane_tests++;
        return plane_intersect(s, i, o, d, t, hit);
    default:
        if (cx) cx->c.triangle_tests++;
        return triangle_intersect(s, i, o, d, t, hit);
    }
}

/* Scene.Intersect, MyApplication.cs:352-368: strict `<` keeps the lowest index on ties. */
static int scene_intersect(ctx_t *cx, const rt_scene *s, v3 o, v3 d, float *t_out, v3 *hit_out)
{
    const float epsilon = 0.0001f;
    int best = -1;
    float distance = 3.40282347e+38f; /* float.MaxValue */
    for (int i = 0; i < s->num_prims; i++) {
        float t;
        v3 hit;
        if (prim_intersect(cx, s, i, o, d, &t, &hit) && t < distance && t > epsilon) {
            distance = t;
            best = i;
            *t_out = t;
            *hit_out = hit;
        }
    }
    return best;
}

/* Scene.IsInShadow, MyApplication.cs:370-380: first occluder in insertion order ends the scan. */
static int scene_in_shadow(ctx_t *cx, const rt_scene *s, v3 o, v3 d, float epsilon, float distance)
{
    for (int i = 0; i < s->num_prims; i++) {
        float t;
        v3 hit;
        if (prim_intersect(cx, s, i, o, d, &t, &hit) && t > epsilon && t < distance) return 1;
    }
    return 0;
}

/* ---- Plane checkerboard / texture --------------------------------------------------- */

/* Plane.GetCheckerboardColor, objects.cs:176-188: sin and the product are double. */
static int checker_white(v3 hit)
{
    const float scale = 7.5f;
    float sines = (float)((sin((double)(hit.x * scale)) * sin((double)(hit.y * scale))) *
                          sin((double)(hit.z * scale)));
    return sines < 0.0f ? 0 : 1;
}

/* Plane.GetTextureCoordinates, objects.cs:190-204.  C# float % is fmodf. */
static void texture_coords(v3 hit, float *u_out, float *v_out)
{
    float u = fmodf((hit.x / 5.0f) - 0.5f, 1.0f);
    float v = fmodf((hit.y / 5.0f) - 0.2f, 1.0f);
    u = -u;
    if (u < 0.0f) u += 1.0f;
    if (v < 0.0f) v += 1.0f;
    *u_out = u;
    *v_out = v;
}

/* Texture.GetColor, objects.cs:348-353 + LoadTexture :343 (pixels[x,y] = byte / 255f).
 * The reference throws IndexOutOfRange for an address outside the image (NaN/inf
 * coordinates); the restatement clamps instead and the tests never go there. */
static v4 texture_color(const orc_texture *t, v3 hit)
{
    float u, v;
    texture_coords(hit, &u, &v);
    int x = net_f2i(u * (float)(t->width - 1));
    int y = net_f2i(v * (float)(t->height - 1));
    if (x < 0) x = 0;
    if (x > t->width - 1) x = t->width - 1;
    if (y < 0) y = 0;
    if (y > t->height - 1) y = t->height - 1;
    const uint8_t *px = t->rgba8 + 4 * ((size_t)y * (size_t)t->width + (size_t)x);
    return v4_make(px[0] / 255.0f, px[1] / 255.0f, px[2] / 255.0f, px[3] / 255.0f);
}

/* ---- Primitive.Shade, objects.cs:44-96 --------------------------------------------- */

static v3 prim_normal(const rt_scene *s, int i, v3 hit)
{
    if (s->prim_types[i] == RT_PRIM_SPHERE) /* objects.cs:128-131 */
        return v3_normalized(v3_sub(hit, v3_load(s->prim_positions + 3 * i)));
    return v3_load(s->prim_normals + 3 * i); /* objects.cs:171-174, 270-273 */
}

static v4 shade(ctx_t *cx, int prim, v3 hit, v3 view, int depth)
{
    const rt_scene *s = cx->s;
    cx->c.shaded_hits++;

    v3 normal = prim_normal(s, prim, hit);
    v4 final = v4_make(0.0f, 0.0f, 0.0f, 0.0f);

    v4 mat_diffuse = v4_load(s->prim_diffuse + 4 * prim);
    v4 mat_glossy = v4_load(s->prim_glossy + 4 * prim);
    v4 mat_ambient = v4_load(s->prim_ambient + 4 * prim);
    float mat_specular = s->prim_specular[prim];
    int is_plane = s->prim_types[prim] == RT_PRIM_PLANE;
    int tex_slot = (is_plane && s->prim_texture) ? s->prim_texture[prim] : -1;
    if (tex_slot >= cx->ntex) tex_slot = -1;
    int checker = is_plane && s->prim_checkerboard[prim] != 0;

    /* Memoised mode: the reflected colour does not depend on the light, so one trace per
     * level gives the same value the reference recomputes inside the light loop. */
    int have_memo = 0, memo_valid = 0;
    v4 memo = v4_make(0.0f, 0.0f, 0.0f, 0.0f);

    for (int l = 0; l < s->num_lights; l++) {
        v3 lp = v3_load(s->light_positions + 3 * l);
        v4 lc = v4_load(s->light_colors + 4 * l);
        v3 shadow_ray = v3_normalized(v3_sub(lp, hit));
        float distance = v3_length(v3_sub(lp, hit));
        float diffuse = net_maxf(v3_dot(normal, shadow_ray), 0.0f);
        float glossy = 1.0f;

        cx->c.rays_shadow++;
        if (scene_in_shadow(cx, s, hit, shadow_ray, 0.0001f, distance)) {
            diffuse = 0.0f;
            glossy = 0.0f;
        }

        v4 diffuse_color = v4_scale(mat_diffuse, diffuse);
        if (is_plane) {
            if (tex_slot >= 0)
                diffuse_color = v4_scale(texture_color(&cx->tex[tex_slot], hit), diffuse);
            else if (checker) {
                float cw = checker_white(hit) ? 1.0f : 0.0f;
                diffuse_color = v4_scale(v4_make(cw, cw, cw, 1.0f), diffuse);
            }
        }

        v3 reflection = v3_sub(shadow_ray, v3_scale(normal, 2.0f * v3_dot(shadow_ray, normal)));
        float n = 250.0f;
        float spec_pow =
            (float)pow((double)net_maxf(v3_dot(reflection, view), 0.0f), (double)n);
        v4 glossy_color = v4_scale(v4_scale(mat_glossy, spec_pow), glossy);

        if (mat_specular != 0.0f) {
            int valid;
            v4 reflected = v4_make(0.0f, 0.0f, 0.0f, 0.0f);
            if (cx->mode == ORC_MODE_MEMOISED && have_memo) {
                valid = memo_valid;
                reflected = memo;
            } else {
                v3 specular_dir =
                    v3_normalized(v3_sub(view, v3_scale(normal, 2.0f * v3_dot(view, normal))));
                float t2;
                v3 hit2;
                int prim2;
                if (depth < cx->max_depth) {
                    cx->c.rays_secondary++;
                    prim2 = scene_intersect(cx, s, hit, specular_dir, &t2, &hit2);
                } else {
                    /* objects.cs:83-84: traced, then discarded by `depth < 5`. */
                    cx->c.rays_secondary_dead++;
                    prim2 = cx->mode == ORC_MODE_LITERAL
                                ? scene_intersect(cx, s, hit, specular_dir, &t2, &hit2)
                                : -1;
                }
                valid = prim2 >= 0 && depth < cx->max_depth;
                if (valid) reflected = shade(cx, prim2, hit2, specular_dir, depth + 1);
                have_memo = 1;
                memo_valid = valid;
                memo = reflected;
            }
            if (valid) {
                /* finalColor += (Vector4)Shade(..) * material.specular * 1 / (distance*distance) */
                v4 term = v4_div(v4_scale(v4_scale(reflected, mat_specular), 1.0f),
                                 distance * distance);
                final = v4_add(final, term);
            }
        }

        /* finalColor += light.color * 1 / (distance*distance) * (diffuseColor + glossyColor)
         *               + material.ambientColor */
        v4 light_term = v4_mul(v4_div(v4_scale(lc, 1.0f), distance * distance),
                               v4_add(diffuse_color, glossy_color));
        final = v4_add(final, v4_add(light_term, mat_ambient));
    }
    return v4_clamp01(final);
}

/* ---- Raytracer.GetRayDirection, MyApplication.cs:428-452 --------------------------- */

static v3 ray_direction(const rt_camera *cam, int x, int y, int w, int h)
{
    const float deg2rad = 3.14159274f / 180.0f; /* MathHelper: MathF.PI / 180f */
    float aspect = (float)w / (float)h;
    float angle_h = (cam->fov_deg * aspect) * deg2rad;
    float angle_v = cam->fov_deg * deg2rad;
    float sx = (((float)x / (float)w) * 2.0f) - 1.0f;
    float sy = (((float)y / (float)h) * 2.0f) - 1.0f;
    sx *= (float)tan((double)(angle_h / 2.0f));
    sy *= (float)tan((double)(angle_v / 2.0f));
    v3 dir = v3_load(cam->direction);
    v3 up = v3_load(cam->up);
    v3 left = v3_normalized(v3_cross(up, dir));
    v3 r = v3_add(v3_add(dir, v3_scale(up, sy)), v3_scale(left, sx));
    return v3_normalized(r);
}

/* Primitive.MixColor, objects.cs:35-41. */
static int32_t mix_color(v4 c)
{
    uint32_t red = (uint32_t)net_f2i(c.x * 255.0f);
    uint32_t green = (uint32_t)net_f2i(c.y * 255.0f);
    uint32_t blue = (uint32_t)net_f2i(c.z * 255.0f);
    return (int32_t)((red << 16) + (green << 8) + blue); /* C# int arithmetic wraps */
}

/* ---- public wrappers ---------------------------------------------------------------- */

static void store3(float *o, v3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

void orc_ray_direction(const rt_camera *cam, int x, int y, int w, int h, float out_dir[3])
{
    store3(out_dir, ray_direction(cam, x, y, w, h));
}

int orc_intersect(const rt_scene *s, const float o[3], const float d[3], float *out_t,
                  float out_hit[3])
{
    float t = 0.0f;
    v3 hit = {0, 0, 0};
    int p = scene_intersect(NULL, s, v3_load(o), v3_load(d), &t, &hit);
    if (p >= 0) {
        if (out_t) *out_t = t;
        if (out_hit) store3(out_hit, hit);
    }
    return p;
}

int orc_in_shadow(const rt_scene *s, const float o[3], const float d[3], float eps, float dist)
{
    return scene_in_shadow(NULL, s, v3_load(o), v3_load(d), eps, dist);
}

int orc_prim_intersect(const rt_scene *s, int prim, const float o[3], const float d[3],
                       float *out_t, float out_hit[3])
{
    float t = 0.0f;
    v3 hit = {0, 0, 0};
    if (prim < 0 || prim >= s->num_prims) return 0;
    int r = prim_intersect(NULL, s, prim, v3_load(o), v3_load(d), &t, &hit);
    if (r) {
        if (out_t) *out_t = t;
        if (out_hit) store3(out_hit, hit);
    }
    return r;
}

void orc_shade(const rt_scene *s, const orc_texture *tex, int ntex, int prim, const float hit[3],
               const float view[3], int depth, int max_depth, int mode, float out_rgba[4])
{
    ctx_t cx;
    memset(&cx, 0, sizeof cx);
    cx.s = s;
    cx.tex = tex;
    cx.ntex = ntex;
    cx.max_depth = max_depth;
    cx.mode = mode;
    v4 c = shade(&cx, prim, v3_load(hit), v3_load(view), depth);
    out_rgba[0] = c.x;
    out_rgba[1] = c.y;
    out_rgba[2] = c.z;
    out_rgba[3] = c.w;
}

int32_t orc_mix_color(const float rgba[4]) { return mix_color(v4_load(rgba)); }

void orc_texture_coords(const float hit[3], float *u, float *v)
{
    texture_coords(v3_load(hit), u, v);
}

void orc_texel(const float hit[3], int tex_w, int tex_h, int *tx, int *ty)
{
    float u, v;
    texture_coords(v3_load(hit), &u, &v);
    *tx = net_f2i(u * (float)(tex_w - 1));
    *ty = net_f2i(v * (float)(tex_h - 1));
}

int orc_checker_white(const float hit[3]) { return checker_white(v3_load(hit)); }

int orc_debug_ray(const rt_scene *s, const rt_camera *cam, int x, int y, int w, int h, float out_dir[3],
                  float *out_t, float out_hit[3], float out_normal[3], float out_reflected[3])
{
    const v3 d = ray_direction(cam, x, y, w, h);                       /* :461 */
    store3(out_dir, d);
    float t = 0.0f;
    v3 hit = {0, 0, 0};
    const int p = scene_intersect(NULL, s, v3_load(cam->position), d, &t, &hit);   /* :464 */
    if (p < 0) return -1;
    const v3 n = prim_normal(s, p, hit);                               /* :471 */
    const v3 r = v3_sub(d, v3_scale(n, 2.0f * v3_dot(d, n)));          /* :472 */
    *out_t = t;
    store3(out_hit, hit);
    store3(out_normal, n);
    store3(out_reflected, r);
    return p;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static void counters_add(orc_counters *a, const orc_counters *b)
{
    a->rays_primary += b->rays_primary;
    a->rays_shadow += b->rays_shadow;
    a->rays_secondary += b->rays_secondary;
    a->rays_secondary_dead += b->rays_secondary_dead;
    a->shaded_hits += b->shaded_hits;
    a->sphere_tests += b->sphere_tests;
    a->plane_tests += b->plane_tests;
    a->triangle_tests += b->triangle_tests;
}

/* Raytracer.Render, MyApplication.cs:402-426; rows go to threads like the outer Parallel.For. */
int orc_render(const rt_scene *s, const orc_texture *tex, int ntex, const rt_camera *cam, int w,
               int h, int row_begin, int row_end, int row_step, int max_depth, int mode,
               int nthreads, int32_t *out, orc_counters *counters)
{
    if (!s || !cam || !out || w <= 0 || h <= 0 || row_begin < 0 || row_end > h ||
        row_begin > row_end || row_step <= 0 || max_depth < 0)
        return -1;
    orc_counters total;
    memset(&total, 0, sizeof total);
    int nrows = (row_end - row_begin + row_step - 1) / row_step;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif

#pragma omp parallel num_threads(nthreads)
    {
        ctx_t cx;
        memset(&cx, 0, sizeof cx);
        cx.s = s;
        cx.tex = tex;
        cx.ntex = ntex;
        cx.max_depth = max_depth;
        cx.mode = mode;
#pragma omp for schedule(dynamic, 1)
        for (int r = 0; r < nrows; r++) {
            int y = row_begin + r * row_step;
            for (int x = 0; x < w; x++) {
                v3 origin = v3_load(cam->position);
                v3 direction = ray_direction(cam, x, y, w, h);
                float t;
                v3 hit;
                cx.c.rays_primary++;
                int prim = scene_intersect(&cx, s, origin, direction, &t, &hit);
                if (prim >= 0)
                    out[(size_t)y * (size_t)w + (size_t)x] =
                        mix_color(shade(&cx, prim, hit, direction, 0));
                else
                    out[(size_t)y * (size_t)w + (size_t)x] = 0;
            }
        }
#pragma omp critical
        counters_add(&total, &cx.c);
    }
    if (counters) *counters = total;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * FXAA, screen_fxaa_fs.glsl:80-140 (see rt_oracle.h).
 * The routine restated here is Copyright (c) 2011 by Armin Ronacher (webgl-meincraft), some rights
 * reserved, three-clause BSD licence; notice, conditions and disclaimer: THIRD_PARTY_NOTICES.md at
 * the root of this repository.
 * ---------------------------------------------------------------------------------------- */
typedef struct { float r, g, b; } fx_rgb;

static int fx_wrap(int i, int n)   /* GL_REPEAT */
{
    i %= n;
    return i < 0 ? i + n : i;
}

static fx_rgb fx_texel(const int32_t *img, int w, int x, int y)
{
    const uint32_t p = (uint32_t)img[(size_t)y * (size_t)w + (size_t)x];
    fx_rgb c;
    c.r = (float)((p >> 16) & 255u) / 255.0f;
    c.g = (float)((p >> 8) & 255u) / 255.0f;
    c.b = (float)(p & 255u) / 255.0f;
    return c;
}

static float fx_mix(float a, float b, float t) { return a + (b - a) * t; }

/* texture2D(tex, (u, v)) with GL_LINEAR / GL_REPEAT. */
static fx_rgb fx_sample(const int32_t *img, int w, int h, float u, float v)
{
    const float xs = u * (float)w - 0.5f, ys = v * (float)h - 0.5f;
    const float xf = floorf(xs), yf = floorf(ys);
    const float fx = xs - xf, fy = ys - yf;
    const int x0 = fx_wrap((int)xf, w), x1 = fx_wrap((int)xf + 1, w);
    const int y0 = fx_wrap((int)yf, h), y1 = fx_wrap((int)yf + 1, h);
    const fx_rgb t00 = fx_texel(img, w, x0, y0), t10 = fx_texel(img, w, x1, y0);
    const fx_rgb t01 = fx_texel(img, w, x0, y1), t11 = fx_texel(img, w, x1, y1);
    fx_rgb c;
    c.r = fx_mix(fx_mix(t00.r, t10.r, fx), fx_mix(t01.r, t11.r, fx), fy);
    c.g = fx_mix(fx_mix(t00.g, t10.g, fx), fx_mix(t01.g, t11.g, fx), fy);
    c.b = fx_mix(fx_mix(t00.b, t10.b, fx), fx_mix(t01.b, t11.b, fx), fy);
    return c;
}

static float fx_luma(fx_rgb c) { return (c.r * 0.299f + c.g * 0.587f) + c.b * 0.114f; }   /* dot(rgb, luma) */

static uint32_t fx_byte(float c)
{
    if (!(c > 0.0f)) return 0u;
    if (c > 1.0f) c = 1.0f;
    return (uint32_t)(c * 255.0f + 0.5f);
}

void orc_fxaa(const int32_t *in, int w, int h, int32_t *out)
{
    const float ivx = 1.0f / (float)w, ivy = 1.0f / (float)h;                    /* :94 */
    const float reduce_min = 1.0f / 128.0f, reduce_mul = 1.0f / 8.0f, span_max = 8.0f;
    const float ta = (float)(1.0 / 3.0 - 0.5), tb = (float)(2.0 / 3.0 - 0.5);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            const float fcx = (float)x + 0.5f, fcy = (float)y + 0.5f;           /* uv * resolution, :428 */
            const fx_rgb nw = fx_sample(in, w, h, (fcx - 1.0f) * ivx, (fcy - 1.0f) * ivy);   /* :95-103 */
            const fx_rgb ne = fx_sample(in, w, h, (fcx + 1.0f) * ivx, (fcy - 1.0f) * ivy);
            const fx_rgb sw = fx_sample(in, w, h, (fcx - 1.0f) * ivx, (fcy + 1.0f) * ivy);
            const fx_rgb se = fx_sample(in, w, h, (fcx + 1.0f) * ivx, (fcy + 1.0f) * ivy);
            const float mu = fcx * ivx, mv = fcy * ivy;
            const fx_rgb m = fx_sample(in, w, h, mu, mv);                        /* :104 */
            const float l_nw = fx_luma(nw), l_ne = fx_luma(ne), l_sw = fx_luma(sw), l_se = fx_luma(se),
                        l_m = fx_luma(m);                                        /* :106-111 */
            const float l_min = fminf(l_m, fminf(fminf(l_nw, l_ne), fminf(l_sw, l_se)));   /* :112-113 */
            const float l_max = fmaxf(l_m, fmaxf(fmaxf(l_nw, l_ne), fmaxf(l_sw, l_se)));
            float dx = -((l_nw + l_ne) - (l_sw + l_se));                         /* :116-117 */
            float dy = (l_nw + l_sw) - (l_ne + l_se);
            const float dir_reduce = fmaxf((((l_nw + l_ne) + l_sw) + l_se) * (0.25f * reduce_mul), reduce_min);
            const float rcp = 1.0f / (fminf(fabsf(dx), fabsf(dy)) + dir_reduce);  /* :122 */
            dx = fminf(span_max, fmaxf(-span_max, dx * rcp)) * ivx;              /* :123-125 */
            dy = fminf(span_max, fmaxf(-span_max, dy * rcp)) * ivy;
            const fx_rgb a1 = fx_sample(in, w, h, mu + dx * ta, mv + dy * ta);   /* :127-129 */
            const fx_rgb a2 = fx_sample(in, w, h, mu + dx * tb, mv + dy * tb);
            const fx_rgb b1 = fx_sample(in, w, h, mu + dx * -0.5f, mv + dy * -0.5f);   /* :130-132 */
            const fx_rgb b2 = fx_sample(in, w, h, mu + dx * 0.5f, mv + dy * 0.5f);
            fx_rgb ca, cb;
            ca.r = 0.5f * (a1.r + a2.r), ca.g = 0.5f * (a1.g + a2.g), ca.b = 0.5f * (a1.b + a2.b);
            cb.r = ca.r * 0.5f + 0.25f * (b1.r + b2.r);
            cb.g = ca.g * 0.5f + 0.25f * (b1.g + b2.g);
            cb.b = ca.b * 0.5f + 0.25f * (b1.b + b2.b);
            const float l_b = fx_luma(cb);                                       /* :134-138 */
            const fx_rgb c = (l_b < l_min || l_b > l_max) ? ca : cb;
            /* alpha: texColor.a, the 0 byte of the 0x00RRGGBB texel */
            out[(size_t)y * (size_t)w + (size_t)x] = (int32_t)((fx_byte(c.r) << 16) | (fx_byte(c.g) << 8) | fx_byte(c.b));
        }
    }
}
