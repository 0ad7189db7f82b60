/*
 * rt_hip.h — C ABI of the MI355X (gfx950) render-loop library.
 *
 * The library replaces the body of the host's per-pixel CPU loop
 *     Raytracer.Render()                        /root/reference/MyApplication.cs:402-426
 * and everything that loop calls (GetRayDirection :428-452, Scene.Intersect /
 * IsInShadow :352-380, Primitive.Shade objects.cs:44-96, the Sphere / Plane /
 * Triangle Intersect routines objects.cs:109-273, checkerboard and texture lookup
 * objects.cs:176-213,348-353, MixColor objects.cs:35-41).  The host keeps its
 * window, camera controls, scene construction and present path; it hands the
 * library a flattened scene once and a camera per frame, and receives the
 * 0x00RRGGBB framebuffer that Surface.pixels (surface.cs:9-19) holds.
 *
 * Conventions
 *   - blittable types only: plain pointers, 32-bit ints and floats, no C++ types;
 *   - every call returns RT_OK (0) or a negative rt_status; no exception and no
 *     abort crosses the boundary; rt_last_error() returns the message the host
 *     would put into `throw new Exception(msg)` (the reference's only error
 *     convention, template.cs:102-106);
 *   - the caller owns every host pointer it passes; the library copies inside the
 *     call and never retains the pointer;
 *   - one caller thread per context (the reference renders from the single window
 *     thread, template.cs:185-204); contexts are independent of each other.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_API_VERSION 7

/* Primitive type codes — the reference's own flatten uses exactly these
 * (MyApplication.cs:161,166,173). */
#define RT_PRIM_SPHERE   0
#define RT_PRIM_PLANE    1
#define RT_PRIM_TRIANGLE 2

/* The reference hard-codes `depth < 5` (objects.cs:84). */
#define RT_REFERENCE_MAX_DEPTH 5
/* Upper bound the library accepts for max_depth (levels 0..RT_MAX_DEPTH). */
#define RT_MAX_DEPTH 16
#define RT_MAX_TEXTURES 8

typedef enum rt_status {
    RT_OK                 =  0,
    RT_ERR_INVALID_ARG    = -1,  /* NULL pointer, bad size, bad type code ...            */
    RT_ERR_NO_DEVICE      = -2,  /* no usable gfx950 device / bad device id             */
    RT_ERR_OUT_OF_MEMORY  = -3,  /* host or device allocation failed                    */
    RT_ERR_HIP            = -4,  /* a HIP runtime call failed (message has the detail)  */
    RT_ERR_NO_SCENE       = -5,  /* rt_render before rt_set_scene                        */
    RT_ERR_BAD_TEXTURE    = -6,  /* a primitive names a texture slot that is not loaded */
    RT_ERR_LIMIT          = -7,  /* request exceeds a fixed limit of an entry point     */
    RT_ERR_STALLED        = -8   /* the device gave up on a frame: its pixels are incomplete */
} rt_status;

/* Camera fields read by the render loop (MyApplication.cs:259-275: position,
 * direction, up, fov in degrees).  Three consecutive floats = OpenTK Vector3. */
typedef struct rt_camera {
    float position[3];
    float direction[3];
    float up[3];
    float fov_deg;
} rt_camera;

/*
 * Flattened scene, structure of arrays.  Field-for-field the arrays the reference
 * builds in MyApplication.Init (MyApplication.cs:142-176), with two changes: the
 * capacity is dynamic (the reference fixes it at 10, :48-62) and a per-primitive
 * texture slot is added (the reference keeps the texture on the Plane object,
 * objects.cs:146, and never flattens it).
 *
 * Primitive i is described by the i-th element of every prim_* array; order is the
 * host's insertion order and decides ties in closest-hit (lowest index wins,
 * MyApplication.cs:361).  Arrays that do not apply to a primitive's type are
 * ignored for that primitive but must still be addressable.
 */
typedef struct rt_scene {
    int32_t num_prims;
    int32_t num_lights;
    const int32_t *prim_types;        /* [n]   RT_PRIM_*                                        */
    const float   *prim_positions;    /* [3n]  sphere centre                                    */
    const float   *prim_normals;      /* [3n]  plane / triangle unit normal as the host's
                                               constructors computed it (objects.cs:150,233)    */
    const float   *prim_diffuse;      /* [4n]  Material.diffuseColor RGBA                       */
    const float   *prim_glossy;       /* [4n]  Material.glossyColor                             */
    const float   *prim_ambient;      /* [4n]  Material.ambientColor                            */
    const float   *prim_specular;     /* [n]   Material.specular                                */
    const float   *prim_radius;       /* [n]   Sphere.radius                                    */
    const float   *prim_d;            /* [n]   Plane.d (plane: n.p + d = 0)                     */
    const float   *prim_v0;           /* [3n]  Triangle vertices                                */
    const float   *prim_v1;           /* [3n]                                                   */
    const float   *prim_v2;           /* [3n]                                                   */
    const int32_t *prim_checkerboard; /* [n]   Material.useCheckerboard as 0/1                  */
    const int32_t *prim_texture;      /* [n]   texture slot (planes only) or -1; NULL = all -1  */
    const float   *light_positions;   /* [3L]  Light.position                                   */
    const float   *light_colors;      /* [4L]  Light.color (reference: 2,2,2,1)                 */
} rt_scene;

/* Ray / work accounting of the last rt_render* call on the context.
 *   primary    one per pixel                                   (MyApplication.cs:415)
 *   shadow     one per shaded hit and light                    (objects.cs:56)
 *   secondary  one per shaded hit with specular != 0 at a level below max_depth —
 *              the reflection trace whose result is used, counted once per level
 *              (objects.cs:83; the reference repeats it once per light)
 * sphere/plane/triangle_tests count the Intersect calls the reference's algorithm makes
 * (an occluded light ray stops at its first occluder in primitive order,
 * MyApplication.cs:373-378), whatever the kernels actually evaluate; like the times they are
 * collected only by calls made in profiling mode 1 (otherwise the traversal kernels keep no
 * work counters — the fields then hold the primary rays' share only — and a light ray stops at
 * whichever occluder it meets first, which changes no pixel but leaves "first in primitive
 * order" unknown).  The ray counts are always exact.
 * Times are milliseconds of the last call, measured with HIP events on the stream the
 * kernels ran on and only in profiling modes 1 and 2 (rt_set_profiling): kernel_ms spans
 * the whole frame pipeline; primary_ms, trace_ms and shade_ms split it into the primary
 * kernel, the per-level ray traversal kernels and the per-level shading kernels.  With
 * several devices each is the maximum over the devices. */
typedef struct rt_stats {
    uint64_t rays_primary;
    uint64_t rays_shadow;
    uint64_t rays_secondary;
    uint64_t shaded_hits;
    uint64_t sphere_tests;
    uint64_t plane_tests;
    uint64_t triangle_tests;
    uint64_t exact_candidates; /* (ray origin, sphere) pairs the conservative sphere filter
                                  passed on to the exact Intersect code: a cost figure      */
    uint64_t filter_ray_tests; /* (ray, sphere or bounding ball) pairs the filter evaluated:
                                  what the kernels executed in place of sphere_tests; groups
                                  of spheres whose bounding ball no ray of a wave can meet
                                  are skipped; for scenes of more than 1024 spheres the cells
                                  the reflection rays' walks through the spheres' uniform grid
                                  visit count here too, and the spheres those walks test as
                                  exact_candidates                                          */
    uint64_t filter_origin_terms; /* (ray origin, sphere or ball) pairs for which the filter's
                                  per-origin term was formed (secondary and shadow rays)    */
    uint64_t primary_filter_ray_tests; /* the primary rays' share of filter_ray_tests ...   */
    uint64_t primary_exact_candidates; /* ... and of exact_candidates                        */
    uint64_t shadow_candidates; /* (light ray, sphere) pairs evaluated with the exact Intersect code
                                  out of the per-light direction tables: what light rays cost
                                  in place of their sphere_tests                            */
    double   kernel_ms;
    double   primary_ms;
    double   trace_ms;
    double   shade_ms;
    double   gather_ms;       /* rt_render with several devices: blocks -> first device */
    double   copy_ms;         /* rt_render: device -> pixels_out; when the copy overlapped
                                 the kernels (see rt_render), the time of the whole call */
    int32_t  num_devices;
    int32_t  reserved;
} rt_stats;

/* One ray of the host's top-down debug view (Raytracer.Debug, MyApplication.cs:454-486):
 * what its loop body computes before it draws. */
typedef struct rt_debug_ray {
    float   direction[3];   /* GetRayDirection(x, row, w, h)                     (:461)      */
    int32_t prim;           /* scene.Intersect: host primitive index, -1 = miss  (:464)      */
    float   hit[3];         /* isect.hitPoint                                    (:468-470)  */
    float   t;              /* isect.distance                                                */
    float   normal[3];      /* isect.prim.GetNormal(hitPoint)                    (:471)      */
    float   reflected[3];   /* direction - 2 * Dot(direction, normal) * normal   (:472)      */
} rt_debug_ray;

typedef struct rt_ctx rt_ctx;

/* Library / ABI version (RT_API_VERSION of the build). */
int rt_api_version(void);

/* Number of HIP devices visible to the process, or a negative rt_status. */
int rt_device_count(void);

/*
 * Create a context on the given devices (device_ids == NULL or n_devices <= 0:
 * device 0 only).  With more than one device, rt_render() splits the frame by rows —
 * the image-space partition the reference already uses for its thread pool
 * (Parallel.For over rows, MyApplication.cs:408): device g of G takes the bands g,
 * g + G, ... of 16 rows (contiguous blocks when the frame has fewer than 2 G bands),
 * and every device copies its bands into the host's frame itself, over its own PCIe
 * link, with one strided copy (RT_HIP_HOST_GATHER=0: into the first device's frame over
 * xGMI, from where one copy brings the frame to the host).
 * The tuning aids and test knobs of the library (environment variables RT_HIP_*) are read
 * here, by rt_set_scene and by rt_reload_tuning, never on the way of a frame.
 * Replaces: the Raytracer constructor (MyApplication.cs:392-400).
 */
int rt_create(rt_ctx **out, const int *device_ids, int n_devices);

/* Free every device and host resource of the context.  NULL is ignored. */
void rt_destroy(rt_ctx *ctx);

/* Message of the last failing call on ctx; with ctx == NULL, of the last failing
 * rt_create on this thread.  Never NULL; valid until the next call on ctx. */
const char *rt_last_error(const rt_ctx *ctx);

/*
 * Copy the flattened scene to every device of the context.
 * Replaces: Scene.primitives / Scene.lights as read by the loop
 * (MyApplication.cs:341-350) — call once after the host's Init() (template.cs:155)
 * and again whenever the host edits the scene.
 */
int rt_set_scene(rt_ctx *ctx, const rt_scene *scene);

/*
 * Load an RGBA8 image (row-major, row 0 first, 4 bytes per texel in R,G,B,A order)
 * into a texture slot.  Replaces: Texture.LoadTexture (objects.cs:331-346), which
 * stores Bitmap.GetPixel(x,y) as R/255f.. in pixels[x,y]; lookups are point
 * samples with the address arithmetic of Plane.GetTextureCoordinates and
 * Texture.GetColor (objects.cs:190-204,348-353).
 */
int rt_set_texture(rt_ctx *ctx, int slot, const uint8_t *rgba8, int width, int height);

/*
 * Render one frame of width x height pixels into pixels_out (host memory,
 * width*height int32, row-major, row 0 = top of the window, value 0x00RRGGBB,
 * 0 where the primary ray hits nothing).  Blocking: every element is written
 * before the call returns.  max_depth bounds the mirror recursion exactly like the
 * literal 5 in `depth < 5` (objects.cs:84); pass RT_REFERENCE_MAX_DEPTH for the
 * reference's behaviour.
 * The copy to pixels_out runs in chunks through page-locked staging memory, the DMA of one
 * chunk overlapping the host-side move of the previous one (or straight into pixels_out: rt_pin_pixels,
 * rt_alloc_pixels); on a context with one device and profiling disabled, a frame of 3 M pixels or more is
 * rendered as up to eight row blocks whose pipelines overlap (a block starts while the one before it works
 * through its small deep levels, in a second set of streams and ray queues: twice the device memory
 * rt_render_rows_device documents for a block), each block copied while the later ones render (same pixels,
 * same rt_get_stats counts).
 * Replaces: Raytracer.Render() (MyApplication.cs:402-426).
 */
int rt_render(rt_ctx *ctx, const rt_camera *camera, int width, int height,
              int max_depth, int32_t *pixels_out);

/*
 * Page-lock the host's framebuffer so that rt_render's DMA writes pixels_out itself instead of
 * internal staging memory that the caller's thread then has to move on (at 3840x2160 that move
 * runs at about the speed of the DMA and mostly hides behind it).  `pixels` .. `pixels + count` must stay allocated
 * and at the same address until rt_unpin_pixels or rt_destroy: in the reference that is
 * Surface.pixels, allocated once in the constructor and never reallocated (surface.cs:14-19);
 * a C# host pins the array with GCHandle.Alloc(pixels, GCHandleType.Pinned) and passes
 * AddrOfPinnedObject().  rt_render uses the direct path whenever pixels_out .. pixels_out +
 * width*height lies inside the registered range, the staged path otherwise.  One range per
 * context; registering another replaces it.  Where the system refuses to page-lock user
 * memory the call fails with RT_ERR_HIP and changes nothing: rt_render keeps working through
 * its staged copy.  No counterpart in the reference (the CPU loop
 * writes the array directly, MyApplication.cs:420).
 */
int rt_pin_pixels(rt_ctx *ctx, int32_t *pixels, int64_t count);
int rt_unpin_pixels(rt_ctx *ctx);

/*
 * The same hand-off for hosts that can present from a native pointer: a framebuffer of `count` pixels in
 * page-locked memory the LIBRARY allocates (works where the system refuses to page-lock user memory).
 * rt_render into *pixels_out (or any sub-range of it) writes it with the DMA itself, no staging, no move
 * by the caller's thread.  The buffer replaces whatever range rt_pin_pixels registered, and lives until
 * rt_unpin_pixels, the next rt_alloc_pixels / rt_pin_pixels, or rt_destroy.  In the reference the host
 * would hand this pointer to GL.TexImage2D in place of screen.pixels (template.cs:199) and let Surface's
 * drawing helpers write through it.  No counterpart in the reference.
 */
int rt_alloc_pixels(rt_ctx *ctx, int64_t count, int32_t **pixels_out);

/*
 * Render rows [row_begin, row_end) of a width x height frame on the context's
 * first device into device memory: dev_pixels_out points to (row_end-row_begin) *
 * width int32 on that device; element (y - row_begin) * width + x is pixel (x, y)
 * of the full frame.  The work is enqueued on hip_stream (a hipStream_t; NULL =
 * a non-blocking stream the context owns, which nothing else is ordered against: pass the
 * stream the consumer of the pixels runs on) and the call returns without synchronising; per-frame
 * scratch is owned by the context, so calls on one context must not overlap.
 * Device memory: every render entry keeps the frame's ray queues in the context, sized for the
 * worst case (every pixel alive at every level) and grown, never shrunk, when a larger frame
 * comes: 48 bytes * pixels of the call (rounded up to 64) * (max_depth + 1), i.e. 2.0 GB for
 * 3840x2160 at max_depth 4 and 14.3 GB for 7680x4320 at max_depth 8 (scenes of more than 1024 spheres or at
 * least eight lights: one level more; scenes of at most four lights: 16 bytes * pixels * max_depth * lights
 * more, for light terms computed ahead; more lights: 16 bytes * 65536 * max_depth * lights); RT_ERR_OUT_OF_MEMORY when
 * the device cannot give that.  A rank of a row partition allocates for its own rows only.
 * This is the per-rank entry of the one-process-per-GPU row partition: rank r of
 * G renders its block and the framebuffer blocks are gathered with one
 * collective.  It is also Render() for hosts that present straight from device
 * memory.  Same semantics per pixel as rt_render (MyApplication.cs:410-424).
 */
int rt_render_rows_device(rt_ctx *ctx, const rt_camera *camera, int width, int height,
                          int row_begin, int row_end, int max_depth,
                          int32_t *dev_pixels_out, void *hip_stream);

/*
 * rt_render_rows_device for callers that cannot allocate (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc): same arguments, same pixels, but the frame's
 * ray queues live in a fixed array of RT_NOALLOC_SCRATCH_BYTES that the library holds in every
 * GPU's memory from load to process exit, and the call makes no allocation of any kind.  The
 * frame must fit: with P = (row_end - row_begin) * width rounded up to a multiple of 64,
 *     P * (max_depth + 1) * 48 + RT_NOALLOC_CONTROL_BYTES  <=  RT_NOALLOC_SCRATCH_BYTES
 * (e.g. 370 x 370 pixels at max_depth 4, or 1024 x 110 at max_depth 5; every frame this admits is accepted),
 * else RT_ERR_LIMIT and nothing is enqueued (rt_get_stats keeps describing the frame before); render a taller frame as several calls of fewer
 * rows.  ALL contexts of a process share a GPU's array: calls that overlap on one GPU (two
 * contexts, or two streams not ordered against each other) race on it; calls ordered one after
 * the other are independent, each clears what it needs.  The context must have been used once
 * before the no-allocation phase begins (rt_set_scene, and one render call of either kind so
 * that the runtime has loaded the kernels).
 * No counterpart in the reference (its loop allocates freely, objects.cs:125); exists for
 * callers that cannot allocate.  rt_get_stats after such a call reads the frame's counts out of the shared
 * array: ask before any other context renders into it.
 */
#define RT_NOALLOC_SCRATCH_BYTES (32u << 20)
/* What the frame's control block (work counters, level schedule, record counts) and the alignment of the three
 * record arrays take of the array at most. */
#define RT_NOALLOC_CONTROL_BYTES (304u << 10)
int rt_render_rows_device_noalloc(rt_ctx *ctx, const rt_camera *camera, int width, int height,
                                  int row_begin, int row_end, int max_depth,
                                  int32_t *dev_pixels_out, void *hip_stream);

/*
 * rt_render_rows_device for an interleaved partition of the frame: the frame is cut into bands of band_rows
 * consecutive rows (band b = rows [b * band_rows, (b + 1) * band_rows), the last one shorter)
 * and this call renders the bands b = part, part + parts, part + 2 * parts, ...  The bands are
 * written one after the other: dev_pixels_out points to rt_band_rows(height, band_rows, part,
 * parts) * width int32 on the context's first device.  With parts ranks each taking one part
 * every rank gets the same mix of cheap and expensive image regions (rows cost very different
 * amounts: mirrors, the reflecting ceiling), which contiguous blocks do not give; the host
 * interleaves the gathered blocks back.  parts = 1 renders the whole frame.
 * No counterpart in the reference (its thread pool balances rows dynamically,
 * MyApplication.cs:408); per pixel the semantics of rt_render (MyApplication.cs:410-424).
 */
int rt_render_bands_device(rt_ctx *ctx, const rt_camera *camera, int width, int height,
                           int band_rows, int part, int parts, int max_depth,
                           int32_t *dev_pixels_out, void *hip_stream);

/*
 * Tell the context that `contexts` contexts (1..8, this one included) render on its devices AT
 * THE SAME TIME, each on a stream of its own: later render calls size their resident launches
 * for 1/contexts of the device, so that the frames of the contexts run side by side instead of
 * one after the other.  What it is for: small frames (a rank's share of a frame split over
 * several GPUs) are bound by their chain of dependent reflection levels, during which most of
 * the device idles; two or three such frames in flight, each in a context of its own, fill it.
 * Default 1.  No counterpart in the reference (one frame at a time, MyApplication.cs:179-193).
 */
int rt_set_device_share(rt_ctx *ctx, int contexts);

/* Rows of a height-row frame that fall to `part` of `parts` with bands of band_rows rows, or
 * RT_ERR_INVALID_ARG. */
int rt_band_rows(int height, int band_rows, int part, int parts);

/*
 * Outcome of the frames enqueued with rt_render_rows_device / rt_render_bands_device since the
 * last call: waits for the stream the last of them was enqueued on (which must still exist; the
 * caller ordered the earlier ones before it, as calls on one context must not overlap) and
 * returns RT_OK, or RT_ERR_STALLED when the device-side schedule of some frame gave up — a
 * workgroup waited longer than the stall limit (one second; environment variable
 * RT_HIP_STALL_MS, see rt_reload_tuning) for the next level of reflection rays to open — in which case that frame's
 * pixels are incomplete.  The report is consumed by the call.  rt_render makes the same check
 * itself before it returns.  Costs nothing on the device: a stalled workgroup writes the
 * report into page-locked host memory, nobody else writes anything.
 * No counterpart in the reference (its loop cannot lose work; this one runs on a device-side
 * schedule and returns before the frame is done).
 */
int rt_frame_status(rt_ctx *ctx);

/*
 * Optional post-pass: the anti-aliasing filter the host's present path applies to the frame
 * AFTER the hand-off (fxaa() in shaders/screen_fxaa_fs.glsl:80-140, called from main()
 * :426-431 on the uploaded texture), over a width x height frame of 0x00RRGGBB ints in
 * device memory on the context's first device; input and output must not overlap.  For
 * hosts that present from device memory or write frames to disk without a GL context; the
 * unchanged C# host keeps its shader and does not call this.  The sampler is the one
 * surface.cs:36-38 configures (LINEAR, REPEAT), evaluated in FP32; a GL implementation's own
 * filter-weight precision makes its output differ from this in the last bits.  Enqueued on
 * hip_stream (NULL: the context's stream) without synchronising.
 */
int rt_fxaa_device(rt_ctx *ctx, const int32_t *dev_pixels_in, int32_t *dev_pixels_out,
                   int width, int height, void *hip_stream);

/*
 * The rays of the host's debug view: for x = 0, x_step, 2 x_step, ... < width the primary ray
 * of pixel (x, row) of a width x height frame, its closest hit, the surface normal there and
 * the mirror direction, written to out[0 .. count) in host memory (count = ceil(width /
 * x_step) <= max_rays, returned through out_count).  Blocking.  The host keeps drawing the
 * lines itself (Surface.Line); this replaces the GetRayDirection / scene.Intersect /
 * GetNormal work of Raytracer.Debug's loop (MyApplication.cs:459-478; the reference uses
 * row 320 and x_step 10).
 */
int rt_debug_rays(rt_ctx *ctx, const rt_camera *camera, int width, int height, int row, int x_step,
                  int max_rays, rt_debug_ray *out, int *out_count);

/* Counters and timings of the last render call.  Synchronises the context's
 * devices (needed to read the device-side counters). */
int rt_get_stats(rt_ctx *ctx, rt_stats *out);

/* Reads the library's tuning aids and test knobs (environment variables RT_HIP_*: INTEGRATION.md, "Environment variables") again;
 * rt_create and rt_set_scene do the same.  No frame reads the environment.  No counterpart in the reference. */
int rt_reload_tuning(rt_ctx *ctx);

/* Profiling mode of later render calls on the context: 0 (default) none; 1 the exact
 * Intersect-call and work counters and event timing (the counting instances of the kernels run:
 * ~2 % slower, and a light ray looks at the whole scene); 2 event timing only (the kernels of
 * mode 0 run, bracketed by events; rt_render does not overlap its copy with rendering). */
int rt_set_profiling(rt_ctx *ctx, int enable);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
