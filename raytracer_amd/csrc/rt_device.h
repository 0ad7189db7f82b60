// rt_device.h — data layout shared by the host side (rt_api.cpp) and the kernels
// (rt_kernels.hip) of librt_hip.so.  gfx950 only.
//
// Scene in HBM (structure of arrays, one array per primitive class; *_orig maps back to
// the host's primitive index, which is what decides ties in closest-hit, MyApplication.cs:361,
// and which occluder a shadow ray meets first, :373-378).  Planes and triangles keep the
// host's order.  Spheres are stored in a spatial order: every aligned run of RT_GROUP spheres
// is a compact group with a bounding ball, every aligned run of 1024 a compact chunk, so
// that a wave whose rays all miss a ball skips the group (rt_kernels.hip, "Sphere filter").
//   sph      float4 {cx, cy, cz, r*r}        one per sphere, what the exact evaluation reads
//   sph_f    float4 {p'x, p'y, p'z, W}       the sphere filter's view of the same spheres:
//                                            p' = centre - sph_center, W = r*r - k1*|p'|^2
//   grp_f    float4 {P'x, P'y, P'z, W}       one bounding ball per group, same form
//   sup_f, chk_f                             one bounding ball per RT_SUPER (64) spheres and per
//                                            RT_CHUNK (1024) spheres: two coarser levels, same form
//   sup_b, chk_b                             the balls around the same runs and chunks as {P, R} in the
//                                            scene's own coordinates (the primary rays' cone test)
//   pl       float4 {nx, ny, nz, d}          one per plane
//   tri      float4 x3 {v0}, {e1}, {e2}      e1 = v1-v0, e2 = v2-v0 (objects.cs:238-239)
//   geo      float4 per host primitive: sphere centre or plane/triangle normal
//   mat      float4 x3 per host primitive: {diffuse.rgb, specular},
//            {glossy.rgb, flags}, {ambient.rgb, -}
//   light_pos / light_col  float4 per light
// The filter arrays are staged in LDS by each workgroup and read from there into VGPRs
// (one ds_read_b128 per sphere per wave, all lanes the same address); the small arrays
// (planes, triangles, lights) are wave-uniform and read through the scalar cache.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_DEV_MAX_TEXTURES 8

// mat[3*i+1].w holds these flags (bit pattern of an int stored in a float slot).
#define RT_MF_TYPE_MASK   0x3      // RT_PRIM_*
#define RT_MF_CHECKER     0x4      // plane with Material.useCheckerboard
#define RT_MF_TEX_SHIFT   4        // texture slot + 1 in bits 4..7 (0 = none)
#define RT_MF_TEX_MASK    0xF0

// sph, sph_f and sph_orig are padded to a multiple of RT_SPH_PAD entries (a whole
// number of trips of bounding balls); the padding is {0, 0, 0, -inf}, which no ray passes in
// the sphere filter, and so are the balls of groups made of padding only.
#define RT_GROUP 8
#define RT_SPH_PAD (RT_GROUP * 8)
#define RT_SUPER RT_SPH_PAD   // spheres per ball of the second level = groups of one GROUP-loop trip
#define RT_CHUNK 1024         // spheres staged in LDS at a time; one ball of the third level each
// k1 = 1 - 2^-18, the filter's relative margin (shared by the host's W and the kernels).
#define RT_FILTER_K1 0.999996185302734375f

#define RT_LT_TIERS 4   // at most so many tiers of the lights' tables (DevScene::lt_tiers)

struct DevScene {
    int ns, np, nt, nl, nprims;
    int ns_padded;
    float sph_center[3];      // the filter's coordinate origin (centre of the spheres' bounding box)
    float sph_l1max;          // >= max over spheres of |p'|_1
    float sph_sc;             // >= 2 max |p'|^2 + max r*r   (p', r over spheres and balls)
    float grp_rmax;           // >= the largest bounding-ball radius of a group,
    float sup_rmax;           //    of a run of RT_SUPER spheres,
    float chk_rmax;           //    of a chunk
    int nsup_padded;          // entries of sup_f: ns_padded / RT_SUPER rounded up to a multiple of 8
    int nchk_padded;          // entries of chk_f: chunks rounded up to a multiple of 8
    const float4 *sph;
    const float4 *sph_f;
    const float4 *grp_f;      // ns_padded / RT_GROUP bounding balls
    const float4 *sup_f;      // nsup_padded balls (padding: never-pass entries)
    const float4 *chk_f;      // nchk_padded balls
    // Balls around the same runs as {Px, Py, Pz, R} in the scene's own coordinates (R rounded up;
    // -1 = no members, +inf = not representable): what the primary rays' cone test reads.
    const float4 *sup_b;
    const float4 *chk_b;
    const int *sph_orig;
    const float4 *pl;
    const int *pl_orig;
    const float4 *tri;
    const int *tri_orig;
    const float4 *geo;
    const float4 *mat;
    const float4 *light_pos;
    const float4 *light_col;
    // class_prefix[3*i + c] = number of primitives of class c among host primitives
    // [0, i): lets a kernel turn "first occluder index" into per-class test counts.
    const int *class_prefix;
    // Per-light direction tables (rt_api.cpp, build_light_tables): for light l and the cube-map bin
    // of the direction from the light to a ray's origin, lt_cells[(l * 6 + face) * lt_n * lt_n + v * lt_n
    // + u] = {first entry, entries} of the spheres that can occlude such a ray: lt_sph {centre, r*r}
    // and lt_orig (host primitive index) per entry.  Valid for origins with |o - sph_center|_1 <=
    // lt_near; lt_n = 0: no tables (no spheres or no lights).
    // The tables come in lt_tiers tiers by the ray's length: tier t (all lights' tables of tier t follow those
    // of tier t - 1 in lt_cells) holds what can occlude a ray whose squared length |origin - light|^2, as the
    // kernel forms it, is at most lt_tier_q[t]; the last tier has no such bound.  The reference's rounding
    // errors grow with the distance between a ray's origin and a sphere, and so does the margin a sphere is
    // listed with: short rays get short lists.
    const uint2 *lt_cells;
    const float4 *lt_sph;
    const int *lt_orig;
    const float *lt_reach;    // per entry: a lower bound of the distance from the light to any point of the
                              // (widened) sphere; a cell's entries are sorted by it
    int lt_n;
    float lt_near;
    int lt_tiers;
    float lt_tier_q[RT_LT_TIERS - 1];
    // Uniform grid over the spheres (rt_api.cpp, build_sphere_grid; rt_kernels.hip, "Grid walk"): cell
    // (x, y, z) = gr_cells[(z * gr_n[1] + y) * gr_n[0] + x] = {first entry, entries} of the spheres whose
    // widened ball touches the cell: gr_sph {centre, r*r} and gr_orig (host primitive index) per entry.  The
    // lists hold every sphere the reference can accept for a ray whose ORIGIN lies in the box gr_reg_lo ..
    // gr_reg_hi; other origins evaluate every sphere.  gr_n[0] = 0: no grid.
    // gr_tier_t >= 0: a second set of lists follows the first in gr_cells (cell + gr_n[0] * gr_n[1] * gr_n[2]),
    // shorter ones, good for the cells a ray leaves at a parameter of at most gr_tier_t.
    const uint2 *gr_cells;
    const float4 *gr_sph;
    const int *gr_orig;
    float gr_tier_t;
    int gr_n[3];
    float gr_lo[3];           // the grid's lower corner
    float gr_ih[3];           // cells per unit of length
    float gr_reg_lo[3], gr_reg_hi[3];
    // Position bins (k_bin_*): cell of a point = floor((p - bin_lo) * bin_scale) per axis, clamped to
    // 0 .. RT_BIN_AXIS - 1; the box is the spheres' extent.  bin_scale[0] = 0: no box.
    float bin_lo[3], bin_scale[3];
    const uchar4 *tex[RT_DEV_MAX_TEXTURES];
    int tex_w[RT_DEV_MAX_TEXTURES];
    int tex_h[RT_DEV_MAX_TEXTURES];
};

// Per-frame camera constants, computed on the host with the float/double sequence of
// GetRayDirection (MyApplication.cs:431-448); only sx, sy vary per pixel.
struct DevCamera {
    float pos[3];
    float dir[3];
    float up[3];
    float left[3];   // Normalized(Cross(up, dir))
    float tan_h;     // (float)Math.Tan(angleH / 2)
    float tan_v;     // (float)Math.Tan(angleV / 2)
    int width, height;
    // The rows of this render call: local row l (0 <= l < rows, row l of the output block) is row
    //   row_begin + ((l / band) * parts + part) * band + l % band
    // of the frame: every parts-th band of `band` rows.  A contiguous block is parts = 1, part = 0.
    int row_begin, rows;
    int band, parts, part;
};

// Device-side counters of one render call.  DevFrame::counters holds RT_COUNTER_SLOTS of
// them: every workgroup adds its totals to the slot of its blockIdx.x with plain loads and
// stores (launches are stream-ordered and no grid is larger than RT_COUNTER_SLOTS), and the
// host sums the slots when statistics are asked for.  One counter word shared by the whole
// grid would take only ~90 atomics per microsecond, a visible tail on a 2 ms frame.
#define RT_COUNTER_SLOTS 4096
struct DevCounters {
    unsigned long long sphere_tests;
    unsigned long long plane_tests;
    unsigned long long triangle_tests;
    unsigned long long exact_candidates;   // (ray origin, sphere) pairs evaluated exactly
    unsigned long long filter_ray_tests;   // (ray, sphere or ball) pairs the filter evaluated, plus the cells the
                                           // reflection rays' grid walks visited
    unsigned long long filter_origin_terms;   // (origin, sphere or ball) per-origin terms formed
    unsigned long long primary_filter_ray_tests;   // the primary kernel's share of filter_ray_tests
    unsigned long long primary_exact_candidates;   // ... and of exact_candidates
    unsigned long long shadow_candidates;   // (light ray, sphere) pairs evaluated exactly from the lights' tables
};

// Tile schedule of k_trace_levels (rt_kernels.hip), zeroed before every render call.
#define RT_DEV_MAX_LEVELS 18   // RT_MAX_DEPTH + 2
#define RT_DEV_CU_SLOTS 2048   // 8 XCCs x 256 (SE, SH, CU) codes
struct DevSched {
    int next_a[RT_DEV_MAX_LEVELS];   // next unclaimed tile of level k's reflecting records
    int next_b[RT_DEV_MAX_LEVELS];   // ... of its plain records
    int done_a[RT_DEV_MAX_LEVELS];   // finished reflecting tiles of level k
    int open_level;                  // levels <= open_level have their final record counts
    int error;                       // set when a workgroup gave up waiting for a level
    int next_primary;                // k_primary: tiles handed out beyond each workgroup's first
    // Timeline of the launch in ticks of the constant 100 MHz clock (wall_clock64): first claim,
    // the opening of every level, last exit.  Written only by builds with -DRT_SCHED_TIMELINE and
    // printed by the host when RT_HIP_SCHED_TRACE is set.
    unsigned long long t_first, t_last;
    unsigned long long t_open[RT_DEV_MAX_LEVELS];
    // Workgroups of k_trace_levels that have started on each compute unit (index: XCC id, then
    // the SE / SH / CU bits of HW_ID): a workgroup's count on arrival is its age rank there.
    int cu_arrivals[RT_DEV_CU_SLOTS];
};

// Hit records: the bounce stack of every pixel, kept in HBM.  Level k of the frame
// owns slots [k*P, (k+1)*P) (P = pixels of the render call, rounded up to RT_LEVEL_ALIGN so
// that levels never share a cache line in any per-slot array).  Records whose surface
// needs a reflection trace ("A": specular != 0 and k < max_depth) are appended from
// the front of the level, the others ("B") from the back, so each level is two dense
// queues and no index arrays are needed.
//   recA[slot] = {hit.xyz, prim}          prim = host primitive index (int bits)
//   recB[slot] = {view.xyz, parent}       parent = slot of the level k-1 record, or
//                                         the pixel index at level 0 (int bits)
//   refl[slot] = {rgb of the reflected Shade(), valid}   written by the child
#define RT_LEVEL_ALIGN 64
// The per-light occluder lists (lt_sph, lt_reach, lt_orig) end with this many entries that never occlude:
// a lane may load up to that far past the end of its own list.
#define RT_LT_PAD 8
// The grid's entry arrays (gr_sph, gr_orig) end with this many entries that can never be hit: the walk loads
// its entries in pairs.
#define RT_GRID_PAD 4
#define RT_BIN_AXIS 16                                         // position bins per axis
#define RT_BINS (RT_BIN_AXIS * RT_BIN_AXIS * RT_BIN_AXIS)      // per queue
struct DevFrame {
    float4 *recA;
    float4 *recB;
    float4 *refl;
    int *cnt;              // cnt[2*k] = A records of level k, cnt[2*k+1] = B records
    DevCounters *counters;
    DevSched *sched;
    int P;                 // level stride: pixels of this render call rounded up to RT_LEVEL_ALIGN
    int slots;             // P * (max_depth + 1)
    int max_depth;
    int any_hit;           // 1: a light ray may stop at ANY occluder (the Intersect-call counters are not wanted)
    int chain_policy;      // 1: k_trace_levels keeps the tiles the next level waits for to the workgroups that
                           // arrived first on their compute units, and the first of each free for them
    int num_cus;
    // Host-visible outcome (page-locked host memory the device writes through the bus, and only
    // when something went wrong): status[0] != 0 = a workgroup of k_trace_levels gave up waiting
    // for a level to open, status[1] = that level.  Never cleared by the device.
    int *status;
    // Page-locked host memory too, or null: what the last launch of a frame tells the host about the frame, as a hint for
    // how to launch the next one (hint[0] = hint_tag, hint[1 + i] = cnt[i]); nothing's correctness rests on it.
    int *hint;
    int hint_tag;
    unsigned stall_ticks;  // how long a workgroup waits without progress, in ticks of the 100 MHz wall clock
    int debug_withhold;    // test knob: the first reflecting tile of this level is never counted as done (-1: off)
    // Records of a level >= 1 put into the order of their hit points' position bins before the level is
    // traced and shaded (k_bin_*): bin_min > 0 enables it for levels of at least that many records.
    int bin_min;
    uint2 *bin_aux;        // per record of the level being binned: {bin, rank within the bin}
    int stage_base;        // frames with bin_min > 0: first slot of a spare level of recA / recB behind the last one,
                           // what the binning passes move a level's records through
    int staged;            // bit k: the records of level k >= 1 are first written there (laid out like a level; the grid
                           // walk's launches), and k_bin_scatter moves them to the level's own slots
    int *bin_hist;         // [2 * RT_BINS] records per bin (reflecting queue, plain queue); zero between uses
    int *bin_base;         // [2 * RT_BINS] first position of each bin in its queue
    // Light terms of the reflecting records, computed ahead of their children (k_lit): per slot of the
    // levels below max_depth and per light {what the light adds directly (rgb), distance * distance}.
    // Null: k_shade runs the whole light loop itself.
    float4 *lit;
    int lit_stride;        // records per level that lit has room for
    int lit_below;         // > 0: only for levels of fewer than that many reflecting records (the others run their
                           // whole light loop once their children are shaded)
    int path_shade;        // 1: a reflecting record whose child reflects too is marked (refl.w = RT_REFL_PENDING), and the
                           // sums of the levels from path_lo down run path by path in one launch (k_shade_paths)
    int light_waves;       // > 0: queues of fewer than light_split_below records are shaded with the lights spread
    int light_split_below; // over that many waves per 64 records (k_shade_lights)
    int spread_below;      // one-chunk scenes: levels with fewer reflecting tiles than this take 64 records per workgroup
                           // and spread the scene over its waves (k_reflect)
    int grid_split_below;  // scenes with a grid: a level of fewer reflecting records than this has every ray's walk cut
                           // into segments, about this many walks in all (0: never)
    int split_chunks;      // 1: levels with fewer than split_below reflecting tiles are traced one (tile, chunk)
    int split_below;       // job at a time (k_reflect_chunks): new reflecting records get their closest-hit word
                           // initialised in refl[]
};

// Launch wrappers (rt_kernels.hip).  All enqueue on `stream` and return at once.
void rtk_primary(const DevScene &s, const DevCamera &cam, const DevFrame &f, int32_t *pixels,
                 int grid_blocks, hipStream_t stream);
// reflecting: the level's records that need a reflection trace (queue A) or the others (B).
void rtk_trace(const DevScene &s, const DevFrame &f, int level, bool reflecting, int32_t *pixels, int grid_blocks,
               hipStream_t stream);
// The reflecting records of a level of a many-chunk scene as (tile, chunk) jobs, then the pass that
// completes them (planes, triangles, next level's records).  Needs DevFrame::split_chunks.
void rtk_reflect_chunks(const DevScene &s, const DevFrame &f, int level, int job_blocks, int tile_blocks, hipStream_t stream);
// The records of level `level` reordered by position bin (four launches; see DevFrame::bin_min).
void rtk_bin_level(const DevScene &s, const DevFrame &f, int level, int grid_blocks, hipStream_t stream);
// Every level in one launch; grid_blocks = workgroups resident at once on the device
// (rtk_trace_levels_resident_blocks), fewer if the frame has fewer tiles.
// (of the instance rtk_trace_levels launches for a frame with that chain policy / counting mode)
int rtk_trace_levels_resident_blocks(const DevScene &s, int num_cus, bool reserve, bool count);
void rtk_trace_levels(const DevScene &s, const DevFrame &f, int32_t *pixels, int grid_blocks, hipStream_t stream);
// alone: no rtk_shade_lights(level, 1) follows; the launch then shades the queue whatever its length.
void rtk_shade_level(const DevScene &s, const DevFrame &f, int level, bool alone, int32_t *pixels,
                     int grid_blocks, hipStream_t stream);
// The sums of the reflecting records of every level from level_lo down, path by path (needs DevFrame::path_shade
// and the light terms of all those levels in DevFrame::lit).
void rtk_shade_paths(const DevScene &s, const DevFrame &f, int level_lo, int32_t *pixels, int grid_blocks, hipStream_t stream);
// The plain records of levels level_lo .. level_hi in one launch, with the light terms of their reflecting
// records when DevFrame::lit is set.
void rtk_plain_levels(const DevScene &s, const DevFrame &f, int level_lo, int level_hi, int32_t *pixels, int grid_blocks,
                      hipStream_t stream);
// The light terms of a level's reflecting records into DevFrame::lit (which must not be null).
void rtk_lit_level(const DevScene &s, const DevFrame &f, int level, int grid_blocks, hipStream_t stream);
// The queues of levels level_lo .. level_hi with the lights spread over DevFrame::light_waves waves per 64 records;
// does nothing for a queue that is not shorter than DevFrame::light_split_below.  queue 0: the plain records, shaded;
// 1: the reflecting records, shaded (their children are); 2: the reflecting records' light terms into DevFrame::lit;
// 3: 0 and 2.
void rtk_shade_lights(const DevScene &s, const DevFrame &f, int level_lo, int level_hi, int queue, int32_t *pixels,
                      int grid_blocks, hipStream_t stream);
// Raytracer.Debug's rays: 16 floats per ray (rt_debug_ray's 14 words, then two of padding).
void rtk_debug_rays(const DevScene &s, const DevCamera &cam, int row, int x_step, int count, float *out,
                    hipStream_t stream);
// The two launches of the test whether two streams share a hardware queue (rt_api.cpp, streams_overlap): words[0]
// = the flag (zero before), words[1] = what the waiter saw.
void rtk_probe_wait(int *words, unsigned long long ticks, hipStream_t stream);
void rtk_probe_set(int *words, hipStream_t stream);
// The present path's FXAA filter over a w x h frame (in and out must not overlap).
void rtk_fxaa(const int32_t *in, int32_t *out, int w, int h, hipStream_t stream);
