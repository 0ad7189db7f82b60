// rt_kernels.hip — gfx950 kernels of the render loop.
//
// Arithmetic contract: every value that can reach a pixel is evaluated operation by
// operation in the order the reference's C# evaluates it (no FMA contraction: this file
// is built with -ffp-contract=off; IEEE divide and square root), and double precision is
// used exactly where the reference calls System.Math (sin, pow).  The one place that
// departs from it is the sphere FILTER below, which only decides which (ray, sphere) pairs
// get the exact evaluation; it is proven to let through every pair the exact evaluation
// could accept, so it cannot change a result.
//
// Pipeline per frame (one HIP stream, in order):
//   setup_camera   per sphere: the filter's per-sphere term for the camera origin
//   primary        per pixel: ray direction, closest hit -> level-0 hit record
//   trace_level k  per hit record: if the surface is specular and k < max_depth, its
//                  reflection ray (closest hit) walks the scene; its light rays (any-hit)
//                  look up the spheres that can occlude them in per-light direction tables
//   shade_level k  (k = max_depth .. 0) Primitive.Shade for every record of the level;
//                  the colour goes to the parent record or, at level 0, to the pixel
// The per-pixel bounce stack lives in HBM as compacted per-level queues (rt_device.h),
// so every wave of every level is dense whatever the neighbouring pixels do.
//
// Instruction selection follows what the chip measures (tools/valu_issue_bench.hip):
// v_mul/add/sub/fma_f32 and v_and/or_b32 with VGPR-only operands issue every ~2.1 cycles
// per wave; the same instructions with an SGPR operand, every v_pk_*_f32, v_cmp_*,
// v_min/max_f32 and v_max3_f32 take ~4.1.  Hence: sphere data is staged in LDS and read
// into VGPRs (ds_read_b128, all lanes one address) instead of being used from SGPRs, the
// filter is plain FMAs, and its verdicts are combined with integer AND/OR on sign bits.
#include "rt_device.h"
#include "filter_asm.inc"

#include <limits.h>

#define RT_EPS 0.0001f
#define RT_FLT_MAX 3.40282347e+38f
// Margins of the sphere filter (derivation: "Sphere filter" below); u = 2^-24.
#define FILTER_K1 RT_FILTER_K1               // 1 - 2^-18 = 1 - 64u
#define FILTER_FM_SCALE 9.5367431640625e-07f // 2^-20   = 16u
#define FILTER_E_SCALE 9.5367431640625e-07f  // 2^-20   = 16u
#define FILTER_BALL_SLACK 1.5e-3f            // >= sqrt(28.6u) + 4.1u, see "Bounding balls"

// ---------------------------------------------------------------------------------------
// OpenTK.Mathematics Vector3 operators, restated (Dot / Cross / Length / Normalized).
// ---------------------------------------------------------------------------------------
struct f3 {
    float x, y, z;
};
// Scene arrays are read through the constant address space: their addresses are
// wave-uniform and no kernel writes them while it reads them, so the loads become scalar
// loads (s_load_dwordx4..x16 into SGPRs, broadcast to the lanes for free) even in loops that
// also store to other buffers, where the compiler could not otherwise prove them unclobbered.
typedef float f4raw __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) f4raw *const_f4_ptr;
typedef const __attribute__((address_space(4))) int *const_int_ptr;
__device__ __forceinline__ float4 cload(const float4 *p, int i)
{
    const f4raw v = ((const_f4_ptr)(unsigned long long)p)[i];
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int cload(const int *p, int i) { return ((const_int_ptr)(unsigned long long)p)[i]; }
typedef const __attribute__((address_space(4))) float *const_float_ptr;
__device__ __forceinline__ float cload(const float *p, int i) { return ((const_float_ptr)(unsigned long long)p)[i]; }

// IEEE-754 correctly rounded FP32 divide and square root.  Plain `/` and sqrtf() are
// that under -fhip-fp32-correctly-rounded-divide-sqrt (set explicitly by the build);
// the __fdiv_rn / __fsqrt_rn intrinsics are NOT (they map to the 1-ulp native forms).
__device__ __forceinline__ float ieee_div(float a, float b) { return a / b; }
__device__ __forceinline__ float ieee_sqrt(float a) { return sqrtf(a); }
__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return ((a.x * b.x) + (a.y * b.y)) + (a.z * b.z); }
__device__ __forceinline__ f3 cross3(f3 a, f3 b)
{
    return mk3((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x));
}
__device__ __forceinline__ float length3(f3 a) { return ieee_sqrt(dot3(a, a)); }
__device__ __forceinline__ f3 normalized3(f3 a)
{
    float scale = ieee_div(1.0f, length3(a));
    return mk3(a.x * scale, a.y * scale, a.z * scale);
}
__device__ __forceinline__ f3 xyz(float4 v) { return mk3(v.x, v.y, v.z); }

// System.Math.Max(float, float): IEEE 754-2019 maximum (NaN propagates, +0 > -0).
__device__ __forceinline__ float net_maxf(float a, float b)
{
    if (a != b) {
        if (!(a != a)) return b < a ? a : b;
        return a;
    }
    return (__float_as_int(b) < 0) ? a : b;
}

// C# (int)float on x64 (cvttss2si): truncate; NaN / out of range -> INT_MIN.
__device__ __forceinline__ int net_f2i(float f)
{
    if (!(f >= -2147483648.0f && f < 2147483648.0f)) return INT_MIN;
    return (int)f;
}

// Sphere.Intersect objects.cs:113-121 from x = Dot(d, o-p), c and a = Dot(d, d).
// Returns false when the discriminant is negative; otherwise t (which may be <= 0).
__device__ __forceinline__ bool sphere_root(float x, float c, float a, float &t)
{
    float b = 2.0f * x;
    float disc = (b * b) - ((4.0f * a) * c);
    if (disc < 0.0f) return false;
    t = ieee_div((-b) - ieee_sqrt(disc), 2.0f * a);
    return true;
}

// Cheap exact filter in front of sphere_root.  With m = x*|x| (= -x^2 for x < 0):
//   -m >= a*c  <=>  x < 0 and x*x >= a*c   (or x >= 0 with c <= 0: origin inside)
// x*x < a*c is exactly `discriminant < 0` (b*b = 4*fl(x*x) and 4*a*c = 4*fl(a*c) are
// exact scalings), and x >= 0 gives -b - sqrt(disc) <= 0, i.e. t <= 0, which both
// Scene.Intersect and IsInShadow discard (t > epsilon).  So every sphere this filter
// drops is one the reference drops too; the ones it keeps go through sphere_root.
__device__ __forceinline__ bool sphere_maybe(float x, float ac)
{
    float m = x * fabsf(x);
    return (-m) >= ac;
}

// Plane.Intersect objects.cs:157-164 from numer = -Dot(n, o) - d.
__device__ __forceinline__ bool plane_t(f3 n, float numer, f3 d, float &t)
{
    float denom = dot3(n, d);
    if (fabsf(denom) <= 0.0001f) return false;
    t = ieee_div(numer, denom);
    return !(t <= 0.0001f);
}

// Triangle.Intersect objects.cs:240-261 (e1, e2 and s = o - v0 prepared by the caller).
__device__ __forceinline__ bool triangle_t(f3 e1, f3 e2, f3 sv, f3 d, float &t)
{
    f3 h = cross3(d, e2);
    float a = dot3(e1, h);
    if (a > -0.0001f && a < 0.0001f) return false;
    float f = ieee_div(1.0f, a);
    float u = f * dot3(sv, h);
    if (u < 0.0f || u > 1.0f) return false;
    f3 q = cross3(sv, e1);
    float v = f * dot3(d, q);
    if (v < 0.0f || (u + v) > 1.0f) return false;
    t = f * dot3(e2, q);
    return t > 0.0001f;
}

// Closest-hit bookkeeping across primitive classes: the reference scans in host order
// with a strict `<` (MyApplication.cs:361), i.e. it keeps the lexicographic minimum of
// (t, host index).  Classes are scanned one after the other here, so ties compare the
// host index explicitly.
struct Closest {
    float t;
    int prim;
};
__device__ __forceinline__ void closest_offer(Closest &c, float t, int prim)
{
    if (t > RT_EPS && (t < c.t || (t == c.t && prim < c.prim))) {
        c.t = t;
        c.prim = prim;
    }
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Adds a workgroup's totals to its slot of DevFrame::counters (rt_device.h).  v[i] is the
// calling wave's total of counter word i (read from lane 0); sh is a __shared__ array of
// COUNTER_WORDS words.  Every thread of the workgroup must call it (barriers inside).
#define COUNTER_WORDS (int)(sizeof(DevCounters) / sizeof(unsigned long long))
__device__ __forceinline__ void flush_counters(const DevFrame &f, unsigned long long *sh,
                                               const unsigned long long (&v)[COUNTER_WORDS])
{
    if (threadIdx.x < COUNTER_WORDS) sh[threadIdx.x] = 0ull;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < COUNTER_WORDS; i++)
            if (v[i]) atomicAdd(&sh[i], v[i]);
    }
    __syncthreads();
    if (threadIdx.x < COUNTER_WORDS && sh[threadIdx.x]) {
        unsigned long long *slot = reinterpret_cast<unsigned long long *>(&f.counters[blockIdx.x % RT_COUNTER_SLOTS]);
        slot[threadIdx.x] += sh[threadIdx.x];
    }
}

__device__ __forceinline__ bool material_specular_nonzero(const DevScene &s, int prim)
{
    return cload(s.mat, 3 * prim).w != 0.0f;
}

// A hit record goes to memory with a device-scope store (sc1: written through the XCD's L2),
// so that workgroups on other XCDs can read it in the same launch without anybody having to
// write a whole L2 back (k_trace_levels).
__device__ __forceinline__ void store_record(float4 *dst, const f4raw v)
{
    // The s_nop covers the hazard "store of more than 64 bits, then a write of its data
    // registers", which the compiler cannot see through inline assembly.
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
}

// A 32-bit word to memory with a device-scope store, like store_record.
__device__ __forceinline__ void store_word_through(uint32_t *dst, uint32_t v)
{
    asm volatile("global_store_dword %0, %1, off sc1" : : "v"(dst), "v"(v) : "memory");
}

// The two halves of a hit record with device-scope loads (sc1: served from memory, not from
// whatever this XCD's L2 or this CU's L1 may hold of the line): the counterpart of store_record
// for records that another workgroup wrote earlier in the same launch.
__device__ __forceinline__ void load_record(const float4 *a, const float4 *b, float4 &ra, float4 &rb)
{
    f4raw va, vb;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\t"
                 "global_load_dwordx4 %1, %3, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(va), "=&v"(vb)
                 : "v"(a), "v"(b)
                 : "memory");
    ra = make_float4(va.x, va.y, va.z, va.w);
    rb = make_float4(vb.x, vb.y, vb.z, vb.w);
}

// refl[slot].w of a reflecting record: 0 = its ray met nothing, 1 = xyz is the child's colour, and (frames with
// DevFrame::path_shade) RT_REFL_PENDING = the child reflects too and has not been summed yet (mark_parent).
#define RT_REFL_PENDING 2.0f

// Append up to N new hit records per thread to level `level`, with ONE atomic per queue per
// WORKGROUP: a single counter word takes only ~90 returning atomics per microsecond, and a
// frame appends hundreds of thousands of waves' worth of records.  Every thread of the
// workgroup must call it (two barriers inside); blk is a __shared__ int[2 * waves + 2] or larger.
//   A records (need a reflection trace) fill the level from the front, B records from the back.
template <int N, bool THROUGH = true, bool STAGE = false>
__device__ __forceinline__ void push_records(const DevFrame &f, const DevScene &s, int level, int *blk,
                                             const bool (&valid)[N], const int (&prim)[N], const f3 (&hit)[N],
                                             const f3 (&view)[N], const int (&parent)[N])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    bool is_a[N];
    int off[N];
    int wave_a = 0, wave_b = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
        // The reference traces the reflection inside its light loop (objects.cs:49,80-83):
        // no lights, no trace.
        is_a[k] = valid[k] && level < f.max_depth && s.nl > 0 && material_specular_nonzero(s, prim[k]);
        const unsigned long long ma = __ballot(is_a[k]), mb = __ballot(valid[k] && !is_a[k]);
        off[k] = is_a[k] ? wave_a + __popcll(ma & below) : wave_b + __popcll(mb & below);
        wave_a += __popcll(ma);
        wave_b += __popcll(mb);
    }
    if (lane == 0) {
        blk[2 * wave] = wave_a;
        blk[2 * wave + 1] = wave_b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int ta = 0, tb = 0;
        for (int w = 0; w < nwaves; w++) {
            const int a = blk[2 * w], b = blk[2 * w + 1];
            blk[2 * w] = ta;
            blk[2 * w + 1] = tb;
            ta += a;
            tb += b;
        }
        // Both queues' counters in ONE returning atomic (they are the two halves of an aligned
        // 64-bit word; neither half can carry: a level holds fewer than 2^31 records).
        unsigned long long before = 0ull;
        if (ta | tb)
            before = atomicAdd(reinterpret_cast<unsigned long long *>(&f.cnt[2 * level]),
                               ((unsigned long long)(unsigned)tb << 32) | (unsigned long long)(unsigned)ta);
        blk[2 * nwaves] = (int)(unsigned)before;
        blk[2 * nwaves + 1] = (int)(unsigned)(before >> 32);
    }
    __syncthreads();
    const int base_a = blk[2 * nwaves] + blk[2 * wave], base_b = blk[2 * nwaves + 1] + blk[2 * wave + 1];
    // STAGE (the grid walk's launches): a frame whose deeper levels are put into bin order stages their records in
    // a spare level (DevFrame::staged, stage_base); k_bin_scatter moves them to their slots.
    const int first = STAGE && level >= 1 && ((f.staged >> level) & 1) ? f.stage_base : level * f.P;
#pragma unroll
    for (int k = 0; k < N; k++) {
        if (valid[k]) {
            const int slot = is_a[k] ? (first + base_a + off[k]) : (first + f.P - 1 - (base_b + off[k]));
            // THROUGH: the records are read later in this launch, by other XCDs (device-scope
            // stores); otherwise by a later launch, which sees plain stores.
            if (THROUGH) {
                store_record(&f.recA[slot], f4raw{hit[k].x, hit[k].y, hit[k].z, __int_as_float(prim[k])});
                store_record(&f.recB[slot], f4raw{view[k].x, view[k].y, view[k].z, __int_as_float(parent[k])});
            } else {
                f.recA[slot] = make_float4(hit[k].x, hit[k].y, hit[k].z, __int_as_float(prim[k]));
                f.recB[slot] = make_float4(view[k].x, view[k].y, view[k].z, __int_as_float(parent[k]));
            }
            // Frames whose reflection rays are traced chunk by chunk (k_reflect_chunks) keep the
            // closest hit so far in the first two words of the record's refl slot: nothing yet.
            if (f.split_chunks && is_a[k]) {
                const float none = __uint_as_float(0xffffffffu);
                const int own = slot - first + level * f.P;   // (refl is not staged)
                if (THROUGH) store_record(&f.refl[own], f4raw{none, none, 0.0f, 0.0f});
                else f.refl[own] = make_float4(none, none, 0.0f, 0.0f);
            }
        }
    }
    __syncthreads();  // blk may be reused by the caller's next trip
}

// ---------------------------------------------------------------------------------------
// Sphere filter.
//
// For a ray (o, d) and a sphere (p, r) the reference computes, in FP32 without contraction
// (objects.cs:112-121):  a = d.d,  x = d.(o-p),  c = (o-p).(o-p) - r*r,  disc = 4(x*x - a*c),
// and only a pair with disc >= 0 and x < 0 can produce t > epsilon.  Call the computed values
// x_ref, c_ref and the exact ones (real arithmetic on the same FP32 inputs) X, Q = |o-p|^2.
// The filter works in coordinates relative to a per-scene centre z (DevScene::sph_center):
//   o' = fl(o - z),  p' = fl(p - z)  (p' rounded once on the host),  w = o' - p' exactly,
// so o - p = w + eps with |eps_i| <= u(|o'_i| + |p'_i|), u = 2^-24.  With the bounds
//   B1 = |o'|_1 + max_j |p'_j|_1   and   B2 = |o'|_2 + |p'|_2 <= B1
// it evaluates, all in FMAs:
//   per ray      e  = fm - d.o'                     fm = 16u * B1
//   per pair     x' = e + d.p'                      = fm - d.w up to 6 roundings <= 6.1u*B1
//   per sphere   -c'' = W + B.p' - C0               W = fl(r*r - k1*|p'|^2) (host, from double),
//                                                   B = fl(2*k1*o'),  C0 = fl(k1*|o'|^2) - E,
//                                                   k1 = 1 - 64u,  E = 16u * (2|o'|^2 + S),
//                                                   S >= 2 max|p'|^2 + max r*r  (DevScene::sph_sc)
//                i.e. c'' = k1*|w|^2 - r*r - E up to roundings <= 9.1u(r*r + B2^2) + 7u*E
//   verdict      v = fma(x', |x'|, -c'');  the pair passes when v's sign bit is clear.
// No accepted pair is lost:
//   (1) x_ref < 0  =>  x' >= |x_ref|.  x_ref = X + eta with |eta| <= 4.1u|o-p| <= 4.1u*B1
//       (one rounding per component of o-p, three products, two sums); X = d.w + d.eps with
//       |d.eps| <= 1.1u*B1.  Hence x' >= fm - X - 7.2u*B1 >= -X - eta = |x_ref| because
//       fm = 16u*B1 >= 11.3u*B1.
//   (2) disc_ref >= 0  =>  x_ref^2 >= T := Q(1 - 14.2u) - r*r.  For c_ref > 0: fl(x_ref^2) >=
//       fl(a_ref*c_ref) with a_ref >= 1 - 6u (d is a normalised vector), c_ref >=
//       (q_ref - r*r)(1 - u), q_ref >= Q(1 - 5.1u); for c_ref <= 0, T <= 0 anyway.
//   (3) c'' <= T.  Q >= |w|^2 - 2u*B2^2, so T >= |w|^2(1 - 14.2u) - 2u*B2^2 - r*r, while the
//       computed c'' <= k1|w|^2 - r*r - E(1 - 7u) + 9.1u(r*r + B2^2); the difference is
//       <= -49.8u|w|^2 - E(1 - 7u) + 11.1u*B2^2 + 9.1u*r*r <= 0 because B2^2 <= 2|o'|^2 +
//       2 max|p'|^2 and therefore E >= 16u(1 - 5u)(B2^2 + r*r).
//   With (1)-(3): x'|x'| = x'^2 >= x_ref^2 >= T >= c'', so v >= 0 and its sign bit is clear.
// What the margins cost: the sphere looks larger by E in r*r, a few 1e-5 for scene extents
// of a few units.  A sphere with non-finite data gets W = -inf (v = -inf, never passes); the
// reference never accepts such a sphere either.  The argument assumes |o'|^2, |p'|^2 do not
// overflow FP32; where they do, the reference's own c is infinite or NaN and it accepts nothing.
// Per sphere the words v of all rays are ANDed; a sphere whose combined word is >= 0 (as an
// integer) in a lane goes to that lane's exact code, which repeats the reference's
// evaluation literally (sphere_maybe / sphere_root).
//
// Bounding balls.  The spheres are stored in groups of RT_GROUP spatial neighbours, each with
// a ball (P, R) such that |p_i - P| + r_i <= R for its members (rt_api.cpp).  Before a wave
// evaluates a group's spheres it asks whether any lane's ray can meet the ball; if none can,
// the group is skipped.  This loses no accepted pair: if the reference accepts (ray, sphere
// i), then by (1) X_i <= 4.1u*B and by (2) the distance of p_i from the ray's line satisfies
//   perp_i^2 = Q_i - X_i^2/|d|^2 <= r_i^2 + 28.6u*B^2,   B >= |o - p_i|,
// (14.2u*Q from T, 8.3u*B^2 from x_ref vs X, 6.1u*Q from |d|^2 >= 1 - 6u), so with
// sg = (sqrt(28.6u) + 4.1u)*B the ball of radius R' = R + sg around P has
//   centre ahead or abeam (X_g <= 0):  perp_g <= perp_i + |P - p_i| <= R', i.e. X_g^2 >= Q_g - R'^2;
//   centre behind (X_g > 0):           Q_g <= (|P - p_i| + perp_i + 4.1u*B)^2 <= R'^2: o is inside.
// The ball test therefore is "x' >= 0 and x'^2 >= c''_g, or c''_g <= 0" with the same
// evaluation and the same evaluation margins as above, for the sphere (P, R) and
// E_g = E + 2*max R*sg + sg^2 in place of E (so that c''_g <= Q_g(1 - 14.2u) - R'^2):
//   v1 = fma(x', x', -c''_g);  reject = (sign(v1) | sign(x')) & [c''_g > 0].
// ---------------------------------------------------------------------------------------
struct FilterOrigin {
    f4raw lane;     // (Bx, By, Bz, C0): the sphere loops' per-origin operand
    f4raw lane_g;   // (Bx, By, Bz, C0 of the groups' bounding balls),
    f4raw lane_s;   //   ... of the balls around RT_SUPER spheres,
    f4raw lane_c;   //   ... of the chunks' balls
    f3 oc;          // o' = o - centre
    float fm;       // front margin
};

__device__ __forceinline__ FilterOrigin filter_origin(const DevScene &s, f3 o)
{
    FilterOrigin fo;
    fo.oc = mk3(o.x - s.sph_center[0], o.y - s.sph_center[1], o.z - s.sph_center[2]);
    fo.fm = FILTER_FM_SCALE * (((fabsf(fo.oc.x) + fabsf(fo.oc.y)) + fabsf(fo.oc.z)) + s.sph_l1max);
    const float oo = __builtin_fmaf(fo.oc.z, fo.oc.z, __builtin_fmaf(fo.oc.y, fo.oc.y, fo.oc.x * fo.oc.x));
    const float e = FILTER_E_SCALE * __builtin_fmaf(2.0f, oo, s.sph_sc);
    const float k2 = 2.0f * FILTER_K1;
    fo.lane = f4raw{k2 * fo.oc.x, k2 * fo.oc.y, k2 * fo.oc.z, (FILTER_K1 * oo) - e};
    // Balls: radius widened by sg (see "Bounding balls" above): E_g = E + 2*Rmax*sg + sg^2.
    const float sg = FILTER_BALL_SLACK * (((fabsf(fo.oc.x) + fabsf(fo.oc.y)) + fabsf(fo.oc.z)) + s.sph_l1max);
    const float eg = e + __builtin_fmaf(2.0f * s.grp_rmax, sg, sg * sg);
    const float es = e + __builtin_fmaf(2.0f * s.sup_rmax, sg, sg * sg);
    const float ec = e + __builtin_fmaf(2.0f * s.chk_rmax, sg, sg * sg);
    fo.lane_g = f4raw{fo.lane.x, fo.lane.y, fo.lane.z, (FILTER_K1 * oo) - eg};
    fo.lane_s = f4raw{fo.lane.x, fo.lane.y, fo.lane.z, (FILTER_K1 * oo) - es};
    fo.lane_c = f4raw{fo.lane.x, fo.lane.y, fo.lane.z, (FILTER_K1 * oo) - ec};
    return fo;
}

// (e, dx, dy, dz): one ray's operand of the filter loops.
__device__ __forceinline__ f4raw filter_ray(const FilterOrigin &fo, f3 d)
{
    const float e = __builtin_fmaf(-d.z, fo.oc.z, __builtin_fmaf(-d.y, fo.oc.y, __builtin_fmaf(-d.x, fo.oc.x, fo.fm)));
    return f4raw{e, d.x, d.y, d.z};
}

// ---------------------------------------------------------------------------------------
// The filter loops themselves are gfx950 assembly with a fixed, VGPR-bank-conflict-free
// register plan (filter_asm.inc, generated by tools/gen_filter_asm.py): hipcc's register
// allocation ignores banks, and a three-source instruction with two sources in one bank
// issues at half rate.  Per staged chunk a wave runs a GROUP loop over the chunk's bounding
// balls, ORs the lanes' pass bits over the wave, runs a SEL loop over the groups some lane
// may hit (lanes queue one word per trip with a candidate), and then evaluates each lane's
// (short) queue exactly here in C++.

// Staged per workgroup (WalkLds): a chunk of SPH_CHUNK filter entries {p', W}, the bounding balls of its groups and of its runs of RT_SUPER
// spheres, each array followed by FA_LDS_SLACK never-pass entries for the GROUP loops'
// read-ahead.  Scenes with more than one chunk also stage the chunks' own balls, GRP_CHUNK at a
// time, in the groups' array before they walk the chunks those select.
#define SPH_CHUNK RT_CHUNK
#define GRP_CHUNK (SPH_CHUNK / RT_GROUP)
#define GRP_LDS_ENTRIES (GRP_CHUNK + FA_LDS_SLACK)
#define SUP_CHUNK (SPH_CHUNK / RT_SUPER)
#define SUP_LDS_ENTRIES (SUP_CHUNK + FA_LDS_SLACK)
#define GRP_WORDS (GRP_CHUNK / 32)
// Queues: QUEUE_SLOTS words per thread, slot-major (slot s of thread t at int index 256*s + t).
// GROUP loops leave trip j's reject mask (ball m of the trip = bit 7-m, 0 = could hit) in
// slot j.  SEL loops append words that name a trip of FA_TRIP spheres and which of them the
// lane's rays could hit: (first index << 8) | reject mask; they return early when some
// thread's queue is full.
#define QUEUE_SLOTS 16
#define EXACT_BATCH 2   // candidates per lane whose sphere data is in flight together
#define QUEUE_INTS (QUEUE_SLOTS * 256)
static_assert(GRP_CHUNK / FA_TRIP <= QUEUE_SLOTS, "one queue slot per trip of balls");
static_assert(RT_GROUP == FA_TRIP && RT_SPH_PAD == FA_TRIP * RT_GROUP, "a trip of a SEL loop is one group");
static_assert(RT_SUPER == FA_TRIP * RT_GROUP, "a trip of a GROUP loop over the groups' balls is one coarse ball");
static_assert(SUP_CHUNK % FA_TRIP == 0, "the coarse balls of a full chunk are whole trips");

struct WalkLds {
    float4 sph[SPH_CHUNK];
    float4 grp[GRP_LDS_ENTRIES];
    float4 sup[SUP_LDS_ENTRIES];
    int cand[QUEUE_INTS];
    unsigned wg_bits[GRP_WORDS];   // chunks some wave of the workgroup has to walk
    float ref[4];                  // ... seen from here (thread 0's ray origin),
    float key[GRP_CHUNK];          // ... their distances from it
    int order[GRP_CHUNK];          // ... and the order they are walked in, nearest first
};

// Walks a thread's queued words: next() returns the next sphere index (local to the staged
// chunk) the lane has to evaluate exactly, or -1 when its queue is exhausted.
struct CandidateWalk {
    const int *cand;
    int nq, j, trip;
    unsigned pass;
    __device__ __forceinline__ CandidateWalk(const int *cand_, int nq_) : cand(cand_), nq(nq_), j(0), trip(0), pass(0u) {}
    __device__ __forceinline__ int next()
    {
        if (pass == 0u) {
            if (j >= nq) return -1;
            const unsigned w = (unsigned)cand[j * 256 + threadIdx.x];
            j++;
            trip = (int)(w >> 8);
            pass = (w & 0xffu) ^ 0xffu;   // never 0: only lanes with a candidate queue a word
        }
        const int b = 31 - __clz((int)pass);
        pass &= ~(1u << b);
        return trip + 7 - b;
    }
};

// Filter entries of spheres [base, base + count) and their balls (count: a multiple of RT_SUPER).
__device__ __forceinline__ void stage_chunk(WalkLds &L, const DevScene &s, int base, int count)
{
    const float4 never = make_float4(0.f, 0.f, 0.f, -INFINITY);
    for (int i = threadIdx.x; i < count; i += blockDim.x)
        L.sph[i] = s.sph_f[base + i];
    const int ng = count / RT_GROUP;
    for (int i = threadIdx.x; i < ng; i += blockDim.x)
        L.grp[i] = s.grp_f[base / RT_GROUP + i];
    if (threadIdx.x < FA_LDS_SLACK) L.grp[ng + threadIdx.x] = never;
    // Coarse balls: padded to whole trips (the source array is padded the same way).
    const int ns = count / RT_SUPER;
    if ((int)threadIdx.x < SUP_LDS_ENTRIES) {
        float4 e = never;
        if ((int)threadIdx.x < ns) e = s.sup_f[base / RT_SUPER + threadIdx.x];
        L.sup[threadIdx.x] = e;
    }
}

// Balls [first, first + count) of the chunks (count: a multiple of FA_TRIP, at most GRP_CHUNK).
__device__ __forceinline__ void stage_chunk_balls(WalkLds &L, const DevScene &s, int first, int count)
{
    for (int i = threadIdx.x; i < count + FA_LDS_SLACK; i += blockDim.x) {
        float4 e = make_float4(0.f, 0.f, 0.f, -INFINITY);
        if (i < count) e = s.chk_f[first + i];
        L.grp[i] = e;
    }
    if (threadIdx.x < GRP_WORDS) L.wg_bits[threadIdx.x] = 0u;
}

// Byte offset of an LDS object inside the workgroup's LDS allocation (what ds_read takes).
__device__ __forceinline__ unsigned lds_offset(const void *p)
{
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char *)p;
}

// Bitwise OR over the 64 lanes of a wave (all of them active), as a wave-uniform value:
// butterfly inside each row of 16 lanes, then row 0 -> 1, 2 -> 3 and rows 0-1 -> 2-3, which
// leaves the total in lane 63.
__device__ __forceinline__ unsigned wave_or(unsigned v)
{
    int x = (int)v;
    x |= __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);    // quad_perm:[1,0,3,2]
    x |= __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);    // quad_perm:[2,3,0,1]
    x |= __builtin_amdgcn_update_dpp(0, x, 0x124, 0xf, 0xf, false);   // row_ror:4
    x |= __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false);   // row_ror:8
    x |= __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
    x |= __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane(x, 63);
}

// The generated loops take the K rays of a lane at once.  Primary rays (TRACE false, K = 4): the
// staged entries already hold -c'' and the four verdicts are combined into one reject mask.
// Trace rays share a per-lane origin operand (`lane`) and come in two families:
//   combined (SPLIT false, K <= 3)  one reject mask for the K rays of a record;
//   split    (SPLIT true,  K <= 4)  the GROUP loop keeps one reject mask per ray (bits 8k..8k+7 of
//                                   the stored word are ray k's) and the SEL loops run one ray
//                                   at a time (rq[0]) over the groups that ray's mask selects.
template <int K, bool TRACE, bool SPLIT>
__device__ __forceinline__ void fa_group(int &i, unsigned &addr, unsigned &qaddr, const f4raw (&rq)[4],
                                         const f4raw lane, const int cnt)
{
#define FA_RUN(NAME, ...)                                                  \
    asm volatile(NAME##_ASM                                                \
                 : [i] "+s"(i), [addr] "+v"(addr), [qaddr] "+v"(qaddr)     \
                 : __VA_ARGS__, [cnt] "s"(cnt)                             \
                 : NAME##_CLOBBERS)
    if constexpr (!TRACE) {
        static_assert(K == 4, "the primary loops take four rays");
        FA_RUN(FA_PGROUP4, FA_DN0(rq[0]), FA_DN1(rq[1]), FA_DN2(rq[2]), FA_DN3(rq[3]));
    } else if constexpr (K == 1) {
        FA_RUN(FA_GROUP1, FA_DN0(rq[0]), FA_LANE(lane));
    } else if constexpr (!SPLIT) {
        static_assert(K <= 3, "combined loops exist for up to three rays");
        if constexpr (K == 2) FA_RUN(FA_GROUP2, FA_DN0(rq[0]), FA_DN1(rq[1]), FA_LANE(lane));
        if constexpr (K == 3) FA_RUN(FA_GROUP3, FA_DN0(rq[0]), FA_DN1(rq[1]), FA_DN2(rq[2]), FA_LANE(lane));
    } else {
        static_assert(K <= 4, "split loops exist for up to four rays");
        if constexpr (K == 2) FA_RUN(FA_GROUPS2, FA_DN0(rq[0]), FA_DN1(rq[1]), FA_LANE(lane));
        if constexpr (K == 3) FA_RUN(FA_GROUPS3, FA_DN0(rq[0]), FA_DN1(rq[1]), FA_DN2(rq[2]), FA_LANE(lane));
        if constexpr (K == 4)
            FA_RUN(FA_GROUPS4, FA_DN0(rq[0]), FA_DN1(rq[1]), FA_DN2(rq[2]), FA_DN3(rq[3]), FA_LANE(lane));
    }
#undef FA_RUN
}

// GSEL loop: the trips of the ball array at LDS offset addr0 that the bitmap names (bit b of wlo =
// trip b ^ 7), masks stored like a GROUP loop's (slot s of the lane at qaddr0 + s * FA_QUEUE_STRIDE).
template <int K, bool TRACE, bool SPLIT>
__device__ __forceinline__ void fa_gsel(unsigned long long wlo, const unsigned qaddr0, const f4raw (&rq)[4],
                                        const f4raw lane, const unsigned addr0)
{
    unsigned long long whi = 0ull;
    int trip0 = 0;
#define FA_RUN(NAME, ...)                                                    \
    asm volatile(NAME##_ASM                                                  \
                 : [wlo] "+s"(wlo), [whi] "+s"(whi), [trip0] "+s"(trip0)     \
                 : __VA_ARGS__, [qaddr] "v"(qaddr0), [addr0] "s"(addr0)      \
                 : NAME##_CLOBBERS)
    if constexpr (!TRACE) {
        FA_RUN(FA_PGSEL4, FA_DN0(rq[0]), FA_DN1(rq[1]), FA_DN2(rq[2]), FA_DN3(rq[3]));
    } else if constexpr (K == 1) {
        FA_RUN(FA_GSEL1, FA_DN0(rq[0]), FA_LANE(lane));
    } else if constexpr (!SPLIT) {
        if constexpr (K == 2) FA_RUN(FA_GSEL2, FA_DN0(rq[0]), FA_DN1(rq[1]), FA_LANE(lane));
        if constexpr (K == 3) FA_RUN(FA_GSEL3, FA_DN0(rq[0]), FA_DN1(rq[1]), FA_DN2(rq[2]), FA_LANE(lane));
    } else {
        if constexpr (K == 2) FA_RUN(FA_GSELS2, FA_DN0(rq[0]), FA_DN1(rq[1]), FA_LANE(lane));
        if constexpr (K == 3) FA_RUN(FA_GSELS3, FA_DN0(rq[0]), FA_DN1(rq[1]), FA_DN2(rq[2]), FA_LANE(lane));
        if constexpr (K == 4)
            FA_RUN(FA_GSELS4, FA_DN0(rq[0]), FA_DN1(rq[1]), FA_DN2(rq[2]), FA_DN3(rq[3]), FA_LANE(lane));
    }
#undef FA_RUN
}

// SEL loop for the K rays rq[0..K-1] (primary: K = 4; split trace walks call it with K = 1).
template <int K, bool TRACE>
__device__ __forceinline__ void fa_sel(unsigned long long &wlo, unsigned long long &whi, int &trip0, unsigned &qaddr,
                                       const f4raw (&rq)[4], const f4raw lane, const unsigned addr0,
                                       const unsigned qlim)
{
#define FA_RUN(NAME, ...)                                                                    \
    asm volatile(NAME##_ASM                                                                  \
                 : [wlo] "+s"(wlo), [whi] "+s"(whi), [trip0] "+s"(trip0), [qaddr] "+v"(qaddr) \
                 : __VA_ARGS__, [addr0] "s"(addr0), [qlim] "s"(qlim)                          \
                 : NAME##_CLOBBERS)
    if constexpr (!TRACE) {
        FA_RUN(FA_PSEL4, FA_DN0(rq[0]), FA_DN1(rq[1]), FA_DN2(rq[2]), FA_DN3(rq[3]));
    } else {
        static_assert(K >= 1 && K <= 3, "one combined SEL loop per ray count 1..3");
        if constexpr (K == 1) FA_RUN(FA_SEL1, FA_DN0(rq[0]), FA_LANE(lane));
        if constexpr (K == 2) FA_RUN(FA_SEL2, FA_DN0(rq[0]), FA_DN1(rq[1]), FA_LANE(lane));
        if constexpr (K == 3) FA_RUN(FA_SEL3, FA_DN0(rq[0]), FA_DN1(rq[1]), FA_DN2(rq[2]), FA_LANE(lane));
    }
#undef FA_RUN
}

// RT_PHASE_CLOCKS (profiling builds only): the wave's clock ticks per phase of a tile replace
// the work counters in DevCounters, word p = phase p: 0 claim, 1 record setup, 2 GROUP loops,
// 3 bitmap, 4 SEL loops, 5 exact pass, 6 planes and triangles, 7 results and record push.
#ifdef RT_PHASE_CLOCKS
struct PhaseClock {
    unsigned long long t_last, clk[9];
};
#define PHASE_MARK(tcref, p)                                            \
    {                                                                   \
        const unsigned long long now_ = __builtin_readcyclecounter();   \
        (tcref).clk[p] += now_ - (tcref).t_last;                        \
        (tcref).t_last = now_;                                          \
    }
#else
struct PhaseClock {};
#define PHASE_MARK(tcref, p)
#endif

// What the filter loops of a walk ran, in entries (spheres or balls) per lane.
struct FilterWork {
    int candidates;   // calls of the exact evaluation (this lane's)
    int terms;        // per-origin terms formed (wave-uniform)
    int ray_tests;    // (ray, entry) pairs evaluated (wave-uniform)
};

template <int N>
struct RayIndex {
    static constexpr int value = N;
};

// The sphere trips a wave-uniform bitmap names (bit b of wlo = trip b ^ 7, whi 64 trips later),
// for the rays rq[0..K-1], then the exact evaluation of every
// lane's queued candidates: exact(sphere {centre, r*r}, host index), EXACT_BATCH candidates per
// lane at a time with their data fetched together.  Returns the trips run.
template <int K, bool TRACE, class Exact>
__device__ __forceinline__ int run_selected(WalkLds &L, const float4 *sph, const int *orig,
                                            const f4raw (&rq)[4], const f4raw lane, unsigned long long wlo,
                                            unsigned long long whi, int &candidates, PhaseClock &pc, Exact &&exact)
{
    const int *cand = L.cand;
    const unsigned q0 = lds_offset(L.cand) + 4u * threadIdx.x;
    const unsigned qlim = lds_offset(L.cand) + QUEUE_SLOTS * FA_QUEUE_STRIDE;
    int trip0 = 0, trips = 0;
#pragma clang loop unroll(disable)
    while ((wlo | whi) != 0ull) {
        unsigned qaddr = q0;
        trips += __popcll(wlo) + __popcll(whi);
        fa_sel<K, TRACE>(wlo, whi, trip0, qaddr, rq, lane, lds_offset(L.sph), qlim);
        trips -= __popcll(wlo) + __popcll(whi);   // what a full queue left for the next call
        PHASE_MARK(pc, 4)
        CandidateWalk walk(cand, (int)((qaddr - q0) / FA_QUEUE_STRIDE));
        int li[EXACT_BATCH];
#pragma unroll
        for (int k = 0; k < EXACT_BATCH; k++) li[k] = k == 0 || li[k - 1] >= 0 ? walk.next() : -1;
#pragma clang loop unroll(disable)
        while (__ballot(li[0] >= 0) != 0ull) {
            float4 sp[EXACT_BATCH];
            int og[EXACT_BATCH];
#pragma unroll
            for (int k = 0; k < EXACT_BATCH; k++) {
                const int i = li[k] >= 0 ? li[k] : 0;
                sp[k] = sph[i];
                og[k] = orig[i];
            }
#pragma unroll
            for (int k = 0; k < EXACT_BATCH; k++) {
                if (li[k] >= 0) {
                    candidates++;
                    exact(sp[k], og[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < EXACT_BATCH; k++)
                li[k] = (k == 0 ? li[EXACT_BATCH - 1] : li[k - 1]) >= 0 ? walk.next() : -1;
        }
        PHASE_MARK(pc, 5)
    }
    return trips;
}

// The wave-wide pass bits of the first `ntrips` trips a GROUP loop left in the queue slots
// (bit 8t + b of word w = entry 8(4w + t) + 7 - b of the array the loop walked), one bitmap
// per ray when the loop kept the rays apart (NB = K), else one for all (NB = 1).
// live[k] = 0 takes ray k of this lane out of the vote (its answer is already known), ~0 keeps it.
template <int NB, int NW>
__device__ __forceinline__ void gather_pass_bits(const int *cand, const int ntrips, unsigned (&words)[NB][NW],
                                                 const unsigned (&live)[4])
{
    // One mask for all rays (NB = 1): the lane votes while any of its rays is live.
    const unsigned all = live[0] | live[1] | live[2] | live[3];
#pragma unroll
    for (int wd = 0; wd < NW; wd++) {
        unsigned bits[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) bits[k] = 0u;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int trip = 4 * wd + t;
            if (trip < ntrips) {
                const unsigned w = ~(unsigned)cand[trip * 256 + threadIdx.x];
#pragma unroll
                for (int k = 0; k < NB; k++) bits[k] |= ((w >> (8 * k)) & 0xffu) << (8 * t);
            }
        }
#pragma unroll
        for (int k = 0; k < NB; k++) words[k][wd] = wave_or(bits[k] & (NB == 1 ? all : live[k]));
    }
}

// One staged chunk of `cnt` spheres (a multiple of RT_SPH_PAD) for K rays per lane, three levels
// of conservative tests, each deciding per WAVE what the next one looks at:
//   1. a GROUP loop over the balls around runs of RT_SUPER spheres (at most two trips);
//   2. GROUP-loop trips over the groups' balls, only for the runs some lane's ray may hit (one
//      trip = the FA_TRIP groups of one run);
//   3. SEL loops over the spheres of the groups some lane may hit, which queue candidates, and
//      the exact evaluation of each lane's queue.
//   SPLIT false: one bitmap for the K rays of a lane; exact(RayIndex<-1>, sphere, host index)
//     evaluates all of them.  Right when the rays are close to each other: the four pixels of
//     a primary lane, the two or three rays of a record in a scene with two lights.
//   SPLIT true (trace rays): with more lights the rays of a record point in unrelated
//     directions, so at level 3 every ray k has its own bitmap and runs only the groups IT may
//     hit, one ray per SEL loop; exact(RayIndex<k>, sphere, host index) evaluates ray k.
// run_mask: the runs of RT_SUPER spheres this call is to look at, in the bit order of level 1's
// result (bit 8t + 7 - m = run 8t + m); the spheres of the other runs are left to other callers.
// lane_s / lane_g: the per-origin operands of the two ball levels (trace rays).  The sphere's
// {centre, r*r} and host primitive index come from sph / orig (the chunk's part of
// DevScene::sph / sph_orig).  The order of the calls is not the spheres' host order.  Every lane
// of the wave must be active.
template <int K, bool TRACE, bool SPLIT, class Exact>
__device__ __forceinline__ void walk_chunk(WalkLds &L, const int cnt, const float4 *sph, const int *orig,
                                           const f4raw (&rq)[4], const f4raw lane_s, const f4raw lane_g,
                                           const f4raw lane, FilterWork &fw, PhaseClock &pc, Exact &&exact,
                                           const unsigned (&live)[4], const unsigned run_mask = ~0u)
{
    static_assert(TRACE || !SPLIT, "primary rays use the combined loops");
    constexpr bool split = SPLIT && K > 1;
    constexpr int NB = split ? K : 1;
    const int ngroups = cnt / RT_GROUP, ntrips = ngroups / FA_TRIP;
    const unsigned q0 = lds_offset(L.cand) + 4u * threadIdx.x;
    PHASE_MARK(pc, 1)
    // Level 1: which runs of RT_SUPER spheres (= which trips of level 2) can any ray of the wave hit?
    unsigned runs;
    {
        const int nsup = (ntrips + FA_TRIP - 1) / FA_TRIP * FA_TRIP;   // staged padded to whole trips
        int i = 0;
        unsigned addr = lds_offset(L.sup), qaddr = q0;
        fa_group<K, TRACE, split>(i, addr, qaddr, rq, lane_s, nsup);
        fw.terms += TRACE ? nsup : 0;
        fw.ray_tests += nsup * K;
        unsigned any[NB][1];
        gather_pass_bits<NB, 1>(L.cand, nsup / FA_TRIP, any, live);
        runs = 0u;
#pragma unroll
        for (int k = 0; k < NB; k++) runs |= any[k][0];   // bit 8t + 7 - m = run 8t + m
        runs &= run_mask;   // (a wave that shares the chunk with others walks only its own runs)
    }
    // Level 2: the groups' balls of those runs; the slots of the other trips read "all rejected".
#pragma unroll
    for (int j = 0; j < SUP_CHUNK; j++)
        if (j < ntrips) L.cand[j * 256 + threadIdx.x] = -1;
    if (runs != 0u) {
        fa_gsel<K, TRACE, split>((unsigned long long)runs, q0, rq, lane_g, lds_offset(L.grp));
        const int ran = __popc(runs);
        fw.terms += TRACE ? ran * FA_TRIP : 0;
        fw.ray_tests += ran * FA_TRIP * K;
    }
    PHASE_MARK(pc, 2)
    static_assert(GRP_WORDS == 4, "the bitmap of a chunk is two 64-bit halves");
    unsigned words[NB][GRP_WORDS];
    gather_pass_bits<NB, GRP_WORDS>(L.cand, ntrips, words, live);
    PHASE_MARK(pc, 3)
    // Level 3.
    if constexpr (!split) {
        const int trips = run_selected<K, TRACE>(L, sph, orig, rq, lane,
                                                 (unsigned long long)words[0][0] | ((unsigned long long)words[0][1] << 32),
                                                 (unsigned long long)words[0][2] | ((unsigned long long)words[0][3] << 32),
                                                 fw.candidates, pc,
                                                 [&](const float4 sp, const int og) { exact(RayIndex<-1>(), sp, og); });
        fw.terms += TRACE ? trips * FA_TRIP : 0;
        fw.ray_tests += trips * FA_TRIP * K;
    } else {
        int trips = 0;
        auto one_ray = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k < K) {
                const f4raw one[4] = {rq[k], rq[k], rq[k], rq[k]};
                trips += run_selected<1, true>(L, sph, orig, one, lane,
                                               (unsigned long long)words[k][0] | ((unsigned long long)words[k][1] << 32),
                                               (unsigned long long)words[k][2] | ((unsigned long long)words[k][3] << 32),
                                               fw.candidates, pc,
                                               [&](const float4 sp, const int og) { exact(kc, sp, og); });
            }
        };
        one_ray(RayIndex<0>());
        one_ray(RayIndex<1>());
        one_ray(RayIndex<2>());
        one_ray(RayIndex<3>());
        fw.terms += trips * FA_TRIP;
        fw.ray_tests += trips * FA_TRIP;
    }
}

// Can the ray (o, d) meet the ball {P, R} (DevScene::chk_b)?  By "Bounding balls" above the
// forward half-line of a ray that hits a member comes within R' = R + 1.31e-3 (|o - P| + R) of P,
// so the ray misses when o is outside and either P is behind it or farther than R' from its
// line.  FP32 evaluation with room (1e-5 |v| on d.v, 1e-4 on R'^2); anything not finite meets.
__device__ __forceinline__ bool ray_meets_ball(const f3 o, const f3 d, const float4 b)
{
    const f3 v = mk3(b.x - o.x, b.y - o.y, b.z - o.z);
    const float D2 = __builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x));
    const float D = __builtin_amdgcn_sqrtf(D2);
    const float R = __builtin_fmaf(1.5e-3f, D + b.w, b.w);
    const float R2 = R * R * 1.0001f + 1e-5f * D2;
    const float c = __builtin_fmaf(d.z, v.z, __builtin_fmaf(d.y, v.y, d.x * v.x));
    const bool outside = D2 > R2 * 1.002f;
    const bool miss = c < -1e-5f * D || __builtin_fmaf(-c, c, D2) > R2;
    return !(outside && miss);
}

// ... and do so at a distance of at most `reach` (every point of the widened ball is at least
// |o - P| - R' from o; the hit the reference computes lies within it)?
__device__ __forceinline__ bool ray_reaches_ball(const f3 o, const f3 d, const float4 b, const float reach)
{
    const f3 v = mk3(b.x - o.x, b.y - o.y, b.z - o.z);
    const float D = __builtin_amdgcn_sqrtf(__builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x)));
    const float R = __builtin_fmaf(1.5e-3f, D + b.w, b.w);
    return !((D - R) * 0.999f > reach) && ray_meets_ball(o, d, b);
}

// Can the ray still gain from chunk `chunk`, given a hit at distance best_t: does it reach the
// chunk's ball and, inside it, the ball of any of its runs of RT_SUPER spheres?  Asked before a
// workgroup stages and walks a chunk: the chunks' balls overlap their neighbours' by far (1024
// spheres in a cube give a ball of nearly the cube's diagonal), so a ray "meets" many chunks it
// has no business in, and a visit costs 30-40 us.
__device__ __forceinline__ bool chunk_may_hit(const DevScene &s, const int chunk, const f3 o, const f3 d,
                                              const float best_t)
{
    const float4 cb = cload(s.chk_b, chunk);
    if (!(cb.w >= 0.0f) || !ray_reaches_ball(o, d, cb, best_t)) return false;
    const int nsup = s.ns_padded / RT_SUPER, s0 = chunk * (RT_CHUNK / RT_SUPER);
    const int s1 = min(s0 + RT_CHUNK / RT_SUPER, nsup);
    for (int si = s0; si < s1; si++) {
        const float4 sb = cload(s.sup_b, si);
        if (sb.w >= 0.0f && ray_reaches_ball(o, d, sb, best_t)) return true;
    }
    return false;
}

// The chunks of a scene with more than one: `visit(base, cnt)` is called, by the whole
// workgroup, for every chunk whose ball some ray of some wave may hit (the chunks' balls are
// tested GRP_CHUNK at a time with a GROUP loop, and the waves' pass bits are ORed in LDS because
// staging a chunk is the workgroup's business).  Every thread of the workgroup must call it.
// need(chunk) (per lane): can this lane still gain from the chunk?  A chunk nobody needs (any more)
// is neither staged nor walked.
template <int K, bool TRACE, bool SPLIT, class Need, class Visit>
__device__ __forceinline__ void for_each_chunk(WalkLds &L, const DevScene &s, const f4raw (&rq)[4], const f4raw lane_c,
                                               const f3 oc, FilterWork &fw, const unsigned (&live)[4], Need &&need,
                                               Visit &&visit)
{
    constexpr bool split = SPLIT && K > 1;
    constexpr int NB = split ? K : 1;
    const int nsp = s.ns_padded, nchk_padded = s.nchk_padded;
    for (int first = 0; first < nchk_padded; first += GRP_CHUNK) {
        const int n = min(GRP_CHUNK, nchk_padded - first);
        __syncthreads();   // whoever still reads the staged arrays is done
        stage_chunk_balls(L, s, first, n);
        __syncthreads();
        int i = 0;
        unsigned addr = lds_offset(L.grp), qaddr = lds_offset(L.cand) + 4u * threadIdx.x;
        fa_group<K, TRACE, split>(i, addr, qaddr, rq, lane_c, n);
        fw.terms += TRACE ? n : 0;
        fw.ray_tests += n * K;
        unsigned words[NB][GRP_WORDS];
        gather_pass_bits<NB, GRP_WORDS>(L.cand, n / FA_TRIP, words, live);
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int wd = 0; wd < GRP_WORDS; wd++) {
                unsigned all = 0u;
#pragma unroll
                for (int k = 0; k < NB; k++) all |= words[k][wd];
                if (all) atomicOr(&L.wg_bits[wd], all);
            }
        }
        if (threadIdx.x == 0) L.ref[0] = oc.x, L.ref[1] = oc.y, L.ref[2] = oc.z;
        __syncthreads();
        // The chunks to walk, nearest first (the rays of a tile start close together): a light
        // ray that meets an occluder early stops voting early (`live`), and the order cannot
        // change a result.  Thread c ranks chunk c among the selected ones by the distance of its
        // ball's centre from thread 0's origin.
        const int c = threadIdx.x;
        bool selected = false;
        float key = INFINITY;
        if (c < n) {
            // bit 8t + 7 - m of word w = entry 8(4w + t) + m
            selected = (L.wg_bits[c >> 5] >> (((c & 31) & ~7) + 7 - (c & 7))) & 1u;
            selected = selected && (first + c) * SPH_CHUNK < nsp;
            const float4 e = L.grp[c];
            const float dx = e.x - L.ref[0], dy = e.y - L.ref[1], dz = e.z - L.ref[2];
            if (selected) key = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            if (key != key) key = 0.0f;   // an unrepresentable ball passes always: walk it first
            L.key[c] = key;
        }
        __syncthreads();
        if (selected) {
            int rank = 0;
            for (int u = 0; u < n; u++) {
                const float ku = L.key[u];
                rank += (ku < key || (ku == key && u < c)) ? 1 : 0;
            }
            L.order[rank] = c;
        }
        int count = 0;
#pragma unroll
        for (int wd = 0; wd < GRP_WORDS; wd++) count += __popc(__builtin_amdgcn_readfirstlane((int)L.wg_bits[wd]));
        __syncthreads();
        // (selected chunks beyond the last sphere do not exist: only padding entries of the ball
        // array, which never pass, lie there, so `count` is the number of ranks handed out.)
#pragma clang loop unroll(disable)
        for (int i = 0; i < count; i++) {
            const int chunk = first + __builtin_amdgcn_readfirstlane(L.order[i]);   // workgroup-uniform
            const int base = chunk * SPH_CHUNK;
            if (!__syncthreads_or(need(chunk) ? 1 : 0)) continue;
            visit(base, min(SPH_CHUNK, nsp - base));
        }
    }
}

// GetRayDirection's per-pixel part, MyApplication.cs:438-449.
__device__ __forceinline__ f3 primary_direction(const DevCamera &cam, int px, int py)
{
    float sx = (ieee_div((float)px, (float)cam.width) * 2.0f) - 1.0f;
    float sy = (ieee_div((float)py, (float)cam.height) * 2.0f) - 1.0f;
    sx *= cam.tan_h;
    sy *= cam.tan_v;
    const f3 cdir = mk3(cam.dir[0], cam.dir[1], cam.dir[2]);
    const f3 cup = mk3(cam.up[0], cam.up[1], cam.up[2]);
    const f3 cleft = mk3(cam.left[0], cam.left[1], cam.left[2]);
    return normalized3(add3(add3(cdir, scale3(cup, sy)), scale3(cleft, sx)));
}

// ---------------------------------------------------------------------------------------
// Cone culling for primary rays.
//
// The 256 rays of a wave (a 16x16 pixel tile) leave one point o in nearly one direction.  Let a
// be one of them and beta >= the angle between a and any of the others (measured on the rays
// themselves: beta = 2 asin(chord / 2) of the longest chord |d_k - a|, widened).  If the
// reference accepts (ray, sphere i) then, with D = |o - p_i| (see "Bounding balls" above): the
// distance of p_i from the ray's line is at most R_i = sqrt(r_i^2 + 28.6u D^2) and the centre is
// at most 4.1u D behind the origin; so either D <= R_i (to within 1e-6: the origin is inside) or
// the angle between d and p_i - o is at most asin(R_i / D), and then the angle between a and
// p_i - o is at most beta + asin(R_i / D).  The same holds for a ball (P, R) around spheres with
// |p_i - P| + r_i <= R and the radius R' = R + 1.31e-3 (|o - P| + R) >= R + (R_i - r_i): the
// ray's forward half-line comes within R' of P.  Hence a sphere or ball whose centre direction
// v = P - o has  a.v < cos(beta) sqrt(|v|^2 - R'^2) - sin(beta) R'  (and |v| > R') is met by no
// ray of the wave.  One LANE tests one ball or sphere: 64 of them per step, against the few
// instructions of the test, instead of every lane testing every ball with its own rays; the
// spheres that pass (a handful per tile) go to the exact, literal Sphere.Intersect of all 256
// rays.  Evaluation margins: the test is made in FP32 on the scene's own coordinates, so v is an
// exactly rounded difference; 1e-5 |v| on the right-hand side, 1e-5 rad on beta and 0.1 % on
// "inside" are far above the sum of the roundings (about 20u |v|).  Anything not
// finite passes; a sphere with non-finite data or r*r < 0 (padding) is never hit by the reference
// and is skipped.  A ball or sphere every point of which is farther away than the wave's farthest
// closest hit so far cannot change any ray's result either.
// ---------------------------------------------------------------------------------------
struct TileCone {
    f3 a;
    float cosb, sinb;   // bounds: cosb <= cos(beta), sinb >= sin(beta)
    float chord;        // >= |d_k - a| for every ray of the wave
    bool open;          // no usable cone (degenerate camera): everything passes
};

__device__ __forceinline__ float wave_max_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// v = centre - origin, R = the (already widened) radius; far = every ray of the wave has a hit
// at t <= far.
__device__ __forceinline__ bool cone_meets(const TileCone &tc, const f3 v, const float R, const float far)
{
    const float D2 = __builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x));
    const float D = __builtin_amdgcn_sqrtf(D2);
    const float h2 = __builtin_fmaf(-R, R, D2);
    const bool inside = !(h2 > 2e-3f * D2);                      // D <= 1.001 R, or not finite
    const float c = __builtin_fmaf(tc.a.z, v.z, __builtin_fmaf(tc.a.y, v.y, tc.a.x * v.x));
    const float bound = (tc.cosb * __builtin_amdgcn_sqrtf(h2) - tc.sinb * R) - 1e-5f * D;
    const bool miss = c < bound;                                  // false when anything is NaN
    const bool beyond = (D - R) * 0.999f > far;
    return (tc.open || inside || !miss) && !beyond;
}

// A bounding ball {P, R}: widened by 1.5e-3 (|o - P| + R), see above.
__device__ __forceinline__ bool ball_in_cone(const TileCone &tc, const float4 b, const f3 o, const float far)
{
    const f3 v = mk3(b.x - o.x, b.y - o.y, b.z - o.z);
    const float D = __builtin_amdgcn_sqrtf(__builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x)));
    return cone_meets(tc, v, __builtin_fmaf(1.5e-3f, D + b.w, b.w), far);
}

// ---------------------------------------------------------------------------------------
// primary: Raytracer.Render's loop body up to scene.Intersect (MyApplication.cs:412-415)
// plus GetRayDirection (:428-452).  Each lane owns a 2x2 pixel quad, a wave 16x16 pixels, a
// workgroup 32x32.  The wave finds the spheres its tile's cone can meet by walking the
// bounding balls one lane per ball (chunks, then runs of RT_SUPER spheres, then the 64 spheres
// of a run), and every lane evaluates those spheres exactly for its four rays.
// ---------------------------------------------------------------------------------------
#define PRIM_PX 4   // pixels per lane (2x2)

#ifndef RT_PRIMARY_WAVES
#define RT_PRIMARY_WAVES 4   // (with workgroups of eight waves: 5 spills, 6 is slower)
#endif
// A workgroup of RT_PRIMARY_THREADS / 64 waves takes a tile of PRIM_TILE_W x 32 pixels per trip (a wave: 16 x 16).
// Eight waves: a trip costs two contended returning atomics (the tile claim, the records' slots), and at 3840x2160
// their rate on one address bounds the kernel: 0.123 ms with four waves per workgroup (8100 trips), 0.095 with
// eight, 0.103 with sixteen (the barrier then waits for the slowest of sixteen tiles).
#ifndef RT_PRIMARY_THREADS
#define RT_PRIMARY_THREADS 512
#endif
#define PRIM_WAVES_X (RT_PRIMARY_THREADS / 128)
#define PRIM_TILE_W (16 * PRIM_WAVES_X)
#define PRIM_CLAIM (RT_PRIMARY_THREADS / 32 + 2)   // blk words of the tile claims, behind push_records' own
__global__ void __launch_bounds__(RT_PRIMARY_THREADS) __attribute__((amdgpu_waves_per_eu(RT_PRIMARY_WAVES)))
k_primary(DevScene s, DevCamera cam, DevFrame f, int32_t *pixels)
{
    __shared__ int blk[PRIM_CLAIM + 2];
    __shared__ unsigned long long cnt_sh[COUNTER_WORDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rows = cam.rows;
    const int tiles_x = (cam.width + PRIM_TILE_W - 1) / PRIM_TILE_W, tiles_y = (rows + 31) >> 5;
    const int ntiles = tiles_x * tiles_y;
    const int nsp = s.ns_padded;
    const int nchk = (nsp + RT_CHUNK - 1) / RT_CHUNK, nsup = nsp / RT_SUPER;
    const f3 o = mk3(cam.pos[0], cam.pos[1], cam.pos[2]);
    int cone_tests = 0, candidates = 0;   // steps of 64 cone tests (wave-uniform); exact calls (this lane's)
    // A scene of one chunk: its 16 balls stay in registers (lane = ball) for every tile.
    const float4 no_ball = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
    const float4 only_chunk = nchk == 1 && lane < RT_CHUNK / RT_SUPER && lane < nsup ? s.sup_b[lane] : no_ball;
    // A workgroup's first tile is its index; the others are claimed from a counter (tiles cost
    // very different amounts), each claim sent before the tile it follows so that its answer
    // travels while that tile is traced.  blk[PRIM_CLAIM], blk[PRIM_CLAIM + 1]: the answer, alternating per trip.
    int trip = 0;
#ifdef RT_PRIMARY_CLOCKS
    unsigned long long pclk[8] = {}, ptl = __builtin_readcyclecounter();
#define PMARK(p) { const unsigned long long now_ = __builtin_readcyclecounter(); pclk[p] += now_ - ptl; ptl = now_; }
#else
#define PMARK(p)
#endif
    for (int tile = blockIdx.x; tile < ntiles; trip++) {
        int claimed = 0;
        if (threadIdx.x == 0) claimed = (int)gridDim.x + atomicAdd(&f.sched->next_primary, 1);
        const int tx = tile % tiles_x, ty = tile / tiles_x;
        const int x0 = tx * PRIM_TILE_W + (wave % PRIM_WAVES_X) * 16 + (lane & 7) * 2;
        const int y0 = ty * 32 + (wave / PRIM_WAVES_X) * 16 + (lane >> 3) * 2;   // row of the output block
        bool inside[PRIM_PX];
        int px[PRIM_PX], py[PRIM_PX], ly[PRIM_PX];
        f3 d[PRIM_PX];
        float a[PRIM_PX], best_t[PRIM_PX];
        int best_prim[PRIM_PX];   // host index of the closest primitive so far
#pragma unroll
        for (int k = 0; k < PRIM_PX; k++) {
            const int x = x0 + (k & 1), y = y0 + (k >> 1);
            inside[k] = x < cam.width && y < rows;
            // Pixels outside the block trace the block's last pixel and write nothing.
            px[k] = x < cam.width ? x : cam.width - 1;
            ly[k] = y < rows ? y : rows - 1;
            py[k] = cam.row_begin + ((ly[k] / cam.band) * cam.parts + cam.part) * cam.band + ly[k] % cam.band;
            d[k] = primary_direction(cam, px[k], py[k]);
            a[k] = dot3(d[k], d[k]);
            best_t[k] = RT_FLT_MAX;
            best_prim[k] = INT_MAX;
        }
        PMARK(0)
        // The tile's cone: axis = the ray next to the tile's centre, beta from the longest chord.
        TileCone tc;
        tc.a = mk3(__shfl(d[0].x, 36), __shfl(d[0].y, 36), __shfl(d[0].z, 36));
        {
            float ch2 = 0.0f;
#pragma unroll
            for (int k = 0; k < PRIM_PX; k++) {
                const f3 e = sub3(d[k], tc.a);
                const float q = __builtin_fmaf(e.z, e.z, __builtin_fmaf(e.y, e.y, e.x * e.x));
                ch2 = q > ch2 || q != q ? q : ch2;   // (a NaN chord sticks)
            }
            float chord = __builtin_amdgcn_sqrtf(ch2);
            const unsigned long long bad = __ballot(chord != chord);
            chord = wave_max_f(chord);
            tc.chord = chord;
            // 2 asin(c / 2) <= c (1 + c*c / 6) for c <= 1.4
            const float beta = chord * __builtin_fmaf(chord * chord, 1.0f / 6.0f, 1.0f) * 1.00001f + 1e-5f;
            tc.open = bad != 0ull || !(beta <= 1.0f);
            tc.sinb = beta;                                        // sin(beta) <= beta
            tc.cosb = __builtin_fmaf(-0.5f * beta, beta, 1.0f) - 1e-7f;   // cos(beta) >= 1 - beta*beta / 2
        }
        PMARK(1)
        // Exact evaluation of a candidate: objects.cs:112-121 literally; among equal
        // distances the lower host index wins, as the strict `<` of MyApplication.cs:361
        // decides when the primitives are scanned in host order.
        auto exact = [&](const float4 sp, const int orig) {
            const f3 op = sub3(o, xyz(sp));
            const float c = dot3(op, op) - sp.w;
#pragma unroll
            for (int k = 0; k < PRIM_PX; k++) {
                const float xx = dot3(d[k], op);
                float t;
                if (sphere_maybe(xx, a[k] * c) && sphere_root(xx, c, a[k], t) && t > RT_EPS &&
                    (t < best_t[k] || (t == best_t[k] && orig < best_prim[k]))) {
                    best_t[k] = t;
                    best_prim[k] = orig;
                }
            }
        };
        float far = RT_FLT_MAX;   // every ray of the wave has its closest hit at t <= far
#pragma clang loop unroll(disable)
        for (int c0 = 0; c0 < nchk; c0 += 64) {
            unsigned long long cm = 1ull;
            if (nchk > 1) {
                bool cpass = false;
                if (c0 + lane < nchk) {
                    const float4 b = s.chk_b[c0 + lane];
                    cpass = b.w >= 0.0f && ball_in_cone(tc, b, o, far);
                }
                cone_tests++;
                cm = __ballot(cpass);
            }
#pragma clang loop unroll(disable)
            while (cm != 0ull) {
                const int chunk = c0 + (__ffsll((long long)cm) - 1);
                cm &= cm - 1ull;
                // the chunk's runs of RT_SUPER spheres: RT_CHUNK / RT_SUPER = 16 balls, lanes 0..15
                const int si = chunk * (RT_CHUNK / RT_SUPER) + lane;
                float4 b = only_chunk;
                if (nchk > 1 && lane < RT_CHUNK / RT_SUPER && si < nsup) b = s.sup_b[si];
                const bool spass = b.w >= 0.0f && ball_in_cone(tc, b, o, far);
                cone_tests++;
                unsigned long long sm = __ballot(spass);
                PMARK(2)
#pragma clang loop unroll(disable)
                while (sm != 0ull) {
                    const int run = chunk * (RT_CHUNK / RT_SUPER) + (__ffsll((long long)sm) - 1);
                    sm &= sm - 1ull;
                    const int i = run * RT_SUPER + lane;   // (< ns_padded: a whole number of runs)
                    const float4 sp = s.sph[i];
                    const int orig = s.sph_orig[i];
                    const f3 v = sub3(xyz(sp), o);
                    const float D2 = __builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x));
                    // R_i^2 = r*r + 28.6u D^2, with room: 4e-6 D^2
                    const float R = __builtin_amdgcn_sqrtf(__builtin_fmaf(4e-6f, D2, sp.w));
                    const bool hpass = sp.w >= 0.0f && sp.w < INFINITY && cone_meets(tc, v, R, far);
                    cone_tests++;
                    unsigned long long hm = __ballot(hpass);
                    PMARK(3)
#pragma clang loop unroll(disable)
                    while (hm != 0ull) {
                        const int j = __ffsll((long long)hm) - 1;
                        hm &= hm - 1ull;
                        exact(make_float4(__shfl(sp.x, j), __shfl(sp.y, j), __shfl(sp.z, j), __shfl(sp.w, j)), __shfl(orig, j));
                        candidates++;
                    }
                    PMARK(4)
                }
                if (nchk > 1) far = wave_max_f(fmaxf(fmaxf(best_t[0], best_t[1]), fmaxf(best_t[2], best_t[3])));
            }
        }
        // Planes (Plane.Intersect objects.cs:157-164).  t = numer / denom decides only when it can
        // still win: the quotient from the hardware reciprocal is within 1e-6 (relative) of the
        // correctly rounded one, so a plane whose quotient is clearly behind the epsilon or clearly
        // beyond the closest hit so far needs no divide.
        // A plane that faces away from every ray of the tile is skipped for the whole wave: with
        // c = n.a and |n.(d_k - a)| <= |n|_1 chord, all denominators n.d_k (and their FP32 values,
        // off by at most 3u |n|_1) share the sign of c when |c| > |n|_1 (chord + 2e-6), and a
        // quotient of opposite signs (or a zero numerator) is <= 0, which Plane.Intersect rejects.
        for (int j = 0; j < s.np; j++) {
            const float4 pl = cload(s.pl, j);
            const f3 n = xyz(pl);
            const float numer = (-dot3(n, o)) - pl.w;
            const int orig = cload(s.pl_orig, j);
            {
                const float c = __builtin_fmaf(n.z, tc.a.z, __builtin_fmaf(n.y, tc.a.y, n.x * tc.a.x));
                const float spread = ((fabsf(n.x) + fabsf(n.y)) + fabsf(n.z)) * (tc.chord * 1.001f + 2e-6f);
                if (!tc.open && ((c > spread && numer <= 0.0f) || (c < -spread && numer >= 0.0f))) continue;
            }
#pragma unroll
            for (int k = 0; k < PRIM_PX; k++) {
                const float denom = dot3(n, d[k]);
                if (fabsf(denom) <= 0.0001f) continue;
                const float q = numer * __builtin_amdgcn_rcpf(denom);
                if (q < RT_EPS * (1.0f - 1e-6f) || q * (1.0f - 1e-6f) > best_t[k]) continue;
                const float t = ieee_div(numer, denom);
                if (!(t <= 0.0001f) && t > RT_EPS && (t < best_t[k] || (t == best_t[k] && orig < best_prim[k]))) {
                    best_t[k] = t;
                    best_prim[k] = orig;
                }
            }
        }
        for (int j = 0; j < s.nt; j++) {
            const f3 v0 = xyz(cload(s.tri, 3 * j)), e1 = xyz(cload(s.tri, 3 * j + 1)), e2 = xyz(cload(s.tri, 3 * j + 2));
            const int orig = cload(s.tri_orig, j);
            const f3 sv = sub3(o, v0);
#pragma unroll
            for (int k = 0; k < PRIM_PX; k++) {
                float t;
                if (triangle_t(e1, e2, sv, d[k], t) && t > RT_EPS &&
                    (t < best_t[k] || (t == best_t[k] && orig < best_prim[k]))) {
                    best_t[k] = t;
                    best_prim[k] = orig;
                }
            }
        }
        bool rec_valid[PRIM_PX];
        PMARK(5)
        int rec_prim[PRIM_PX], rec_parent[PRIM_PX];
        f3 rec_hit[PRIM_PX];
#pragma unroll
        for (int k = 0; k < PRIM_PX; k++) {
            const bool hit = inside[k] && best_prim[k] != INT_MAX;
            const int pix = ly[k] * cam.width + px[k];
            rec_valid[k] = hit;
            rec_prim[k] = hit ? best_prim[k] : 0;
            rec_parent[k] = pix;
            rec_hit[k] = add3(o, scale3(d[k], best_t[k]));
            if (inside[k] && !hit) pixels[pix] = 0;  // MyApplication.cs:423
        }
        if (threadIdx.x == 0) blk[PRIM_CLAIM + (trip & 1)] = claimed;
        push_records<PRIM_PX, false>(f, s, 0, blk, rec_valid, rec_prim, rec_hit, d, rec_parent);   // (barriers)
        tile = blk[PRIM_CLAIM + (trip & 1)];
        PMARK(6)
    }
    // Counter words in the order of DevCounters: cone tests count as filter evaluations.
    const unsigned long long cand_total = (unsigned long long)wave_sum(candidates);
    const unsigned long long tests = 64ull * (unsigned long long)cone_tests;
#ifdef RT_PRIMARY_CLOCKS
    const unsigned long long v[COUNTER_WORDS] = {pclk[0], pclk[1], pclk[2], pclk[3], pclk[4], pclk[5], pclk[6], cand_total, tests};
    flush_counters(f, cnt_sh, v);
#else
    const unsigned long long v[COUNTER_WORDS] = {0ull, 0ull, 0ull, cand_total, tests, 0ull, tests, cand_total, 0ull};
#ifndef RT_PHASE_CLOCKS
    flush_counters(f, cnt_sh, v);
#endif
#endif
#undef PMARK
}

// ---------------------------------------------------------------------------------------
// trace_level: for each hit record, for specular surfaces below the depth bound,
// scene.Intersect along the mirror direction (objects.cs:82-83) — a walk of the scene through
// the filter loops — and IsInShadow towards every light (objects.cs:51-60) through the lights'
// direction tables (light_occluder below, called while the record is shaded).
// ---------------------------------------------------------------------------------------
// Per-lane work counters of a kernel.
struct TestCount {
    int sph, pl, tri;
    int exact;        // (record, sphere) pairs this lane evaluated exactly
    int filt_terms;   // spheres and balls this lane formed the filter's per-origin term for
    int filt_rays;    // (ray, sphere or ball) pairs this lane's filter evaluated
    int shadow;       // (light ray, sphere) pairs this lane evaluated exactly from the lights' tables
    int grid;         // cells this lane's grid walks visited
};

template <int LC, bool SEC>
struct Rays {
    f3 sh_d[LC > 0 ? LC : 1];
    float sh_a[LC > 0 ? LC : 1];
    float sh_dist[LC > 0 ? LC : 1];
    int sh_first[LC > 0 ? LC : 1];  // lowest host index among occluders, INT_MAX if none
    f3 sec_d;
    float sec_a;
    Closest sec;
};

template <int LC, bool SEC, bool SPLIT, bool MULTI>
__device__ __forceinline__ void traverse(const DevScene &s, WalkLds &L,
                                         const f3 o, Rays<LC, SEC> &r, TestCount &tc, PhaseClock &pc,
                                         const bool any_hit)
{
    constexpr int NR = LC + (SEC ? 1 : 0);
    static_assert(NR >= 1 && NR <= 4, "one GROUP loop per ray count 1..4");
    // Ray operands of the filter loop: light rays first, the reflection ray last.
    const FilterOrigin fo = filter_origin(s, o);
    f4raw rq[4];
#pragma unroll
    for (int l = 0; l < 4; l++) rq[l] = f4raw{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < LC; l++) rq[l] = filter_ray(fo, r.sh_d[l]);
    if (SEC) rq[LC] = filter_ray(fo, r.sec_d);

    float best_t = RT_FLT_MAX;
    int best_prim = INT_MAX;   // host index of the closest sphere so far
    FilterWork fw = {0, 0, 0};
    const int nsp = s.ns_padded;
    // Exact evaluation of a candidate (Sphere.Intersect objects.cs:112-121, literally) for ray
    // k of the record (lights first, then the reflection ray), or for all rays (k = -1).
    // Closest hit: among equal distances the lower host index wins (MyApplication.cs:361);
    // shadow rays: the lowest host index among the occluders is the one the reference
    // meets first (MyApplication.cs:373-378).
    auto exact = [&](auto kc, const float4 sp, const int orig) {
        constexpr int k = decltype(kc)::value;
        const f3 op = sub3(o, xyz(sp));
        const float c = dot3(op, op) - sp.w;
        float t;
        if constexpr (SEC && (k < 0 || k == LC)) {
            const float x = dot3(r.sec_d, op);
            if (sphere_maybe(x, r.sec_a * c) && sphere_root(x, c, r.sec_a, t) && t > RT_EPS &&
                (t < best_t || (t == best_t && orig < best_prim))) {
                best_t = t;
                best_prim = orig;
            }
        }
#pragma unroll
        for (int l = 0; l < LC; l++) {
            if (k >= 0 && l != k) continue;
            const float x = dot3(r.sh_d[l], op);
            if (sphere_maybe(x, r.sh_a[l] * c) && sphere_root(x, c, r.sh_a[l], t) && t > RT_EPS &&
                t < r.sh_dist[l])
                r.sh_first[l] = min(r.sh_first[l], orig);
        }
    };
    unsigned live[4] = {~0u, ~0u, ~0u, ~0u};
    if constexpr (!MULTI) {   // the scene's spheres are one chunk, staged once per workgroup
        if (nsp > 0)
            walk_chunk<NR, true, SPLIT>(L, nsp, s.sph, s.sph_orig, rq, fo.lane_s, fo.lane_g, fo.lane, fw, pc, exact, live);
    } else {
        for_each_chunk<NR, true, SPLIT>(L, s, rq, fo.lane_c, fo.oc, fw, live,
                                        [&](const int chunk) { return !SEC || LC > 0 || chunk_may_hit(s, chunk, o, r.sec_d, best_t); },
                                        [&](const int base, const int cnt) {
            __syncthreads();
            stage_chunk(L, s, base, cnt);
            __syncthreads();
            // A light ray that has met an occluder needs no further look at the scene unless the
            // caller wants the reference's Intersect-call counts, which depend on WHICH occluder
            // comes first in host order: its lane stops voting for groups on that ray's behalf.
            if (any_hit) {
#pragma unroll
                for (int l = 0; l < LC; l++)
                    if (r.sh_first[l] != INT_MAX) live[l] = 0u;
            }
            walk_chunk<NR, true, SPLIT>(L, cnt, s.sph + base, s.sph_orig + base, rq, fo.lane_s, fo.lane_g, fo.lane,
                                        fw, pc, exact, live);
        });
    }
    tc.exact += fw.candidates;
    tc.filt_terms += fw.terms;
    tc.filt_rays += fw.ray_tests;
    if (SEC) {
        r.sec.t = best_t;
        r.sec.prim = best_prim;
    }
    // ---- planes --------------------------------------------------------------------
    PHASE_MARK(pc, 5)
    for (int j = 0; j < s.np; j++) {
        const float4 pl = cload(s.pl, j);
        const f3 n = xyz(pl);
        const float numer = (-dot3(n, o)) - pl.w;
        const int orig = cload(s.pl_orig, j);
        float t;
        if (SEC) {
            if (plane_t(n, numer, r.sec_d, t)) closest_offer(r.sec, t, orig);
        }
#pragma unroll
        for (int l = 0; l < LC; l++) {
            if (plane_t(n, numer, r.sh_d[l], t) && t > RT_EPS && t < r.sh_dist[l])
                r.sh_first[l] = min(r.sh_first[l], orig);
        }
    }
    // ---- triangles -----------------------------------------------------------------
    for (int j = 0; j < s.nt; j++) {
        const f3 v0 = xyz(cload(s.tri, 3 * j)), e1 = xyz(cload(s.tri, 3 * j + 1)), e2 = xyz(cload(s.tri, 3 * j + 2));
        const f3 sv = sub3(o, v0);
        const int orig = cload(s.tri_orig, j);
        float t;
        if (SEC) {
            if (triangle_t(e1, e2, sv, r.sec_d, t)) closest_offer(r.sec, t, orig);
        }
#pragma unroll
        for (int l = 0; l < LC; l++) {
            if (triangle_t(e1, e2, sv, r.sh_d[l], t) && t > RT_EPS && t < r.sh_dist[l])
                r.sh_first[l] = min(r.sh_first[l], orig);
        }
    }
}

// Surface normal of a record: Sphere.GetNormal objects.cs:128-131, else the stored
// plane / triangle normal (objects.cs:171-174, 270-273).
__device__ __forceinline__ f3 record_normal(const DevScene &s, int prim, int type, f3 hit)
{
    const float4 g = cload(s.geo, prim);
    if (type == 0) return normalized3(sub3(hit, xyz(g)));
    return xyz(g);
}


// A record's reflection ray (objects.cs:82-83): closest hit over the whole scene.
template <bool MULTI>
__device__ __forceinline__ void trace_reflection(const DevScene &s, WalkLds &L, const f3 hit, const f3 view,
                                                 const f3 normal, Closest &sec_out, f3 &sec_dir, TestCount &tc,
                                                 PhaseClock &pc)
{
    Rays<0, true> r;
    // objects.cs:82: (viewDir - 2 * Dot(viewDir, normal) * normal).Normalized()
    r.sec_d = normalized3(sub3(view, scale3(normal, 2.0f * dot3(view, normal))));
    r.sec_a = dot3(r.sec_d, r.sec_d);
    r.sec.t = RT_FLT_MAX;
    r.sec.prim = INT_MAX;
    traverse<0, true, false, MULTI>(s, L, hit, r, tc, pc, true);
    sec_out = r.sec;
    sec_dir = r.sec_d;
    tc.sph += s.ns;
    tc.pl += s.np;
    tc.tri += s.nt;
}

// ---------------------------------------------------------------------------------------
// Light rays (IsInShadow, MyApplication.cs:370-380, called once per light from objects.cs:56).
// Every light ray ends at a light, so the spheres that can occlude it are found in that light's
// direction table (rt_device.h, DevScene::lt_cells; built and proven conservative in rt_api.cpp,
// build_light_tables): the bin of the direction from the light to the ray's origin lists them,
// and the literal Sphere.Intersect runs over that list.  Origins farther from the scene than the
// tables were built for evaluate every sphere.  Planes and triangles are few and tested one by
// one as the reference does.
// ---------------------------------------------------------------------------------------

// Cell of light l's table for the direction w (from the light to the ray's origin; any length).
__device__ __forceinline__ uint2 light_table_cell(const DevScene &s, const int l, const f3 w)
{
    const float ax = fabsf(w.x), ay = fabsf(w.y), az = fabsf(w.z);
    int axis;
    float m, a, b;   // major component, then the two others in cyclic order
    if (ax >= ay && ax >= az) axis = 0, m = w.x, a = w.y, b = w.z;
    else if (ay >= az) axis = 1, m = w.y, a = w.z, b = w.x;
    else axis = 2, m = w.z, a = w.x, b = w.y;
    const int face = 2 * axis + (m < 0.0f ? 1 : 0);
    const float inv = __builtin_amdgcn_rcpf(m);   // 1 ulp: far inside the tables' angular margin
    const float half = 0.5f * (float)s.lt_n;
    const float fu = (a * inv + 1.0f) * half, fv = (b * inv + 1.0f) * half;
    const int top = s.lt_n - 1;
    const int iu = !(fu >= 0.0f) ? 0 : (fu >= (float)top ? top : (int)fu);   // NaN -> 0
    const int iv = !(fv >= 0.0f) ? 0 : (fv >= (float)top ? top : (int)fv);
    // The tier of the ray's length: the first whose bound |w|^2 does not exceed (NaN: the last, unbounded one).
    const float q = (w.x * w.x + w.y * w.y) + w.z * w.z;
    int tier = 0;
    while (tier < s.lt_tiers - 1 && !(q <= s.lt_tier_q[tier])) tier++;
    return s.lt_cells[((size_t)((tier * s.nl + l) * 6 + face) * (size_t)s.lt_n + (size_t)iv) * (size_t)s.lt_n + (size_t)iu];
}

// One light ray against one sphere: true = Sphere.Intersect (objects.cs:112-121) gives a t in (eps, dist).
// The IEEE square root and divide of the literal evaluation cost ~60 instructions, and a wave executes them
// whenever ANY lane has a candidate, so the candidate's t is first bracketed with the one-ulp hardware
// square root and reciprocal (the bracket and its bound: "Grid walk" below), which settles the answer
// unless t lies within the bracket's width of one of the two bounds; only then is the literal evaluation
// made (also for operands that are not finite: every comparison below is then false).
__device__ __forceinline__ bool sphere_occludes(const float4 sp, const f3 o, const f3 d, const float a, const float dist)
{
    const f3 op = sub3(o, xyz(sp));
    const float c = dot3(op, op) - sp.w;
    const float x = dot3(d, op);
    if (!sphere_maybe(x, a * c)) return false;
    const float bb = 2.0f * x;
    const float disc = (bb * bb) - ((4.0f * a) * c);   // (as sphere_root forms it)
    if (disc < 0.0f) return false;
    const float sq = __builtin_amdgcn_sqrtf(disc);
    const float ta = ((-bb) - sq) * __builtin_amdgcn_rcpf(2.0f * a);
    const float delta = 1e-6f * ((fabsf(bb) + sq) + fabsf(ta)) + 1e-18f;
    if (ta + delta <= RT_EPS || ta - delta >= dist) return false;
    if (ta - delta > RT_EPS && ta + delta < dist) return true;
    float t;
    return sphere_root(x, c, a, t) && t > RT_EPS && t < dist;
}

// Is the ray from `hit` towards light l (direction d, a = d.d, length dist) occluded?  Returns the
// lowest host index among the occluders found, INT_MAX if none.  COUNT: the reference's
// Intersect-call counts are kept, which need the FIRST occluder in host order
// (MyApplication.cs:373-378), so every candidate is evaluated; otherwise any occluder will do and
// the search stops at the first one.  far: the origin lies outside the region the tables serve
// (any_far: so does some lane's of this wave).  Every lane of the wave must call it.
// Does a plane occlude a light ray?  Plane.Intersect (objects.cs:157-164) accepts t = numer / denom
// when t > 0.0001, and IsInShadow then asks for epsilon < t < dist.  The quotient from the
// hardware reciprocal is within a few ulp of the correctly rounded one, so it settles the answer
// unless it lies within a relative 1e-6 of one of the two bounds; only then is the exact divide made.
__device__ __forceinline__ bool plane_occludes(const f3 n, const float numer, const f3 d, const float dist)
{
    const float denom = dot3(n, d);
    if (fabsf(denom) <= 0.0001f) return false;
    const float q = numer * __builtin_amdgcn_rcpf(denom);
    const float lo = RT_EPS, slack = 1e-6f;
    if (q < lo * (1.0f - slack) || q > dist * (1.0f + slack)) return false;
    if (q > lo * (1.0f + slack) && q < dist * (1.0f - slack)) return true;
    const float t = ieee_div(numer, denom);   // (also where q or dist is not finite)
    return !(t <= 0.0001f) && t > RT_EPS && t < dist;
}

// Table entries per lane in flight together.  (3840x2160, 1000 spheres, 2 lights: 0.542 ms per frame with 2, 0.550
// with 3, 0.555 with 4 or 1; 65536 spheres, 16 lights, tables in tiers: 2.66, 2.63, 2.65 and 3.05 ms.)
#ifndef LIGHT_BATCH
#define LIGHT_BATCH 2
#endif

template <bool COUNT>
__device__ __forceinline__ int light_occluder(const DevScene &s, const uint2 cell, const f3 hit, const f3 d,
                                              const float a, const float dist, const bool far, const bool any_far,
                                              const bool active, TestCount &tc)
{
    int first = INT_MAX;
    // The list's first entries are fetched before the planes and triangles are tested, so that their
    // way from memory is covered by that arithmetic (cell: the light's table cell for this origin,
    // which the caller fetched while it shaded the previous light).  Every lane loads, whatever its
    // list's length: the lists are followed by RT_LT_PAD entries that never occlude, an index is
    // held to "one past this lane's list", and what a lane has no use for is never looked at
    // (k + j < cnt below) - straight-line loads instead of a branch around every one of them.
    const bool tables = s.lt_n > 0;   // (wave-uniform)
    const float4 *list = s.lt_sph + cell.x;
    const float *reach = s.lt_reach + cell.x;
    const int len = (int)cell.y;
    const int cnt0 = tables && active && !far ? len : 0;
    float4 nxt[LIGHT_BATCH];
    float nxt_reach[LIGHT_BATCH];
    static_assert(LIGHT_BATCH <= RT_LT_PAD, "the first batch is loaded without looking at the list's length");
    if (tables) {
#pragma unroll
        for (int j = 0; j < LIGHT_BATCH; j++) {
            nxt[j] = list[j];
            nxt_reach[j] = reach[j];
        }
    }
    for (int j = 0; j < s.np; j++) {
        const float4 pl = cload(s.pl, j);
        const f3 n = xyz(pl);
        if (plane_occludes(n, (-dot3(n, hit)) - pl.w, d, dist)) first = min(first, cload(s.pl_orig, j));
    }
    for (int j = 0; j < s.nt; j++) {
        const f3 v0 = xyz(cload(s.tri, 3 * j)), e1 = xyz(cload(s.tri, 3 * j + 1)), e2 = xyz(cload(s.tri, 3 * j + 2));
        float t;
        if (triangle_t(e1, e2, sub3(hit, v0), d, t) && t > RT_EPS && t < dist) first = min(first, cload(s.tri_orig, j));
    }
    if (tables) {
        int cnt = !COUNT && first != INT_MAX ? 0 : cnt0;
        // LIGHT_BATCH entries per step, the next step's loads in flight while this step's entries are
        // evaluated: a list is walked at the pace of the arithmetic, not of one memory round trip
        // per entry.
        // A list is sorted by how close its spheres come to the light (lt_reach): the first one that
        // stays farther from the light than the ray's own origin ends the walk.
#pragma clang loop unroll(disable)
        for (int k = 0; __ballot(k < cnt) != 0ull; k += LIGHT_BATCH) {
            float4 sp[LIGHT_BATCH];
            float sp_reach[LIGHT_BATCH];
#pragma unroll
            for (int j = 0; j < LIGHT_BATCH; j++) {
                sp[j] = nxt[j];
                sp_reach[j] = nxt_reach[j];
            }
            if (__ballot(k + LIGHT_BATCH < cnt) != 0ull) {
#pragma unroll
                for (int j = 0; j < LIGHT_BATCH; j++) {
                    const int i = min(k + LIGHT_BATCH + j, len);
                    nxt[j] = list[i];
                    nxt_reach[j] = reach[i];
                }
            }
#pragma unroll
            for (int j = 0; j < LIGHT_BATCH; j++) {
                if (k + j < cnt) {
                    if (sp_reach[j] > dist) {
                        cnt = 0;   // this sphere and all after it lie beyond the ray's origin
                    } else {
                        if (COUNT) tc.shadow++;
                        if (sphere_occludes(sp[j], hit, d, a, dist)) {
                            if (COUNT) first = min(first, s.lt_orig[cell.x + k + j]);
                            else first = 0, cnt = 0;   // any occluder will do
                        }
                    }
                }
            }
        }
        if (any_far) {
            // Origins the tables do not serve meet every sphere; the wave takes such rays one at a
            // time, 64 spheres per step (ns_padded is a multiple of 64; padding never occludes).
            const int lane = threadIdx.x & 63;
            unsigned long long todo = __ballot(active && far && (COUNT || first == INT_MAX));
#pragma clang loop unroll(disable)
            while (todo != 0ull) {
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                const f3 ro = mk3(__shfl(hit.x, src), __shfl(hit.y, src), __shfl(hit.z, src));
                const f3 rd = mk3(__shfl(d.x, src), __shfl(d.y, src), __shfl(d.z, src));
                const float ra = __shfl(a, src), rdist = __shfl(dist, src);
                int found = INT_MAX;
#pragma clang loop unroll(disable)
                for (int base = 0; base < s.ns_padded; base += 64) {
                    if (sphere_occludes(s.sph[base + lane], ro, rd, ra, rdist)) found = min(found, s.sph_orig[base + lane]);
                    if (!COUNT && __ballot(found != INT_MAX) != 0ull) break;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) found = min(found, __shfl_xor(found, o, 64));
                if (lane == src) {
                    first = min(first, found);
                    if (COUNT) tc.shadow += s.ns_padded;
                }
            }
        }
    }
    if (COUNT && active) {
        // Tests the reference executes for this light: primitives up to and including
        // the first occluder (MyApplication.cs:373-378), all of them when unoccluded.
        const int upto = first != INT_MAX ? first + 1 : s.nprims;
        tc.sph += cload(s.class_prefix, 3 * upto + 0);
        tc.pl += cload(s.class_prefix, 3 * upto + 1);
        tc.tri += cload(s.class_prefix, 3 * upto + 2);
    }
    return first;
}

// ---------------------------------------------------------------------------------------
// Shading: Primitive.Shade objects.cs:44-96 for one hit record, light by light as the reference
// does it: the light ray (IsInShadow, through the light's direction table), the surface colour
// under that light, the Phong term, the reflected colour its child record left in refl[], the
// accumulation in the reference's order, the clamp.  The colour goes to the parent record or, at
// level 0, to the pixel.  The alpha channel is not carried: it never reaches an RGB value or
// MixColor.
// ---------------------------------------------------------------------------------------
// (float)Math.Pow(x, 250) for the Phong term (objects.cs:77-78).  x^250 by repeated
// squaring in double: 250 = 128+64+32+16+8+2.  Relative error <= ~250 * 2^-53, i.e.
// the value rounds to the same float as a correctly rounded pow except within
// ~3e-14 (relative) of a float rounding boundary.
__device__ __forceinline__ float pow250(float xf)
{
    // Below 0.65 the power is below 2e-47, far under half the smallest FP32 denormal (7e-46):
    // the conversion to float gives +0 whatever the last bits of the double were.
    if (xf >= 0.0f && xf < 0.65f) return 0.0f;
    const double x = (double)xf;
    const double x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8, x32 = x16 * x16,
                 x64 = x32 * x32, x128 = x64 * x64;
    return (float)(((((x128 * x64) * x32) * x16) * x8) * x2);
}

// Sign of Math.Sin(a) for a double a: +1, -1, or 0 when the cheap route cannot vouch for it.
// With t = a / pi, sin(a) is negative exactly when floor(t) is odd.  t is computed in double
// with an absolute error below 2e-10 for |a| < 2^20; when t is farther than 1e-8 from an
// integer, floor(t) is the true floor and the sign is certain (any sine routine accurate to an
// ulp returns that sign, the reference's included).
__device__ __forceinline__ int sin_sign(const double a)
{
    if (!(fabs(a) < 1048576.0)) return 0;
    const double t = a * 0.31830988618379067154;   // 1 / pi
    const double fl = floor(t), frac = t - fl;
    if (!(frac > 1e-8 && frac < 1.0 - 1e-8)) return 0;
    return ((long long)fl & 1ll) ? -1 : 1;
}

// Plane.GetCheckerboardColor objects.cs:176-188: 1 = white.  The reference forms
// (float)(Math.Sin(x * 7.5f) * Math.Sin(y * 7.5f) * Math.Sin(z * 7.5f)) and tests `< 0`.  The
// product of three sines whose signs are certain and whose arguments are at least 1e-8 * pi away
// from a zero is in magnitude above 1e-25, so its float is not a zero and its sign is the product
// of the signs; otherwise the sines themselves are evaluated.
__device__ __forceinline__ bool checker_white(f3 hit)
{
    const float scale = 7.5f;
    const double ax = (double)(hit.x * scale), ay = (double)(hit.y * scale), az = (double)(hit.z * scale);
    const int sx = sin_sign(ax), sy = sin_sign(ay), sz = sin_sign(az);
    if (sx != 0 && sy != 0 && sz != 0) return sx * sy * sz > 0;
    const float sines = (float)((sin(ax) * sin(ay)) * sin(az));
    return !(sines < 0.0f);
}

// Plane.GetTextureCoordinates objects.cs:190-204 + Texture.GetColor :348-353 +
// LoadTexture :343.  `% 1` is fmodf(x, 1) = x - trunc(x), exact in FP32.  The reference
// throws for addresses outside the image (NaN / inf coordinates); clamp instead.
__device__ __forceinline__ f3 texture_color(const DevScene &s, int slot, f3 hit)
{
    const float fu = ieee_div(hit.x, 5.0f) - 0.5f, fv = ieee_div(hit.y, 5.0f) - 0.2f;
    float u = fu - truncf(fu), v = fv - truncf(fv);
    u = -u;
    if (u < 0.0f) u += 1.0f;
    if (v < 0.0f) v += 1.0f;
    const int w = s.tex_w[slot], h = s.tex_h[slot];
    int tx = net_f2i(u * (float)(w - 1)), ty = net_f2i(v * (float)(h - 1));
    tx = min(max(tx, 0), w - 1);
    ty = min(max(ty, 0), h - 1);
    const uchar4 p = s.tex[slot][(size_t)ty * (size_t)w + (size_t)tx];
    return mk3(ieee_div((float)p.x, 255.0f), ieee_div((float)p.y, 255.0f),
               ieee_div((float)p.z, 255.0f));
}

__device__ __forceinline__ float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }


// What Shade needs of a record before it looks at the lights.
struct ShadeRecord {
    f3 hit, view, normal;
    f3 base;          // surface colour under a light: material, texture or checkerboard (objects.cs:62-74)
    f3 mat_glossy, mat_ambient;
    f3 refl_term;     // (Vector4)Shade(child) * material.specular * 1, when have_refl
    bool have_refl;   // the record traced a reflection ray that hit something (objects.cs:80-88)
    bool far;         // the origin lies outside the region the lights' tables serve
    bool any_far;     // ... some active lane's of this wave does
    int parent;
};

// reflecting: the record is in its level's queue A (it traced a reflection ray, and refl[slot]
// holds the child's colour if that ray hit something).  through: the record was written earlier
// in this launch by another workgroup (device-scope loads).  Every lane of the wave must call it.
__device__ __forceinline__ ShadeRecord shade_setup(const DevScene &s, const DevFrame &f, const int slot,
                                                   const bool active, const bool reflecting, const bool through)
{
    ShadeRecord r;
    float4 ra, rb;
    if (through) load_record(&f.recA[slot], &f.recB[slot], ra, rb);
    else ra = f.recA[slot], rb = f.recB[slot];
    r.hit = xyz(ra);
    r.view = xyz(rb);
    const int prim = __float_as_int(ra.w);
    r.parent = __float_as_int(rb.w);
    const float4 m0 = cload(s.mat, 3 * prim), m1 = cload(s.mat, 3 * prim + 1), m2 = cload(s.mat, 3 * prim + 2);
    const int flags = __float_as_int(m1.w);
    const int type = flags & RT_MF_TYPE_MASK;
    r.normal = record_normal(s, prim, type, r.hit);
    const float specular = m0.w;
    r.mat_glossy = xyz(m1);
    r.mat_ambient = xyz(m2);
    // The same for every light, so looked up once.
    r.base = xyz(m0);
    if (type == 1) {
        const int tex = (flags & RT_MF_TEX_MASK) >> RT_MF_TEX_SHIFT;
        if (tex != 0) r.base = texture_color(s, tex - 1, r.hit);
        else if (flags & RT_MF_CHECKER) {
            const float cw = checker_white(r.hit) ? 1.0f : 0.0f;
            r.base = mk3(cw, cw, cw);
        }
    }
    r.have_refl = false;
    r.refl_term = mk3(0.0f, 0.0f, 0.0f);
    if (reflecting) {
        const float4 rf = f.refl[slot];
        r.have_refl = rf.w != 0.0f;
        r.refl_term = scale3(scale3(xyz(rf), specular), 1.0f);
    }
    // Does the lights' tables' region hold this origin?  (A NaN origin is "far" and then meets
    // nothing, like in the reference.)
    const f3 oc = mk3(r.hit.x - s.sph_center[0], r.hit.y - s.sph_center[1], r.hit.z - s.sph_center[2]);
    r.far = s.lt_n > 0 && !(((fabsf(oc.x) + fabsf(oc.y)) + fabsf(oc.z)) <= s.lt_near);
    r.any_far = s.lt_n > 0 && __ballot(active && r.far) != 0ull;
    return r;
}

// One pass of the reference's light loop (objects.cs:51-91) for light l: what it adds to finalColor,
// in its order: first `reflected` (only when the record has a reflected colour), then `direct`.
// cell: the light's table cell for this origin (light_table_cell).  Every lane of the wave must call it.
// d2: distance * distance, the divisor of the reflected colour (for callers that add it later).
template <bool COUNT>
__device__ __forceinline__ void light_terms(const DevScene &s, const ShadeRecord &r, const int l, const uint2 cell,
                                            const bool active, TestCount &tc, f3 &reflected, f3 &direct, float &d2)
{
    const f3 lp = xyz(cload(s.light_pos, l)), lc = xyz(cload(s.light_col, l));
    const f3 to_light = sub3(lp, r.hit);
    const f3 shadow_ray = normalized3(to_light);            // objects.cs:51
    const float distance = length3(to_light);               // objects.cs:52
    float diffuse = net_maxf(dot3(r.normal, shadow_ray), 0.0f);
    float glossy = 1.0f;
    // IsInShadow(hitPoint, shadowRay, 0.0001f, distance), objects.cs:56
    const int occluder = light_occluder<COUNT>(s, cell, r.hit, shadow_ray, dot3(shadow_ray, shadow_ray), distance,
                                               r.far, r.any_far, active, tc);
    if (occluder != INT_MAX) {
        diffuse = 0.0f;
        glossy = 0.0f;
    }
    const f3 diffuse_color = scale3(r.base, diffuse);
    const f3 reflection = sub3(shadow_ray, scale3(r.normal, 2.0f * dot3(shadow_ray, r.normal)));
    const float sp = pow250(net_maxf(dot3(reflection, r.view), 0.0f));
    const f3 glossy_color = scale3(scale3(r.mat_glossy, sp), glossy);
    d2 = distance * distance;
    reflected = mk3(0.0f, 0.0f, 0.0f);
    if (r.have_refl) reflected = mk3(ieee_div(r.refl_term.x, d2), ieee_div(r.refl_term.y, d2), ieee_div(r.refl_term.z, d2));
    // light.color * 1 / (distance*distance) * (diffuseColor + glossyColor) + ambient
    // (per component; the reference's lights are (2, 2, 2): equal components share the quotient)
    f3 li;
    li.x = ieee_div(lc.x * 1.0f, d2);
    if (lc.y == lc.x && lc.z == lc.x) li.y = li.z = li.x;   // wave-uniform
    else li.y = ieee_div(lc.y * 1.0f, d2), li.z = ieee_div(lc.z * 1.0f, d2);
    const f3 dg = add3(diffuse_color, glossy_color);
    direct = add3(mk3(li.x * dg.x, li.y * dg.y, li.z * dg.z), r.mat_ambient);
}

// The clamp of objects.cs:94 and where the colour goes: the pixel (level 0) or the parent record.
__device__ __forceinline__ void shade_store(const DevFrame &f, const int level, const int parent, f3 final,
                                            const bool through, int32_t *pixels)
{
    final = mk3(clamp01(final.x), clamp01(final.y), clamp01(final.z));
    if (level == 0) {
        // Primitive.MixColor objects.cs:35-41 -> Surface.pixels (MyApplication.cs:420)
        const uint32_t r = (uint32_t)net_f2i(final.x * 255.0f);
        const uint32_t g = (uint32_t)net_f2i(final.y * 255.0f);
        const uint32_t b = (uint32_t)net_f2i(final.z * 255.0f);
        pixels[parent] = (int32_t)((r << 16) + (g << 8) + b);
    } else if (through) {
        store_record(&f.refl[parent], f4raw{final.x, final.y, final.z, 1.0f});   // (see reflection_tile)
    } else {
        f.refl[parent] = make_float4(final.x, final.y, final.z, 1.0f);
    }
}

// Primitive.Shade for one record per lane, the lights one after the other.
template <bool COUNT>
__device__ __forceinline__ void shade_record(const DevScene &s, const DevFrame &f, const int level, const int slot,
                                             const bool active, const bool reflecting, const bool through,
                                             int32_t *pixels, TestCount &tc)
{
    const ShadeRecord r = shade_setup(s, f, slot, active, reflecting, through);
    f3 final = mk3(0.0f, 0.0f, 0.0f);
    const int nl = s.nl;
    // The table cell of light l + 1 is fetched while light l is shaded.
    uint2 cell_next = make_uint2(0u, 0u);
    if (s.lt_n > 0 && nl > 0) cell_next = light_table_cell(s, 0, sub3(r.hit, xyz(cload(s.light_pos, 0))));
    for (int l = 0; l < nl; l++) {
        const uint2 cell = cell_next;
        if (s.lt_n > 0 && l + 1 < nl) cell_next = light_table_cell(s, l + 1, sub3(r.hit, xyz(cload(s.light_pos, l + 1))));
        f3 reflected, direct;
        float d2;
        light_terms<COUNT>(s, r, l, cell, active, tc, reflected, direct, d2);
        if (r.have_refl) final = add3(final, reflected);
        final = add3(final, direct);
    }
    if (active) shade_store(f, level, r.parent, final, through, pixels);
}

// A reflecting record's light loop does not depend on its child: what every light adds directly, and
// the divisor of the reflected colour, are computed as soon as the record exists (lit_record / k_lit,
// beside the plain records' shading, while the chain of reflection levels is the frame's critical
// path) and kept in DevFrame::lit; once the child's colour is there only the sum is left
// (combine_record): the same additions in the same order as shade_record makes them.  Every lane of
// the wave must call lit_record.
// Where DevFrame::lit keeps the light terms of the reflecting record in `slot` of `level` (lit_stride records
// per level: with many lights only the short queues have their terms computed ahead).
__device__ __forceinline__ size_t lit_first(const DevFrame &f, const int level, const int slot, const int nl)
{
    return ((size_t)level * (size_t)f.lit_stride + (size_t)(slot - level * f.P)) * (size_t)nl;
}

// Frames with DevFrame::path_shade: a reflecting record tells its parent that the pixel's path goes on below it
// (k_shade_paths).  Done where the record's light terms are computed - beside the chain of levels, not on it: in
// push_records the same store cost the 3840x2160 frame 3 % - and after the parent's "no reflected colour yet",
// which the launch that created this record stored.
__device__ __forceinline__ void mark_parent(const DevFrame &f, const int level, const int parent)
{
    if (f.path_shade && level >= 1) reinterpret_cast<float *>(&f.refl[parent])[3] = RT_REFL_PENDING;
}

template <bool COUNT>
__device__ __forceinline__ void lit_record(const DevScene &s, const DevFrame &f, const int level, const int slot,
                                           const bool active, TestCount &tc)
{
    const ShadeRecord r = shade_setup(s, f, slot, active, false, false);
    if (active) mark_parent(f, level, r.parent);
    const int nl = s.nl;
    uint2 cell_next = make_uint2(0u, 0u);
    if (s.lt_n > 0 && nl > 0) cell_next = light_table_cell(s, 0, sub3(r.hit, xyz(cload(s.light_pos, 0))));
    for (int l = 0; l < nl; l++) {
        const uint2 cell = cell_next;
        if (s.lt_n > 0 && l + 1 < nl) cell_next = light_table_cell(s, l + 1, sub3(r.hit, xyz(cload(s.light_pos, l + 1))));
        f3 reflected, direct;
        float d2;
        light_terms<COUNT>(s, r, l, cell, active, tc, reflected, direct, d2);
        if (active) f.lit[lit_first(f, level, slot, nl) + (size_t)l] = make_float4(direct.x, direct.y, direct.z, d2);
    }
}

// rf: the record's refl entry (the child's colour, w != 0 when there is one).
__device__ __forceinline__ f3 combine_sum(const DevScene &s, const DevFrame &f, const int level, const int slot,
                                          const float4 rf, int &parent)
{
    const int prim = __float_as_int(f.recA[slot].w);
    parent = __float_as_int(f.recB[slot].w);
    const float specular = cload(s.mat, 3 * prim).w;
    const bool have_refl = rf.w != 0.0f;
    const f3 refl_term = scale3(scale3(xyz(rf), specular), 1.0f);   // (Vector4)Shade(..) * material.specular * 1
    f3 final = mk3(0.0f, 0.0f, 0.0f);
    const int nl = s.nl;
    for (int l = 0; l < nl; l++) {
        const float4 t = f.lit[lit_first(f, level, slot, nl) + (size_t)l];
        if (have_refl) final = add3(final, mk3(ieee_div(refl_term.x, t.w), ieee_div(refl_term.y, t.w), ieee_div(refl_term.z, t.w)));
        final = add3(final, xyz(t));
    }
    return final;
}

__device__ __forceinline__ void combine_record(const DevScene &s, const DevFrame &f, const int level, const int slot,
                                               int32_t *pixels)
{
    int parent;
    const f3 final = combine_sum(s, f, level, slot, f.refl[slot], parent);
    shade_store(f, level, parent, final, false, pixels);
}

// Are the light terms of a level of `na` reflecting records computed ahead of their children (DevFrame::lit)?
__device__ __forceinline__ bool lit_ahead(const DevFrame &f, const int na)
{
    return f.lit != nullptr && (f.lit_below <= 0 || na < f.lit_below);
}

// One tile = 256 consecutive records of one queue of one level, one record per lane.
// reflection_tile: records that need a reflection trace (queue "A") walk the scene with their
// reflection ray and push the records of the next level; they are shaded later, when their
// children are (k_shade).  n = records in the queue.  Every thread of the workgroup must call it
// (barriers inside).
template <bool MULTI>
__device__ __forceinline__ void reflection_tile(const DevScene &s, const DevFrame &f, WalkLds &L, int *blk,
                                                const int level, const int tile, const int n,
                                                TestCount &tc, PhaseClock &pc)
{
    PHASE_MARK(pc, 0)
    const int i = tile * 256 + threadIdx.x;
    const bool active = i < n;
    const int ii = active ? i : n - 1;
    const int slot = level * f.P + ii;
    float4 ra, rb;
    load_record(&f.recA[slot], &f.recB[slot], ra, rb);
    const f3 hit = xyz(ra), view = xyz(rb);
    const int prim = __float_as_int(ra.w);
    const int type = __float_as_int(cload(s.mat, 3 * prim + 1).w) & RT_MF_TYPE_MASK;
    const f3 normal = record_normal(s, prim, type, hit);
    Closest sec = {RT_FLT_MAX, INT_MAX};
    f3 sec_dir = mk3(0.0f, 0.0f, 0.0f);
    TestCount ltc = {0, 0, 0, 0, 0, 0, 0};
    trace_reflection<MULTI>(s, L, hit, view, normal, sec, sec_dir, ltc, pc);
    PHASE_MARK(pc, 6)
    if (active) {
        tc.sph += ltc.sph;
        tc.pl += ltc.pl;
        tc.tri += ltc.tri;
        tc.exact += ltc.exact;
    }
    tc.filt_terms += ltc.filt_terms;   // inactive lanes run the loops too
    tc.filt_rays += ltc.filt_rays;
    const bool child = active && sec.prim != INT_MAX;
    const f3 hit2 = add3(hit, scale3(sec_dir, sec.t));   // objects.cs:123,166,263
    // "No reflected colour yet": written through to memory like the records, because the child's
    // colour may be written later in this same launch from another XCD (plain_tile), and two
    // caches holding the line dirty would write it back in either order.
    if (active) store_record(&f.refl[slot], f4raw{0.0f, 0.0f, 0.0f, 0.0f});
    const bool rv[1] = {child};
    const int rp[1] = {child ? sec.prim : 0}, rparent[1] = {slot};
    const f3 rh[1] = {hit2}, rd[1] = {sec_dir};
    push_records<1>(f, s, level + 1, blk, rv, rp, rh, rd, rparent);
    PHASE_MARK(pc, 7)
}

// A closest hit as one word: t > 0, so the keys' order is the order of (t, host index).
__device__ __forceinline__ unsigned long long hit_key(const float t, const int prim)
{
    return ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(unsigned)prim;
}

// The same for a level of few records, 64 records per workgroup instead of 256: every wave takes
// the SAME 64 records and a quarter of the scene (the runs of RT_SUPER spheres r with r % 4 = its
// number, every fourth plane and triangle), the waves' closest hits meet in LDS as 64-bit keys
// (hit_key: their order is the order Scene.Intersect's strict `<` implies), and wave 0 pushes the
// children.  A wave alone on its SIMD issues an instruction every ~5 cycles, and the rays of a deep
// level have so little in common that a wave ends up testing most of the scene's spheres: a tile
// of a 1000-sphere scene takes ~40 us however few tiles the level has; four waves with a quarter of
// the spheres each take a third of that.  best: a __shared__ array of 64 keys.
__device__ __forceinline__ void reflection_tile_spread(const DevScene &s, const DevFrame &f, WalkLds &L, int *blk,
                                                       unsigned long long *best, const int level, const int tile,
                                                       const int n, TestCount &tc, PhaseClock &pc)
{
    PHASE_MARK(pc, 0)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = tile * 64 + lane;
    const bool active = i < n;
    const int slot = level * f.P + (active ? i : n - 1);
    float4 ra, rb;
    load_record(&f.recA[slot], &f.recB[slot], ra, rb);
    const f3 hit = xyz(ra), view = xyz(rb);
    const int prim = __float_as_int(ra.w);
    const int type = __float_as_int(cload(s.mat, 3 * prim + 1).w) & RT_MF_TYPE_MASK;
    const f3 normal = record_normal(s, prim, type, hit);
    // objects.cs:82: (viewDir - 2 * Dot(viewDir, normal) * normal).Normalized()
    const f3 dir = normalized3(sub3(view, scale3(normal, 2.0f * dot3(view, normal))));
    const float a = dot3(dir, dir);
    if (wave == 0) best[lane] = ~0ull;
    __syncthreads();
    Closest sec = {RT_FLT_MAX, INT_MAX};
    FilterWork fw = {0, 0, 0};
    if (s.ns_padded > 0) {
        const FilterOrigin fo = filter_origin(s, hit);
        f4raw rq[4];
        rq[0] = filter_ray(fo, dir);
#pragma unroll
        for (int l = 1; l < 4; l++) rq[l] = f4raw{0.f, 0.f, 0.f, 0.f};
        // Sphere.Intersect objects.cs:112-121, literally; equal distances: the lower host index.
        auto exact = [&](auto, const float4 sp, const int orig) {
            const f3 op = sub3(hit, xyz(sp));
            const float c = dot3(op, op) - sp.w;
            const float x = dot3(dir, op);
            float t;
            if (sphere_maybe(x, a * c) && sphere_root(x, c, a, t)) closest_offer(sec, t, orig);
        };
        const unsigned live[4] = {~0u, ~0u, ~0u, ~0u};
        walk_chunk<1, true, false>(L, s.ns_padded, s.sph, s.sph_orig, rq, fo.lane_s, fo.lane_g, fo.lane, fw, pc, exact,
                                   live, 0x88888888u >> wave);
    }
    PHASE_MARK(pc, 5)
    for (int j = wave; j < s.np; j += 4) {
        const float4 pl = cload(s.pl, j);
        const f3 nn = xyz(pl);
        float t;
        if (plane_t(nn, (-dot3(nn, hit)) - pl.w, dir, t)) closest_offer(sec, t, cload(s.pl_orig, j));
    }
    for (int j = wave; j < s.nt; j += 4) {
        const f3 v0 = xyz(cload(s.tri, 3 * j)), e1 = xyz(cload(s.tri, 3 * j + 1)), e2 = xyz(cload(s.tri, 3 * j + 2));
        float t;
        if (triangle_t(e1, e2, sub3(hit, v0), dir, t)) closest_offer(sec, t, cload(s.tri_orig, j));
    }
    if (sec.prim != INT_MAX) atomicMin(&best[lane], hit_key(sec.t, sec.prim));
    __syncthreads();
    PHASE_MARK(pc, 6)
    const unsigned long long key = best[lane];
    const float t = key == ~0ull ? RT_FLT_MAX : __uint_as_float((unsigned)(key >> 32));
    const int hit_prim = key == ~0ull ? INT_MAX : (int)(unsigned)key;
    if (active && wave == 0) {
        tc.sph += s.ns;
        tc.pl += s.np;
        tc.tri += s.nt;
    }
    tc.exact += active ? fw.candidates : 0;
    tc.filt_terms += fw.terms;
    tc.filt_rays += fw.ray_tests;
    const bool child = active && wave == 0 && hit_prim != INT_MAX;
    const f3 hit2 = add3(hit, scale3(dir, t));   // objects.cs:123,166,263
    if (active && wave == 0) store_record(&f.refl[slot], f4raw{0.0f, 0.0f, 0.0f, 0.0f});   // (see reflection_tile)
    const bool rv[1] = {child};
    const int rp[1] = {child ? hit_prim : 0}, rparent[1] = {slot};
    const f3 rh[1] = {hit2}, rd[1] = {dir};
    push_records<1>(f, s, level + 1, blk, rv, rp, rh, rd, rparent);
    PHASE_MARK(pc, 7)
}

// plain_tile: records without a reflection trace (queue "B") depend on nothing further: they are
// shaded on the spot (light rays included) and their colour goes to the parent record or pixel.
template <bool COUNT>
__device__ __forceinline__ void plain_tile(const DevScene &s, const DevFrame &f, const int level, const int tile,
                                           const int n, int32_t *pixels, TestCount &tc, PhaseClock &pc)
{
    PHASE_MARK(pc, 0)
    const int i = tile * 256 + threadIdx.x;
    const bool active = i < n;
    const int ii = active ? i : n - 1;
    shade_record<COUNT>(s, f, level, (level + 1) * f.P - 1 - ii, active, false, true, pixels, tc);
    PHASE_MARK(pc, 5)
}

__device__ __forceinline__ void flush_test_counts(const DevFrame &f, unsigned long long *cnt_sh, const TestCount &tc)
{
    static_assert(COUNTER_WORDS == 9, "the initialiser below follows DevCounters word for word");
    const int ws = wave_sum(tc.sph), wp = wave_sum(tc.pl), wt = wave_sum(tc.tri), we = wave_sum(tc.exact),
              wsh = wave_sum(tc.shadow), wg = wave_sum(tc.grid);
    // the filter counts are the same in every lane of the wave; the cells of grid walks count with them
    const unsigned long long v[COUNTER_WORDS] = {(unsigned long long)ws, (unsigned long long)wp, (unsigned long long)wt,
                                                 (unsigned long long)we,
                                                 64ull * (unsigned long long)tc.filt_rays + (unsigned long long)wg,
                                                 64ull * (unsigned long long)tc.filt_terms, 0ull, 0ull,
                                                 (unsigned long long)wsh};
    flush_counters(f, cnt_sh, v);
}

// One queue of one level per launch.  k_reflect: the records that need a reflection trace
// (they walk the staged spheres); k_plain: the others, shaded on the spot — no LDS, and held to
// RT_PLAIN_WAVES waves per SIMD worth of registers, because the work is a chain of short
// dependent memory reads (record, table cell, occluder list) that only other waves can hide.
template <bool MULTI, bool COUNT>
__global__ void __launch_bounds__(256) k_reflect(DevScene s, DevFrame f, int level)
{
    __shared__ WalkLds L;
    __shared__ int blk[12];
    __shared__ unsigned long long cnt_sh[COUNTER_WORDS];
    const int n = f.cnt[2 * level];
    const int ntiles = (n + 255) >> 8;
    if (MULTI && f.split_chunks && ntiles < f.split_below) return;   // k_reflect_chunks takes this level
    // One chunk and too few tiles to fill the device: 64 records per workgroup, the scene spread over its waves.
    const bool spread = !MULTI && ntiles < f.spread_below;
    const int nwork = spread ? (n + 63) >> 6 : ntiles;
    if (!MULTI && (int)blockIdx.x < nwork) {
        stage_chunk(L, s, 0, s.ns_padded);
        __syncthreads();
    }
    TestCount tc = {0, 0, 0, 0, 0, 0, 0};
    PhaseClock pc = {};
#ifdef RT_PHASE_CLOCKS
    pc.t_last = __builtin_readcyclecounter();
#endif
    if constexpr (!MULTI) {
        if (spread) {
            __shared__ unsigned long long best[64];
            for (int tile = blockIdx.x; tile < nwork; tile += gridDim.x)
                reflection_tile_spread(s, f, L, blk, best, level, tile, n, tc, pc);
        }
    }
    if (!spread)
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) reflection_tile<MULTI>(s, f, L, blk, level, tile, n, tc, pc);
#ifdef RT_PHASE_CLOCKS
    if ((int)blockIdx.x < nwork) flush_counters(f, cnt_sh, pc.clk);
#else
    if constexpr (COUNT) flush_test_counts(f, cnt_sh, tc);
#endif
}

// ---------------------------------------------------------------------------------------
// Reflection rays of a scene of many chunks, one (tile, chunk) pair per job.  After the first
// bounce the rays of a tile point everywhere, so the tile's workgroup would stage and walk nearly
// every chunk of the scene one after the other (k_reflect: ~15 us per chunk, 64 chunks); here
// the pairs are independent jobs spread over the device.  That pays when the level has too few
// tiles to fill the device (below DevFrame::split_below: deep levels cost ~1 ms each whatever their
// size, against 30-500 us as jobs); a level with more tiles keeps k_reflect, whose workgroups walk
// their chunks nearest first and stop at the closest hit.  Both are launched for every level and
// the one whose turn it is not returns at once.  A job tests the tile's rays against the
// chunk's ball, and if any can meet it stages the chunk, walks it, and merges each ray's closest
// hit into the record's refl slot with a 64-bit atomic minimum of (bits of t) << 32 | host index:
// t > 0, so the order of the keys is the order of (t, index) that Scene.Intersect's strict `<`
// implies (MyApplication.cs:361).  k_reflect_finish then adds planes and triangles, and pushes
// the next level's records, as k_reflect does.
// ---------------------------------------------------------------------------------------

// The mirror direction of a record and what it starts from.
struct ReflectionRay {
    f3 hit, dir;
    float a;
    int slot;
    bool active;
};
__device__ __forceinline__ ReflectionRay reflection_ray(const DevScene &s, const DevFrame &f, const int level,
                                                        const int tile, const int n, const bool through)
{
    ReflectionRay r;
    const int i = tile * 256 + threadIdx.x;
    r.active = i < n;
    r.slot = level * f.P + (r.active ? i : n - 1);
    float4 ra, rb;
    if (through) load_record(&f.recA[r.slot], &f.recB[r.slot], ra, rb);
    else ra = f.recA[r.slot], rb = f.recB[r.slot];
    r.hit = xyz(ra);
    const f3 view = xyz(rb);
    const int prim = __float_as_int(ra.w);
    const int type = __float_as_int(cload(s.mat, 3 * prim + 1).w) & RT_MF_TYPE_MASK;
    const f3 normal = record_normal(s, prim, type, r.hit);
    // objects.cs:82: (viewDir - 2 * Dot(viewDir, normal) * normal).Normalized()
    r.dir = normalized3(sub3(view, scale3(normal, 2.0f * dot3(view, normal))));
    r.a = dot3(r.dir, r.dir);
    return r;
}

template <bool COUNT>
__global__ void __launch_bounds__(256) k_reflect_chunks(DevScene s, DevFrame f, int level)
{
    __shared__ WalkLds L;
    __shared__ unsigned long long cnt_sh[COUNTER_WORDS];
    const int n = f.cnt[2 * level];
    const int ntiles = (n + 255) >> 8;
    if (ntiles >= f.split_below) return;   // enough tiles to fill the device: k_reflect takes this level
    const int nsp = s.ns_padded, nchk = (nsp + RT_CHUNK - 1) / RT_CHUNK;
    const long long jobs = (long long)ntiles * (long long)nchk;
    TestCount tc = {0, 0, 0, 0, 0, 0, 0};
    PhaseClock pc = {};
    const unsigned live[4] = {~0u, ~0u, ~0u, ~0u};
    int tile_held = -1;
    ReflectionRay r = {};
#pragma clang loop unroll(disable)
    for (long long job = blockIdx.x; job < jobs; job += gridDim.x) {
        // Jobs of one tile are neighbours: chunk = job % nchk.
        const int tile = (int)(job / nchk), chunk = (int)(job % nchk);
        if (tile != tile_held) {
            r = reflection_ray(s, f, level, tile, n, false);
            tile_held = tile;
        }
        // (the closest hit other jobs have merged so far: a plain load may see an older, farther one)
        const unsigned long long seen = *reinterpret_cast<const unsigned long long *>(&f.refl[r.slot]);
        const float so_far = seen == ~0ull ? RT_FLT_MAX : __uint_as_float((unsigned)(seen >> 32));
        const bool meets = r.active && chunk_may_hit(s, chunk, r.hit, r.dir, so_far);
        if (!__syncthreads_or(meets ? 1 : 0)) continue;   // (also: everybody is done with the staged chunk)
        const int base = chunk * RT_CHUNK, cnt = min(RT_CHUNK, nsp - base);
        stage_chunk(L, s, base, cnt);
        __syncthreads();
        const FilterOrigin fo = filter_origin(s, r.hit);
        f4raw rq[4];
        rq[0] = filter_ray(fo, r.dir);
#pragma unroll
        for (int l = 1; l < 4; l++) rq[l] = f4raw{0.f, 0.f, 0.f, 0.f};
        float best_t = RT_FLT_MAX;
        int best_prim = INT_MAX;
        // Sphere.Intersect objects.cs:112-121, literally; equal distances: the lower host index.
        auto exact = [&](auto, const float4 sp, const int orig) {
            const f3 op = sub3(r.hit, xyz(sp));
            const float c = dot3(op, op) - sp.w;
            const float x = dot3(r.dir, op);
            float t;
            if (sphere_maybe(x, r.a * c) && sphere_root(x, c, r.a, t) && t > RT_EPS &&
                (t < best_t || (t == best_t && orig < best_prim))) {
                best_t = t;
                best_prim = orig;
            }
        };
        FilterWork fw = {0, 0, 0};
        walk_chunk<1, true, false>(L, cnt, s.sph + base, s.sph_orig + base, rq, fo.lane_s, fo.lane_g, fo.lane, fw, pc,
                                   exact, live);
        if (r.active && best_prim != INT_MAX)
            atomicMin(reinterpret_cast<unsigned long long *>(&f.refl[r.slot]), hit_key(best_t, best_prim));
        tc.exact += r.active ? fw.candidates : 0;
        tc.filt_terms += fw.terms;
        tc.filt_rays += fw.ray_tests;
    }
    if constexpr (COUNT) flush_test_counts(f, cnt_sh, tc);
}

template <bool COUNT>
__global__ void __launch_bounds__(256) k_reflect_finish(DevScene s, DevFrame f, int level)
{
    __shared__ int blk[12];
    __shared__ unsigned long long cnt_sh[COUNTER_WORDS];
    const int n = f.cnt[2 * level];
    const int ntiles = (n + 255) >> 8;
    if (ntiles >= f.split_below) return;
    TestCount tc = {0, 0, 0, 0, 0, 0, 0};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const ReflectionRay r = reflection_ray(s, f, level, tile, n, false);
        // the closest sphere, as the jobs of k_reflect_chunks left it
        const unsigned long long key = *reinterpret_cast<const unsigned long long *>(&f.refl[r.slot]);
        Closest sec = {RT_FLT_MAX, INT_MAX};
        if (key != ~0ull) sec.t = __uint_as_float((unsigned)(key >> 32)), sec.prim = (int)(unsigned)key;
        for (int j = 0; j < s.np; j++) {
            const float4 pl = cload(s.pl, j);
            const f3 nn = xyz(pl);
            float t;
            if (plane_t(nn, (-dot3(nn, r.hit)) - pl.w, r.dir, t)) closest_offer(sec, t, cload(s.pl_orig, j));
        }
        for (int j = 0; j < s.nt; j++) {
            const f3 v0 = xyz(cload(s.tri, 3 * j)), e1 = xyz(cload(s.tri, 3 * j + 1)), e2 = xyz(cload(s.tri, 3 * j + 2));
            float t;
            if (triangle_t(e1, e2, sub3(r.hit, v0), r.dir, t)) closest_offer(sec, t, cload(s.tri_orig, j));
        }
        if (r.active) {
            tc.sph += s.ns;
            tc.pl += s.np;
            tc.tri += s.nt;
        }
        const bool child = r.active && sec.prim != INT_MAX;
        const f3 hit2 = add3(r.hit, scale3(r.dir, sec.t));   // objects.cs:123,166,263
        if (r.active) f.refl[r.slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // no reflected colour yet
        const bool rv[1] = {child};
        const int rp[1] = {child ? sec.prim : 0}, rparent[1] = {r.slot};
        const f3 rh[1] = {hit2}, rd[1] = {r.dir};
        push_records<1, false>(f, s, level + 1, blk, rv, rp, rh, rd, rparent);
    }
    if constexpr (COUNT) flush_test_counts(f, cnt_sh, tc);
}

// ---------------------------------------------------------------------------------------
// Grid walk: the closest sphere along a reflection ray from the spheres' uniform grid (rt_device.h,
// DevScene::gr_*; built, and proven to list every sphere the reference can accept, in rt_api.cpp,
// build_sphere_grid).  One ray per lane: the cells along the ray are visited nearest first
// (three-dimensional DDA in cell coordinates; the exit parameter of a cell is recomputed from the
// plane's index at every step, so no error accumulates), Sphere.Intersect runs over each visited
// cell's list, and the walk stops when the closest hit so far - planes and triangles were offered
// before the walk - lies before the parameter at which the ray leaves the cell.  A sphere listed in
// several cells is met several times, which changes nothing: the result is the lexicographic
// minimum of (t, host index) over everything evaluated, exactly what Scene.Intersect's scan with its
// strict `<` keeps (MyApplication.cs:352-368).  Entries are loaded two at a time without looking
// at the list's end: lists start at even entries, and what follows a list is never-hit padding or
// another cell's spheres, and evaluating more spheres cannot change the result either.
//
// Rays are uneven (one meets a sphere in its first cell, its neighbour crosses the room), and a
// wave pays for its slowest lane, so two things keep the lanes busy:
//   * A workgroup takes a tile of up to GRID_TILE_MAX records and its lanes take rays from it as
//     they become free (a wave refills when half its lanes are idle); the results meet in LDS and
//     are pushed as the next level's records together, one counter atomic per tile.
//   * The IEEE square root and divide of Sphere.Intersect (~60 instructions, executed by a wave
//     whenever ANY lane has a candidate) are put off: a candidate's t is first bracketed with the
//     one-ulp hardware square root and reciprocal,
//         t_a = (-b - sqrt~(disc)) * rcp~(2a),   |t - t_a| <= delta = 1e-6 (|b| + sqrt~(disc) + |t_a|) + 1e-18
//     (t = the reference's own FP32 result: both evaluations share b, disc and 2a; the two square
//     roots differ by <= 3u sqrt(disc), the two subtractions round once each, <= 2u (|b| + sqrt),
//     the quotients by <= 4u |t|, u = 2^-24; 1e-18 covers an operand the hardware flushes to
//     zero), and the lane keeps ONE closest candidate as [lo, hi]: a new one with lo > hi_kept is
//     strictly farther and dropped, one with hi < lo_kept replaces it, and only brackets that
//     overlap (or straddle the epsilon of MyApplication.cs:359) are resolved with the literal
//     evaluation on the spot.  The kept candidate gets its literal evaluation once, when the lane's
//     walk is over, together with the other lanes that finished since the last refill.
// Origins outside the region the lists were built for evaluate every sphere of the scene.
//
// Small levels are as long as their longest walk - up to ~150 cells, one memory round trip each, ~100 us however
// few the rays - while nearly all lanes of the device idle.  There a ray's walk is cut into `segs` SEGMENTS of its
// parameter range [entry into the grid, exit or closest plane hit], each walked by a lane of its own: segment k
// starts at the cell that holds the point at the segment's start (a hundredth of a cell early, so that rounding
// cannot open a gap between two segments' cells) and ends after the first cell the ray leaves beyond the segment's
// end.  Every cell the ray crosses is visited by the segment its entry parameter falls into, every candidate is a
// true hit of the ray, and the lanes' results meet in a 64-bit minimum of (t, host index): the lexicographic
// minimum over everything evaluated, as before.  A segment that starts behind a hit already found is not walked.
// ---------------------------------------------------------------------------------------
#define GRID_TILE_MAX 512
#define GRID_SEGS_MAX 8
#ifndef GRID_REFILL_BELOW
#define GRID_REFILL_BELOW 32
#endif
struct GridLds {
    // per record of the tile: the closest hit as bits(t) << 32 | host index (t > 0, so the order of the keys is the
    // order Scene.Intersect's strict `<` implies; no hit: t = FLT_MAX, index INT_MAX), merged with a 64-bit minimum
    // when several lanes walk parts of one ray,
    unsigned long long key[GRID_TILE_MAX];
    float dx[GRID_TILE_MAX], dy[GRID_TILE_MAX], dz[GRID_TILE_MAX];   // and the mirror direction
    unsigned next;                // walks of the tile handed out so far
};
#define GRID_NO_HIT_KEY (((unsigned long long)0x7F7FFFFFu << 32) | 0x7FFFFFFFull)   // (FLT_MAX, INT_MAX)

// Sphere.Intersect objects.cs:112-121, literally; false: no hit the reference accepts.
__device__ __forceinline__ bool grid_exact(const float4 sp, const f3 o, const f3 d, const float a, float &t)
{
    const f3 op = sub3(o, xyz(sp));
    const float c = dot3(op, op) - sp.w;
    const float x = dot3(d, op);
    return sphere_maybe(x, a * c) && sphere_root(x, c, a, t) && t > RT_EPS;
}

// The closest candidate of a lane: [lo, hi] brackets the reference's t; e >= 0: entry of gr_sph whose
// literal evaluation is still owed, e < 0: lo = hi = the literal t (or nothing yet: prim = INT_MAX).
struct GridBest {
    float lo, hi;
    int prim, e;
};

__device__ __forceinline__ void grid_candidate(const DevScene &s, const float4 sp, const int orig, const unsigned e,
                                               const f3 o, const f3 d, const float a, GridBest &b)
{
    const f3 op = sub3(o, xyz(sp));
    const float c = dot3(op, op) - sp.w;
    const float x = dot3(d, op);
    if (!sphere_maybe(x, a * c)) return;
    const float bb = 2.0f * x;
    const float disc = (bb * bb) - ((4.0f * a) * c);   // (as sphere_root forms it)
    if (disc < 0.0f) return;
    const float sq = __builtin_amdgcn_sqrtf(disc);
    const float ta = ((-bb) - sq) * __builtin_amdgcn_rcpf(2.0f * a);
    const float delta = 1e-6f * ((fabsf(bb) + sq) + fabsf(ta)) + 1e-18f;
    const float lo = ta - delta, hi = ta + delta;
    if (hi <= RT_EPS || lo > b.hi || orig == b.prim) return;   // not a hit, strictly farther, or met before
    if (lo > RT_EPS && hi < b.lo) {   // strictly closer
        b.lo = lo, b.hi = hi, b.prim = orig, b.e = (int)e;
        return;
    }
    // The brackets overlap (or one straddles epsilon): the literal evaluation decides.
    if (b.e >= 0) {
        const float4 kept = s.gr_sph[b.e];
        const f3 kp = sub3(o, xyz(kept));
        float tk = b.lo;
        (void)sphere_root(dot3(d, kp), dot3(kp, kp) - kept.w, a, tk);   // (a hit: its bracket lies above epsilon)
        b.lo = b.hi = tk;
        b.e = -1;
    }
    float t;
    if (sphere_root(x, c, a, t) && t > RT_EPS && (t < b.lo || (t == b.lo && orig < b.prim)))
        b.lo = b.hi = t, b.prim = orig, b.e = -1;
}

// One axis of the ray in cell coordinates: where the ray is inside the grid's slab.
__device__ __forceinline__ void grid_slab(const float g, const float dg, const float r, const float n, float &tn, float &tf)
{
    const float ta = (0.0f - g) * r, tb = (n - g) * r;
    tn = fminf(ta, tb);
    tf = fmaxf(ta, tb);
    if (dg == 0.0f) {
        const bool in = g >= 0.0f && g <= n;
        tn = in ? -RT_FLT_MAX : RT_FLT_MAX;
        tf = in ? RT_FLT_MAX : -RT_FLT_MAX;
    }
}

#define GRID_EMPTY 0      // the lane has no ray
#define GRID_WALKING 1
#define GRID_FINISHED 2   // the walk is over, the kept candidate's literal evaluation and the result are owed

// The reflection rays of records [first, first + count) of a level's reflecting queue, count <= GRID_TILE_MAX,
// and the next level's records they create.  Every thread of the workgroup must call it.
// THROUGH: the records are read and written within one launch by several XCDs (k_trace_levels); otherwise
// a launch per level.
// segs: walks per ray (1: the whole ray; count * segs <= GRID_TILE_MAX * GRID_SEGS_MAX).
template <bool COUNT, bool THROUGH>
__device__ __forceinline__ void reflection_tile_grid(const DevScene &s, const DevFrame &f, GridLds &G, int *blk,
                                                     const int level, const int first, const int count, TestCount &tc,
                                                     const int segs = 1)
{
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) G.next = 0u;
    for (int j = threadIdx.x; j < count; j += blockDim.x) G.key[j] = GRID_NO_HIT_KEY;
    __syncthreads();
    const int walks = count * segs;
    float seg_end = RT_FLT_MAX;   // a walk of a segment ends after the first cell the ray leaves at or beyond this
    const int n0 = s.gr_n[0], n1 = s.gr_n[1], n2 = s.gr_n[2], ncells = n0 * n1 * n2;
    float leave = 0.f;   // parameter at which the ray leaves its current cell
    int state = GRID_EMPTY, my = 0, left = 0, cell = 0;
    unsigned e = 0u, end = 0u;
    f3 o = mk3(0.f, 0.f, 0.f), d = mk3(0.f, 0.f, 0.f);
    float a = 0.f, gx = 0.f, gy = 0.f, gz = 0.f, rx = 0.f, ry = 0.f, rz = 0.f, tmx = 0.f, tmy = 0.f, tmz = 0.f;
    GridBest best = {RT_FLT_MAX, RT_FLT_MAX, INT_MAX, -1};
    bool exhausted = false;
    int cells = 0, tests = 0;
    for (;;) {
        const int walking = __popcll(__ballot(state == GRID_WALKING));
        // Refill when half the lanes are idle; once the tile has no rays left, settle the finished lanes
        // together when the last walk is over.
        if (exhausted ? (walking == 0 && __any(state == GRID_FINISHED)) : walking <= GRID_REFILL_BELOW) {
            // ---- results of the lanes that finished since the last refill ----
            if (state == GRID_FINISHED) {
                float t = best.lo;
                if (best.e >= 0) {
                    const float4 sp = s.gr_sph[best.e];
                    const f3 op = sub3(o, xyz(sp));
                    (void)sphere_root(dot3(d, op), dot3(op, op) - sp.w, a, t);   // (a hit: its bracket lies above epsilon)
                }
                const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned)best.prim;
                if (segs == 1) G.key[my] = key;
                else if (best.prim != INT_MAX) atomicMin(&G.key[my], key);
                G.dx[my] = d.x, G.dy[my] = d.y, G.dz[my] = d.z;   // (every segment of a ray writes the same direction)
                state = GRID_EMPTY;
            }
            // ---- new rays for the idle lanes ----
            if (!exhausted) {
                const unsigned long long idle = __ballot(state == GRID_EMPTY);
                unsigned base = 0u;
                if (lane == 0) base = atomicAdd(&G.next, (unsigned)__popcll(idle));
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                exhausted = base + (unsigned)__popcll(idle) >= (unsigned)walks;
                const int mine = (int)base + __popcll(idle & ((1ull << lane) - 1ull));
                if (state == GRID_EMPTY && mine < walks) {
                    // (the first segments of all the tile's rays are handed out first)
                    const int seg = segs == 1 ? 0 : mine / count;
                    my = segs == 1 ? mine : mine - seg * count;
                    const int slot = level * f.P + first + my;
                    float4 ra, rb;
                    if (THROUGH) load_record(&f.recA[slot], &f.recB[slot], ra, rb);
                    else ra = f.recA[slot], rb = f.recB[slot];
                    o = xyz(ra);
                    const f3 view = xyz(rb);
                    const int prim = __float_as_int(ra.w);
                    const int type = __float_as_int(cload(s.mat, 3 * prim + 1).w) & RT_MF_TYPE_MASK;
                    const f3 normal = record_normal(s, prim, type, o);
                    // objects.cs:82: (viewDir - 2 * Dot(viewDir, normal) * normal).Normalized()
                    d = normalized3(sub3(view, scale3(normal, 2.0f * dot3(view, normal))));
                    a = dot3(d, d);
                    Closest sec = {RT_FLT_MAX, INT_MAX};
                    for (int j = 0; j < s.np; j++) {
                        const float4 pl = cload(s.pl, j);
                        const f3 nn = xyz(pl);
                        float t;
                        if (plane_t(nn, (-dot3(nn, o)) - pl.w, d, t)) closest_offer(sec, t, cload(s.pl_orig, j));
                    }
                    for (int j = 0; j < s.nt; j++) {
                        const f3 v0 = xyz(cload(s.tri, 3 * j)), e1 = xyz(cload(s.tri, 3 * j + 1)), e2 = xyz(cload(s.tri, 3 * j + 2));
                        float t;
                        if (triangle_t(e1, e2, sub3(o, v0), d, t)) closest_offer(sec, t, cload(s.tri_orig, j));
                    }
                    best.lo = best.hi = sec.t, best.prim = sec.prim, best.e = -1;
                    state = GRID_FINISHED;   // unless the walk below starts
                    const bool served = o.x >= s.gr_reg_lo[0] && o.x <= s.gr_reg_hi[0] && o.y >= s.gr_reg_lo[1] &&
                                        o.y <= s.gr_reg_hi[1] && o.z >= s.gr_reg_lo[2] && o.z <= s.gr_reg_hi[2];
                    if (!served) {
                        // An origin the lists were not built for: every sphere of the scene (the ray's first walk only).
                        for (int i = 0; i < (seg == 0 ? s.ns_padded : 0); i++) {
                            float t;
                            if (grid_exact(s.sph[i], o, d, a, t)) {
                                const int orig = s.sph_orig[i];
                                if (t < best.lo || (t == best.lo && orig < best.prim)) best.lo = best.hi = t, best.prim = orig;
                            }
                        }
                        if (seg == 0) tests += s.ns_padded;
                    } else {
                        // The ray in cell coordinates and where it is inside the grid.
                        gx = (o.x - s.gr_lo[0]) * s.gr_ih[0], gy = (o.y - s.gr_lo[1]) * s.gr_ih[1], gz = (o.z - s.gr_lo[2]) * s.gr_ih[2];
                        const float ddx = d.x * s.gr_ih[0], ddy = d.y * s.gr_ih[1], ddz = d.z * s.gr_ih[2];
                        // (a zero component: its reciprocal is an infinity of the direction's sign, which the
                        // slab test and the exit parameters below treat as "never crossed")
                        rx = __builtin_amdgcn_rcpf(ddx), ry = __builtin_amdgcn_rcpf(ddy), rz = __builtin_amdgcn_rcpf(ddz);
                        float tnx, tfx, tny, tfy, tnz, tfz;
                        grid_slab(gx, ddx, rx, (float)n0, tnx, tfx);
                        grid_slab(gy, ddy, ry, (float)n1, tny, tfy);
                        grid_slab(gz, ddz, rz, (float)n2, tnz, tfz);
                        const float t_in = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f)), t_out = fminf(fminf(tfx, tfy), tfz);
                        bool walk = t_in <= t_out && !(best.hi < t_in);   // the ray meets the grid before its closest hit so far
                        float t_from = t_in;
                        seg_end = RT_FLT_MAX;
                        if (segs > 1 && walk) {
                            // This lane's part of the walk: segment `seg` of [t_in, where the ray leaves the grid or meets
                            // its closest plane or triangle]; the last segment has no end of its own.
                            const float t_to = fminf(t_out, best.hi), step = (t_to - t_in) * (1.0f / (float)segs);
                            if (seg > 0) {
                                // a hundredth of a cell early along the axis the ray crosses cells fastest on
                                const float back = 0.01f * fminf(fminf(fabsf(rx), fabsf(ry)), fabsf(rz));
                                t_from = fmaxf(t_in, (t_in + step * (float)seg) - back);
                                // (not walked when a hit before this segment is known already; the plain read may miss
                                // a minimum another wave is about to merge, which costs a walk, not the result)
                                walk = !(__uint_as_float((unsigned)(G.key[my] >> 32)) < t_from) && step > 0.0f;
                            }
                            if (seg + 1 < segs) seg_end = t_in + step * (float)(seg + 1);
                        }
                        if (walk) {
                            const int ix = min(max((int)floorf(gx + t_from * ddx), 0), n0 - 1),
                                      iy = min(max((int)floorf(gy + t_from * ddy), 0), n1 - 1),
                                      iz = min(max((int)floorf(gz + t_from * ddz), 0), n2 - 1);
                            cell = ix | (iy << 10) | (iz << 20);
                            // parameter at which the ray leaves the current cell along each axis
                            tmx = ddx == 0.0f ? RT_FLT_MAX : ((float)(ix + (rx > 0.0f ? 1 : 0)) - gx) * rx;
                            tmy = ddy == 0.0f ? RT_FLT_MAX : ((float)(iy + (ry > 0.0f ? 1 : 0)) - gy) * ry;
                            tmz = ddz == 0.0f ? RT_FLT_MAX : ((float)(iz + (rz > 0.0f ? 1 : 0)) - gz) * rz;
                            leave = fminf(fminf(tmx, tmy), tmz);
                            const uint2 cur = s.gr_cells[(iz * n1 + iy) * n0 + ix + (leave <= s.gr_tier_t ? ncells : 0)];
                            e = cur.x, end = cur.x + cur.y;
                            left = n0 + n1 + n2 + 3;   // (a ray of NaNs must end too)
                            cells++;
                            state = GRID_WALKING;
                        }
                    }
                }
            }
        }
        if (!__any(state == GRID_WALKING)) {
            if (exhausted && !__any(state == GRID_FINISHED)) break;   // nothing left in the tile for this wave
            continue;
        }
        // ---- one step of every walking lane: two entries of its cell, then on to the next cell ----
        // (Tried and dropped, each slower by 10-50 %: the next cell's list fetched one step ahead; several
        // cells per step for the lanes that cross empty ones; a 64-byte block per cell holding its first two
        // spheres, so that moving on and testing is one fetch - empty cells then cost two tests each.  The
        // loop is bound by the instructions it issues, not by the latency of its loads.)
        if (state == GRID_WALKING) {
            // When this step ends the cell's list, the next cell's word is fetched BEFORE the entries are
            // looked at, so that the two loads are in flight together (a walk is a chain of dependent loads,
            // and the deep levels of a frame are as long as their longest walk); whether the walk goes on is
            // only known after the entries, and a word fetched in vain is dropped.
            const bool last = e + 2u >= end;
            int ncell = 0;
            float ntm = 0.f, nleave = 0.f;
            bool bx = false, by = false, inside = false;
            uint2 nxt = make_uint2(0u, 0u);
            if (last) {
                bx = tmx <= tmy && tmx <= tmz, by = !bx && tmy <= tmz;
                int ix = cell & 1023, iy = (cell >> 10) & 1023, iz = cell >> 20;
                if (bx) ix += rx > 0.0f ? 1 : -1;
                else if (by) iy += ry > 0.0f ? 1 : -1;
                else iz += rz > 0.0f ? 1 : -1;
                inside = (unsigned)ix < (unsigned)n0 && (unsigned)iy < (unsigned)n1 && (unsigned)iz < (unsigned)n2;
                ncell = ix | (iy << 10) | (iz << 20);
                if (bx) ntm = ((float)(ix + (rx > 0.0f ? 1 : 0)) - gx) * rx, nleave = fminf(fminf(ntm, tmy), tmz);
                else if (by) ntm = ((float)(iy + (ry > 0.0f ? 1 : 0)) - gy) * ry, nleave = fminf(fminf(tmx, ntm), tmz);
                else ntm = ((float)(iz + (rz > 0.0f ? 1 : 0)) - gz) * rz, nleave = fminf(fminf(tmx, tmy), ntm);
                // (the short lists while the ray leaves the cell early enough: rt_device.h, gr_tier_t)
                if (inside) nxt = s.gr_cells[(iz * n1 + iy) * n0 + ix + (nleave <= s.gr_tier_t ? ncells : 0)];
            }
            if (e < end) {
                const float4 s0 = s.gr_sph[e], s1 = s.gr_sph[e + 1];
                const int2 og = *reinterpret_cast<const int2 *>(&s.gr_orig[e]);
                grid_candidate(s, s0, og.x, e, o, d, a, best);
                grid_candidate(s, s1, og.y, e + 1u, o, d, a, best);
                e += 2u;
                tests += 2;
            }
            if (last) {
                if (best.hi < leave || !inside || --left <= 0 || leave >= seg_end) {
                    state = GRID_FINISHED;
                } else {
                    cell = ncell;
                    if (bx) tmx = ntm;
                    else if (by) tmy = ntm;
                    else tmz = ntm;
                    leave = nleave;
                    e = nxt.x, end = nxt.x + nxt.y;
                    cells++;
                }
            }
        }
    }
    __syncthreads();
    // ---- the next level's records ----
    constexpr int PER = GRID_TILE_MAX / 256;
    bool rv[PER];
    int rp[PER], rparent[PER];
    f3 rh[PER], rd[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int j = k * 256 + (int)threadIdx.x;
        const bool active = j < count;
        const int jj = active ? j : 0;
        const int slot = level * f.P + first + jj;
        const unsigned long long key = G.key[jj];
        const int hit_prim = (int)(unsigned)key;
        const float t = __uint_as_float((unsigned)(key >> 32));
        rd[k] = mk3(G.dx[jj], G.dy[jj], G.dz[jj]);
        f3 hit = mk3(0.f, 0.f, 0.f);
        if (active) {
            float4 ra, rb;
            if (THROUGH) load_record(&f.recA[slot], &f.recB[slot], ra, rb);
            else ra = f.recA[slot];
            hit = xyz(ra);
            // no reflected colour yet (see reflection_tile for the store's form)
            if (THROUGH) store_record(&f.refl[slot], f4raw{0.0f, 0.0f, 0.0f, 0.0f});
            else f.refl[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        rv[k] = active && hit_prim != INT_MAX;
        rp[k] = rv[k] ? hit_prim : 0;
        rparent[k] = slot;
        rh[k] = add3(hit, scale3(rd[k], t));   // objects.cs:123,166,263
        if (COUNT && active) {
            tc.sph += s.ns;
            tc.pl += s.np;
            tc.tri += s.nt;
        }
    }
    if (COUNT) {
        tc.exact += tests;
        tc.grid += cells;
    }
    push_records<PER, THROUGH, !THROUGH>(f, s, level + 1, blk, rv, rp, rh, rd, rparent);
}

#ifndef RT_GRID_WAVES
#define RT_GRID_WAVES 6
#endif
template <bool COUNT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RT_GRID_WAVES)))
k_reflect_grid(DevScene s, DevFrame f, int level)
{
    __shared__ GridLds G;
    __shared__ int blk[12];
    __shared__ unsigned long long cnt_sh[COUNTER_WORDS];
    const int n = f.cnt[2 * level];
    // Levels too small to fill the device: every ray's walk in segments, so many that the level comes to about
    // grid_split_below walks (DevFrame), and tiles of 256 walks.
    const int segs = n > 0 && n < f.grid_split_below ? max(1, min(GRID_SEGS_MAX, f.grid_split_below / n)) : 1;
    // Large levels: GRID_TILE_MAX records per tile (half as many contended counter atomics); small ones: 256, so
    // that more workgroups share the level.
    const int per_tile = segs > 1 ? 256 / segs : n >= GRID_TILE_MAX * (int)gridDim.x ? GRID_TILE_MAX : 256;
    TestCount tc = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long long first = (long long)blockIdx.x * per_tile; first < n; first += (long long)gridDim.x * per_tile)
        reflection_tile_grid<COUNT, false>(s, f, G, blk, level, (int)first, min(per_tile, n - (int)first), tc, segs);
    if constexpr (COUNT) flush_test_counts(f, cnt_sh, tc);
}

// ---------------------------------------------------------------------------------------
// Record binning.  The records of level 0 lie in the order of the pixels, so the 64 records of a
// wave hit neighbouring points: their light rays share table cells and occluder lists, their
// reflection rays share chunks.  The records of deeper levels lie in the order their parents'
// tiles happened to push them, and neighbours have nothing in common: on the 64k-sphere scene a
// plain record of level 1 costs 2.6 ns against 0.55 ns at level 0, with the same number of exact
// tests per ray.  Before a level >= 1 of enough records is used, its records are therefore put
// into the order of the position bins of their hit points (16 x 16 x 16 cells over the spheres'
// extent, in Morton order so that consecutive bins are neighbours): count per bin with each
// record's rank taken from the counting atomic, prefix sums, scatter into a spare level, copy back.  The grid
// walk's launches write a level's records to the spare level in the first place (push_records, STAGE), and the
// scatter moves them to their slots: no pass over the records beyond the one that orders them.  The order of records changes nothing observable: a record is shaded by itself, children
// are created after the move and name their parents by the new slots, and a record's own parent
// field moves with it.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned spread4(unsigned v)   // bits abcd -> a00b00c00d
{
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6);
}
__device__ __forceinline__ int position_bin(const DevScene &s, const f3 p)
{
    unsigned c[3];
    const float v[3] = {p.x, p.y, p.z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float t = (v[k] - s.bin_lo[k]) * s.bin_scale[k];
        c[k] = !(t > 0.0f) ? 0u : (t >= (float)(RT_BIN_AXIS - 1) ? (unsigned)(RT_BIN_AXIS - 1) : (unsigned)t);   // NaN -> 0
    }
    return (int)(spread4(c[0]) | (spread4(c[1]) << 1) | (spread4(c[2]) << 2));
}
// Record i of the level's two queues taken together (reflecting first): its slot, and where push_records
// staged it in the binning passes' buffers.
__device__ __forceinline__ int binned_slot(const DevFrame &f, const int level, const int na, const int i)
{
    return i < na ? level * f.P + i : (level + 1) * f.P - 1 - (i - na);
}
__device__ __forceinline__ int staged_at(const DevFrame &f, const int na, const int i)
{
    return i < na ? i : f.P - 1 - (i - na);
}

// A workgroup counts a block of BIN_BLOCK records in LDS first and adds each bin it met to the global
// histogram with ONE atomic, whose return value places the block's records of that bin behind those
// of other blocks: neighbouring records often share a bin, and one atomic per record on such a bin's
// counter serialises (688k records: 340 us against 30).
#define BIN_PER_THREAD 8
#define BIN_BLOCK (256 * BIN_PER_THREAD)
__global__ void __launch_bounds__(256) k_bin_count(DevScene s, DevFrame f, int level)
{
    __shared__ int local[2 * RT_BINS];
    const int na = f.cnt[2 * level], n = na + f.cnt[2 * level + 1];
    if (n < f.bin_min) return;
    for (int first = blockIdx.x * BIN_BLOCK; first < n; first += gridDim.x * BIN_BLOCK) {
        for (int b = threadIdx.x; b < 2 * RT_BINS; b += 256) local[b] = 0;
        __syncthreads();
        int bin[BIN_PER_THREAD], rank[BIN_PER_THREAD];
#pragma unroll
        for (int r = 0; r < BIN_PER_THREAD; r++) {
            const int i = first + r * 256 + threadIdx.x;
            bin[r] = -1;
            if (i < n) {
                const float4 ra = f.recA[f.staged ? f.stage_base + staged_at(f, na, i) : binned_slot(f, level, na, i)];
                bin[r] = (i < na ? 0 : RT_BINS) + position_bin(s, xyz(ra));
                rank[r] = atomicAdd(&local[bin[r]], 1);
            }
        }
        __syncthreads();
        for (int b = threadIdx.x; b < 2 * RT_BINS; b += 256) {
            const int c = local[b];
            if (c) local[b] = atomicAdd(&f.bin_hist[b], c);   // now: where this block's records of the bin start
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < BIN_PER_THREAD; r++) {
            const int i = first + r * 256 + threadIdx.x;
            if (bin[r] >= 0) f.bin_aux[i] = make_uint2((unsigned)bin[r], (unsigned)(local[bin[r]] + rank[r]));
        }
        __syncthreads();
    }
}

// One workgroup of 1024: exclusive prefix sums of the two queues' histograms, which are zeroed for
// the next level.
__global__ void __launch_bounds__(1024) k_bin_scan(DevFrame f, int level)
{
    __shared__ int wave_total[16];
    const int n = f.cnt[2 * level] + f.cnt[2 * level + 1];
    if (n < f.bin_min) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PER = RT_BINS / 1024;
    for (int q = 0; q < 2; q++) {
        int *hist = f.bin_hist + q * RT_BINS, *base = f.bin_base + q * RT_BINS;
        int v[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            v[k] = hist[threadIdx.x * PER + k];
            hist[threadIdx.x * PER + k] = 0;
            sum += v[k];
        }
        int incl = sum;   // inclusive scan over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wave_total[wave] = incl;
        __syncthreads();
        int before = incl - sum;
        for (int w = 0; w < wave; w++) before += wave_total[w];
#pragma unroll
        for (int k = 0; k < PER; k++) {
            base[threadIdx.x * PER + k] = before;
            before += v[k];
        }
        __syncthreads();
    }
}

// The level's records into bin order.  Staged records (DevFrame::staged) go from the spare level straight to their
// slots - as they lie when the level is too short for the order -; others are put into order in the spare level
// and copied back by k_bin_copy.
__global__ void __launch_bounds__(256) k_bin_scatter(DevFrame f, int level)
{
    const int na = f.cnt[2 * level], n = na + f.cnt[2 * level + 1];
    const bool in_order = n >= f.bin_min;
    if (!f.staged && !in_order) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        int to = i;
        if (in_order) {
            const uint2 a = f.bin_aux[i];
            to = (i < na ? 0 : na) + f.bin_base[a.x] + (int)a.y;   // position among the level's records
        }
        const int from = f.staged ? f.stage_base + staged_at(f, na, i) : binned_slot(f, level, na, i);
        const int slot = f.staged ? binned_slot(f, level, na, to) : f.stage_base + to;
        f.recA[slot] = f.recA[from];
        f.recB[slot] = f.recB[from];
    }
}

__global__ void __launch_bounds__(256) k_bin_copy(DevFrame f, int level)
{
    const int na = f.cnt[2 * level], n = na + f.cnt[2 * level + 1];
    if (n < f.bin_min) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int slot = binned_slot(f, level, na, i);
        f.recA[slot] = f.recA[f.stage_base + i];
        f.recB[slot] = f.recB[f.stage_base + i];
    }
}

#ifndef RT_PLAIN_WAVES
#define RT_PLAIN_WAVES 5
#endif
template <bool COUNT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RT_PLAIN_WAVES)))
k_plain(DevScene s, DevFrame f, int level, int32_t *pixels)
{
    __shared__ unsigned long long cnt_sh[COUNTER_WORDS];
    const int n = f.cnt[2 * level + 1];
    if (f.light_waves > 0 && n < f.light_split_below) return;   // k_shade_lights takes this queue
    const int ntiles = (n + 255) >> 8;
    TestCount tc = {0, 0, 0, 0, 0, 0, 0};
    PhaseClock pc = {};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) plain_tile<COUNT>(s, f, level, tile, n, pixels, tc, pc);
    if constexpr (COUNT) flush_test_counts(f, cnt_sh, tc);
}

// The last small levels of a frame in ONE launch (levels level .. level_hi: their plain records, and
// their reflecting records' light terms when DevFrame::lit is set): a launch of its own for each
// would cost more in start-up than in work.  (A kernel of its own, so that k_plain keeps its
// registers.)
template <bool COUNT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RT_PLAIN_WAVES)))
k_plain_tail(DevScene s, DevFrame f, int level, int level_hi, int32_t *pixels)
{
    __shared__ unsigned long long cnt_sh[COUNTER_WORDS];
    TestCount tc = {0, 0, 0, 0, 0, 0, 0};
    PhaseClock pc = {};
    // The tiles of all the queues as ONE list of jobs - level by level, the plain queue and then the reflecting
    // queue's light terms -, so that a workgroup's turns are spread over the queues instead of every workgroup
    // working through one tile of each queue in a row (four chains of dependent reads one after the other: the
    // launch took 37 us for half a million records of the 3840x2160 frame's levels 2 and 3).
    // (the queues' lengths once per workgroup, in LDS: read from memory for every job they cost a level of a million
    // records a third of its time)
    __shared__ int q_nb[RT_DEV_MAX_LEVELS], q_na[RT_DEV_MAX_LEVELS], q_tiles_b[RT_DEV_MAX_LEVELS], q_tiles_a[RT_DEV_MAX_LEVELS];
    if ((int)threadIdx.x >= level && (int)threadIdx.x <= level_hi) {
        const int lv = threadIdx.x;
        const int nb = f.cnt[2 * lv + 1], na = f.cnt[2 * lv];
        q_nb[lv] = nb, q_na[lv] = na;
        q_tiles_b[lv] = !(f.light_waves > 0 && nb < f.light_split_below) ? (nb + 255) >> 8 : 0;   // (else k_shade_lights takes this queue)
        q_tiles_a[lv] = lv < f.max_depth && lit_ahead(f, na) && !(f.light_waves > 0 && na < f.light_split_below) ? (na + 255) >> 8 : 0;
    }
    __syncthreads();
    int total = 0;
    for (int lv = level; lv <= level_hi; lv++) total += q_tiles_b[lv] + q_tiles_a[lv];
    for (int job = blockIdx.x; job < total; job += gridDim.x) {
        int tile = job;
        for (int lv = level; lv <= level_hi; lv++) {
            const int nb = q_nb[lv], na = q_na[lv];
            const int tiles_b = q_tiles_b[lv];
            if (tile < tiles_b) {
                plain_tile<COUNT>(s, f, lv, tile, nb, pixels, tc, pc);
                break;
            }
            tile -= tiles_b;
            const int tiles_a = q_tiles_a[lv];
            if (tile < tiles_a) {
                const int i = tile * 256 + threadIdx.x;
                lit_record<COUNT>(s, f, lv, lv * f.P + (i < na ? i : na - 1), i < na, tc);
                break;
            }
            tile -= tiles_a;
        }
    }
    if constexpr (COUNT) flush_test_counts(f, cnt_sh, tc);
}

// The light terms of a level's reflecting records (lit_record), one record per lane.
template <bool COUNT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RT_PLAIN_WAVES)))
k_lit(DevScene s, DevFrame f, int level)
{
    __shared__ unsigned long long cnt_sh[COUNTER_WORDS];
    const int n = f.cnt[2 * level];
    if (!lit_ahead(f, n)) return;
    if (f.light_waves > 0 && n < f.light_split_below) return;   // k_shade_lights (reflecting = 2) takes this queue
    const int ntiles = (n + 255) >> 8;
    TestCount tc = {0, 0, 0, 0, 0, 0, 0};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int i = tile * 256 + threadIdx.x;
        lit_record<COUNT>(s, f, level, level * f.P + (i < n ? i : n - 1), i < n, tc);
    }
    if constexpr (COUNT) flush_test_counts(f, cnt_sh, tc);
}

// ---------------------------------------------------------------------------------------
// trace_levels: every level of the frame in ONE launch of resident workgroups that take
// tiles from a shared schedule (DevSched).  Level k+1's records exist once all reflecting
// tiles ("A") of level k are done, so the A tiles form a serial chain of ever smaller
// levels; the plain tiles ("B", most of the work) have no successors.  A workgroup takes
// the next A tile of the newest open level when there is one, otherwise a B tile of any
// open level, and sleeps only when neither exists while A tiles are still in flight.  The
// chip therefore stays full across level boundaries, where separate launches per level
// leave it to a shrinking tail of workgroups.
//
// Protocol (cache maintenance is the cost to avoid: a device-scope release fence writes the
// whole L2 of the XCD back and an acquire fence invalidates it, and a fence pair per tile was
// measured at ~45 us per tile, more than the tile's own work):
//   * hit records are written with device-scope stores (push_records, store_record), which
//     go through to memory, and counted with device-scope atomics; a workgroup that finished
//     an A tile waits for its stores (vmcnt 0 in every wave, then a barrier) and only then
//     counts the tile as done; the workgroup that counts the level's last tile publishes
//     open_level = k+1.  No write-back fence is needed.
//   * a taker (thread 0) reads open_level with a device-scope load on every claim and reads a
//     newly opened level's record counts before it hands tiles of that level to the workgroup.
//     No cache invalidation is needed either: the records are read with device-scope loads
//     (load_record), a reflected colour and the zero it replaces are both written through
//     (store_record), and the levels' slot ranges start at multiples of RT_LEVEL_ALIGN slots,
//     so no cache line of a per-slot array holds slots of two levels.
// The chain ends at the first open level without A records (level max_depth never has any);
// a workgroup leaves when the chain has ended and no B tile is left to claim.  A workgroup
// that has waited DevFrame::stall_ticks (wall clock, about a second) for a level to open without
// seeing any progress sets sched->error, reports the level it waited for in the host-visible
// status block (DevFrame::status) and leaves, so that no launch can spin forever and the host
// learns that the frame is incomplete whichever entry point enqueued it.
// ---------------------------------------------------------------------------------------
#define SCHED_JOB_EXIT 0
#define SCHED_JOB_A 1
#define SCHED_JOB_B 2

__device__ __forceinline__ int sched_load(const int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Thread 0's view of the schedule.
struct Taker {
    int rank;             // how many workgroups of the launch were on this compute unit before this one
    int seen_open;        // levels <= seen_open: record counts read
    int b_level;          // lowest level whose B queue may still hold an unclaimed tile
    bool a_left;          // level seen_open may still hold an unclaimed A tile
};

// RESERVE: the chain tiles are kept to the oldest workgroups of the compute units (see the claim
// loop); an instance of its own, because the bookkeeping costs the plain instance ~1.5 % of a
// large frame in register pressure.
// COUNT: the work counters of rt_stats are kept (profiling); without them the compiler drops
// the per-lane sums and the per-light prefix lookups they are made of.
// GRID: the reflection rays walk the spheres' grid (reflection_tile_grid) instead of the staged scene.
template <bool MULTI, bool RESERVE, bool COUNT, bool GRID = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4)))
k_trace_levels(DevScene s, DevFrame f, int32_t *pixels)
{
    __shared__ WalkLds L;
    __shared__ GridLds G;   // (only one of the two is used, and only that one allocated)
    __shared__ int blk[12];
    __shared__ unsigned long long cnt_sh[COUNTER_WORDS];
    __shared__ int job[4];
    // Record counts of the seen levels and what is left to claim of their plain tiles (thread 0).
    __shared__ int lv_a[RT_DEV_MAX_LEVELS], lv_b[RT_DEV_MAX_LEVELS], lv_jobs[RT_DEV_MAX_LEVELS];
    DevSched *const sc = f.sched;
    const int D = f.max_depth;
    bool staged = false;
    TestCount tc = {0, 0, 0, 0, 0, 0, 0};
    PhaseClock pc = {};
#ifdef RT_PHASE_CLOCKS
    pc.t_last = __builtin_readcyclecounter();
#endif
    Taker tk = {0, -1, 0, false};
    if constexpr (RESERVE) {
        if (threadIdx.x == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            tk.rank = atomicAdd(&sc->cu_arrivals[((xcc & 7u) << 8) | ((hw >> 8) & 255u)], 1);
        }
    }
#ifdef RT_SCHED_TIMELINE   // profiling builds only: two atomics per workgroup on one word cost ~20 us
    if (threadIdx.x == 0) atomicCAS(&sc->t_first, 0ull, (unsigned long long)wall_clock64());
#endif
    for (;;) {
        if (threadIdx.x == 0) {
            int kind = SCHED_JOB_EXIT, lvl = 0, tile = 0, n = 0;
            unsigned long long wait_since = 0ull;   // wall clock at the first idle poll of this wait
            for (;;) {
                // What this workgroup believes is claimable, A tiles of the newest level first.
                int *ctr = nullptr;
                int cand_kind = SCHED_JOB_EXIT, cand_lvl = 0, cand_jobs = 0;
                // The SIMDs issue for their oldest waves first (s_setprio changes nothing there):
                // a tile the next level waits for belongs in a workgroup that arrived early on
                // its compute unit — the first of each while they suffice for the level's tiles.
                bool chain_wg = true, keep_free = false;
                if constexpr (RESERVE) {
                    chain_wg = tk.rank == 0 || (tk.seen_open >= 0 && tk.rank * f.num_cus < ((lv_a[tk.seen_open] + 255) >> 8));
                    // While the chain lasts the first workgroup of a compute unit keeps itself free
                    // for the next level's tiles instead of taking a plain job.
                    keep_free = tk.rank == 0 && tk.seen_open >= 0 && lv_a[tk.seen_open] > 0;
                }
                if (tk.a_left && chain_wg) {
                    ctr = &sc->next_a[tk.seen_open];
                    cand_kind = SCHED_JOB_A, cand_lvl = tk.seen_open, cand_jobs = (lv_a[tk.seen_open] + 255) >> 8;
                } else if (!keep_free) {
                    while (tk.b_level <= tk.seen_open && lv_jobs[tk.b_level] < 0) tk.b_level++;
                    if (tk.b_level <= tk.seen_open) {
                        ctr = &sc->next_b[tk.b_level];
                        cand_kind = SCHED_JOB_B, cand_lvl = tk.b_level, cand_jobs = lv_jobs[tk.b_level];
                    }
                }
                // One round trip: the level check and the claim travel together.
                const int open = sched_load(&sc->open_level);
                const int t = ctr ? atomicAdd(ctr, 1) : 0;
                const bool claimed = ctr && t < cand_jobs;
                if (ctr && !claimed) {   // that queue is exhausted
                    if (cand_kind == SCHED_JOB_A) tk.a_left = false;
                    else lv_jobs[cand_lvl] = -1;
                }
                const bool opened = open > tk.seen_open;
                if (opened) {
                    for (int k = tk.seen_open + 1; k <= open; k++) {
                        lv_a[k] = k < D ? sched_load(&f.cnt[2 * k]) : 0;
                        lv_b[k] = sched_load(&f.cnt[2 * k + 1]);
                        const int jobs = (lv_b[k] + 255) >> 8;
                        lv_jobs[k] = jobs > 0 ? jobs : -1;   // -1: nothing (left) to claim
                    }
                    tk.seen_open = open;
                    tk.a_left = lv_a[open] > 0;
                }
                if (claimed) {
                    kind = cand_kind, lvl = cand_lvl, tile = t;
                    n = kind == SCHED_JOB_A ? lv_a[lvl] : lv_b[lvl];
                    break;
                }
                if (ctr || opened) {
                    wait_since = 0ull;   // progress: a queue ran dry or a level opened
                    continue;
                }
                // Nothing to claim in the levels seen so far.
                if (tk.seen_open >= 0 && lv_a[tk.seen_open] == 0) break;   // the chain has ended: leave
                // A tiles of the newest level are in flight (or the frame has not started): wait.
                const unsigned long long now = wall_clock64();
                if (wait_since == 0ull) wait_since = now;
                else if (now - wait_since > (unsigned long long)f.stall_ticks) {
                    atomicExch(&sc->error, 1);
                    // Host-visible (page-locked host memory): the level, then the flag.
                    __hip_atomic_store(&f.status[1], tk.seen_open + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(&f.status[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                __builtin_amdgcn_s_sleep(127);
            }
            job[0] = kind;
            job[1] = lvl;
            job[2] = tile;
            job[3] = n;
        }
        __syncthreads();
        const int kind = job[0], lvl = job[1], tile = job[2], n = job[3];
        if (kind == SCHED_JOB_EXIT) break;
        // Only reflection rays walk the staged spheres: stage them before the first such tile.
        if constexpr (!GRID) {
            if (!MULTI && !staged && kind == SCHED_JOB_A) {
                stage_chunk(L, s, 0, s.ns_padded);
                staged = true;
            }
        }
        __syncthreads();   // job[] read by all, spheres staged
        if (kind == SCHED_JOB_A) {
            __builtin_amdgcn_s_setprio(3);   // the chain of levels is the launch's critical path
            if constexpr (GRID) reflection_tile_grid<COUNT, true>(s, f, G, blk, lvl, tile * 256, min(256, n - tile * 256), tc);
            else reflection_tile<MULTI>(s, f, L, blk, lvl, tile, n, tc, pc);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's record stores have reached memory
            __builtin_amdgcn_s_setprio(0);
            __syncthreads();
            if (threadIdx.x == 0) {
                const int tiles_a = (n + 255) >> 8;
                // (debug_withhold: a test knob that loses one tile's count, so that the next level never opens)
                if (!(lvl == f.debug_withhold && tile == 0) && atomicAdd(&sc->done_a[lvl], 1) + 1 == tiles_a) {
#ifdef RT_SCHED_TIMELINE
                    sc->t_open[lvl + 1] = (unsigned long long)wall_clock64();
#endif
                    __hip_atomic_store(&sc->open_level, lvl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        } else {
            plain_tile<COUNT>(s, f, lvl, tile, n, pixels, tc, pc);
        }
    }
#ifdef RT_SCHED_TIMELINE
    if (threadIdx.x == 0) atomicMax(&sc->t_last, (unsigned long long)wall_clock64());
#endif
#ifdef RT_PHASE_CLOCKS
    PHASE_MARK(pc, 0)
    flush_counters(f, cnt_sh, pc.clk);
#else
    if constexpr (COUNT) flush_test_counts(f, cnt_sh, tc);
#endif
}

// ---------------------------------------------------------------------------------------
// shade_level: the records of a level that traced a reflection ray (queue "A"), once the level
// below has been shaded (launches run from the deepest level up); the plain records were shaded
// by the level kernel.
// ---------------------------------------------------------------------------------------
template <bool COUNT>
__global__ void __launch_bounds__(256) k_shade(DevScene s, DevFrame f, int level, int alone, int32_t *pixels)
{
    __shared__ unsigned long long cnt_sh[COUNTER_WORDS];
    const int na = f.cnt[2 * level];
    // The frame's last launch leaves the levels' record counts where the host can read them without asking the device
    // (DevFrame::hint).
    if (level == 0 && f.hint != nullptr && blockIdx.x == 0 && (int)threadIdx.x <= 2 * (f.max_depth + 1)) {
        const int v = threadIdx.x == 0 ? f.hint_tag : f.cnt[threadIdx.x - 1];
        __hip_atomic_store(&f.hint[threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (lit_ahead(f, na)) {   // the light terms are there (k_lit): only the sums are left
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < na; i += gridDim.x * blockDim.x)
            combine_record(s, f, level, level * f.P + i, pixels);
        return;
    }
    // k_shade_lights takes this queue - unless the host has issued no such launch for this level (alone: it expects the
    // light terms to be there, on a hint; whatever the queue turns out to be, it is shaded here)
    if (!alone && f.light_waves > 0 && na < f.light_split_below) return;
    const int stride = gridDim.x * blockDim.x;
    TestCount tc = {0, 0, 0, 0, 0, 0, 0};
    // (whole waves: the light rays of a record are traced by its wave together)
    for (int i0 = blockIdx.x * blockDim.x; i0 < na; i0 += stride) {
        const int i = i0 + threadIdx.x;
        const bool active = i < na;
        shade_record<COUNT>(s, f, level, level * f.P + (active ? i : na - 1), active, true, false, pixels, tc);
    }
    if constexpr (COUNT) flush_test_counts(f, cnt_sh, tc);
}

// ---------------------------------------------------------------------------------------
// The sums of every level from level_lo down in ONE launch, for frames whose light terms are all computed ahead
// (DevFrame::lit with lit_below <= 0) and whose reflecting records carry RT_REFL_PENDING (DevFrame::path_shade).
// A reflection ray has one child, so the reflecting records of a pixel form a path; the thread of the path's last
// reflecting record (the one whose child does not reflect: its refl entry is final when this launch starts) sums
// that record, hands the clamped colour to the parent's sum in a register, and so on up to level_lo, whose result
// goes where k_shade would put it.  The same sums in the same order as one k_shade launch per level, without the
// launches' start-up between the levels (~5 us each at the end of the frame, when nothing else runs).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_shade_paths(DevScene s, DevFrame f, int level_lo, int32_t *pixels)
{
    __shared__ int na[RT_DEV_MAX_LEVELS];
    const int deepest = f.max_depth - 1;
    if ((int)threadIdx.x <= deepest) na[threadIdx.x] = (int)threadIdx.x >= level_lo ? f.cnt[2 * threadIdx.x] : 0;
    __syncthreads();
    int total = 0;
    for (int k = level_lo; k <= deepest; k++) total += na[k];
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        // deepest level first: the longest paths start first
        int level = deepest, j = i;
        while (level > level_lo && j >= na[level]) {
            j -= na[level];
            level--;
        }
        int slot = level * f.P + j;
        float4 rf = f.refl[slot];
        if (rf.w == RT_REFL_PENDING) continue;   // the path goes on below this record
        for (;;) {
            int parent;
            f3 final = combine_sum(s, f, level, slot, rf, parent);
            if (level == level_lo) {
                shade_store(f, level, parent, final, false, pixels);
                break;
            }
            rf = make_float4(clamp01(final.x), clamp01(final.y), clamp01(final.z), 1.0f);   // as shade_store hands it on
            slot = parent;
            level--;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Shading with the lights spread over the waves of a workgroup, for queues too short to fill the
// device with one record per lane: a lone wave issues an instruction only every ~4.5 cycles, and
// a record's light loop is one dependent chain of ~400 instructions per light, so a level of a few
// thousand records takes as long as its longest chain (16 lights: ~200 us) however few they are.
// Here a workgroup takes 64 records; wave w computes what lights w, w + W, ... add to the colour
// (light_terms: the same arithmetic as shade_record), the addends meet in LDS, and wave 0 adds them
// in the reference's order, so the sum is the same float for float.  Launched beside k_plain
// (reflecting = 0) and k_shade (reflecting = 1); whichever of the two forms the queue's length
// does not call for returns at once (DevFrame::light_waves, light_split_below).
// ---------------------------------------------------------------------------------------
#define LIGHT_WAVES_MAX 16
template <bool COUNT>
__global__ void __launch_bounds__(64 * LIGHT_WAVES_MAX) k_shade_lights(DevScene s, DevFrame f, int level_lo, int level_hi,
                                                                      int reflecting, int32_t *pixels)
{
    __shared__ float terms[LIGHT_WAVES_MAX][6][64];
    __shared__ unsigned long long cnt_sh[COUNTER_WORDS];
    // reflecting = 0: the plain records, shaded; 1: the reflecting records, shaded (their children are);
    // 2: the light terms of the reflecting records into DevFrame::lit, ahead of their children (what k_lit does for
    // long queues; nothing is summed or stored elsewhere); 3: 0 and 2 in one launch (both are chains of dependent
    // reads that a short queue cannot hide: side by side they cost the time of one).
    // Whichever queue is too long for this form, or has its light terms computed the other way, is left out.
    // Levels level_lo .. level_hi in one launch: the tiles of their queues are ONE list of jobs, level by level, the
    // plain queue and then the reflecting one (a short level's launch is as long as one record's chain of lights:
    // several levels side by side cost the time of one).
    const bool want_plain = reflecting == 0 || reflecting == 3, want_ahead = reflecting == 2 || reflecting == 3;
    // (the queues' lengths once per workgroup: several levels in LDS, one level - most launches, and the shortest - in
    // registers: the barrier behind the LDS costs a launch of a few tiles 1.5 us)
    __shared__ int q_nb[RT_DEV_MAX_LEVELS], q_na[RT_DEV_MAX_LEVELS], q_tiles_b[RT_DEV_MAX_LEVELS], q_tiles_a[RT_DEV_MAX_LEVELS];
    auto count_tiles = [&](const int lv, const int nb, const int na, int &tiles_b, int &tiles_a) {
        tiles_b = want_plain && nb < f.light_split_below ? (nb + 63) >> 6 : 0;
        tiles_a = (reflecting == 1 && na < f.light_split_below && !lit_ahead(f, na)) ||
                          (want_ahead && lv < f.max_depth && na < f.light_split_below && lit_ahead(f, na))
                      ? (na + 63) >> 6 : 0;
    };
    const bool one_level = level_lo == level_hi;
    int one_nb = 0, one_na = 0, one_tiles_b = 0, one_tiles_a = 0, total = 0;
    if (one_level) {
        one_nb = f.cnt[2 * level_lo + 1], one_na = f.cnt[2 * level_lo];
        count_tiles(level_lo, one_nb, one_na, one_tiles_b, one_tiles_a);
        total = one_tiles_b + one_tiles_a;
    } else {
        if ((int)threadIdx.x >= level_lo && (int)threadIdx.x <= level_hi) {
            const int lv = threadIdx.x;
            const int nb = f.cnt[2 * lv + 1], na = f.cnt[2 * lv];
            q_nb[lv] = nb, q_na[lv] = na;
            count_tiles(lv, nb, na, q_tiles_b[lv], q_tiles_a[lv]);
        }
        __syncthreads();
        for (int lv = level_lo; lv <= level_hi; lv++) total += q_tiles_b[lv] + q_tiles_a[lv];
    }
    auto tiles_of = [&](const int lv, int &tiles_b, int &tiles_a, int &nb, int &na) {
        if (one_level) nb = one_nb, na = one_na, tiles_b = one_tiles_b, tiles_a = one_tiles_a;
        else nb = q_nb[lv], na = q_na[lv], tiles_b = q_tiles_b[lv], tiles_a = q_tiles_a[lv];
    };
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    const int nl = s.nl;
    TestCount tc = {0, 0, 0, 0, 0, 0, 0};
    for (int job = blockIdx.x; job < total; job += gridDim.x) {
        int level = level_lo, tiles_b, tiles_a, nb, na, in_level = job;
        for (;; level++) {   // (ends: job < total)
            tiles_of(level, tiles_b, tiles_a, nb, na);
            if (in_level < tiles_b + tiles_a) break;
            in_level -= tiles_b + tiles_a;
        }
        const bool queue_a = in_level >= tiles_b;              // (uniform over the workgroup)
        const bool ahead = queue_a && want_ahead;
        const int tile = queue_a ? in_level - tiles_b : in_level, n = queue_a ? na : nb;
        const int i = tile * 64 + lane;
        const bool active = i < n;
        const int ii = active ? i : n - 1;
        const int slot = queue_a ? level * f.P + ii : (level + 1) * f.P - 1 - ii;
        const bool through = !queue_a;   // as k_plain and k_shade read and write their records
        const ShadeRecord r = shade_setup(s, f, slot, active, queue_a && !ahead, through);
        if (ahead && wave == 0 && active) mark_parent(f, level, r.parent);
        f3 final = mk3(0.0f, 0.0f, 0.0f);
        for (int l0 = 0; l0 < nl; l0 += W) {
            const int l = l0 + wave;
            if (l < nl) {
                uint2 cell = make_uint2(0u, 0u);
                if (s.lt_n > 0) cell = light_table_cell(s, l, sub3(r.hit, xyz(cload(s.light_pos, l))));
                f3 reflected, direct;
                float d2;
                light_terms<COUNT>(s, r, l, cell, active, tc, reflected, direct, d2);
                if (ahead && active) f.lit[lit_first(f, level, slot, nl) + (size_t)l] = make_float4(direct.x, direct.y, direct.z, d2);
                terms[wave][0][lane] = reflected.x, terms[wave][1][lane] = reflected.y, terms[wave][2][lane] = reflected.z;
                terms[wave][3][lane] = direct.x, terms[wave][4][lane] = direct.y, terms[wave][5][lane] = direct.z;
            }
            if (ahead) continue;   // (uniform over the workgroup)
            __syncthreads();
            if (wave == 0) {
                const int m = min(W, nl - l0);
                for (int w = 0; w < m; w++) {
                    if (r.have_refl) final = add3(final, mk3(terms[w][0][lane], terms[w][1][lane], terms[w][2][lane]));
                    final = add3(final, mk3(terms[w][3][lane], terms[w][4][lane], terms[w][5][lane]));
                }
            }
            __syncthreads();
        }
        if (wave == 0 && active && !ahead) shade_store(f, level, r.parent, final, through, pixels);
    }
    if constexpr (COUNT) flush_test_counts(f, cnt_sh, tc);
}

// ---------------------------------------------------------------------------------------
// fxaa: the filter the host's present path applies to the frame after the hand-off
// (screen_fxaa_fs.glsl:80-140, called from main() :426-431), as an optional post-pass over the
// 0x00RRGGBB frame in device memory.  Pixel (x, y) is the fragment with uv * resolution =
// (x + 0.5, y + 0.5); texture2D is the sampler of surface.cs:36-38 (LINEAR, the default REPEAT
// wrap, texel = byte / 255) evaluated in FP32, operation by operation; colours go back to bytes
// by round-to-nearest and the alpha byte stays 0 (texColor.a).  Bandwidth-bound: 4 bytes read
// (neighbours come from cache) and 4 written per pixel.
//
// The filter follows, operation by operation, the FXAA routine of that shader, which is
// Copyright (c) 2011 by Armin Ronacher (webgl-meincraft), some rights reserved, distributed under
// a three-clause BSD licence; the notice, conditions and disclaimer are reproduced in
// THIRD_PARTY_NOTICES.md at the root of this repository.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int fx_wrap(int i, int n)
{
    i %= n;
    return i < 0 ? i + n : i;
}
__device__ __forceinline__ f3 fx_texel(const int32_t *img, int w, int x, int y)
{
    const uint32_t p = (uint32_t)img[(size_t)y * (size_t)w + (size_t)x];
    return mk3(ieee_div((float)((p >> 16) & 255u), 255.0f), ieee_div((float)((p >> 8) & 255u), 255.0f),
               ieee_div((float)(p & 255u), 255.0f));
}
__device__ __forceinline__ float fx_mix(float a, float b, float t) { return a + (b - a) * t; }
__device__ __forceinline__ f3 fx_sample(const int32_t *img, int w, int h, float u, float v)
{
    const float xs = u * (float)w - 0.5f, ys = v * (float)h - 0.5f;
    const float xf = floorf(xs), yf = floorf(ys);
    const float fx = xs - xf, fy = ys - yf;
    const int x0 = fx_wrap((int)xf, w), x1 = fx_wrap((int)xf + 1, w);
    const int y0 = fx_wrap((int)yf, h), y1 = fx_wrap((int)yf + 1, h);
    const f3 t00 = fx_texel(img, w, x0, y0), t10 = fx_texel(img, w, x1, y0);
    const f3 t01 = fx_texel(img, w, x0, y1), t11 = fx_texel(img, w, x1, y1);
    return mk3(fx_mix(fx_mix(t00.x, t10.x, fx), fx_mix(t01.x, t11.x, fx), fy),
               fx_mix(fx_mix(t00.y, t10.y, fx), fx_mix(t01.y, t11.y, fx), fy),
               fx_mix(fx_mix(t00.z, t10.z, fx), fx_mix(t01.z, t11.z, fx), fy));
}
__device__ __forceinline__ float fx_luma(f3 c) { return (c.x * 0.299f + c.y * 0.587f) + c.z * 0.114f; }
__device__ __forceinline__ uint32_t fx_byte(float c)
{
    if (!(c > 0.0f)) return 0u;
    if (c > 1.0f) c = 1.0f;
    return (uint32_t)(c * 255.0f + 0.5f);
}

__global__ void __launch_bounds__(256) k_fxaa(const int32_t *in, int32_t *out, int w, int h)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= w || y >= h) return;
    const float ivx = ieee_div(1.0f, (float)w), ivy = ieee_div(1.0f, (float)h);   // :94
    const float reduce_min = 1.0f / 128.0f, reduce_mul = 1.0f / 8.0f, span_max = 8.0f;
    const float ta = (float)(1.0 / 3.0 - 0.5), tb = (float)(2.0 / 3.0 - 0.5);
    const float fcx = (float)x + 0.5f, fcy = (float)y + 0.5f;
    const f3 nw = fx_sample(in, w, h, (fcx - 1.0f) * ivx, (fcy - 1.0f) * ivy);     // :95-103
    const f3 ne = fx_sample(in, w, h, (fcx + 1.0f) * ivx, (fcy - 1.0f) * ivy);
    const f3 sw = fx_sample(in, w, h, (fcx - 1.0f) * ivx, (fcy + 1.0f) * ivy);
    const f3 se = fx_sample(in, w, h, (fcx + 1.0f) * ivx, (fcy + 1.0f) * ivy);
    const float mu = fcx * ivx, mv = fcy * ivy;
    const f3 m = fx_sample(in, w, h, mu, mv);                                     // :104
    const float l_nw = fx_luma(nw), l_ne = fx_luma(ne), l_sw = fx_luma(sw), l_se = fx_luma(se), l_m = fx_luma(m);
    const float l_min = fminf(l_m, fminf(fminf(l_nw, l_ne), fminf(l_sw, l_se)));  // :112-113
    const float l_max = fmaxf(l_m, fmaxf(fmaxf(l_nw, l_ne), fmaxf(l_sw, l_se)));
    float dx = -((l_nw + l_ne) - (l_sw + l_se));                                  // :116-117
    float dy = (l_nw + l_sw) - (l_ne + l_se);
    const float dir_reduce = fmaxf((((l_nw + l_ne) + l_sw) + l_se) * (0.25f * reduce_mul), reduce_min);
    const float rcp = ieee_div(1.0f, fminf(fabsf(dx), fabsf(dy)) + dir_reduce);   // :122
    dx = fminf(span_max, fmaxf(-span_max, dx * rcp)) * ivx;                       // :123-125
    dy = fminf(span_max, fmaxf(-span_max, dy * rcp)) * ivy;
    const f3 a1 = fx_sample(in, w, h, mu + dx * ta, mv + dy * ta);                // :127-129
    const f3 a2 = fx_sample(in, w, h, mu + dx * tb, mv + dy * tb);
    const f3 b1 = fx_sample(in, w, h, mu + dx * -0.5f, mv + dy * -0.5f);          // :130-132
    const f3 b2 = fx_sample(in, w, h, mu + dx * 0.5f, mv + dy * 0.5f);
    const f3 ca = mk3(0.5f * (a1.x + a2.x), 0.5f * (a1.y + a2.y), 0.5f * (a1.z + a2.z));
    const f3 cb = mk3(ca.x * 0.5f + 0.25f * (b1.x + b2.x), ca.y * 0.5f + 0.25f * (b1.y + b2.y),
                      ca.z * 0.5f + 0.25f * (b1.z + b2.z));
    const float l_b = fx_luma(cb);                                                // :134-138
    const f3 c = (l_b < l_min || l_b > l_max) ? ca : cb;
    out[(size_t)y * (size_t)w + (size_t)x] = (int32_t)((fx_byte(c.x) << 16) | (fx_byte(c.y) << 8) | fx_byte(c.z));
}

// ---------------------------------------------------------------------------------------
// debug_rays: the loop body of Raytracer.Debug (MyApplication.cs:459-478) for the rays
// x = 0, x_step, ... of one pixel row: ray direction, Scene.Intersect, GetNormal, mirror
// direction.  A hundred rays per call: every lane scans the whole scene with the literal
// Intersect code; no filter, no staging.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_debug_rays(DevScene s, DevCamera cam, int row, int x_step, int count,
                                                   float *out /* 16 floats per ray */)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const f3 o = mk3(cam.pos[0], cam.pos[1], cam.pos[2]);
    const f3 d = primary_direction(cam, i * x_step, row);
    const float a = dot3(d, d);
    Closest c = {RT_FLT_MAX, INT_MAX};
    for (int j = 0; j < s.ns_padded; j++) {
        const float4 sp = s.sph[j];
        const f3 op = sub3(o, xyz(sp));
        const float cc = dot3(op, op) - sp.w;
        const float x = dot3(d, op);
        float t;
        if (sphere_maybe(x, a * cc) && sphere_root(x, cc, a, t)) closest_offer(c, t, s.sph_orig[j]);
    }
    for (int j = 0; j < s.np; j++) {
        const float4 pl = s.pl[j];
        const f3 n = xyz(pl);
        float t;
        if (plane_t(n, (-dot3(n, o)) - pl.w, d, t)) closest_offer(c, t, s.pl_orig[j]);
    }
    for (int j = 0; j < s.nt; j++) {
        const f3 v0 = xyz(s.tri[3 * j]), e1 = xyz(s.tri[3 * j + 1]), e2 = xyz(s.tri[3 * j + 2]);
        float t;
        if (triangle_t(e1, e2, sub3(o, v0), d, t)) closest_offer(c, t, s.tri_orig[j]);
    }
    float *r = out + 16 * (size_t)i;
    r[0] = d.x, r[1] = d.y, r[2] = d.z;
    const bool hit = c.prim != INT_MAX;
    r[3] = __int_as_float(hit ? c.prim : -1);
    f3 p = mk3(0.f, 0.f, 0.f), n = p, m = p;
    if (hit) {
        p = add3(o, scale3(d, c.t));
        const int type = __float_as_int(s.mat[3 * c.prim + 1].w) & RT_MF_TYPE_MASK;
        n = record_normal(s, c.prim, type, p);
        m = sub3(d, scale3(n, 2.0f * dot3(d, n)));   // direction - 2 * Dot(direction, normal) * normal
    }
    r[4] = p.x, r[5] = p.y, r[6] = p.z;
    r[7] = hit ? c.t : 0.0f;
    r[8] = n.x, r[9] = n.y, r[10] = n.z;
    r[11] = m.x, r[12] = m.y, r[13] = m.z;
    r[14] = r[15] = 0.0f;
}

// ---------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------
void rtk_debug_rays(const DevScene &s, const DevCamera &cam, int row, int x_step, int count, float *out,
                    hipStream_t stream)
{
    hipLaunchKernelGGL(k_debug_rays, dim3((count + 63) / 64), dim3(64), 0, stream, s, cam, row, x_step, count, out);
}

// Do two streams share a hardware queue?  k_probe_wait waits (one lane, at most `ticks` of the 100 MHz wall clock)
// for the word k_probe_set writes, and reports whether it saw it: launches of one hardware queue run one after
// the other, so a setter enqueued behind the waiter on the same queue comes too late.
__global__ void k_probe_wait(int *words, unsigned long long ticks)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    int seen = 0;
    while ((seen = __hip_atomic_load(&words[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && wall_clock64() - t0 < ticks)
        __builtin_amdgcn_s_sleep(16);
    words[1] = seen;
}
__global__ void k_probe_set(int *words)
{
    if (threadIdx.x == 0) __hip_atomic_store(&words[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
void rtk_probe_wait(int *words, unsigned long long ticks, hipStream_t stream)
{
    hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(64), 0, stream, words, ticks);
}
void rtk_probe_set(int *words, hipStream_t stream) { hipLaunchKernelGGL(k_probe_set, dim3(1), dim3(64), 0, stream, words); }

void rtk_fxaa(const int32_t *in, int32_t *out, int w, int h, hipStream_t stream)
{
    hipLaunchKernelGGL(k_fxaa, dim3((w + 31) / 32, (h + 7) / 8), dim3(256), 0, stream, in, out, w, h);
}

// Scenes whose spheres fit one LDS chunk use the kernels that stage it once per workgroup.
static bool multi_chunk(const DevScene &s) { return s.ns_padded > SPH_CHUNK; }

void rtk_primary(const DevScene &s, const DevCamera &cam, const DevFrame &f, int32_t *pixels, int grid_blocks,
                 hipStream_t stream)
{
    const int ntiles = ((cam.width + PRIM_TILE_W - 1) / PRIM_TILE_W) * ((cam.rows + 31) / 32);  // pixels per workgroup trip
    grid_blocks = grid_blocks * 256 / RT_PRIMARY_THREADS;   // (the caller counts workgroups of four waves)
    if (grid_blocks < 1) grid_blocks = 1;
    const dim3 g(ntiles < grid_blocks ? ntiles : grid_blocks), b(RT_PRIMARY_THREADS);
    hipLaunchKernelGGL(k_primary, g, b, 0, stream, s, cam, f, pixels);
}

void rtk_trace(const DevScene &s, const DevFrame &f, int level, bool reflecting, int32_t *pixels, int grid_blocks,
               hipStream_t stream)
{
    const dim3 g(grid_blocks), b(256);
    const bool count = f.any_hit == 0;
    if (!reflecting) {
        if (count) hipLaunchKernelGGL(k_plain<true>, g, b, 0, stream, s, f, level, pixels);
        else hipLaunchKernelGGL(k_plain<false>, g, b, 0, stream, s, f, level, pixels);
    } else if (s.gr_n[0] > 0) {
        if (count) hipLaunchKernelGGL(k_reflect_grid<true>, g, b, 0, stream, s, f, level);
        else hipLaunchKernelGGL(k_reflect_grid<false>, g, b, 0, stream, s, f, level);
    } else if (multi_chunk(s)) {
        if (count) hipLaunchKernelGGL((k_reflect<true, true>), g, b, 0, stream, s, f, level);
        else hipLaunchKernelGGL((k_reflect<true, false>), g, b, 0, stream, s, f, level);
    } else {
        if (count) hipLaunchKernelGGL((k_reflect<false, true>), g, b, 0, stream, s, f, level);
        else hipLaunchKernelGGL((k_reflect<false, false>), g, b, 0, stream, s, f, level);
    }
}

void rtk_bin_level(const DevScene &s, const DevFrame &f, int level, int grid_blocks, hipStream_t stream)
{
    const dim3 g(grid_blocks), b(256);
    hipLaunchKernelGGL(k_bin_count, g, b, 0, stream, s, f, level);
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, stream, f, level);
    hipLaunchKernelGGL(k_bin_scatter, g, b, 0, stream, f, level);
    if (!f.staged) hipLaunchKernelGGL(k_bin_copy, g, b, 0, stream, f, level);
}

void rtk_reflect_chunks(const DevScene &s, const DevFrame &f, int level, int job_blocks, int tile_blocks, hipStream_t stream)
{
    const dim3 b(256);
    if (f.any_hit == 0) {
        hipLaunchKernelGGL(k_reflect_chunks<true>, dim3(job_blocks), b, 0, stream, s, f, level);
        hipLaunchKernelGGL(k_reflect_finish<true>, dim3(tile_blocks), b, 0, stream, s, f, level);
    } else {
        hipLaunchKernelGGL(k_reflect_chunks<false>, dim3(job_blocks), b, 0, stream, s, f, level);
        hipLaunchKernelGGL(k_reflect_finish<false>, dim3(tile_blocks), b, 0, stream, s, f, level);
    }
}

typedef void (*trace_levels_kernel)(DevScene, DevFrame, int32_t *);
template <bool RESERVE, bool COUNT>
static trace_levels_kernel trace_levels_pick(const DevScene &s)
{
    if (s.gr_n[0] > 0) return k_trace_levels<false, RESERVE, COUNT, true>;
    return multi_chunk(s) ? k_trace_levels<true, RESERVE, COUNT> : k_trace_levels<false, RESERVE, COUNT>;
}
static trace_levels_kernel trace_levels_for(const DevScene &s, bool reserve = false, bool count = true)
{
    if (reserve) return count ? trace_levels_pick<true, true>(s) : trace_levels_pick<true, false>(s);
    return count ? trace_levels_pick<false, true>(s) : trace_levels_pick<false, false>(s);
}

int rtk_trace_levels_resident_blocks(const DevScene &s, int num_cus, bool reserve, bool count)
{
    int per_cu = 0;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trace_levels_for(s, reserve, count), 256, 0);
    if (e != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    return per_cu * num_cus;
}

void rtk_trace_levels(const DevScene &s, const DevFrame &f, int32_t *pixels, int grid_blocks, hipStream_t stream)
{
    hipLaunchKernelGGL(trace_levels_for(s, f.chain_policy != 0, f.any_hit == 0), dim3(grid_blocks), dim3(256), 0, stream, s, f,
                       pixels);
}

void rtk_plain_levels(const DevScene &s, const DevFrame &f, int level_lo, int level_hi, int32_t *pixels, int grid_blocks,
                      hipStream_t stream)
{
    const dim3 g(grid_blocks), b(256);
    if (f.any_hit == 0) hipLaunchKernelGGL(k_plain_tail<true>, g, b, 0, stream, s, f, level_lo, level_hi, pixels);
    else hipLaunchKernelGGL(k_plain_tail<false>, g, b, 0, stream, s, f, level_lo, level_hi, pixels);
}

void rtk_lit_level(const DevScene &s, const DevFrame &f, int level, int grid_blocks, hipStream_t stream)
{
    if (f.any_hit == 0) hipLaunchKernelGGL(k_lit<true>, dim3(grid_blocks), dim3(256), 0, stream, s, f, level);
    else hipLaunchKernelGGL(k_lit<false>, dim3(grid_blocks), dim3(256), 0, stream, s, f, level);
}

void rtk_shade_lights(const DevScene &s, const DevFrame &f, int level_lo, int level_hi, int queue, int32_t *pixels,
                      int grid_blocks, hipStream_t stream)
{
    const dim3 g(grid_blocks), b(64 * f.light_waves);
    if (f.any_hit == 0) hipLaunchKernelGGL(k_shade_lights<true>, g, b, 0, stream, s, f, level_lo, level_hi, queue, pixels);
    else hipLaunchKernelGGL(k_shade_lights<false>, g, b, 0, stream, s, f, level_lo, level_hi, queue, pixels);
}

void rtk_shade_paths(const DevScene &s, const DevFrame &f, int level_lo, int32_t *pixels, int grid_blocks, hipStream_t stream)
{
    hipLaunchKernelGGL(k_shade_paths, dim3(grid_blocks), dim3(256), 0, stream, s, f, level_lo, pixels);
}

void rtk_shade_level(const DevScene &s, const DevFrame &f, int level, bool alone, int32_t *pixels,
                     int grid_blocks, hipStream_t stream)
{
    if (f.any_hit == 0) hipLaunchKernelGGL(k_shade<true>, dim3(grid_blocks), dim3(256), 0, stream, s, f, level, alone ? 1 : 0, pixels);
    else hipLaunchKernelGGL(k_shade<false>, dim3(grid_blocks), dim3(256), 0, stream, s, f, level, alone ? 1 : 0, pixels);
}
