// rt_api.cpp — host side of librt_hip.so: contexts, scene packing, per-frame camera
// constants, the frame pipeline and the C ABI of include/rt_hip.h.
//
// Built with -ffp-contract=off: the few FP32 values computed here (camera constants,
// r*r, triangle edges) follow the reference's operation order like the kernels do.
#include "../../include/rt_hip.h"
#include "rt_device.h"

#include <algorithm>
#include <atomic>
#if !defined(__HIP_DEVICE_COMPILE__)
#include <emmintrin.h>   // SSE2 streaming stores (baseline x86-64), host pass only
#endif
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

// Scratch of rt_render_rows_device_noalloc, the entry point that allocates nothing: the ray
// queues and control block of ONE frame of bounded size.  Every GPU the library
// runs on holds its own copy of this array from the moment the library is loaded until the
// process ends.  In host code the array's name is only a handle for hipGetSymbolAddress: the
// address the kernels get is the one that call returned for the device in question
// (scratch_address below), never the name itself.
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];

namespace {

// Device address of `scratch` on each device, asked for once per device with that device current.
constexpr int RT_SCRATCH_TABLE = 64;
std::mutex g_scratch_lock;
unsigned char *g_scratch_address[RT_SCRATCH_TABLE] = {};

constexpr int RT_CHAIN_RESERVED_TILES_PER_WORKGROUP = 6;
// Below this many level-0 tiles per workgroup of the level kernel that the whole device holds, a
// frame of a context that shares its device with three or more others is rendered "fused"
// (enqueue_rows).  Measured on MI355X, 3840x2160 / G, ms per frame with 1 / 2 / 3 / 4 frames in
// flight - G = 8 (4 tiles): levels 0.20 / 0.16 / 0.13 / 0.127, fused 0.33 / 0.18 / 0.14 / 0.106;
// G = 4 (8 tiles): levels 0.26 / 0.23 / 0.18 / 0.17, fused 0.42 / 0.26 / 0.21 / 0.18; G = 1: levels
// 0.61, fused 1.07.
constexpr int RT_FUSED_BELOW_TILES_PER_WORKGROUP = 6;

// Counter slots + level schedule + two record counts per level (DevFrame::counters, sched, cnt).
constexpr size_t FRAME_CONTROL_USED =
    RT_COUNTER_SLOTS * sizeof(DevCounters) + sizeof(DevSched) + 2 * RT_DEV_MAX_LEVELS * sizeof(int);
// (rounded so that the fill that clears it is one kernel: a size that is no multiple of 16 bytes
// gets a second fill for its tail)
constexpr size_t FRAME_CONTROL_BYTES = (FRAME_CONTROL_USED + 255) / 256 * 256;
static_assert(FRAME_CONTROL_BYTES + 3 * 256 <= RT_NOALLOC_CONTROL_BYTES,
              "the fit rule rt_hip.h documents for rt_render_rows_device_noalloc must cover the control block and the "
              "rounding of the three record arrays");

// Level schedule + record counts: the part of the control block a render call leaves its
// outcome in (has the schedule stalled, how many records did each level hold).
constexpr size_t FRAME_OUTCOME_BYTES = sizeof(DevSched) + 2 * RT_DEV_MAX_LEVELS * sizeof(int);

// rt_render on one device renders a large frame as up to RT_MAX_FRAME_PARTS row blocks one after
// the other, so that the copy of a finished block to the host runs while the next one renders.
// Small blocks cost more in rendering (the chain of levels does not shrink with the block) than
// their copy hides: measured on MI355X, two blocks pay from ~3 M pixels (2560x1440: 1.45 -> 1.30
// ms; 1920x1080: 0.86 -> 1.02 ms), and beyond that blocks of ~4 M pixels are best (3840x2160:
// two blocks 2.40 ms, four 2.52; 7680x4320: eight blocks 7.7 ms against 10.6 in one piece).
constexpr int RT_MAX_FRAME_PARTS = 8;
// Rows per band of the interleaved partition over several devices: one wave of the primary
// kernel covers 16 x 16 pixels.
constexpr int RT_BAND_ROWS = 16;
constexpr long long RT_MIN_SPLIT_PIXELS = 3000000, RT_PART_PIXELS = 4000000;
// Helper threads that share the host-side move of a staged frame with the caller's thread, for frames of at
// least so many pixels (a smaller frame's move is over before a sleeping thread has woken up).
constexpr int RT_MOVER_THREADS = 3, RT_MAX_MOVER_THREADS = 15;
constexpr long long RT_MOVERS_MIN_PIXELS = 1500000;
constexpr int RT_PATH_SHADE_MAX_PIXELS = 12 << 20;   // (a 3840x2160 frame has 8.3 M)

// Messages live in fixed buffers so that reporting a failure can never itself fail.
constexpr size_t RT_ERROR_CHARS = 512;
thread_local char g_create_error[RT_ERROR_CHARS] = "";

// Default patience of a workgroup of the level kernel that waits for a level to open, in
// milliseconds (RT_HIP_STALL_MS overrides): far above anything a healthy frame shows (levels open
// every few tens of microseconds), short enough that a lost frame is reported at once.
constexpr unsigned RT_STALL_MS_DEFAULT = 1000;
constexpr size_t RT_LIT_MAX_BYTES = (size_t)16 << 30;
constexpr size_t RT_STATUS_BYTES = 256;   // DevFrame::status (16 words), then DevFrame::hint (a tag and 2 words per level)
static_assert(64 + sizeof(int) * (1 + 2 * RT_DEV_MAX_LEVELS) <= RT_STATUS_BYTES, "the hint block holds every level's counts");
constexpr int RT_LIGHT_SPLIT_LARGE_SCENE = 32768;   // spheres (padded) from which long queues too are shaded with the lights spread

struct DeviceState {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[5] = {};  // begin, after primary, after trace, after shade, after gather copy
    // Frames rendered as one launch per level and queue: the plain records' launches run on a
    // second stream beside the reflecting records' (enqueue_rows), ordered by these events.
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_level[RT_MAX_DEPTH + 1] = {};
    // The caller streams the second stream, as it is now, has been checked against (calibrate_second_stream: it
    // shares a hardware queue with none of them).  Handles only, never used again: a caller may have destroyed its
    // stream since.  replaced: how often the second stream was exchanged (bounded).  probe: the check's two words.
    std::vector<hipStream_t> checked_against;
    int replaced = 0;
    int *probe = nullptr;

    int num_cus = 256;
    // k_trace_levels workgroups resident at once, per instance [chain policy][counting] (0: ask
    // again, the scene changed)
    int resident_blocks[2][2] = {};
    bool peer_to_root = true;  // this device can write the first device's memory (or is the first device)
    // Outcome words the device writes only when a frame went wrong (DevFrame::status): page-locked,
    // device-visible host memory, so that reading them costs a finished stream and nothing else.
    int *status = nullptr;
    hipStream_t last_stream = nullptr;   // the stream the last render call was enqueued on

    // scene
    DevScene scene{};
    std::vector<void *> scene_allocs;
    void *tex_mem[RT_MAX_TEXTURES] = {};

    // per-frame scratch
    DevFrame frame{};          // in memory this context allocated (grown on demand)
    DevFrame fixed{};          // in the library's fixed scratch array (rt_render_rows_device_noalloc)
    bool last_fixed = false;   // which of the two the last render call used
    DevFrame &last() { return last_fixed ? fixed : frame; }
    unsigned char *scratch_base = nullptr;   // this device's copy of the fixed scratch array
    size_t cap_slots = 0;
    // Buffers of the record binning (k_bin_*), sized for one level of the largest frame so far.
    uint2 *bin_aux = nullptr;
    int *bin_hist = nullptr;   // histograms, then the prefix sums: 4 * RT_BINS ints
    bool spare_level = false;  // the frame's record arrays hold a spare level (ensure_frame)
    size_t cap_bin = 0, bin_refused = SIZE_MAX;
    // Light terms of the reflecting records (DevFrame::lit), for cap_lit float4s.
    float4 *lit = nullptr;
    size_t cap_lit = 0, lit_refused = SIZE_MAX;   // (lit_refused: the smallest size the device could not give)
    int32_t *out = nullptr;  // device framebuffer block of this device
    size_t cap_out = 0;
    int32_t *bands = nullptr;  // this device's bands of an interleaved partition, stacked (rt_render, several devices)
    size_t cap_bands = 0;

    // last call
    int last_P = 0, last_depth = 0;
    bool timed = false;
    // Frames rendered in parts: the outcome block of every part, copied to page-locked host
    // memory behind the part's kernels (snap_parts of them are valid; 0 = the frame was one call
    // and its outcome is still in device memory).
    char *outcome_snap = nullptr;
    int snap_parts = 0;
};


// Tuning aids and test knobs, all from the environment, read when a context is created, when it gets a
// scene and by rt_reload_tuning - never on the way of a frame.
struct Tuning {
    int chain_policy = -1;       // RT_HIP_CHAIN_POLICY=0/1
    long long stall_ms = 0;      // RT_HIP_STALL_MS (> 0)
    int withhold_level = -1;     // RT_HIP_DEBUG_WITHHOLD_LEVEL: the level whose first reflecting tile is never counted
                                 // as done, which stalls the schedule there on purpose
    int pipeline = 0;            // RT_HIP_PIPELINE=levels (1) | fused (2)
    bool level_kernels = false;  // RT_HIP_LEVEL_KERNELS: "levels" on ONE stream (a kernel trace then shows what each launch costs alone)
    int split_chunks = -1;       // RT_HIP_SPLIT_CHUNKS=0: a tile's chunks stay with its own workgroup
    int split_below = 0;         // RT_HIP_SPLIT_BELOW=n
    int grid_split_below = -1;   // RT_HIP_GRID_SPLIT_BELOW=n (0 = never)
    int spread_below = -1;       // RT_HIP_SPREAD_BELOW=n (0 = never)
    int light_split_below = -1;  // RT_HIP_LIGHT_SPLIT_BELOW=n (0 = never)
    int light_waves = 0;         // RT_HIP_LIGHT_WAVES=1..16
    int bin_min = -1;            // RT_HIP_BIN_MIN=n (0 = never)
    int early_lights = -1;       // RT_HIP_EARLY_LIGHTS=0 turns the light terms ahead of the children off
    int early_lights_max = 16;   // RT_HIP_EARLY_LIGHTS_MAX: the most lights it is used for
    int early_lights_below = -1; // RT_HIP_EARLY_LIGHTS_BELOW: the bound in records (0: every level)
    int tail_levels = 0;         // RT_HIP_TAIL_LEVELS=n: the last n levels below the deepest share one launch per kernel,
                                 // the others have their own
    int level_group = 0;         // RT_HIP_LEVEL_GROUP=n: scenes of three lights and more, levels per group from the fourth on
    int bin_hint = -1;           // RT_HIP_BIN_HINT=0: every level goes through the binning launches, whatever the last frame's counts
    int path_shade = -1;         // RT_HIP_PATH_SHADE=0: one k_shade launch per level even when one launch could sum the paths;
                                 // n > 0: the paths' launch whatever the frame's size, for the levels from n - 1 down
    int reflect_blocks = 0, plain_blocks = 16, light_blocks = 16;   // RT_HIP_REFLECT_BLOCKS / _PLAIN_BLOCKS / _LIGHT_BLOCKS: workgroups per
                                                                    // compute unit (reflect_blocks 0: 8, or 4 for the grid walk)
    size_t copy_chunk_pixels = 0;   // RT_HIP_COPY_CHUNK_KIB
    int frame_parts = 0;         // RT_HIP_FRAME_PARTS: row blocks of rt_render's frame on one device
    int part_bands = -1;         // RT_HIP_PART_BANDS=0: contiguous row blocks instead of interleaved bands
    int host_gather = -1;        // RT_HIP_HOST_GATHER=0: several devices gather on the first one before the copy to the host
    int aux_priority = -1;       // RT_HIP_AUX_PRIORITY=1 (read by rt_create): the second stream at the lowest priority (experiment)
    int calibrate = -1;          // RT_HIP_CALIBRATE_STREAMS=0: take the second stream as it comes
    long long frame_timeout_ms = 0;   // RT_HIP_FRAME_TIMEOUT_MS: how long rt_render / rt_frame_status wait for the device (default 30000)
    int mover_threads = -1;      // RT_HIP_MOVER_THREADS: helper threads of a staged frame's host-side move (0: the caller alone)
    bool handoff_trace = false;  // RT_HIP_HANDOFF_TRACE (set before the first frame): rt_render prints, per row block, when the
                                 // host enqueued it and when the device finished its kernels and its copy
};

Tuning read_tuning()
{
    Tuning t;
    auto num = [](const char *name, long long unset) -> long long {
        const char *v = getenv(name);
        return v && *v ? atoll(v) : unset;
    };
    t.chain_policy = getenv("RT_HIP_CHAIN_POLICY") ? (num("RT_HIP_CHAIN_POLICY", 0) != 0) : -1;
    t.stall_ms = std::max(0ll, num("RT_HIP_STALL_MS", 0));
    t.withhold_level = (int)num("RT_HIP_DEBUG_WITHHOLD_LEVEL", -1);
    if (const char *v = getenv("RT_HIP_PIPELINE")) t.pipeline = strcmp(v, "fused") == 0 ? 2 : 1;
    t.level_kernels = getenv("RT_HIP_LEVEL_KERNELS") != nullptr;
    t.split_chunks = (int)num("RT_HIP_SPLIT_CHUNKS", -1);
    t.split_below = (int)num("RT_HIP_SPLIT_BELOW", 0);
    t.grid_split_below = (int)num("RT_HIP_GRID_SPLIT_BELOW", -1);
    t.spread_below = (int)num("RT_HIP_SPREAD_BELOW", -1);
    t.light_split_below = (int)num("RT_HIP_LIGHT_SPLIT_BELOW", -1);
    t.light_waves = (int)num("RT_HIP_LIGHT_WAVES", 0);
    t.bin_min = (int)num("RT_HIP_BIN_MIN", -1);
    t.early_lights = (int)num("RT_HIP_EARLY_LIGHTS", -1);
    t.early_lights_max = (int)num("RT_HIP_EARLY_LIGHTS_MAX", 16);
    t.early_lights_below = (int)num("RT_HIP_EARLY_LIGHTS_BELOW", -1);
    t.path_shade = (int)num("RT_HIP_PATH_SHADE", -1);
    t.bin_hint = (int)num("RT_HIP_BIN_HINT", -1);
    t.level_group = (int)num("RT_HIP_LEVEL_GROUP", 0);
    t.tail_levels = (int)num("RT_HIP_TAIL_LEVELS", 0);
    auto positive = [&](const char *name, int dflt) { const long long v = num(name, 0); return v > 0 ? (int)v : dflt; };
    t.reflect_blocks = positive("RT_HIP_REFLECT_BLOCKS", 0);
    t.plain_blocks = positive("RT_HIP_PLAIN_BLOCKS", 16);
    t.light_blocks = positive("RT_HIP_LIGHT_BLOCKS", 16);
    t.copy_chunk_pixels = (size_t)std::max(0ll, num("RT_HIP_COPY_CHUNK_KIB", 0)) * 256;
    t.frame_parts = (int)num("RT_HIP_FRAME_PARTS", 0);
    t.part_bands = (int)num("RT_HIP_PART_BANDS", -1);
    t.host_gather = (int)num("RT_HIP_HOST_GATHER", -1);
    t.aux_priority = (int)num("RT_HIP_AUX_PRIORITY", -1);
    t.calibrate = (int)num("RT_HIP_CALIBRATE_STREAMS", -1);
    t.frame_timeout_ms = num("RT_HIP_FRAME_TIMEOUT_MS", 0);
    t.mover_threads = (int)num("RT_HIP_MOVER_THREADS", -1);
    t.handoff_trace = getenv("RT_HIP_HANDOFF_TRACE") != nullptr;
    return t;
}

}  // namespace

// Pixels per piece of rt_render's staged device-to-host copy (4 MiB: a piece's DMA takes ~75 us, long
// enough to amortise its start - 3840x2160 into pageable memory 1.47 ms with pieces of 1 MiB, 1.24 with 2, 1.20
// with 4, 1.27 with 8 -; the last pieces of a frame are a quarter of that, so that the move of the last
// one, which nothing overlaps, is brief).
#define RT_STAGED_COPY_CHUNK_PIXELS ((size_t)1024 * 1024)

namespace {
class HostMovers;
}

struct rt_ctx {
    std::vector<DeviceState> dev;
    char error[RT_ERROR_CHARS] = "";
    bool have_scene = false;
    unsigned scene_serial = 0;     // counts rt_set_scene calls (hints of one scene's frames say nothing about another's)
    bool wedged = false;           // a device did not come back within the time limit: nothing of this context may wait for it again
    int profiling = 0;             // RT_PROFILE_*: 0 off, 1 counters and event timing, 2 event timing only
    int share = 1;                 // contexts rendering on this context's devices at the same time (rt_set_device_share)
    Tuning tune;
    int num_prims = 0, num_lights = 0, ns = 0, np = 0, nt = 0;
    uint32_t textures_used = 0;    // bit i: some primitive names texture slot i
    uint32_t textures_loaded = 0;  // bit i: slot i holds an image
    int32_t *staging = nullptr;    // pinned host staging for rt_render
    size_t cap_staging = 0;
    std::vector<hipEvent_t> copy_ev;   // one per chunk of the framebuffer copy to the host (device 0)
    hipStream_t copy_stream = nullptr; // carries those copies while device 0's stream renders the next part
    hipEvent_t part_ev[RT_MAX_FRAME_PARTS] = {};
    HostMovers *movers = nullptr;      // helper threads of the staged copy's host-side move (created on first use)
    hipEvent_t trace_ev = nullptr;   // RT_HIP_HANDOFF_TRACE: start of the frame on the device
    int32_t *pinned = nullptr;     // the host's own framebuffer, page-locked by rt_pin_pixels
    void *pinned_base = nullptr;   // start of the registered pages
    bool pinned_owned = false;     // the range is page-locked memory of the library's own (rt_alloc_pixels)
    size_t pinned_count = 0;
    double copy_ms = 0.0, gather_ms = 0.0;
    std::vector<int> last_rows;    // row split of the last rt_render (n_dev + 1 entries)
    int last_w = 0, last_h = 0;
    bool last_was_rows_device = false;
};

namespace {

int fail(rt_ctx *ctx, int code, const char *fmt, ...) noexcept
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx ? ctx->error : g_create_error, RT_ERROR_CHARS, fmt, ap);
    va_end(ap);
    return code;
}

// Every entry point's body runs inside this: no C++ exception leaves the library (the host is
// C#, C or ctypes: an exception unwinding into it is undefined behaviour).
template <class Body>
int guarded(rt_ctx *ctx, Body &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(ctx, RT_ERR_OUT_OF_MEMORY, "out of host memory");
    } catch (const std::exception &e) {
        return fail(ctx, RT_ERR_HIP, "unexpected exception: %s", e.what());
    } catch (...) {
        return fail(ctx, RT_ERR_HIP, "unexpected exception");
    }
}

// memcpy with streaming stores: the destination (the caller's framebuffer) is written once and
// not read here, so its lines need not be fetched into the cache first — a third less memory
// traffic than a chunk-sized memcpy, which uses ordinary stores below the C library's own
// streaming threshold.  The caller issues the store fence.
#if defined(__HIP_DEVICE_COMPILE__)   // this file is also parsed for the device, which has no such intrinsics
void copy_streaming(void *, const void *, size_t) {}
void store_fence() {}
#else
void copy_streaming(void *dst_v, const void *src_v, size_t bytes)
{
    char *dst = (char *)dst_v;
    const char *src = (const char *)src_v;
    const size_t head = std::min(bytes, (size_t)(-(uintptr_t)dst & 15u));
    memcpy(dst, src, head);
    dst += head, src += head, bytes -= head;
    size_t i = 0;
    for (; i + 64 <= bytes; i += 64) {
        const __m128i a = _mm_loadu_si128((const __m128i *)(src + i)), b = _mm_loadu_si128((const __m128i *)(src + i + 16)),
                      c = _mm_loadu_si128((const __m128i *)(src + i + 32)), d = _mm_loadu_si128((const __m128i *)(src + i + 48));
        _mm_stream_si128((__m128i *)(dst + i), a);
        _mm_stream_si128((__m128i *)(dst + i + 16), b);
        _mm_stream_si128((__m128i *)(dst + i + 32), c);
        _mm_stream_si128((__m128i *)(dst + i + 48), d);
    }
    memcpy(dst + i, src + i, bytes - i);
}
void store_fence() { _mm_sfence(); }
#endif

// Helper threads that move a staged frame's pieces from the page-locked staging frame into the caller's pageable
// array beside the calling thread: one thread moves ~27 GB/s with streaming stores, the DMA brings 55, so a frame
// that is staged is bound by the move unless it is shared (3840x2160: 1.35 ms with one thread).  The threads are
// started with the first large staged frame, sleep between frames and end with the context.  A system that gives
// no thread leaves the move to the caller alone.
class HostMovers {
public:
    ~HostMovers()
    {
        {
            std::lock_guard<std::mutex> lock(m_);
            quit_ = true;
        }
        cv_.notify_all();
        for (std::thread &t : threads_)
            if (t.joinable()) t.join();
    }
    // Threads available beside the caller (starts them on first use; never throws).
    int helpers(int wanted) noexcept
    {
        while ((int)threads_.size() < wanted) {
            try {
                const int id = (int)threads_.size() + 1;
                threads_.emplace_back([this, id] { loop(id); });
            } catch (...) {
                break;   // (no thread to be had: the caller moves more itself)
            }
        }
        return (int)threads_.size();
    }
    // Runs job(worker, workers) on the caller (worker 0) and on every helper; returns when all are done.
    void run(const std::function<void(int, int)> &job)
    {
        const int workers = (int)threads_.size() + 1;
        {
            std::lock_guard<std::mutex> lock(m_);
            job_ = &job;
            workers_ = workers;
            pending_.store(workers - 1, std::memory_order_relaxed);
            generation_++;
        }
        cv_.notify_all();
        job(0, workers);
        while (pending_.load(std::memory_order_acquire) != 0) std::this_thread::yield();
        std::lock_guard<std::mutex> lock(m_);
        job_ = nullptr;
    }

private:
    void loop(int id)
    {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int, int)> *job = nullptr;
            int workers = 0;
            {
                std::unique_lock<std::mutex> lock(m_);
                cv_.wait(lock, [&] { return quit_ || generation_ != seen; });
                if (quit_) return;
                seen = generation_;
                job = job_;
                workers = workers_;
            }
            if (job && id < workers) {
                (*job)(id, workers);
                pending_.fetch_sub(1, std::memory_order_release);
            }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_;
    const std::function<void(int, int)> *job_ = nullptr;
    int workers_ = 0;
    uint64_t generation_ = 0;
    bool quit_ = false;
    std::atomic<int> pending_{0};
};

#define HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(ctx, e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP, \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                     \
    } while (0)

void free_scene(DeviceState &d)
{
    for (void *p : d.scene_allocs) (void)hipFree(p);
    d.scene_allocs.clear();
    const uchar4 *tex[RT_DEV_MAX_TEXTURES];
    int tw[RT_DEV_MAX_TEXTURES], th[RT_DEV_MAX_TEXTURES];
    memcpy(tex, d.scene.tex, sizeof tex);
    memcpy(tw, d.scene.tex_w, sizeof tw);
    memcpy(th, d.scene.tex_h, sizeof th);
    d.scene = DevScene{};
    memcpy(d.scene.tex, tex, sizeof tex);  // textures outlive the scene
    memcpy(d.scene.tex_w, tw, sizeof tw);
    memcpy(d.scene.tex_h, th, sizeof th);
}

void free_frame(DeviceState &d)
{
    if (d.frame.recA) (void)hipFree(d.frame.recA);
    if (d.frame.recB) (void)hipFree(d.frame.recB);
    if (d.frame.refl) (void)hipFree(d.frame.refl);
    if (d.frame.counters) (void)hipFree(d.frame.counters);   // counters, sched and cnt are one allocation
    if (d.bin_aux) (void)hipFree(d.bin_aux);
    if (d.bin_hist) (void)hipFree(d.bin_hist);
    d.bin_aux = nullptr;
    d.bin_hist = nullptr;
    d.cap_bin = 0;
    if (d.lit) (void)hipFree(d.lit);
    d.lit = nullptr;
    d.cap_lit = 0;
    if (d.out) (void)hipFree(d.out);
    d.frame = DevFrame{};
    d.out = nullptr;
    d.cap_slots = d.cap_out = 0;
}

struct PackedScene {
    std::vector<float4> sph, sph_f, grp_f, sup_f, chk_f, sup_b, chk_b, pl, tri, geo, mat, light_pos, light_col;
    std::vector<int> sph_orig, pl_orig, tri_orig, class_prefix;
    uint32_t textures_used = 0;
    float sph_center[3] = {0.f, 0.f, 0.f};
    float sph_l1max = 0.f, sph_sc = 0.f, grp_rmax = 0.f, sup_rmax = 0.f, chk_rmax = 0.f;
    float bin_lo[3] = {0.f, 0.f, 0.f}, bin_scale[3] = {0.f, 0.f, 0.f};   // position bins (DevScene)
    // Per-light direction tables (build_light_tables).
    std::vector<uint2> lt_cells;
    std::vector<float4> lt_sph;
    std::vector<int> lt_orig;
    std::vector<float> lt_reach;
    int lt_n = 0;
    float lt_near = 0.f;
    int lt_tiers = 0;
    float lt_tier_q[RT_LT_TIERS - 1] = {};
    float gr_tier_t = -1.0f;
    // Uniform grid over the spheres (build_sphere_grid).
    std::vector<uint2> gr_cells;
    std::vector<float4> gr_sph;
    std::vector<int> gr_orig;
    int gr_n[3] = {0, 0, 0};
    float gr_lo[3] = {0.f, 0.f, 0.f}, gr_ih[3] = {0.f, 0.f, 0.f};
    float gr_reg_lo[3] = {0.f, 0.f, 0.f}, gr_reg_hi[3] = {0.f, 0.f, 0.f};
};

bool usable_sphere(const float4 &s)
{
    return std::isfinite(s.x) && std::isfinite(s.y) && std::isfinite(s.z) && std::isfinite(s.w) && s.w >= 0.f;
}

// Spatial order of the spheres: recursive median split along the widest axis of the
// centres, the cut rounded to a whole number of 1024-sphere chunks, then of 64, then of
// groups, so that every aligned group / chunk is spatially compact.  Spheres that can never
// be hit (non-finite data) go last.  Deterministic: ties fall back on the host index.
void spatial_order(const std::vector<float4> &sph, const std::vector<int> &orig, int *idx, size_t n)
{
    if (n <= (size_t)RT_GROUP) return;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i < n; i++) {
        const float4 &s = sph[idx[i]];
        const float c[3] = {s.x, s.y, s.z};
        for (int k = 0; k < 3; k++) lo[k] = std::min(lo[k], c[k]), hi[k] = std::max(hi[k], c[k]);
    }
    int axis = 0;
    for (int k = 1; k < 3; k++)
        if (hi[k] - lo[k] > hi[axis] - lo[axis]) axis = k;
    const size_t unit = n > 1024 ? 1024 : (n > 64 ? 64 : (size_t)RT_GROUP);
    size_t half = ((n / 2 + unit - 1) / unit) * unit;
    if (half >= n) half -= unit;
    auto key = [&](int i) { const float4 &s = sph[i]; return axis == 0 ? s.x : (axis == 1 ? s.y : s.z); };
    std::nth_element(idx, idx + half, idx + n, [&](int a, int b) {
        const float ka = key(a), kb = key(b);
        return ka < kb || (ka == kb && orig[a] < orig[b]);
    });
    spatial_order(sph, orig, idx, half);
    spatial_order(sph, orig, idx + half, n - half);
}

void reorder_spheres(PackedScene &out)
{
    const size_t n = out.sph.size();
    std::vector<int> good, bad;
    for (size_t i = 0; i < n; i++) (usable_sphere(out.sph[i]) ? good : bad).push_back((int)i);
    spatial_order(out.sph, out.sph_orig, good.data(), good.size());
    good.insert(good.end(), bad.begin(), bad.end());
    std::vector<float4> sph(n);
    std::vector<int> orig(n);
    for (size_t i = 0; i < n; i++) sph[i] = out.sph[good[i]], orig[i] = out.sph_orig[good[i]];
    out.sph.swap(sph);
    out.sph_orig.swap(orig);
}

// The sphere filter's view of the spheres (rt_kernels.hip, "Sphere filter"): centres relative
// to the middle of their bounding box, W = r*r - k1*|p'|^2 from double arithmetic on the
// rounded p'; one bounding ball per group of RT_GROUP spheres in the same form; and the
// bounds the kernels' margins are built from (rounded up).  A sphere with non-finite data can
// never be hit (every comparison of objects.cs:116-121 fails) and gets the never-pass entry,
// as does the ball of a group without any usable sphere.
void build_filter_arrays(PackedScene &out)
{
    const size_t n = out.sph.size();
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (const float4 &s : out.sph) {
        if (!usable_sphere(s)) continue;
        const double c[3] = {s.x, s.y, s.z};
        for (int k = 0; k < 3; k++) lo[k] = std::min(lo[k], c[k]), hi[k] = std::max(hi[k], c[k]);
    }
    double cen[3];
    for (int k = 0; k < 3; k++) {
        out.sph_center[k] = lo[k] <= hi[k] ? (float)(0.5 * (lo[k] + hi[k])) : 0.f;
        cen[k] = out.sph_center[k];
    }
    // Position bins over the extent of the centres (a degenerate or unrepresentable extent: one bin).
    for (int k = 0; k < 3; k++) {
        const double width = hi[k] - lo[k];
        const float scale = width > 0.0 ? (float)((double)RT_BIN_AXIS / width) : 0.f;
        out.bin_lo[k] = lo[k] <= hi[k] ? (float)lo[k] : 0.f;
        out.bin_scale[k] = std::isfinite(scale) && std::isfinite(out.bin_lo[k]) ? scale : 0.f;
        if (!std::isfinite(out.bin_lo[k])) out.bin_lo[k] = 0.f;
    }
    const float4 never = make_float4(0.f, 0.f, 0.f, -INFINITY);
    out.sph_f.assign(n, never);
    double l1max = 0.0, p2max = 0.0, r2max = 0.0;
    auto entry = [&](const double c[3], double r2, float4 *dst) {
        const float px = (float)(c[0] - cen[0]), py = (float)(c[1] - cen[1]), pz = (float)(c[2] - cen[2]);
        const double p2 = (double)px * px + (double)py * py + (double)pz * pz;
        const float w = (float)(r2 - (double)RT_FILTER_K1 * p2);
        if (!std::isfinite(w)) return false;
        *dst = make_float4(px, py, pz, w);
        l1max = std::max(l1max, std::fabs((double)px) + std::fabs((double)py) + std::fabs((double)pz));
        p2max = std::max(p2max, p2);
        r2max = std::max(r2max, r2);
        return true;
    };
    for (size_t i = 0; i < n; i++) {
        const float4 &s = out.sph[i];
        if (!usable_sphere(s)) continue;
        const double c[3] = {s.x, s.y, s.z};
        entry(c, (double)s.w, &out.sph_f[i]);
    }
    // Bounding balls of every aligned run of `run` spheres: centre = an approximate centre of the
    // smallest ball around the members (start in the middle of their extent, then step towards
    // the member that sticks out farthest with steps of 1/2, 1/3, 1/4 ... of the way: Badoiu and
    // Clarkson's iteration; ~15 % less r*r than the middle of the extent, and the share of rays
    // that meet a ball goes with r*r), rounded the way the filter will see it; radius = max
    // over members of |centre_i - centre| + r_i against that rounded centre, widened by 1e-6
    // relative, so the ball holds its members whatever the iteration did.  The array is padded
    // with never-pass entries to a multiple of `pad` balls.
    auto balls = [&](size_t run, size_t pad, std::vector<float4> &dst, double &rmax) {
        const size_t nb = (n + run - 1) / run;
        dst.assign((nb + pad - 1) / pad * pad, never);
        for (size_t g = 0; g < nb; g++) {
            const size_t i0 = g * run, i1 = std::min(n, (g + 1) * run);
            double blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
            bool any = false;
            for (size_t i = i0; i < i1; i++) {
                if (out.sph_f[i].w == -INFINITY) continue;
                const float4 &s = out.sph[i];
                const double c[3] = {s.x, s.y, s.z}, r = std::sqrt((double)s.w);
                for (int k = 0; k < 3; k++) blo[k] = std::min(blo[k], c[k] - r), bhi[k] = std::max(bhi[k], c[k] + r);
                any = true;
            }
            if (!any) continue;
            double bc[3];
            for (int k = 0; k < 3; k++) bc[k] = 0.5 * (blo[k] + bhi[k]);
            for (int it = 0; it < 64; it++) {
                double far = -1.0, fc[3] = {bc[0], bc[1], bc[2]};
                for (size_t i = i0; i < i1; i++) {
                    if (out.sph_f[i].w == -INFINITY) continue;
                    const float4 &s = out.sph[i];
                    const double dx = s.x - bc[0], dy = s.y - bc[1], dz = s.z - bc[2];
                    const double dist = std::sqrt(dx * dx + dy * dy + dz * dz) + std::sqrt((double)s.w);
                    if (dist > far) far = dist, fc[0] = s.x, fc[1] = s.y, fc[2] = s.z;
                }
                for (int k = 0; k < 3; k++) bc[k] += (fc[k] - bc[k]) / (double)(it + 2);
            }
            for (int k = 0; k < 3; k++) bc[k] = (double)(float)(bc[k] - cen[k]) + cen[k];
            double rad = 0.0;
            for (size_t i = i0; i < i1; i++) {
                if (out.sph_f[i].w == -INFINITY) continue;
                const float4 &s = out.sph[i];
                const double dx = s.x - bc[0], dy = s.y - bc[1], dz = s.z - bc[2];
                rad = std::max(rad, std::sqrt(dx * dx + dy * dy + dz * dz) + std::sqrt((double)s.w));
            }
            rad *= 1.0 + 1e-6;
            if (entry(bc, rad * rad, &dst[g])) rmax = std::max(rmax, rad);
            else {
                // The ball cannot be represented: keep its members reachable with an always-pass ball.
                dst[g] = make_float4(0.f, 0.f, 0.f, INFINITY);
            }
        }
    };
    double grmax = 0.0, surmax = 0.0, chrmax = 0.0;
    balls(RT_GROUP, 1, out.grp_f, grmax);
    balls(RT_SUPER, 8, out.sup_f, surmax);
    balls(RT_CHUNK, 8, out.chk_f, chrmax);
    const double up = 1.0 + 1e-6;   // far above the single rounding to float
    out.sph_l1max = (float)(l1max * up);
    out.sph_sc = (float)((2.0 * p2max + r2max) * up);
    out.grp_rmax = (float)(grmax * up);
    out.sup_rmax = (float)(surmax * up);
    out.chk_rmax = (float)(chrmax * up);
}

// The bounding balls the primary rays' cone test reads (rt_kernels.hip, "Cone culling"): one per
// aligned run of `run` spheres as {Px, Py, Pz, R} in the scene's own coordinates.  Members are
// the spheres the reference can hit at all (finite data); the centre is found as above and
// rounded to FP32, the radius is measured against the rounded centre and rounded up.  R = -1: no
// members (nothing to hit); R = +inf: not representable (always passes).
void build_plain_balls(const PackedScene &out, size_t run, size_t pad, std::vector<float4> &dst)
{
    const size_t n = out.sph.size(), nb = (n + run - 1) / run;
    dst.assign((nb + pad - 1) / pad * pad, make_float4(0.f, 0.f, 0.f, -1.f));
    for (size_t g = 0; g < nb; g++) {
        const size_t i0 = g * run, i1 = std::min(n, (g + 1) * run);
        double blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
        bool any = false;
        for (size_t i = i0; i < i1; i++) {
            const float4 &s = out.sph[i];
            if (!usable_sphere(s)) continue;
            const double c[3] = {s.x, s.y, s.z}, r = std::sqrt((double)s.w);
            for (int k = 0; k < 3; k++) blo[k] = std::min(blo[k], c[k] - r), bhi[k] = std::max(bhi[k], c[k] + r);
            any = true;
        }
        if (!any) continue;
        double bc[3];
        for (int k = 0; k < 3; k++) bc[k] = 0.5 * (blo[k] + bhi[k]);
        for (int it = 0; it < 64; it++) {
            double far = -1.0, fc[3] = {bc[0], bc[1], bc[2]};
            for (size_t i = i0; i < i1; i++) {
                const float4 &s = out.sph[i];
                if (!usable_sphere(s)) continue;
                const double dx = s.x - bc[0], dy = s.y - bc[1], dz = s.z - bc[2];
                const double dist = std::sqrt(dx * dx + dy * dy + dz * dz) + std::sqrt((double)s.w);
                if (dist > far) far = dist, fc[0] = s.x, fc[1] = s.y, fc[2] = s.z;
            }
            for (int k = 0; k < 3; k++) bc[k] += (fc[k] - bc[k]) / (double)(it + 2);
        }
        const float P[3] = {(float)bc[0], (float)bc[1], (float)bc[2]};
        double rad = 0.0;
        for (size_t i = i0; i < i1; i++) {
            const float4 &s = out.sph[i];
            if (!usable_sphere(s)) continue;
            const double dx = s.x - (double)P[0], dy = s.y - (double)P[1], dz = s.z - (double)P[2];
            rad = std::max(rad, std::sqrt(dx * dx + dy * dy + dz * dz) + std::sqrt((double)s.w));
        }
        const float R = std::nextafter((float)(rad * (1.0 + 1e-6)), INFINITY);
        const bool ok = std::isfinite(P[0]) && std::isfinite(P[1]) && std::isfinite(P[2]) && std::isfinite(R);
        dst[g] = ok ? make_float4(P[0], P[1], P[2], R) : make_float4(0.f, 0.f, 0.f, INFINITY);
    }
}


// ---------------------------------------------------------------------------------------
// Per-light direction tables: which spheres can occlude a light ray.
//
// Every light ray (IsInShadow, MyApplication.cs:370-380, called from objects.cs:56) ends at one
// of a few fixed points, the lights.  Seen from a light L, sphere i covers a cap of directions;
// a ray from a hit point h towards L can only meet sphere i if the direction of h - L lies in
// that cap.  So per light the directions are cut into bins (a cube map: six faces of n x n
// bins, face = largest component of h - L, (u, v) = the other two divided by it) and every bin
// lists the spheres whose cap touches it; the kernel looks up the bin of h - L and runs the
// exact, literal Sphere.Intersect over that short list instead of walking the scene.
//
// The lists are conservative against everything the reference's FP32 evaluation can do.  With
// u = 2^-24, ray origin o = h and B >= |o - p_i| (rt_kernels.hip, "Sphere filter" and "Bounding
// balls"): if the reference accepts (ray, sphere i) then the ray's exact line passes within
//   R_i = sqrt(r_i^2 + 28.6u * B^2)
// of p_i, and the point e = o + t*d at the reference's own t (0 < t < distance, as it demands)
// lies within zeta = 8u * B of the part of the line inside that ball.  With the ray's direction
// d = fl(normalize(L - o)) and distance = fl(|L - o|), o + distance * d misses L by at most
// 4e-7 * |L - o|.  Hence either e is within m of L, and then |p_i - L| <= R_i + zeta + m + 4e-7
// * Dmax (such a sphere is listed in EVERY bin), or the direction of e - L, which is within
// asin((R_i + zeta) / |p_i - L|) of the direction of p_i - L, is within 4e-7 * Dmax / m + 2e-6 of
// the direction of o - L.  The tables are built for origins with |o - centre|_1 <= near (the
// kernel checks it per ray and evaluates the other rays against every sphere): B = near +
// max |p'|_1, Dmax = near + |L - centre|_1, m = 4e-3 * Dmax, and every cap is widened by 2e-4
// rad, which covers the 1e-4 above, the rounding of (u, v) on the device (< 1e-6) and the
// double arithmetic here.
//
// Tiers.  28.6u * B^2 is what the reference's cancellation in |o - p|^2 - r^2 can add to r^2, and for
// small spheres and a B that covers the whole served region it exceeds r^2 itself (r = 0.0175, B = 20:
// R_i = 1.84 r).  A SHORT ray cannot start far from a sphere it meets: if the reference accepts the
// pair at t (0 < t < distance), then with B* = |o - p_i| itself in the bounds above
//   B* <= t |d| + R_i(B*) + zeta(B*) <= distance (1 + 1e-6) + r_i + (sqrt(28.6u) + 8u) B*,
// so B* <= (distance (1 + 1e-6) + r_i) (1 + 2e-3).  The tables are therefore built several times
// over: tier k serves the rays (of served origins) whose length is at most d_k, with B the smaller
// of the bound above at distance = d_k and the region's, with Dmax = min(d_k (1 + 1e-6), the
// region's), and without the spheres that stay farther than d_k from the light (lt_reach); the last
// tier has no d_k and is the table described above.  The kernel picks the first tier whose d_k^2
// (1 - 1e-5) its own |o - L|^2 does not exceed (light_table_cell: that product differs from the
// square of the reference's `distance` by a few ulp).  d_1 = 240 times the median radius (R_i = 1.05
// r_i there), each further tier 1.6 times longer, up to the region's own bound or RT_LT_TIERS tiers.
// (65536 spheres of radius 0.005 .. 0.03, 16 lights: one tier 3.46 ms per frame, tiers of 4.2 and 6.7
// 2.77 ms; a second tier of 10.5 alone, 3.06.)
// ---------------------------------------------------------------------------------------
struct ConeNode {
    double cx, cy, cz, cosb, sinb;   // centre direction (face-local: u, v, axis) and angular radius of a uv square
};

// Bounding cones of the quadtree over one cube face, level by level down to n x n leaves:
// node (level, i, j) covers u in [-1 + 2i/2^level, -1 + 2(i+1)/2^level], v likewise.  The largest
// angle between the centre and a direction of the square is at a corner (a cap is convex under
// the gnomonic projection).
std::vector<std::vector<ConeNode>> face_cones(int n)
{
    std::vector<std::vector<ConeNode>> levels;
    for (int side = 1; side <= n; side *= 2) {
        std::vector<ConeNode> nodes((size_t)side * side);
        const double w = 2.0 / side;
        for (int j = 0; j < side; j++)
            for (int i = 0; i < side; i++) {
                const double u0 = -1.0 + w * i, v0 = -1.0 + w * j, uc = u0 + 0.5 * w, vc = v0 + 0.5 * w;
                const double len = std::sqrt(uc * uc + vc * vc + 1.0);
                ConeNode nd{uc / len, vc / len, 1.0 / len, 1.0, 0.0};
                double cosb = 1.0;
                for (int k = 0; k < 4; k++) {
                    const double u = u0 + (k & 1) * w, v = v0 + (k >> 1) * w;
                    cosb = std::min(cosb, (nd.cx * u + nd.cy * v + nd.cz) / std::sqrt(u * u + v * v + 1.0));
                }
                cosb = std::max(-1.0, std::min(1.0, cosb)) * (1.0 - 1e-12) - 1e-12;   // rounded towards a wider cone
                nd.cosb = cosb;
                nd.sinb = std::sqrt(std::max(0.0, 1.0 - cosb * cosb)) * (1.0 + 1e-12) + 1e-12;
                nodes[(size_t)j * side + i] = nd;
            }
        levels.push_back(std::move(nodes));
    }
    return levels;
}

int light_table_bins(size_t spheres)
{
    static const char *env = getenv("RT_HIP_LIGHT_BINS");   // tuning aid: bins per face edge (a power of two)
    if (env && atoi(env) >= 1) {
        int n = 1;
        while (n < atoi(env) && n < 1024) n *= 2;
        return n;
    }
    // (65536 spheres, 16 lights: 128 bins 4.04 ms per frame, 256 bins 4.20, 64 bins 4.65, 512 bins 4.74)
    return spheres <= 256 ? 32 : spheres <= 4096 ? 64 : spheres <= 262144 ? 128 : 256;
}

// Jobs 0 .. jobs - 1, each exactly once, on up to `workers` threads: the calling thread is one of them, and it
// does all the work itself when the system gives no further thread.  Every thread that was started is joined
// before this returns or throws; a job must not throw.
template <class Job>
void run_jobs(size_t jobs, size_t workers, Job &&job)
{
    std::atomic<size_t> next{0};
    auto drain = [&] {
        for (size_t k = next.fetch_add(1); k < jobs; k = next.fetch_add(1)) job(k);
    };
    struct Joiner {
        std::vector<std::thread> pool;
        ~Joiner()
        {
            for (std::thread &t : pool)
                if (t.joinable()) t.join();
        }
    } joiner;
    try {
        joiner.pool.reserve(workers);
        for (size_t w = 1; w < workers; w++) joiner.pool.emplace_back(drain);
    } catch (...) {
        // (thread or process limit reached: the threads that did start share the jobs with this one)
    }
    drain();
}

void build_light_tables(PackedScene &out)
{
    const size_t ns = out.sph.size(), nl = out.light_pos.size();
    out.lt_cells.clear();
    out.lt_sph.clear();
    out.lt_orig.clear();
    out.lt_reach.clear();
    out.lt_n = 0;
    out.lt_near = 0.f;
    out.lt_tiers = 0;
    if (ns == 0 || nl == 0) return;
    const int n = light_table_bins(ns);
    const double u24 = 1.0 / 16777216.0;
    double cen[3] = {out.sph_center[0], out.sph_center[1], out.sph_center[2]};
    double light_l1 = 0.0;
    for (const float4 &lp : out.light_pos) {
        const double l1 = std::fabs(lp.x - cen[0]) + std::fabs(lp.y - cen[1]) + std::fabs(lp.z - cen[2]);
        if (std::isfinite(l1)) light_l1 = std::max(light_l1, l1);
    }
    // Origins the tables serve: twice the spheres' extent around their centre, and at least as
    // far out as the lights.
    static const char *near_env = getenv("RT_HIP_LIGHT_NEAR");   // tuning aid: the factor below
    const double near = (near_env && atof(near_env) > 0.0 ? atof(near_env) : 2.0) * std::max((double)out.sph_l1max, light_l1);
    // The tiers' ray lengths d_k (the last tier: none).  RT_HIP_LIGHT_TIERS=1 (tuning aid): the last tier only.
    std::vector<double> tier_d;
    {
        std::vector<float> radii;
        for (const float4 &sp : out.sph)
            if (usable_sphere(sp)) radii.push_back(std::sqrt(sp.w));
        const char *tiers_env = getenv("RT_HIP_LIGHT_TIERS");
        const int most = tiers_env && atoi(tiers_env) >= 1 ? std::min(atoi(tiers_env), RT_LT_TIERS) : RT_LT_TIERS;
        if (!radii.empty()) {
            std::nth_element(radii.begin(), radii.begin() + radii.size() / 2, radii.end());
            const char *d1_env = getenv("RT_HIP_LIGHT_TIER_D1");   // tuning aid: the first tier's ray length
            double dk = d1_env && atof(d1_env) > 0.0 ? atof(d1_env) : 240.0 * (double)radii[radii.size() / 2];
            // (a tier that reaches the region's bound gains nothing over the last one)
            while ((int)tier_d.size() + 1 < most && dk > 0.0 && dk < 0.9 * (near + light_l1)) {
                tier_d.push_back(dk);
                const char *ratio_env = getenv("RT_HIP_LIGHT_TIER_RATIO");   // tuning aid
                dk *= ratio_env && atof(ratio_env) > 1.0 ? atof(ratio_env) : 1.6;
            }
        }
        tier_d.push_back(INFINITY);
    }
    const size_t tiers = tier_d.size();
    // (B >= |o - p_i| per sphere: the origin's bound plus the sphere's own distance from the centre, or what
    // the tier's ray length allows)
    auto widened = [&](const float4 &sp, const double dk) {
        double B = near + std::fabs(sp.x - cen[0]) + std::fabs(sp.y - cen[1]) + std::fabs(sp.z - cen[2]);
        if (std::isfinite(dk)) B = std::min(B, (dk * (1.0 + 1e-6) + std::sqrt((double)sp.w)) * (1.0 + 2e-3));
        return (std::sqrt((double)sp.w + 28.6 * u24 * B * B) + 8.0 * u24 * B) * (1.0 + 1e-9);
    };
    const auto cones = face_cones(n);
    const int leaf_level = (int)cones.size() - 1;
    const size_t cells_per_light = (size_t)6 * n * n;
    out.lt_cells.assign(tiers * nl * cells_per_light, uint2{0u, 0u});

    struct PerLight {
        std::vector<uint32_t> cell, sphere;   // (bin, sphere) pairs in the order found
        std::vector<float> reach;             // per sphere: lower bound of the distance light -> widened sphere
        bool failed = false;
    };
    std::vector<PerLight> per(tiers * nl);   // (tier, light) = per[tier * nl + light]
    auto one_light = [&](size_t job) {
        try {
            PerLight &pl = per[job];
            const size_t l = job % nl;
            const double dk = tier_d[job / nl];
            const float4 lp = out.light_pos[l];
            const double L[3] = {lp.x, lp.y, lp.z};
            const double Dmax = std::min(near + std::fabs(L[0] - cen[0]) + std::fabs(L[1] - cen[1]) + std::fabs(L[2] - cen[2]),
                                         dk * (1.0 + 1e-6));
            const double m = 4e-3 * Dmax, dl = 4e-7 * Dmax;
            std::vector<uint32_t> stack;
            // Spheres in the order of how close they come to the light, so that every bin's list
            // comes out sorted that way (the kernel stops a list at the ray's own length).
            pl.reach.assign(ns, 0.f);
            std::vector<uint32_t> by_reach;
            for (size_t i = 0; i < ns; i++) {
                const float4 &sp = out.sph[i];
                if (!usable_sphere(sp)) continue;
                const double dx = sp.x - L[0], dy = sp.y - L[1], dz = sp.z - L[2];
                const double reach = std::sqrt(dx * dx + dy * dy + dz * dz) - widened(sp, dk) - dl;
                // rounded down; a ray of length `distance` can only meet spheres with reach <= distance
                pl.reach[i] = std::isfinite(reach) ? std::nextafter((float)(reach - 1e-6 * std::fabs(reach)), -INFINITY) : -INFINITY;
                if ((double)pl.reach[i] > dk) continue;   // no ray of this tier is that long
                by_reach.push_back((uint32_t)i);
            }
            std::stable_sort(by_reach.begin(), by_reach.end(), [&](uint32_t a, uint32_t b) { return pl.reach[a] < pl.reach[b]; });
            for (const uint32_t i : by_reach) {
                const float4 &sp = out.sph[i];
                const double R = widened(sp, dk);
                const double dx = sp.x - L[0], dy = sp.y - L[1], dz = sp.z - L[2];
                const double dist = std::sqrt(dx * dx + dy * dy + dz * dz);
                double cosa = 1.0, sina = 0.0;
                bool everywhere = !(dist > R + m + dl) || !std::isfinite(dist);
                if (!everywhere) {
                    const double alpha = std::asin(std::min(1.0, R / dist)) + 2e-4;
                    if (alpha >= 1.5) everywhere = true;
                    cosa = std::cos(alpha), sina = std::sin(alpha);
                }
                const double vdir[3] = {dx / dist, dy / dist, dz / dist};
                for (int face = 0; face < 6; face++) {
                    // face-local coordinates: (u, v, axis) with the axis component positive on the face
                    const int axis = face >> 1, ua = (axis + 1) % 3, va = (axis + 2) % 3;
                    const double sgn = (face & 1) ? -1.0 : 1.0;
                    const double a = everywhere ? 0.0 : vdir[ua] * sgn, b = everywhere ? 0.0 : vdir[va] * sgn,
                                 w = everywhere ? 0.0 : vdir[axis] * sgn;
                    stack.clear();
                    stack.push_back(0u);   // (level << 24) | node index does not fit for n = 1024: two words
                    stack.push_back(0u);
                    while (!stack.empty()) {
                        const uint32_t idx = stack.back();
                        stack.pop_back();
                        const uint32_t lv = stack.back();
                        stack.pop_back();
                        const ConeNode &nd = cones[lv][idx];
                        if (!everywhere) {
                            // reject when the angle between the cap's axis and the node's centre exceeds alpha + beta
                            const double c = nd.cx * a + nd.cy * b + nd.cz * w;
                            // (alpha <= 1.5 and beta <= 0.96 rad: the sum stays below pi, where cos decreases)
                            const double cos_sum = cosa * nd.cosb - sina * nd.sinb;   // cos(alpha + beta)
                            if (c < cos_sum - 1e-12) continue;
                        }
                        const uint32_t side = 1u << lv, i0 = idx % side, j0 = idx / side;
                        if ((int)lv == leaf_level) {
                            pl.cell.push_back((uint32_t)face * (uint32_t)(n * n) + idx);
                            pl.sphere.push_back((uint32_t)i);
                        } else {
                            for (uint32_t k = 0; k < 4; k++) {
                                stack.push_back(lv + 1);
                                stack.push_back((2 * j0 + (k >> 1)) * (2 * side) + 2 * i0 + (k & 1));
                            }
                        }
                    }
                }
            }
        } catch (...) {
            per[job].failed = true;
        }
    };
    // A few host threads take the (tier, light) tables one after the other, the last tier's (the longest
    // lists) first.
    const size_t jobs = tiers * nl;
    const size_t workers = std::max<size_t>(1, std::min<size_t>({jobs, (size_t)std::thread::hardware_concurrency(), 16}));
    run_jobs(jobs, workers, [&](size_t k) { one_light(jobs - 1 - k); });
    for (const PerLight &pl : per)
        if (pl.failed) throw std::bad_alloc();
    size_t total = 0;
    for (const PerLight &pl : per) total += pl.cell.size();
    if (total >= (size_t)1 << 31) throw std::length_error("the per-light occluder lists exceed 2^31 entries");
    // (followed by entries that never occlude: the kernels load a little past a list's end)
    out.lt_sph.assign(total + RT_LT_PAD, float4{0.f, 0.f, 0.f, -INFINITY});
    out.lt_orig.assign(total + RT_LT_PAD, INT_MAX);
    out.lt_reach.assign(total + RT_LT_PAD, INFINITY);
    // Every table's lists lie one behind the other; the tables are filled in side by side.
    std::vector<size_t> first(jobs + 1, 0);
    for (size_t l = 0; l < jobs; l++) first[l + 1] = first[l] + per[l].cell.size();
    auto fill = [&](size_t l) {
        const PerLight &pl = per[l];
        uint2 *cells = out.lt_cells.data() + l * cells_per_light;
        for (uint32_t c : pl.cell) cells[c].y++;
        size_t off = first[l];
        for (size_t c = 0; c < cells_per_light; c++) {
            cells[c].x = (uint32_t)off;
            off += cells[c].y;
            cells[c].y = 0;
        }
        for (size_t k = 0; k < pl.cell.size(); k++) {
            uint2 &c = cells[pl.cell[k]];
            out.lt_sph[(size_t)c.x + c.y] = out.sph[pl.sphere[k]];
            out.lt_orig[(size_t)c.x + c.y] = out.sph_orig[pl.sphere[k]];
            out.lt_reach[(size_t)c.x + c.y] = pl.reach[pl.sphere[k]];
            c.y++;
        }
    };
    run_jobs(jobs, workers, [&](size_t k) { fill(jobs - 1 - k); });
    out.lt_n = n;
    out.lt_near = (float)(near * (1.0 - 1e-6));
    out.lt_tiers = (int)tiers;
    for (size_t k = 0; k + 1 < tiers; k++) {
        // (rounded down; the kernel's |o - L|^2 and the square of the reference's `distance` differ by a few ulp)
        out.lt_tier_q[k] = std::nextafter((float)(tier_d[k] * tier_d[k] * (1.0 - 1e-5)), 0.0f);
    }
}

// ---------------------------------------------------------------------------------------
// Uniform grid over the spheres: what a reflection ray's closest-hit search walks instead of the
// whole scene (rt_kernels.hip, "Grid walk").  A cell lists the spheres whose WIDENED ball comes
// within the walk's slack of it; the kernel runs the exact, literal Sphere.Intersect over the lists of the cells the ray
// passes, nearest first, and stops once its closest hit so far lies before the parameter at which
// it leaves the cells visited.  What makes that lose nothing the reference accepts:
//   * If the reference accepts the pair (ray, sphere i) with parameter t_ref, the point o + t_ref d
//     of the real ray lies in the ball of radius R_i = sqrt(r_i^2 + 28.6u B^2) + 8u B around the
//     centre, B >= |o - p_i|, u = 2^-24 (the bound the sphere filter's proof gives for the line's
//     distance from the centre, rt_kernels.hip "Bounding balls", plus the rounding of t_ref itself:
//     the computed discriminant / 4 is at most r^2 - perp^2 + 20u B^2, so the computed hit lies
//     no farther than that root from the point of closest approach).  The lists are built with
//     B = the largest distance between an origin in the served region (gr_reg_lo .. gr_reg_hi) and
//     a sphere centre; the kernel evaluates every sphere for origins outside that region.
//   * The walk's cell sequence is that of a ray whose plane crossings are off by at most a few
//     1e-4 cells (cell coordinates up to ~400 in magnitude, a handful of roundings): every point
//     of the real ray up to the parameter at which the walk leaves a cell lies within GRID_EPS
//     cells (per axis) of a visited cell, and a sphere is listed in every cell whose distance from
//     its centre is at most R_i plus that slack as a length.  So if the hit the reference accepts
//     lies before the parameter at which the walk leaves its last cell, some visited cell lists the
//     sphere; a sphere in none of the visited cells' lists is hit, if at all, beyond that parameter.
// The grid is defined by the FP32 parameters the kernel uses (cell coordinate of x = (x - gr_lo) *
// gr_ih in real arithmetic); the boxes are binned against those same numbers in double.
// No grid (the kernels that walk the LDS-staged scene are used): fewer than RT_GRID_MIN_SPHERES usable
// spheres, extents that are not finite, or lists that would grow out of proportion (a sphere that
// covers thousands of cells), or a scene of one chunk (below).  RT_HIP_GRID_DENSITY = cells
// per sphere (default 2); RT_HIP_GRID_REGION = how far the served region reaches beyond the spheres'
// extent, in units of its largest side (default 1).
// ---------------------------------------------------------------------------------------
constexpr size_t RT_GRID_MIN_SPHERES = 64;
constexpr int RT_GRID_MAX_AXIS = 256;
constexpr double GRID_EPS = 4e-3;   // cells

void build_sphere_grid(PackedScene &out)
{
    out.gr_cells.clear();
    out.gr_sph.clear();
    out.gr_orig.clear();
    out.gr_n[0] = out.gr_n[1] = out.gr_n[2] = 0;
    out.gr_tier_t = -1.0f;
    // Scenes whose spheres fit one LDS chunk keep the kernels that stage it: their wave-wide culling costs fewer
    // instructions per ray than one walk per lane, and a frame is bound by the instructions it issues (1000
    // spheres, 3840x2160: 0.595 ms per frame against 0.616 with the grid; 256x256: 0.16 against 0.20), while
    // scenes of many chunks gain (65536 spheres, 1920x1080: 4.2 against 6.1 ms).  RT_HIP_GRID=2 builds and
    // uses the grid whatever the sphere count, RT_HIP_GRID=0 never (tuning aids, read by rt_set_scene).
    const char *env = getenv("RT_HIP_GRID");
    const int mode = env ? atoi(env) : 1;
    if (mode == 0 || (mode != 2 && out.sph.size() <= (size_t)RT_CHUNK)) return;
    const char *dens_env = getenv("RT_HIP_GRID_DENSITY"), *reg_env = getenv("RT_HIP_GRID_REGION");
    const double density = dens_env && atof(dens_env) > 0.0 ? atof(dens_env) : 2.0;
    // (an origin outside the region evaluates every sphere: with the 64k-sphere scene's room wall 0.37 sides
    // away from the spheres, a region of 0.25 sides costs 84 ms per frame against 3.6; the wider region's wider
    // lists cost 1 %)
    const double reach = reg_env && atof(reg_env) >= 0.0 ? atof(reg_env) : 1.0;
    const double u24 = 1.0 / 16777216.0;
    std::vector<uint32_t> use;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i < out.sph.size(); i++) {
        const float4 &sp = out.sph[i];
        if (!usable_sphere(sp) || out.sph_orig[i] == INT32_MAX) continue;
        use.push_back((uint32_t)i);
        const double c[3] = {sp.x, sp.y, sp.z};
        for (int k = 0; k < 3; k++) lo[k] = std::min(lo[k], c[k]), hi[k] = std::max(hi[k], c[k]);
    }
    if (use.size() < RT_GRID_MIN_SPHERES) return;
    double ext = 0.0;
    for (int k = 0; k < 3; k++) ext = std::max(ext, hi[k] - lo[k]);
    if (!std::isfinite(ext) || !(ext > 0.0)) return;
    // The served region of origins and the bound B it implies (per axis: centre and origin both lie
    // within the region, the centre within the centres' extent).
    double reg_lo[3], reg_hi[3], B2 = 0.0;
    for (int k = 0; k < 3; k++) {
        reg_lo[k] = lo[k] - reach * ext;
        reg_hi[k] = hi[k] + reach * ext;
        const double span = (hi[k] - lo[k]) + reach * ext;
        B2 += span * span;
    }
    const double Bmax = std::sqrt(B2) * (1.0 + 1e-6);
    // Two sets of lists over the same cells.  The second serves the cells a ray leaves at a parameter of at
    // most d1: a hit the reference accepts there lies no farther than d1 (1 + 1e-6) from the origin, so the
    // origin is at most (d1 (1 + 1e-6) + r_i) (1 + 2e-3) from the centre (build_light_tables, "Tiers": the
    // same argument with the reflection ray's unit direction), and with that B instead of the region's the
    // widened ball of a small sphere is barely larger than the sphere (r = 0.0175: 1.05 r against 1.8 r).
    // d1 = 240 times the median radius; no second set when that reaches the region's bound.
    // RT_HIP_GRID_TIER_D1 (tuning aid): d1 itself; 0 = no second set.
    double d1 = 0.0;
    {
        std::vector<float> radii;
        for (const uint32_t i : use) radii.push_back(std::sqrt(out.sph[i].w));
        std::nth_element(radii.begin(), radii.begin() + radii.size() / 2, radii.end());
        const char *d1_env = getenv("RT_HIP_GRID_TIER_D1");
        d1 = d1_env ? atof(d1_env) : 240.0 * (double)radii[radii.size() / 2];
        if (!(d1 > 0.0) || !(d1 < 0.9 * Bmax)) d1 = 0.0;
    }
    auto widened = [&](const float4 &sp, const double dk) {
        const double B = dk > 0.0 ? std::min(Bmax, (dk * (1.0 + 1e-6) + std::sqrt((double)sp.w)) * (1.0 + 2e-3)) : Bmax;
        return (std::sqrt((double)sp.w + 28.6 * u24 * B * B) + 8.0 * u24 * B) * (1.0 + 1e-6) + 1e-6 * B;
    };
    double glo[3] = {INFINITY, INFINITY, INFINITY}, ghi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (const uint32_t i : use) {
        const float4 &sp = out.sph[i];
        const double R = widened(sp, 0.0), c[3] = {sp.x, sp.y, sp.z};
        for (int k = 0; k < 3; k++) glo[k] = std::min(glo[k], c[k] - R), ghi[k] = std::max(ghi[k], c[k] + R);
    }
    double vol = 1.0, side[3];
    for (int k = 0; k < 3; k++) {
        side[k] = ghi[k] - glo[k];
        if (!std::isfinite(side[k]) || !(side[k] > 0.0)) return;
        vol *= side[k];
    }
    const double h = std::cbrt(vol / (density * (double)use.size()));
    if (!std::isfinite(h) || !(h > 0.0)) return;
    int n[3];
    float lo_f[3], ih_f[3];
    for (int k = 0; k < 3; k++) {
        n[k] = (int)std::min<double>(RT_GRID_MAX_AXIS, std::max(1.0, std::ceil(side[k] / h)));
        // One per cent of a cell of room on either side of the boxes, then FP32 parameters that
        // still cover them: the corner rounded down, the scale rounded towards zero.
        const double cell = side[k] / n[k];
        const double l = glo[k] - 0.01 * cell - 1e-6 * std::fabs(glo[k]), hgh = ghi[k] + 0.01 * cell + 1e-6 * std::fabs(ghi[k]);
        lo_f[k] = std::nextafter((float)l, -INFINITY);
        ih_f[k] = std::nextafter((float)((double)n[k] / (hgh - (double)lo_f[k]) * (1.0 - 1e-6)), 0.f);
        if (!std::isfinite(lo_f[k]) || !std::isfinite(ih_f[k]) || !(ih_f[k] > 0.f)) return;
        if (((double)ghi[k] - (double)lo_f[k]) * (double)ih_f[k] > (double)n[k] - 0.005) return;   // (cannot happen)
    }
    const size_t ncells = (size_t)n[0] * n[1] * n[2];
    const size_t sets = d1 > 0.0 ? 2 : 1;
    // Cell ranges of the boxes (those of the region's B: the second set's boxes lie inside them).
    struct Range { int a[3], b[3]; };
    std::vector<Range> ranges(use.size());
    size_t total = 0;
    for (size_t j = 0; j < use.size(); j++) {
        const float4 &sp = out.sph[use[j]];
        const double R = widened(sp, 0.0), c[3] = {sp.x, sp.y, sp.z};
        size_t cover = 1;
        for (int k = 0; k < 3; k++) {
            const double a = ((c[k] - R) - (double)lo_f[k]) * (double)ih_f[k] - GRID_EPS;
            const double b = ((c[k] + R) - (double)lo_f[k]) * (double)ih_f[k] + GRID_EPS;
            ranges[j].a[k] = (int)std::min<double>(n[k] - 1, std::max(0.0, std::floor(a)));
            ranges[j].b[k] = (int)std::min<double>(n[k] - 1, std::max(0.0, std::floor(b)));
            cover *= (size_t)(ranges[j].b[k] - ranges[j].a[k] + 1);
        }
        if (cover > 4096) return;   // a sphere far larger than the cells: the lists would be the scene
        total += cover;
    }
    if (total > 64 * use.size() + 4096 || sets * total >= ((size_t)1 << 30)) return;
    out.gr_cells.assign(sets * ncells, uint2{0u, 0u});
    auto cell_of = [&](int x, int y, int z) { return ((size_t)z * n[1] + y) * n[0] + x; };
    // Of the cells its box covers a sphere is listed in those that come within R (+ the walk's GRID_EPS cells, as
    // a distance) of its centre: the corners of the box do not.
    double eps_len = 0.0;
    for (int k = 0; k < 3; k++) eps_len = std::max(eps_len, GRID_EPS / (double)ih_f[k]);
    eps_len *= 1.7321;
    auto for_cells = [&](size_t set, size_t j, auto &&visit) {
        const Range &r = ranges[j];
        const float4 &sp = out.sph[use[j]];
        const double R = widened(sp, set == 0 ? 0.0 : d1) + eps_len, c[3] = {sp.x, sp.y, sp.z};
        double gap2[3][RT_GRID_MAX_AXIS];   // per axis and cell of the range: squared distance of the centre from the cell's slab
        for (int k = 0; k < 3; k++)
            for (int i = r.a[k]; i <= r.b[k]; i++) {
                const double lo_i = (double)lo_f[k] + (double)i / (double)ih_f[k], hi_i = (double)lo_f[k] + (double)(i + 1) / (double)ih_f[k];
                const double gap = c[k] < lo_i ? lo_i - c[k] : (c[k] > hi_i ? c[k] - hi_i : 0.0);
                gap2[k][i] = gap * gap;
            }
        const double R2 = R * R * (1.0 + 1e-9);
        for (int z = r.a[2]; z <= r.b[2]; z++)
            for (int y = r.a[1]; y <= r.b[1]; y++)
                for (int x = r.a[0]; x <= r.b[0]; x++)
                    if (gap2[0][x] + gap2[1][y] + gap2[2][z] <= R2) visit(out.gr_cells[set * ncells + cell_of(x, y, z)]);
    };
    for (size_t set = 0; set < sets; set++)
        for (size_t j = 0; j < use.size(); j++) for_cells(set, j, [](uint2 &c) { c.y++; });
    // Every list starts at an even entry (the kernel loads its entries in pairs); the gaps and the
    // arrays' ends hold entries that can never be hit.
    size_t off = 0;
    for (uint2 &c : out.gr_cells) {
        c.x = (uint32_t)off;
        off += (c.y + 1u) & ~1u;
        c.y = 0;
    }
    if (off >= ((size_t)1 << 31)) {
        out.gr_cells.clear();
        return;
    }
    out.gr_sph.assign(off + RT_GRID_PAD, float4{0.f, 0.f, 0.f, -INFINITY});
    out.gr_orig.assign(off + RT_GRID_PAD, INT32_MAX);
    for (size_t set = 0; set < sets; set++)
        for (size_t j = 0; j < use.size(); j++)
            for_cells(set, j, [&](uint2 &c) {
                out.gr_sph[(size_t)c.x + c.y] = out.sph[use[j]];
                out.gr_orig[(size_t)c.x + c.y] = out.sph_orig[use[j]];
                c.y++;
            });
    // (rounded down; the walk compares the parameter at which the ray leaves a cell with it)
    out.gr_tier_t = d1 > 0.0 ? std::nextafter((float)(d1 * (1.0 - 1e-5)), 0.0f) : -1.0f;
    for (int k = 0; k < 3; k++) {
        out.gr_n[k] = n[k];
        out.gr_lo[k] = lo_f[k];
        out.gr_ih[k] = ih_f[k];
        // (rounded inwards)
        out.gr_reg_lo[k] = std::nextafter((float)reg_lo[k], INFINITY);
        out.gr_reg_hi[k] = std::nextafter((float)reg_hi[k], -INFINITY);
    }
}

int pack_scene(rt_ctx *ctx, const rt_scene *s, PackedScene &out)
{
    const int n = s->num_prims, nl = s->num_lights;
    out.geo.resize(n);
    out.mat.resize(3 * (size_t)n);
    out.class_prefix.assign(3 * ((size_t)n + 1), 0);
    int cs = 0, cp = 0, ct = 0;
    for (int i = 0; i < n; i++) {
        const int type = s->prim_types[i];
        if (type < RT_PRIM_SPHERE || type > RT_PRIM_TRIANGLE)
            return fail(ctx, RT_ERR_INVALID_ARG, "prim_types[%d] = %d is not a primitive type", i, type);
        out.class_prefix[3 * (size_t)i + 0] = cs;
        out.class_prefix[3 * (size_t)i + 1] = cp;
        out.class_prefix[3 * (size_t)i + 2] = ct;
        int flags = type;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (type == RT_PRIM_SPHERE) {
            const float *p = s->prim_positions + 3 * (size_t)i;
            const float r = s->prim_radius[i];
            out.sph.push_back(make_float4(p[0], p[1], p[2], r * r));  // objects.cs:114 radius*radius
            out.sph_orig.push_back(i);
            g = make_float4(p[0], p[1], p[2], 0.f);
            cs++;
        } else if (type == RT_PRIM_PLANE) {
            const float *nn = s->prim_normals + 3 * (size_t)i;
            out.pl.push_back(make_float4(nn[0], nn[1], nn[2], s->prim_d[i]));
            out.pl_orig.push_back(i);
            g = make_float4(nn[0], nn[1], nn[2], 0.f);
            if (s->prim_checkerboard[i]) flags |= RT_MF_CHECKER;
            const int tex = s->prim_texture ? s->prim_texture[i] : -1;
            if (tex >= RT_MAX_TEXTURES)
                return fail(ctx, RT_ERR_INVALID_ARG, "prim_texture[%d] = %d exceeds %d slots", i, tex,
                            RT_MAX_TEXTURES);
            if (tex >= 0) {
                flags |= (tex + 1) << RT_MF_TEX_SHIFT;
                out.textures_used |= 1u << tex;
            }
            cp++;
        } else {
            const float *v0 = s->prim_v0 + 3 * (size_t)i, *v1 = s->prim_v1 + 3 * (size_t)i,
                        *v2 = s->prim_v2 + 3 * (size_t)i;
            const float *nn = s->prim_normals + 3 * (size_t)i;
            out.tri.push_back(make_float4(v0[0], v0[1], v0[2], 0.f));
            out.tri.push_back(make_float4(v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2], 0.f));  // e1
            out.tri.push_back(make_float4(v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2], 0.f));  // e2
            out.tri_orig.push_back(i);
            g = make_float4(nn[0], nn[1], nn[2], 0.f);
            ct++;
        }
        out.geo[i] = g;
        const float *dc = s->prim_diffuse + 4 * (size_t)i, *gc = s->prim_glossy + 4 * (size_t)i,
                    *ac = s->prim_ambient + 4 * (size_t)i;
        float flags_as_float;
        memcpy(&flags_as_float, &flags, sizeof flags);
        out.mat[3 * (size_t)i + 0] = make_float4(dc[0], dc[1], dc[2], s->prim_specular[i]);
        out.mat[3 * (size_t)i + 1] = make_float4(gc[0], gc[1], gc[2], flags_as_float);
        out.mat[3 * (size_t)i + 2] = make_float4(ac[0], ac[1], ac[2], 0.f);
    }
    while (out.sph.size() % RT_SPH_PAD != 0) {
        out.sph.push_back(make_float4(0.f, 0.f, 0.f, -INFINITY));
        out.sph_orig.push_back(INT32_MAX);
    }
    reorder_spheres(out);
    build_filter_arrays(out);
    build_plain_balls(out, RT_SUPER, 8, out.sup_b);
    build_plain_balls(out, RT_CHUNK, 8, out.chk_b);
    out.class_prefix[3 * (size_t)n + 0] = cs;
    out.class_prefix[3 * (size_t)n + 1] = cp;
    out.class_prefix[3 * (size_t)n + 2] = ct;
    out.light_pos.resize(nl);
    out.light_col.resize(nl);
    for (int l = 0; l < nl; l++) {
        const float *p = s->light_positions + 3 * (size_t)l, *c = s->light_colors + 4 * (size_t)l;
        out.light_pos[l] = make_float4(p[0], p[1], p[2], 0.f);
        out.light_col[l] = make_float4(c[0], c[1], c[2], c[3]);
    }
    // RT_HIP_SCENE_TRACE=1 (diagnostic): what the selectors' tables took to build and hold, on stderr.
    const bool trace = getenv("RT_HIP_SCENE_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    build_light_tables(out);
    const auto t1 = std::chrono::steady_clock::now();
    build_sphere_grid(out);
    const auto t2 = std::chrono::steady_clock::now();
    if (trace)
        fprintf(stderr, "rt_hip: lights' tables %d tiers x %d lights x 6 x %d^2 cells, %zu entries, %.1f ms; grid %d x %d x %d, %zu entries, %.1f ms\n",
                out.lt_tiers, nl, out.lt_n, out.lt_sph.size(), std::chrono::duration<double, std::milli>(t1 - t0).count(),
                out.gr_n[0], out.gr_n[1], out.gr_n[2], out.gr_sph.size(), std::chrono::duration<double, std::milli>(t2 - t1).count());
    return RT_OK;
}

// The arrays of a scene on their way to one device allocation (upload_scene).
struct SceneSlab {
    struct Item {
        const void *src;
        size_t bytes, offset;
        const void **dst;
    };
    std::vector<Item> items;
    size_t total = 0;
    template <typename T>
    void add(const std::vector<T> &host, const T **dev_out)
    {
        // Never an empty array: kernels index element 0 only when a count says so, but a pointer into the
        // allocation keeps every launch argument well formed.
        const size_t bytes = host.size() * sizeof(T), room = std::max<size_t>(bytes, 16);
        const size_t align = room >= ((size_t)2 << 20) ? (size_t)2 << 20 : 4096;
        total = (total + align - 1) / align * align;
        items.push_back(Item{host.data(), bytes, total, reinterpret_cast<const void **>(dev_out)});
        total += room;
    }
    int upload(rt_ctx *ctx, DeviceState &d)
    {
        char *base = nullptr;
        const size_t slack = (size_t)2 << 20;
        HIP_TRY(ctx, hipMalloc((void **)&base, total + slack));
        d.scene_allocs.push_back(base);
        char *start = reinterpret_cast<char *>(((uintptr_t)base + slack - 1) / slack * slack);
        for (const Item &it : items) {
            if (it.bytes) HIP_TRY(ctx, hipMemcpy(start + it.offset, it.src, it.bytes, hipMemcpyHostToDevice));
            *it.dst = start + it.offset;
        }
        return RT_OK;
    }
};

int upload_scene(rt_ctx *ctx, DeviceState &d, const PackedScene &ps, int n, int nl)
{
    HIP_TRY(ctx, hipSetDevice(d.device));
    HIP_TRY(ctx, hipStreamSynchronize(d.stream));
    free_scene(d);
    DevScene &sc = d.scene;
    memset(d.resident_blocks, 0, sizeof d.resident_blocks);
    sc.ns_padded = (int)ps.sph.size();
    sc.ns = ps.class_prefix[3 * (size_t)n];
    sc.np = (int)ps.pl.size();
    sc.nt = (int)ps.tri_orig.size();
    sc.nl = nl;
    sc.nprims = n;
    // Every array of the scene lies in ONE allocation, each at a multiple of 2 MiB or (small ones) 4 KiB from its
    // 2 MiB-aligned start: where the tables land relative to one another and to page boundaries is the same
    // whatever the process allocated and freed before (and rt_set_scene makes one hipMalloc instead of 27).
    SceneSlab slab;
    slab.add(ps.sph, &sc.sph);
    slab.add(ps.sph_f, &sc.sph_f);
    slab.add(ps.grp_f, &sc.grp_f);
    slab.add(ps.sup_f, &sc.sup_f);
    slab.add(ps.chk_f, &sc.chk_f);
    slab.add(ps.sup_b, &sc.sup_b);
    slab.add(ps.chk_b, &sc.chk_b);
    slab.add(ps.sph_orig, &sc.sph_orig);
    slab.add(ps.pl, &sc.pl);
    slab.add(ps.pl_orig, &sc.pl_orig);
    slab.add(ps.tri, &sc.tri);
    slab.add(ps.tri_orig, &sc.tri_orig);
    slab.add(ps.geo, &sc.geo);
    slab.add(ps.mat, &sc.mat);
    slab.add(ps.light_pos, &sc.light_pos);
    slab.add(ps.light_col, &sc.light_col);
    slab.add(ps.class_prefix, &sc.class_prefix);
    slab.add(ps.lt_cells, &sc.lt_cells);
    slab.add(ps.lt_sph, &sc.lt_sph);
    slab.add(ps.lt_orig, &sc.lt_orig);
    slab.add(ps.lt_reach, &sc.lt_reach);
    slab.add(ps.gr_cells, &sc.gr_cells);
    slab.add(ps.gr_sph, &sc.gr_sph);
    slab.add(ps.gr_orig, &sc.gr_orig);
    int rc = slab.upload(ctx, d);
    if (rc != RT_OK) return rc;
    sc.lt_n = ps.lt_n;
    sc.lt_near = ps.lt_near;
    sc.lt_tiers = ps.lt_tiers;
    for (int k = 0; k < RT_LT_TIERS - 1; k++) sc.lt_tier_q[k] = ps.lt_tier_q[k];
    sc.gr_tier_t = ps.gr_tier_t;
    for (int k = 0; k < 3; k++) {
        sc.gr_n[k] = ps.gr_n[k];
        sc.gr_lo[k] = ps.gr_lo[k];
        sc.gr_ih[k] = ps.gr_ih[k];
        sc.gr_reg_lo[k] = ps.gr_reg_lo[k];
        sc.gr_reg_hi[k] = ps.gr_reg_hi[k];
    }
    sc.nsup_padded = (int)ps.sup_f.size();
    sc.nchk_padded = (int)ps.chk_f.size();
    sc.grp_rmax = ps.grp_rmax;
    sc.sup_rmax = ps.sup_rmax;
    sc.chk_rmax = ps.chk_rmax;
    for (int k = 0; k < 3; k++) sc.sph_center[k] = ps.sph_center[k];
    sc.sph_l1max = ps.sph_l1max;
    for (int k = 0; k < 3; k++) sc.bin_lo[k] = ps.bin_lo[k], sc.bin_scale[k] = ps.bin_scale[k];
    sc.sph_sc = ps.sph_sc;
    return RT_OK;
}

// GetRayDirection's per-frame part, MyApplication.cs:431-448.
// The rows a render call covers (DevCamera): `rows` rows starting at row_begin, or every
// parts-th band of `band` rows of the whole frame.
struct RowSet {
    int row_begin, rows, band, parts, part;
};

RowSet contiguous_rows(int y0, int y1) { return RowSet{y0, y1 - y0, 16, 1, 0}; }

// Rows of a height-row frame in the bands b = part, part + parts, ... of band_rows rows each.
int rows_of_part(int height, int band_rows, int part, int parts)
{
    const int bands = (height + band_rows - 1) / band_rows;
    long long rows = 0;
    for (int b = part; b < bands; b += parts) rows += std::min(band_rows, height - b * band_rows);
    return (int)rows;
}

DevCamera make_camera(const rt_camera *cam, int w, int h, const RowSet &rs)
{
    DevCamera c{};
    const float deg2rad = 3.14159274f / 180.0f;  // MathHelper.DegreesToRadians: MathF.PI / 180f
    const float aspect = (float)w / (float)h;
    const float angle_h = (cam->fov_deg * aspect) * deg2rad;
    const float angle_v = cam->fov_deg * deg2rad;
    c.tan_h = (float)std::tan((double)(angle_h / 2.0f));
    c.tan_v = (float)std::tan((double)(angle_v / 2.0f));
    const float *u = cam->up, *d = cam->direction;
    // Vector3.Cross(up, direction).Normalized()
    float lx = (u[1] * d[2]) - (u[2] * d[1]);
    float ly = (u[2] * d[0]) - (u[0] * d[2]);
    float lz = (u[0] * d[1]) - (u[1] * d[0]);
    const float scale = 1.0f / std::sqrt(((lx * lx) + (ly * ly)) + (lz * lz));
    c.left[0] = lx * scale;
    c.left[1] = ly * scale;
    c.left[2] = lz * scale;
    for (int k = 0; k < 3; k++) {
        c.pos[k] = cam->position[k];
        c.dir[k] = d[k];
        c.up[k] = u[k];
    }
    c.width = w;
    c.height = h;
    c.row_begin = rs.row_begin;
    c.rows = rs.rows;
    c.band = rs.band;
    c.parts = rs.parts;
    c.part = rs.part;
    return c;
}

// This device's copy of the fixed scratch array.  `device` must be the current device: the
// address hipGetSymbolAddress returns is the current device's.
hipError_t scratch_address(int device, unsigned char **out)
{
    std::lock_guard<std::mutex> lock(g_scratch_lock);
    if (device >= 0 && device < RT_SCRATCH_TABLE && g_scratch_address[device]) {
        *out = g_scratch_address[device];
        return hipSuccess;
    }
    void *p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    *out = static_cast<unsigned char *>(p);
    if (device >= 0 && device < RT_SCRATCH_TABLE) g_scratch_address[device] = *out;
    return hipSuccess;
}


// Streams, events, the status block and the fixed scratch's address of one device state (the device is current).
// aux_low_priority (RT_HIP_AUX_PRIORITY=1, an experiment): the second stream at the lowest stream priority, i.e. in a
// priority class, and therefore on a hardware queue, no ordinary stream shares.  Not the default: on the pool this
// library was measured on a frame's launches on such a stream now and then never start (the test-suite stalls in
// one of its ~100 frames), like the high-priority copy stream that stalls two processes sharing a GPU.
hipError_t create_device_objects(DeviceState &d, bool aux_low_priority)
{
    hipError_t e = hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking);
    for (int k = 0; k < 5 && e == hipSuccess; k++) e = hipEventCreate(&d.ev[k]);
    int least = 0, greatest = 0;
    if (e == hipSuccess && aux_low_priority && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) {
        e = hipStreamCreateWithPriority(&d.aux, hipStreamNonBlocking, least);
        if (e != hipSuccess) {   // (no such stream to be had: an ordinary one)
            (void)hipGetLastError();
            d.aux = nullptr;
            e = hipSuccess;
        }
    }
    if (e == hipSuccess && !d.aux) e = hipStreamCreateWithFlags(&d.aux, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d.ev_fork, hipEventDisableTiming);
    for (int k = 0; k <= RT_MAX_DEPTH && e == hipSuccess; k++) e = hipEventCreateWithFlags(&d.ev_level[k], hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc((void **)&d.status, RT_STATUS_BYTES, hipHostMallocMapped);
    if (e == hipSuccess) memset(d.status, 0, RT_STATUS_BYTES);
    if (e == hipSuccess) e = scratch_address(d.device, &d.scratch_base);
    if (e == hipSuccess) e = hipMalloc((void **)&d.probe, 64);
    return e;
}

// Everything a device state owns.
void destroy_device_state(DeviceState &d)
{
    (void)hipSetDevice(d.device);
    if (d.stream) (void)hipStreamSynchronize(d.stream);
    if (d.aux) (void)hipStreamSynchronize(d.aux);
    free_scene(d);
    for (int t = 0; t < RT_MAX_TEXTURES; t++)
        if (d.tex_mem[t]) (void)hipFree(d.tex_mem[t]);
    if (d.outcome_snap) (void)hipHostFree(d.outcome_snap);
    free_frame(d);
    for (int k = 0; k < 5; k++)
        if (d.ev[k]) (void)hipEventDestroy(d.ev[k]);
    if (d.ev_fork) (void)hipEventDestroy(d.ev_fork);
    for (hipEvent_t e : d.ev_level)
        if (e) (void)hipEventDestroy(e);
    if (d.aux) (void)hipStreamDestroy(d.aux);
    if (d.stream) (void)hipStreamDestroy(d.stream);
    if (d.status) (void)hipHostFree(d.status);
    if (d.bands) (void)hipFree(d.bands);
    if (d.probe) (void)hipFree(d.probe);
}

// Bytes of the fixed scratch array a frame of P pixels and max_depth + 1 levels needs: the control
// block, then the three record arrays, each starting on a multiple of 256 bytes.
size_t fixed_frame_bytes(size_t slots)
{
    const size_t rec = (slots * sizeof(float4) + 255) / 256 * 256;
    return FRAME_CONTROL_BYTES + 3 * rec;
}

// d.fixed := a frame of P pixels laid out in this device's copy of the fixed scratch array.
// Allocates nothing; a frame that does not fit is refused.
int place_fixed_frame(rt_ctx *ctx, DeviceState &d, int P, int max_depth)
{
    const int levels = max_depth + 1;
    const size_t stride = ((size_t)P + RT_LEVEL_ALIGN - 1) / RT_LEVEL_ALIGN * RT_LEVEL_ALIGN;
    const size_t slots = stride * (size_t)levels;
    const size_t need = fixed_frame_bytes(slots);
    if (need > (size_t)RT_NOALLOC_SCRATCH_BYTES)
        return fail(ctx, RT_ERR_LIMIT, "%d pixels x %d levels need %zu bytes of ray queues; the fixed scratch "
                    "of rt_render_rows_device_noalloc holds %zu (use rt_render_rows_device, or fewer rows per call)",
                    P, levels, need, (size_t)RT_NOALLOC_SCRATCH_BYTES);
    if (!d.scratch_base) return fail(ctx, RT_ERR_HIP, "device %d: the fixed scratch array has no address", d.device);
    unsigned char *p = d.scratch_base;
    const size_t rec = (slots * sizeof(float4) + 255) / 256 * 256;
    DevFrame &f = d.fixed;
    f.counters = reinterpret_cast<DevCounters *>(p);
    f.sched = reinterpret_cast<DevSched *>(p + RT_COUNTER_SLOTS * sizeof(DevCounters));
    static_assert((RT_COUNTER_SLOTS * sizeof(DevCounters) + sizeof(DevSched)) % 8 == 0,
                  "a level's two record counts are updated as one aligned 64-bit word (push_records)");
    f.cnt = reinterpret_cast<int *>(p + RT_COUNTER_SLOTS * sizeof(DevCounters) + sizeof(DevSched));
    p += FRAME_CONTROL_BYTES;
    f.recA = reinterpret_cast<float4 *>(p);
    f.recB = reinterpret_cast<float4 *>(p + rec);
    f.refl = reinterpret_cast<float4 *>(p + 2 * rec);
    f.P = (int)stride;
    f.slots = (int)slots;
    f.max_depth = max_depth;
    return RT_OK;
}

int ensure_frame(rt_ctx *ctx, DeviceState &d, int P, int max_depth)
{
    const int levels = max_depth + 1;
    // Level stride: the pixels rounded up so that no cache line of any per-slot array (16-byte
    // records) holds slots of two levels (see k_trace_levels).
    const size_t stride = ((size_t)P + RT_LEVEL_ALIGN - 1) / RT_LEVEL_ALIGN * RT_LEVEL_ALIGN;
    // Scenes whose deeper levels may be put into bin order (many chunks or many lights) get a spare level behind
    // the last one, where a level's records are staged until the binning pass moves them to their slots.
    d.spare_level = max_depth > 0 && (d.scene.ns_padded > RT_CHUNK || d.scene.nl >= 8);
    const size_t slots = stride * (size_t)(levels + (d.spare_level ? 1 : 0));
    if (slots > (size_t)INT32_MAX)
        return fail(ctx, RT_ERR_LIMIT, "%d pixels x %d levels exceeds 2^31-1 hit-record slots", P, levels);
    if (slots > d.cap_slots) {
        if (d.frame.recA) (void)hipFree(d.frame.recA);
        if (d.frame.recB) (void)hipFree(d.frame.recB);
        if (d.frame.refl) (void)hipFree(d.frame.refl);
        d.frame.recA = d.frame.recB = d.frame.refl = nullptr;
        d.cap_slots = 0;
        HIP_TRY(ctx, hipMalloc((void **)&d.frame.recA, slots * sizeof(float4)));
        HIP_TRY(ctx, hipMalloc((void **)&d.frame.recB, slots * sizeof(float4)));
        HIP_TRY(ctx, hipMalloc((void **)&d.frame.refl, slots * sizeof(float4)));
        d.cap_slots = slots;
    }
    // What every render call zeroes, in one allocation so that it is one fill per frame:
    // the counter slots, the level schedule, the record counts of every possible level.
    if (!d.frame.counters) {
        char *ctl = nullptr;
        HIP_TRY(ctx, hipMalloc((void **)&ctl, FRAME_CONTROL_BYTES));
        d.frame.counters = reinterpret_cast<DevCounters *>(ctl);
        d.frame.sched = reinterpret_cast<DevSched *>(ctl + RT_COUNTER_SLOTS * sizeof(DevCounters));
        d.frame.cnt = reinterpret_cast<int *>(ctl + RT_COUNTER_SLOTS * sizeof(DevCounters) + sizeof(DevSched));
    }
    d.frame.P = (int)stride;
    d.frame.slots = (int)(stride * (size_t)levels);
    d.frame.max_depth = max_depth;
    return RT_OK;
}

int check_render_args(rt_ctx *ctx, const rt_camera *cam, int w, int h, int max_depth)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (ctx->wedged) return fail(ctx, RT_ERR_STALLED, "a device of this context did not come back from an earlier frame; destroy the context");
    if (!cam) return fail(ctx, RT_ERR_INVALID_ARG, "camera is NULL");
    if (w <= 0 || h <= 0) return fail(ctx, RT_ERR_INVALID_ARG, "bad frame size %d x %d", w, h);
    if (max_depth < 0 || max_depth > RT_MAX_DEPTH)
        return fail(ctx, RT_ERR_INVALID_ARG, "max_depth %d outside [0, %d]", max_depth, RT_MAX_DEPTH);
    if (!ctx->have_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_render before rt_set_scene");
    const uint32_t missing = ctx->textures_used & ~ctx->textures_loaded;
    if (missing) {
        int slot = 0;
        while (!((missing >> slot) & 1u)) slot++;
        return fail(ctx, RT_ERR_BAD_TEXTURE, "a primitive uses texture slot %d, which is not loaded", slot);
    }
    return RT_OK;
}

// Waiting for the device with a deadline: a frame that does not come back within RT_HIP_FRAME_TIMEOUT_MS (default
// 30 s; a frame takes milliseconds) is reported as RT_ERR_STALLED, with what the device had reached, instead of
// holding the caller for ever.  (The wait polls: spinning at first, then in steps of 50 us.)
hipError_t wait_until(rt_ctx *ctx, const std::function<hipError_t()> &query)
{
    const auto t0 = std::chrono::steady_clock::now();
    const long long limit_ms = ctx->tune.frame_timeout_ms > 0 ? ctx->tune.frame_timeout_ms : 30000;
    for (unsigned spins = 0;; spins++) {
        const hipError_t e = query();
        if (e != hipErrorNotReady) return e;
        if (spins < 4096) continue;
        const auto waited = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        if (waited >= limit_ms) {
            (void)hipGetLastError();
            return hipErrorNotReady;
        }
        if (waited >= 2) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

// What a frame that did not come back had reached on device state d (for the error message).
void describe_progress(DeviceState &d, char *out, size_t n)
{
    int reached = -1;
    for (int k = 0; k <= d.last_depth && k <= RT_MAX_DEPTH; k++)
        if (d.ev_level[k] && hipEventQuery(d.ev_level[k]) == hipSuccess) reached = k;
    const bool aux_done = !d.aux || hipStreamQuery(d.aux) == hipSuccess;
    (void)hipGetLastError();
    snprintf(out, n, "device %d: %d pixels, max_depth %d; the chain's launches are done up to level %d, the second stream is %s",
             d.device, d.last_P, d.last_depth, reached, aux_done ? "idle" : "still busy");
}

int wait_stream(rt_ctx *ctx, DeviceState &d, hipStream_t stream)
{
    const hipError_t e = wait_until(ctx, [&] { return hipStreamQuery(stream); });
    if (e == hipSuccess) return RT_OK;
    if (e != hipErrorNotReady) return fail(ctx, RT_ERR_HIP, "waiting for device %d failed: %s", d.device, hipGetErrorString(e));
    char what[256];
    describe_progress(d, what, sizeof what);
    ctx->wedged = true;
    return fail(ctx, RT_ERR_STALLED, "the frame did not come back within the time limit (RT_HIP_FRAME_TIMEOUT_MS); %s; the context "
                "must be destroyed", what);
}

// After the device has finished the frames enqueued so far (the caller has synchronised the
// stream): did every workgroup of k_trace_levels see the level it waited for?  The device
// reports a stall in d.status (host memory); the report is consumed here.
int check_schedule(rt_ctx *ctx, DeviceState &d)
{
    static_assert(RT_DEV_MAX_LEVELS >= RT_MAX_DEPTH + 2, "DevSched holds one entry per level");
    static const bool trace = getenv("RT_HIP_SCHED_TRACE") != nullptr;
    if (trace && d.last().sched && d.snap_parts == 0) {
        DevSched sc;
        std::vector<int> cnt(2 * (size_t)(d.last_depth + 2));
        HIP_TRY(ctx, hipMemcpy(&sc, d.last().sched, sizeof sc, hipMemcpyDeviceToHost));
        (void)hipMemcpy(cnt.data(), d.last().cnt, cnt.size() * sizeof(int), hipMemcpyDeviceToHost);
        fprintf(stderr, "[rt_hip] trace launch: %.1f us;", (double)(sc.t_last - sc.t_first) * 0.01);
        for (int k = 1; k <= sc.open_level && k < RT_DEV_MAX_LEVELS; k++)
            fprintf(stderr, " L%d opens at %.1f (A %d B %d)", k, (double)(sc.t_open[k] - sc.t_first) * 0.01,
                    cnt[2 * k], cnt[2 * k + 1]);
        fprintf(stderr, "; level 0: A %d B %d\n", cnt[0], cnt[1]);
    }
    if (!d.status) return RT_OK;
    volatile int *st = d.status;
    if (st[0] != 0) {
        const int level = st[1];
        st[0] = 0;
        st[1] = 0;
        // What the schedule of the LAST frame looked like when it ended (the stalled frame's own, if
        // the caller checks after every frame).
        int claimed = -1, done = -1, records = -1;
        if (d.last().sched && level >= 1 && level < RT_DEV_MAX_LEVELS) {
            DevSched sc;
            int cnt[2 * RT_DEV_MAX_LEVELS];
            if (hipMemcpy(&sc, d.last().sched, sizeof sc, hipMemcpyDeviceToHost) == hipSuccess &&
                hipMemcpy(cnt, d.last().cnt, sizeof cnt, hipMemcpyDeviceToHost) == hipSuccess) {
                claimed = sc.next_a[level - 1];
                done = sc.done_a[level - 1];
                records = cnt[2 * (level - 1)];
            } else {
                (void)hipGetLastError();
            }
        }
        return fail(ctx, RT_ERR_STALLED, "device %d: the level schedule stalled (workgroups waited more than the stall "
                    "limit for level %d to open; level %d: %d reflecting records, %d tile claims, %d tiles done); a frame "
                    "rendered since the last check is incomplete", d.device, level, level - 1, records, claimed, done);
    }
    return RT_OK;
}

// Enqueue the whole frame pipeline for the rows of rs on device state d.
// call.part < 0: the call is a frame of its own.  call.part >= 0: that part of a frame rendered in parts:
// its outcome is copied to d.outcome_snap behind its kernels.
// call.fixed_scratch: the frame's queues live in the library's fixed scratch array and nothing is allocated.
// Do launches on `a` run beside launches on `x`, or does the runtime keep both streams on one hardware queue (it
// maps a process's streams onto GPU_MAX_HW_QUEUES queues, 4 by default, least used first; launches of one queue
// are serial)?  A waiter on x, a setter on a behind it (rt_kernels.hip, k_probe_wait).  Synchronises both streams.
int streams_overlap(rt_ctx *ctx, DeviceState &d, hipStream_t x, hipStream_t a, bool *overlap)
{
    *overlap = true;
    int seen = 0;
    for (int attempt = 0; attempt < 2 && !seen; attempt++) {   // (once more before a "no": a setter may just have been late)
        HIP_TRY(ctx, hipMemsetAsync(d.probe, 0, 8, x));
        HIP_TRY(ctx, hipEventRecord(d.ev_fork, x));
        rtk_probe_wait(d.probe, 30000ull << attempt, x);   // 300 us, then 600
        HIP_TRY(ctx, hipStreamWaitEvent(a, d.ev_fork, 0));
        rtk_probe_set(d.probe, a);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(x));
        HIP_TRY(ctx, hipStreamSynchronize(a));
        int words[2] = {0, 0};
        HIP_TRY(ctx, hipMemcpy(words, d.probe, sizeof words, hipMemcpyDeviceToHost));
        seen = words[1];
    }
    *overlap = seen != 0;
    return RT_OK;
}

// The frame's two streams must not share a hardware queue: the chain of reflection levels and the plain records'
// launches would run one after the other (the 64k-sphere scene 3.71 instead of 2.61 ms per frame in a process
// whose fourth stream carried the frames).  Checked once per caller stream; a second stream that does share the
// caller's queue is replaced by a new one (the runtime gives a new stream the least used queue) until one does
// not, a few times at most.  Costs about a millisecond, the first time a stream is seen.
int calibrate_second_stream(rt_ctx *ctx, DeviceState &d, hipStream_t stream)
{
    // Only `stream`, which the caller vouches for in this call, is ever launched on: streams seen in earlier calls
    // may be gone.  A host that alternates between streams gets each checked when it comes; should they keep
    // displacing one another's second stream (more caller streams than the runtime has queues to spare), the
    // exchanges stop after a few and the streams are taken as they are.
    std::vector<hipStream_t> rejected;
    int rc = RT_OK;
    for (int tries = 0; tries < 6 && d.replaced < 12; tries++) {
        bool overlap = true;
        if ((rc = streams_overlap(ctx, d, stream, d.aux, &overlap)) != RT_OK || overlap) break;
        hipStream_t fresh = nullptr;
        if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        rejected.push_back(d.aux);   // (kept until a good one is found, so that the next one gets another queue)
        d.aux = fresh;
        d.replaced++;
        d.checked_against.clear();   // (what passed with the old second stream must be looked at again when it comes)
    }
    for (hipStream_t s : rejected) (void)hipStreamDestroy(s);
    if (d.checked_against.size() >= 8) d.checked_against.erase(d.checked_against.begin());
    d.checked_against.push_back(stream);
    return rc;
}

struct RowsCall {
    int part = -1;
    bool keep_counters = false;   // the work counters run on from the parts this device state rendered before
    bool fixed_scratch = false;
};

int enqueue_rows(rt_ctx *ctx, DeviceState &d, const rt_camera *cam, int w, int h, const RowSet &rs,
                 int max_depth, int32_t *dev_pixels, hipStream_t stream, const RowsCall &call = RowsCall())
{
    const Tuning &tune = ctx->tune;
    const int part = call.part;
    const bool fixed_scratch = call.fixed_scratch;

    const long long P64 = (long long)rs.rows * (long long)w;
    if (P64 > INT32_MAX) return fail(ctx, RT_ERR_LIMIT, "row block of %lld pixels exceeds 2^31-1", P64);
    const int P = (int)P64;
    if (P == 0) {   // (a frame rendered in parts has no empty part)
        d.last_P = 0;
        d.snap_parts = 0;
        return RT_OK;
    }
    HIP_TRY(ctx, hipSetDevice(d.device));
    if (part >= 0 && !d.outcome_snap)
        HIP_TRY(ctx, hipHostMalloc((void **)&d.outcome_snap, RT_MAX_FRAME_PARTS * FRAME_OUTCOME_BYTES, hipHostMallocDefault));
    int rc = fixed_scratch ? place_fixed_frame(ctx, d, P, max_depth) : ensure_frame(ctx, d, P, max_depth);
    if (rc != RT_OK) return rc;   // (nothing enqueued: what rt_get_stats describes stays the previous frame)
    d.last_P = call.keep_counters ? d.last_P + P : P;
    d.last_depth = max_depth;
    d.snap_parts = part < 0 ? 0 : part + 1;
    d.last_fixed = fixed_scratch;
    DevFrame &fr = d.last();
    const DevCamera dc = make_camera(cam, w, h, rs);
    const int tiles0 = (P + 255) / 256;
    const bool counting = ctx->profiling == 1;   // the counting kernels walk everything
    // Workgroups of the level kernel that fit the device at once, for the instance a large frame
    // launches (the policy decisions below are made against it) ...
    auto resident = [&](bool reserve) {
        int &r = d.resident_blocks[reserve][counting];
        if (r <= 0) r = rtk_trace_levels_resident_blocks(d.scene, d.num_cus, reserve, counting);
        return r;
    };
    // A context that shares its device with others rendering at the same time takes its part of
    // the device: the level kernel's workgroups stay resident for the whole launch, so two
    // full-size launches would run one after the other instead of side by side.
    const int share = std::max(1, ctx->share);
    const int resident_plain = std::max(d.num_cus / 2, resident(false) / share);
    fr.any_hit = counting ? 0 : 1;
    // On still smaller frames (measured: below ~6 tiles per resident workgroup; 4K / 8 GPUs has 4)
    // it pays to reserve the oldest workgroup of every compute unit for the chain: a SIMD issues
    // for its oldest waves first, so a chain tile in a young workgroup crawls while its elders
    // have work.  1/8 of the 4K frame: 0.397 -> 0.362 ms; 1/4 (8 tiles per workgroup): 0.547 ->
    // 0.571, hence the bound.  RT_HIP_CHAIN_POLICY=0/1 (tuning aid) overrides.
    fr.chain_policy = tune.chain_policy >= 0 ? tune.chain_policy
                                             : tiles0 < RT_CHAIN_RESERVED_TILES_PER_WORKGROUP * resident_plain;
    fr.num_cus = d.num_cus;
    // ... and for the instance this frame launches (workgroups beyond what is resident at once
    // would only start when the first ones leave, find nothing to claim and exit).
    const int resident_blocks = std::max(d.num_cus / 2, resident(fr.chain_policy != 0) / share);
    fr.status = d.status;
    const long long stall_ms = tune.stall_ms > 0 ? tune.stall_ms : (long long)RT_STALL_MS_DEFAULT;
    fr.stall_ticks = (unsigned)std::min<long long>(stall_ms * 100000ll, 4000000000ll);   // 100 MHz wall clock
    fr.debug_withhold = tune.withhold_level;
    const bool grid = d.scene.gr_n[0] > 0;   // reflection rays walk the spheres' grid (k_reflect_grid): no chunks, no LDS
    const bool level_kernels = tune.level_kernels;
    // (the fused form pays only where several small frames are in flight on the device at once - there its
    // single launch per frame beats the levels' chains of short launches, 1/8 of the 4K frame with four in
    // flight: 0.106 against 0.127 ms - and the scene is one chunk; a frame alone on the device: 0.33 against
    // 0.20 ms, 256x256 of the 64k-sphere scene: 9.1 against 1.5 ms)
    const bool fused = tune.pipeline ? tune.pipeline == 2
                                     : !level_kernels && share >= 4 && (d.scene.ns_padded <= RT_CHUNK || grid) &&
                                          tiles0 < RT_FUSED_BELOW_TILES_PER_WORKGROUP * resident(false);
    // Scenes of several chunks, one launch per level: the reflection rays of a level are traced as
    // (tile, chunk) jobs (k_reflect_chunks).  RT_HIP_SPLIT_CHUNKS=0 (tuning aid) keeps a tile's chunks
    // to its own workgroup.
    fr.split_chunks = !fused && !grid && tune.split_chunks != 0 && d.scene.ns_padded > RT_CHUNK ? 1 : 0;
    fr.split_below = tune.split_below > 0 ? tune.split_below : std::max(1, d.num_cus * 2 / share);
    // Scenes with a grid, one launch per level: levels that cannot fill the device walk every ray in segments
    // (k_reflect_grid).  RT_HIP_GRID_SPLIT_BELOW=n (tuning aid; 0 = never).  Alone on the device the 64k-sphere
    // scene's levels 2 - 7 take 103, 54, 38, 36, 34, 32 us instead of 129, 131, 140, 114, 99, 95; the 1920x1080 frame
    // keeps the device busy with the plain records' shading beside the chain and gains nothing (2.63 ms either way
    // with a bound of 32768, 2.66 with 131072), smaller frames and a rank's share of a frame do: 1280x720 2.15 -> 1.88
    // ms, 960x540 1.84 -> 1.63, one eighth of the 1920x1080 frame 1.36 -> 1.08.
    fr.grid_split_below = !grid || fused ? 0 : tune.grid_split_below >= 0 ? tune.grid_split_below : 32768 / share;
    // Scenes of one chunk, one launch per level: a level with fewer reflecting tiles than this is traced 64
    // records per workgroup, the scene spread over the workgroup's waves.  RT_HIP_SPREAD_BELOW=n (tuning
    // aid; 0 = never).
    // (on a large frame the plain records' launches keep the device busy beside the chain, and the four times as
    // many workgroups of the spread form only take from them: 3840x2160 0.552 ms with a bound of 128 tiles
    // against 0.562 with 512, 1920x1080 0.259 against 0.250)
    fr.spread_below = fused || grid ? 0 : tune.spread_below >= 0 ? tune.spread_below
                                          : std::max(1, (tiles0 >= 16384 ? d.num_cus / 2 : d.num_cus * 2) / share);
    // Scenes with several lights, one launch per level: short queues are shaded with the lights spread
    // over the waves of a workgroup (k_shade_lights).  RT_HIP_LIGHT_SPLIT_BELOW=n (tuning aid; 0 = never).
    const int light_below_env = tune.light_split_below;
    const int nl = d.scene.nl;
    // (16 lights over 16 waves: 2.44 ms for the 64k-sphere frame; over 12, 10, 8, 7, 6, 5, 4: 2.64, 2.60, 2.36, 2.28, 2.29,
    // 2.29, 2.37 - a workgroup of six waves leaves room for more of them on a compute unit)
    fr.light_waves = fused || nl < 3 || light_below_env == 0 ? 0 : nl >= 6 ? 6 : 4;
    if (fr.light_waves && tune.light_waves >= 1 && tune.light_waves <= 16) fr.light_waves = tune.light_waves;   // RT_HIP_LIGHT_WAVES
    // (The more lights, the longer the queues it pays for: a wave of k_shade_lights looks one light up for 64
    // neighbouring records - one table, nearby cells - where a lane of k_plain goes through all the tables.
    // 16 lights: 4.04 ms per frame with queues below 64 K records, 3.88 below 256 K, 3.64 below 512 K or 1 M, 4.3 for all.)
    // That is the 64k-sphere scene, whose tables are large and whose light rays wait for memory: the sixteen times as
    // many waves hide it.  With smaller tables the spread form's repeated work (every wave sets the record up) is what
    // counts - 16 lights at 1920x1080, depth 6, bounds of 512 K against 64 K records: 2048 spheres 0.858 / 0.732 ms,
    // 4096: 1.193 / 0.965, 16384: 1.415 / 1.236, 32768: the same, 49152: 2.23 at 256 K against 2.42 at 64 K - so
    // scenes of fewer than RT_LIGHT_SPLIT_LARGE_SCENE spheres spread the lights only for queues below 64 K records.
    const int light_below_default = d.scene.ns_padded < RT_LIGHT_SPLIT_LARGE_SCENE ? 65536 : std::max(64, 32 * nl) * 1024;
    fr.light_split_below = light_below_env > 0 ? light_below_env : light_below_default / share;
    // Scenes of several chunks or many lights, one launch per level: the records of a level >= 1 are put
    // into the order of their hit points' position bins before the level is used (k_bin_*).
    // RT_HIP_BIN_MIN=n (tuning aid): for levels of at least n records; 0 = never.
    const int bin_min_env = tune.bin_min;
    fr.bin_min = 0;
    fr.stage_base = 0;
    fr.staged = 0;
    if (!fused && !fixed_scratch && max_depth > 0 && bin_min_env != 0 && (d.scene.ns_padded > RT_CHUNK || nl >= 8) &&
        d.scene.bin_scale[0] > 0.f) {
        const size_t stride = (size_t)fr.P;
        if (stride > d.cap_bin && stride < d.bin_refused) {
            if (d.bin_aux) (void)hipFree(d.bin_aux);
            d.bin_aux = nullptr;
            d.cap_bin = 0;
            // An optimisation only: a device that cannot give the memory renders without it.
            if (hipMalloc((void **)&d.bin_aux, stride * sizeof(uint2)) == hipSuccess)
                d.cap_bin = stride;
            else {
                (void)hipGetLastError();
                d.bin_aux = nullptr;
                d.bin_refused = stride;
            }
        }
        if (!d.bin_hist && stride <= d.cap_bin) {
            if (hipMalloc((void **)&d.bin_hist, 4 * RT_BINS * sizeof(int)) == hipSuccess)
                HIP_TRY(ctx, hipMemsetAsync(d.bin_hist, 0, 4 * RT_BINS * sizeof(int), stream));   // (the kernels zero it after use)
            else {
                (void)hipGetLastError();
                d.bin_hist = nullptr;
            }
        }
        if (stride <= d.cap_bin && d.bin_hist && d.spare_level) {
            fr.bin_min = bin_min_env > 0 ? bin_min_env : 16384;
            fr.stage_base = (max_depth + 1) * fr.P;   // (the spare level: ensure_frame)
            fr.bin_aux = d.bin_aux;
            fr.bin_hist = d.bin_hist;
            fr.bin_base = d.bin_hist + 2 * RT_BINS;
        }
    }
    // Which levels go through the binning launches (bit k: level k).  A level below bin_min records is not put into
    // order, but its three or four launches still stand on the chain (~10 us each: a rank's eighth of the 64k-sphere
    // frame spent a quarter of its time on them).  The frame's last launch leaves the levels' record counts in
    // page-locked memory (DevFrame::hint); a level that held fewer than half of bin_min records in the last finished
    // frame of this scene, size and depth is pushed straight to its slots this time and has no binning launches.  Only
    // speed depends on the hint: a level left out is a level that is not put into order, whatever its length turns out
    // to be, and the counts keep coming, so it is taken in again when it grows.  RT_HIP_BIN_HINT=0 (tuning aid): every level.
    // The same counts tell which levels' reflecting records will have their light terms ahead (fewer than lit_below):
    // the passes at the end of the frame then issue one launch for such a level instead of two (lit_small, below).
    unsigned bin_levels = fr.bin_min > 0 ? ~0u : 0u, lit_small = 0u;
    fr.hint = nullptr;
    fr.hint_tag = 0;
    if ((fr.bin_min > 0 || fr.light_waves > 0) && d.status && tune.bin_hint != 0 && !fixed_scratch) {
        fr.hint = d.status + 16;
        fr.hint_tag = (int)(((ctx->scene_serial * 2654435761u) ^ ((unsigned)P * 40503u) ^ ((unsigned)max_depth * 97u)) | 1u);
        const volatile int *hint = fr.hint;
        if (hint[0] == fr.hint_tag)
            for (int k = 0; k <= max_depth; k++) {
                if (k >= 1 && hint[1 + 2 * k] + hint[2 + 2 * k] < fr.bin_min / 2) bin_levels &= ~(1u << k);
                if (hint[1 + 2 * k] < 32768) lit_small |= 1u << k;
            }
    }
    if (fr.bin_min > 0) fr.staged = grid ? (int)(bin_levels & 0x7fffffffu) : 0;
    // Scenes with few lights, one launch per level: the reflecting records' light terms are computed
    // beside the plain records' shading (k_lit), so that the passes at the end of the frame, which
    // nothing can overlap, only add them up.  Costs 16 bytes per slot of the levels below max_depth and
    // light; not beyond RT_LIT_MAX_BYTES.  RT_HIP_EARLY_LIGHTS=0 (tuning aid) turns it off.
    fr.lit = nullptr;
    {
        // (not for a context that shares its device with others rendering at the same time: there the
        // tail of one frame is filled by the other frames' kernels anyway, and the extra traffic costs 5 %)
        // With more than four lights only for the levels of few reflecting records: their passes at the end of
        // the frame are pure latency (16 lights: ~75 us per level however few its records), while a long queue
        // costs its work wherever it runs (the 16 lights of the 64k-sphere scene with every level ahead: the
        // frame's tail shrinks by 1.2 ms and the stage grows by as much).  RT_HIP_EARLY_LIGHTS_MAX: the most
        // lights it is used for; RT_HIP_EARLY_LIGHTS_BELOW: the bound in records (0: every level); tuning aids.
        const int lit_lights = tune.early_lights_max;
        fr.lit_below = nl <= 4 ? 0 : tune.early_lights_below >= 0 ? tune.early_lights_below : 65536;
        // (a level with its terms ahead holds fewer than lit_below reflecting records)
        fr.lit_stride = fr.lit_below > 0 ? std::min(fr.P, fr.lit_below) : fr.P;
        const size_t want = (size_t)fr.lit_stride * (size_t)max_depth * (size_t)std::max(nl, 1);
        if (!fused && !fixed_scratch && share == 1 && max_depth > 0 && nl >= 1 && nl <= lit_lights &&
            tune.early_lights != 0 &&
            want * sizeof(float4) <= RT_LIT_MAX_BYTES) {
            if (want > d.cap_lit && want < d.lit_refused) {
                if (d.lit) (void)hipFree(d.lit);
                d.lit = nullptr;
                d.cap_lit = 0;
                // An optimisation only: a device that cannot give the memory renders without it.
                if (hipMalloc((void **)&d.lit, want * sizeof(float4)) == hipSuccess) d.cap_lit = want;
                else {
                    (void)hipGetLastError();
                    d.lit = nullptr;
                    d.lit_refused = want;
                }
            }
            if (want <= d.cap_lit) fr.lit = d.lit;
        }
    }
    // The sums of the reflecting records path by path in one launch (k_shade_paths) when every level's light terms
    // are there; the levels above path_lo keep their own launches.  Not for frames beyond RT_PATH_SHADE_MAX_PIXELS:
    // a path's records lie where its levels' queues put them, and the one launch per level reads each level in
    // order (measured, 1k-sphere scene, depth 4: 1920x1080 -5 %, 3840x2160 -0.5 %, 7680x4320 +1.9 %).
    const int path_lo = tune.path_shade > 0 ? std::min(tune.path_shade - 1, std::max(0, max_depth - 2)) : 0;
    fr.path_shade = fr.lit && fr.lit_below <= 0 && fr.bin_min <= 0 && !fused && max_depth >= 2 &&
                    (tune.path_shade > 0 || (tune.path_shade < 0 && P <= RT_PATH_SHADE_MAX_PIXELS));
    d.last_stream = stream;
    const DevFrame &f = fr;
    if (call.keep_counters) HIP_TRY(ctx, hipMemsetAsync(f.sched, 0, FRAME_CONTROL_BYTES - RT_COUNTER_SLOTS * sizeof(DevCounters), stream));
    else HIP_TRY(ctx, hipMemsetAsync(f.counters, 0, FRAME_CONTROL_BYTES, stream));
    d.timed = ctx->profiling != 0;
    if (d.timed) HIP_TRY(ctx, hipEventRecord(d.ev[0], stream));
    // Workgroups per compute unit of the per-level launches (RT_HIP_REFLECT_BLOCKS / RT_HIP_PLAIN_BLOCKS /
    // RT_HIP_LIGHT_BLOCKS, tuning aids, override the defaults).
    // (the grid walk refills its lanes from a tile of 512 records: fewer, longer-lived workgroups; 64k spheres, 16 lights:
    // 2.37 ms with 8 per compute unit, 2.34 with 4, 2.45 with 2)
    const int reflect_blocks = tune.reflect_blocks > 0 ? tune.reflect_blocks : grid ? 4 : 8;
    const int blocks = std::max(1, std::min({(P + 255) / 256, d.num_cus * reflect_blocks / std::max(1, ctx->share),
                                             RT_COUNTER_SLOTS}));
    // k_primary's workgroups claim their 32x32-pixel tiles dynamically: as many workgroups as are
    // resident at once (4 per compute unit) when each gets several tiles, twice that on small frames
    // (measured: 3840x2160 0.161 against 0.170 ms, 1920x1080 0.050 against 0.048 ms).
    const int tiles32 = ((w + 31) / 32) * ((rs.rows + 31) / 32);
    rtk_primary(d.scene, dc, f, dev_pixels,
                std::min((tiles32 >= 16 * d.num_cus ? d.num_cus * 4 : d.num_cus * 8) / share, RT_COUNTER_SLOTS), stream);
    if (d.timed) HIP_TRY(ctx, hipEventRecord(d.ev[1], stream));
    // Two forms of the levels' work.  "levels": one launch per level and queue; the reflecting
    // records' launches (the chain: level k + 1 exists when level k's are done) stay on the caller's
    // stream, the plain records' launches (shaded on the spot, no successors, most of the work, in
    // a kernel of their own that needs few registers and runs at a higher occupancy) go to a second
    // stream and fill the device beside them.  "fused": every level in ONE launch of
    // resident workgroups with a device-side schedule (k_trace_levels): no launch gaps between the
    // levels of a small frame, but plain records then run at the walk kernel's occupancy.
    // RT_HIP_PIPELINE=levels|fused (tuning aid) overrides; RT_HIP_LEVEL_KERNELS=1 (profiling aid)
    // is "levels" on ONE stream, so that a kernel trace shows what each launch costs on its own.
    if (!fused) {
        // (counting kernels add to per-workgroup counter slots without atomics: one stream then)
        const bool two_streams = !level_kernels && !counting && max_depth > 0;
        const int plain_blocks = std::max(1, std::min({tiles0, d.num_cus * tune.plain_blocks / share, RT_COUNTER_SLOTS}));
        const int light_blocks = std::max(1, std::min(d.num_cus * tune.light_blocks / share, RT_COUNTER_SLOTS));
        // The chain stays on the caller's stream (a dependency between two launches of one stream costs
        // ~6 us, across streams ~12, and the chain is the frame's critical path); the plain records'
        // launches go to the second stream, except the deepest level's, which the chain ends with.
        if (two_streams && !fixed_scratch && tune.calibrate != 0 &&
            std::find(d.checked_against.begin(), d.checked_against.end(), stream) == d.checked_against.end() &&
            (rc = calibrate_second_stream(ctx, d, stream)) != RT_OK)
            return rc;
        hipStream_t side = two_streams ? d.aux : stream;
        if (two_streams) {
            HIP_TRY(ctx, hipEventRecord(d.ev_fork, stream));
            HIP_TRY(ctx, hipStreamWaitEvent(d.aux, d.ev_fork, 0));
        }
        for (int k = 0; k <= max_depth; k++) {
            if (k < max_depth) {
                rtk_trace(d.scene, f, k, true, dev_pixels, blocks, stream);
                if (f.split_chunks) rtk_reflect_chunks(d.scene, f, k, std::max(1, d.num_cus * 4 / share), blocks, stream);
                if (f.bin_min > 0 && ((bin_levels >> (k + 1)) & 1u))
                    rtk_bin_level(d.scene, f, k + 1, std::max(1, d.num_cus * 8 / share), stream);
                if (two_streams) HIP_TRY(ctx, hipEventRecord(d.ev_level[k], stream));
            }
            hipStream_t where = k == max_depth ? stream : side;
            if (two_streams && k > 0 && k < max_depth) HIP_TRY(ctx, hipStreamWaitEvent(d.aux, d.ev_level[k - 1], 0));
            // Small levels share launches: every launch of its own would add its start-up to the second stream,
            // which would then end after the chain.  Scenes of one or two lights: the two levels below the deepest
            // go in one launch (and their light terms), issued when the second one's records exist.  With the lights
            // spread over waves (three lights and more) a short level's launch lasts as long as one record's chain of
            // lights however few its records (16 lights: ~65 us): there the levels from the fourth on go in groups
            // of five, each group issued when its last level exists (64k spheres, 16 lights, 8 bounces: 2.64 -> 2.45
            // ms; groups of 2, 3, 4: 2.55, 2.51, 2.51; the last 7 levels as one: 2.95; 1000 spheres under 3, 4, 8 lights
            // at 1920x1080, depth 8: 2, 4.5, 4 % faster).
            // RT_HIP_TAIL_LEVELS=n (tuning aid): the last n levels as one group, the others on their own;
            // RT_HIP_LEVEL_GROUP=n: groups of n.
            int group_first = k, group_last = k;
            if (two_streams && max_depth >= 3 && k < max_depth) {
                const bool groups = tune.tail_levels <= 0 && f.light_waves;
                const int first = tune.tail_levels > 0 ? std::max(1, max_depth - tune.tail_levels)
                                  : groups             ? std::min(3, max_depth - 2)
                                                       : max_depth - 2;
                const int per_group = !groups ? max_depth : tune.level_group > 0 ? tune.level_group : 5;
                if (k >= first) {
                    group_first = first + (k - first) / per_group * per_group;
                    group_last = std::min(max_depth - 1, group_first + per_group - 1);
                }
            }
            if (k < group_last) continue;
            if (group_first < group_last) {
                rtk_plain_levels(d.scene, f, group_first, k, dev_pixels, plain_blocks, where);
                if (f.light_waves) rtk_shade_lights(d.scene, f, group_first, k, f.lit ? 3 : 0, dev_pixels, light_blocks, where);
                continue;
            }
            rtk_trace(d.scene, f, k, false, dev_pixels, plain_blocks, where);
            if (f.light_waves) rtk_shade_lights(d.scene, f, k, k, f.lit && k < max_depth ? 3 : 0, dev_pixels, light_blocks, where);
            if (f.lit && k < max_depth) rtk_lit_level(d.scene, f, k, plain_blocks, where);
        }
        if (two_streams) {   // the reflecting records' shading below needs every plain level (and the light terms)
            HIP_TRY(ctx, hipEventRecord(d.ev_fork, d.aux));
            HIP_TRY(ctx, hipStreamWaitEvent(stream, d.ev_fork, 0));
        }
    } else {
        // Every level in one launch of resident workgroups (k_trace_levels).
        rtk_trace_levels(d.scene, f, dev_pixels, std::max(1, std::min({2 * tiles0, resident_blocks, RT_COUNTER_SLOTS})), stream);
    }
    if (d.timed) HIP_TRY(ctx, hipEventRecord(d.ev[2], stream));
    // The shading kernels are bandwidth-bound and keep no counters: 32 workgroups per compute unit
    // (measured at 3840x2160: 0.128 ms against 0.145 with 8 and 0.143 with 128).
    const int shade_blocks = std::max(1, std::min((P + 255) / 256, d.num_cus * 32));
    // (the reflecting records; the deepest level has none)
    if (f.path_shade) rtk_shade_paths(d.scene, f, path_lo, dev_pixels, shade_blocks, stream);   // (keeps no counters)
    for (int k = (f.path_shade ? path_lo : max_depth) - 1; k >= 0; k--) {
        // (a level whose light terms are expected to be there - a hint - gets the one launch, which then takes the queue
        // whatever it turns out to be; lit_below is 65536 when it is positive, and the hint asks for half of that)
        const bool alone = f.lit && f.lit_below >= 65536 && ((lit_small >> k) & 1u);
        rtk_shade_level(d.scene, f, k, alone, dev_pixels, std::min(shade_blocks, RT_COUNTER_SLOTS), stream);
        if (f.light_waves && (!f.lit || f.lit_below > 0) && !alone)
            rtk_shade_lights(d.scene, f, k, k, 1, dev_pixels, std::max(1, std::min(d.num_cus * 8 / share, RT_COUNTER_SLOTS)), stream);
    }
    if (d.timed) HIP_TRY(ctx, hipEventRecord(d.ev[3], stream));
    if (part >= 0)
        HIP_TRY(ctx, hipMemcpyAsync(d.outcome_snap + (size_t)part * FRAME_OUTCOME_BYTES, f.sched, FRAME_OUTCOME_BYTES,
                                    hipMemcpyDeviceToHost, stream));
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_api_version(void) { return RT_API_VERSION; }

int rt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return RT_ERR_NO_DEVICE;
    return n;
}

int rt_create(rt_ctx **out, const int *device_ids, int n_devices)
{
    if (!out) return fail(nullptr, RT_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    return guarded(nullptr, [&]() -> int {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(nullptr, RT_ERR_NO_DEVICE, "no HIP device is visible");
    std::vector<int> ids;
    if (!device_ids || n_devices <= 0) ids.push_back(0);
    else ids.assign(device_ids, device_ids + n_devices);
    for (int id : ids)
        if (id < 0 || id >= count)
            return fail(nullptr, RT_ERR_NO_DEVICE, "device id %d outside [0, %d)", id, count);
    rt_ctx *ctx = new (std::nothrow) rt_ctx();
    if (!ctx) return fail(nullptr, RT_ERR_OUT_OF_MEMORY, "out of host memory");
    ctx->tune = read_tuning();
    ctx->dev.resize(ids.size());
    for (size_t i = 0; i < ids.size(); i++) {
        DeviceState &d = ctx->dev[i];
        d.device = ids[i];
        hipDeviceProp_t prop;
        hipError_t e = hipSetDevice(d.device);
        if (e == hipSuccess) e = hipGetDeviceProperties(&prop, d.device);
        if (e == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            int rc = fail(nullptr, RT_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only",
                          d.device, prop.gcnArchName);
            rt_destroy(ctx);
            return rc;
        }
        if (e == hipSuccess) e = create_device_objects(d, ctx->tune.aux_priority > 0);
        if (e != hipSuccess) {
            int rc = fail(nullptr, RT_ERR_HIP, "device %d setup failed: %s", d.device, hipGetErrorString(e));
            rt_destroy(ctx);
            return rc;
        }
        d.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    // Peer access towards the gathering device (the first one).  A device without it hands its
    // rows over through page-locked host memory instead (rt_render).  RT_HIP_NO_PEER=1 (test knob)
    // treats every other device as such.
    const bool no_peer = getenv("RT_HIP_NO_PEER") != nullptr;
    for (size_t i = 1; i < ctx->dev.size(); i++) {
        DeviceState &d = ctx->dev[i];
        if (no_peer) {
            d.peer_to_root = false;
            continue;
        }
        if (d.device == ctx->dev[0].device) continue;
        int can = 0;
        d.peer_to_root = false;
        if (hipDeviceCanAccessPeer(&can, d.device, ctx->dev[0].device) == hipSuccess && can) {
            (void)hipSetDevice(d.device);
            const hipError_t e = hipDeviceEnablePeerAccess(ctx->dev[0].device, 0);
            d.peer_to_root = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
            if (e != hipSuccess) (void)hipGetLastError();
        }
    }
    *out = ctx;
    return RT_OK;
    });
}

void rt_destroy(rt_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->wedged) {
        // A device that stopped answering would hold every synchronising call for ever (hipFree included): the
        // context's device memory and streams are abandoned, only the host side goes.
        delete ctx->movers;
        delete ctx;
        return;
    }
    for (DeviceState &d : ctx->dev) destroy_device_state(d);
    if (ctx->staging) (void)hipHostFree(ctx->staging);
    if (!ctx->dev.empty()) (void)hipSetDevice(ctx->dev[0].device);
    for (hipEvent_t e : ctx->copy_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->part_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->trace_ev) (void)hipEventDestroy(ctx->trace_ev);
    delete ctx->movers;
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->pinned) (void)(ctx->pinned_owned ? hipHostFree(ctx->pinned_base) : hipHostUnregister(ctx->pinned_base));
    delete ctx;
}

const char *rt_last_error(const rt_ctx *ctx)
{
    return ctx ? ctx->error : g_create_error;
}

int rt_set_profiling(rt_ctx *ctx, int enable)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (enable < 0 || enable > 2) return fail(ctx, RT_ERR_INVALID_ARG, "profiling mode %d is none of 0, 1, 2", enable);
    ctx->profiling = enable;
    return RT_OK;
}

int rt_reload_tuning(rt_ctx *ctx)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    ctx->tune = read_tuning();
    return RT_OK;
}

int rt_set_device_share(rt_ctx *ctx, int contexts)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (contexts < 1 || contexts > 8) return fail(ctx, RT_ERR_INVALID_ARG, "device share %d outside [1, 8]", contexts);
    ctx->share = contexts;
    return RT_OK;
}

int rt_set_scene(rt_ctx *ctx, const rt_scene *s)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!s) return fail(ctx, RT_ERR_INVALID_ARG, "scene is NULL");
    if (s->num_prims < 0 || s->num_lights < 0)
        return fail(ctx, RT_ERR_INVALID_ARG, "negative primitive or light count");
    if (s->num_prims > 0 &&
        (!s->prim_types || !s->prim_positions || !s->prim_normals || !s->prim_diffuse || !s->prim_glossy ||
         !s->prim_ambient || !s->prim_specular || !s->prim_radius || !s->prim_d || !s->prim_v0 ||
         !s->prim_v1 || !s->prim_v2 || !s->prim_checkerboard))
        return fail(ctx, RT_ERR_INVALID_ARG, "a primitive array is NULL");
    if (s->num_lights > 0 && (!s->light_positions || !s->light_colors))
        return fail(ctx, RT_ERR_INVALID_ARG, "a light array is NULL");
    return guarded(ctx, [&]() -> int {
    ctx->tune = read_tuning();
    PackedScene ps;
    int rc = pack_scene(ctx, s, ps);
    if (rc != RT_OK) return rc;
    ctx->have_scene = false;
    for (DeviceState &d : ctx->dev)
        if ((rc = upload_scene(ctx, d, ps, s->num_prims, s->num_lights)) != RT_OK) return rc;
    ctx->num_prims = s->num_prims;
    ctx->num_lights = s->num_lights;
    ctx->ns = ps.class_prefix[3 * (size_t)s->num_prims];
    ctx->np = (int)ps.pl.size();
    ctx->nt = (int)ps.tri_orig.size();
    ctx->textures_used = ps.textures_used;
    ctx->have_scene = true;
    ctx->scene_serial++;
    return RT_OK;
    });
}

int rt_set_texture(rt_ctx *ctx, int slot, const uint8_t *rgba8, int width, int height)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (slot < 0 || slot >= RT_MAX_TEXTURES)
        return fail(ctx, RT_ERR_INVALID_ARG, "texture slot %d outside [0, %d)", slot, RT_MAX_TEXTURES);
    if (!rgba8 || width <= 0 || height <= 0)
        return fail(ctx, RT_ERR_INVALID_ARG, "bad texture image (%p, %d x %d)", (const void *)rgba8, width, height);
    const size_t bytes = (size_t)width * (size_t)height * 4;
    return guarded(ctx, [&]() -> int {
    for (DeviceState &d : ctx->dev) {
        HIP_TRY(ctx, hipSetDevice(d.device));
        HIP_TRY(ctx, hipStreamSynchronize(d.stream));
        if (d.tex_mem[slot]) (void)hipFree(d.tex_mem[slot]);
        d.tex_mem[slot] = nullptr;
        d.scene.tex[slot] = nullptr;
        ctx->textures_loaded &= ~(1u << slot);
        HIP_TRY(ctx, hipMalloc(&d.tex_mem[slot], bytes));
        HIP_TRY(ctx, hipMemcpy(d.tex_mem[slot], rgba8, bytes, hipMemcpyHostToDevice));
        d.scene.tex[slot] = static_cast<const uchar4 *>(d.tex_mem[slot]);
        d.scene.tex_w[slot] = width;
        d.scene.tex_h[slot] = height;
    }
    ctx->textures_loaded |= 1u << slot;
    return RT_OK;
    });
}

int rt_render_rows_device(rt_ctx *ctx, const rt_camera *cam, int w, int h, int row_begin, int row_end,
                          int max_depth, int32_t *dev_pixels_out, void *hip_stream)
{
    int rc = check_render_args(ctx, cam, w, h, max_depth);
    if (rc != RT_OK) return rc;
    if (row_begin < 0 || row_end > h || row_begin > row_end)
        return fail(ctx, RT_ERR_INVALID_ARG, "rows [%d, %d) outside [0, %d)", row_begin, row_end, h);
    if (!dev_pixels_out && row_end > row_begin) return fail(ctx, RT_ERR_INVALID_ARG, "dev_pixels_out is NULL");
    return guarded(ctx, [&]() -> int {
    DeviceState &d = ctx->dev[0];
    ctx->last_was_rows_device = true;
    ctx->last_w = w;
    ctx->last_h = h;
    ctx->copy_ms = ctx->gather_ms = 0.0;
    for (size_t i = 1; i < ctx->dev.size(); i++) ctx->dev[i].last_P = 0;
    return enqueue_rows(ctx, d, cam, w, h, contiguous_rows(row_begin, row_end), max_depth, dev_pixels_out,
                        hip_stream ? static_cast<hipStream_t>(hip_stream) : d.stream);
    });
}

int rt_render_rows_device_noalloc(rt_ctx *ctx, const rt_camera *cam, int w, int h, int row_begin, int row_end,
                                  int max_depth, int32_t *dev_pixels_out, void *hip_stream)
{
    int rc = check_render_args(ctx, cam, w, h, max_depth);
    if (rc != RT_OK) return rc;
    if (row_begin < 0 || row_end > h || row_begin > row_end)
        return fail(ctx, RT_ERR_INVALID_ARG, "rows [%d, %d) outside [0, %d)", row_begin, row_end, h);
    if (!dev_pixels_out && row_end > row_begin) return fail(ctx, RT_ERR_INVALID_ARG, "dev_pixels_out is NULL");
    return guarded(ctx, [&]() -> int {
    DeviceState &d = ctx->dev[0];
    ctx->last_was_rows_device = true;
    ctx->last_w = w;
    ctx->last_h = h;
    ctx->copy_ms = ctx->gather_ms = 0.0;
    for (size_t i = 1; i < ctx->dev.size(); i++) ctx->dev[i].last_P = 0;
    RowsCall fixed;
    fixed.fixed_scratch = true;
    return enqueue_rows(ctx, d, cam, w, h, contiguous_rows(row_begin, row_end), max_depth, dev_pixels_out,
                        hip_stream ? static_cast<hipStream_t>(hip_stream) : d.stream, fixed);
    });
}

int rt_band_rows(int height, int band_rows, int part, int parts)
{
    if (height < 0 || band_rows <= 0 || parts <= 0 || part < 0 || part >= parts) return RT_ERR_INVALID_ARG;
    return rows_of_part(height, band_rows, part, parts);
}

int rt_render_bands_device(rt_ctx *ctx, const rt_camera *cam, int w, int h, int band_rows, int part, int parts,
                           int max_depth, int32_t *dev_pixels_out, void *hip_stream)
{
    int rc = check_render_args(ctx, cam, w, h, max_depth);
    if (rc != RT_OK) return rc;
    if (band_rows <= 0 || parts <= 0 || part < 0 || part >= parts)
        return fail(ctx, RT_ERR_INVALID_ARG, "bad band partition: %d rows per band, part %d of %d", band_rows, part, parts);
    const RowSet rs{0, rows_of_part(h, band_rows, part, parts), band_rows, parts, part};
    if (!dev_pixels_out && rs.rows > 0) return fail(ctx, RT_ERR_INVALID_ARG, "dev_pixels_out is NULL");
    return guarded(ctx, [&]() -> int {
    DeviceState &d = ctx->dev[0];
    ctx->last_was_rows_device = true;
    ctx->last_w = w;
    ctx->last_h = h;
    ctx->copy_ms = ctx->gather_ms = 0.0;
    for (size_t i = 1; i < ctx->dev.size(); i++) ctx->dev[i].last_P = 0;
    return enqueue_rows(ctx, d, cam, w, h, rs, max_depth, dev_pixels_out,
                        hip_stream ? static_cast<hipStream_t>(hip_stream) : d.stream);
    });
}

static int render_to_host(rt_ctx *ctx, const rt_camera *cam, int w, int h, int max_depth, int32_t *pixels_out)
{
    int rc = RT_OK;
    const Tuning &tune = ctx->tune;
    const long long total = (long long)w * (long long)h;
    if (total > INT32_MAX) return fail(ctx, RT_ERR_LIMIT, "frame of %lld pixels exceeds 2^31-1", total);
    const int G = (int)ctx->dev.size();
    ctx->last_was_rows_device = false;
    ctx->last_w = w;
    ctx->last_h = h;
    const auto t_call = std::chrono::steady_clock::now();

    // Contiguous row blocks, one per device (rows like the outer Parallel.For).
    ctx->last_rows.assign(G + 1, 0);
    for (int g = 0; g <= G; g++) ctx->last_rows[g] = (int)(((long long)h * g) / G);

    DeviceState &root = ctx->dev[0];
    HIP_TRY(ctx, hipSetDevice(root.device));
    bool all_peers = true;
    for (const DeviceState &d : ctx->dev) all_peers = all_peers && d.peer_to_root;
    // Several devices: every device copies its own rows to the host over its own PCIe link (RT_HIP_HOST_GATHER=0,
    // tuning aid: they first meet in the first device's memory, over xGMI, and one copy brings the frame to the host).
    const bool host_gather = G > 1 && tune.host_gather != 0;
    if (!host_gather && (size_t)total > root.cap_out) {
        if (root.out) (void)hipFree(root.out);
        root.out = nullptr;
        root.cap_out = 0;
        HIP_TRY(ctx, hipMalloc((void **)&root.out, (size_t)total * sizeof(int32_t)));
        root.cap_out = (size_t)total;
    }
    // The copy to the host goes straight into the caller's array when it is page-locked (rt_pin_pixels,
    // rt_alloc_pixels), else into page-locked staging memory and from there into the caller's (pageable) array.
    const bool direct = ctx->pinned && pixels_out >= ctx->pinned &&
                        (size_t)(pixels_out - ctx->pinned) + (size_t)total <= ctx->pinned_count;
    if ((!direct || (!host_gather && !all_peers)) && (size_t)total > ctx->cap_staging) {
        if (ctx->staging) (void)hipHostFree(ctx->staging);
        ctx->staging = nullptr;
        ctx->cap_staging = 0;
        HIP_TRY(ctx, hipHostMalloc((void **)&ctx->staging, (size_t)total * sizeof(int32_t), hipHostMallocDefault));
        ctx->cap_staging = (size_t)total;
    }
    int32_t *const host = direct ? pixels_out : ctx->staging;
    // A piece of the frame on its way to the host: `runs` runs of `len` pixels, `stride` pixels apart, the first at
    // pixel `off`; ctx->copy_ev[k] of piece k says that it has arrived.  A staged frame goes in pieces of ~2 MiB:
    // while the DMA engine brings piece k + 1 the caller's thread moves piece k on into the array, so the frame
    // costs the slower of the two transfers, not their sum.  A direct one needs no pieces smaller than what
    // finishes together.
    struct Piece {
        size_t off, len, stride, runs;
    };
    std::vector<Piece> pieces;
    const size_t chunk = direct ? (size_t)total : tune.copy_chunk_pixels ? tune.copy_chunk_pixels : RT_STAGED_COPY_CHUNK_PIXELS;
    auto piece_event = [&](hipStream_t stream, const Piece &pc) -> int {
        if (ctx->copy_ev.size() <= pieces.size()) {
            hipEvent_t e;
            HIP_TRY(ctx, hipEventCreateWithFlags(&e, tune.handoff_trace ? hipEventDefault : hipEventDisableTiming));
            ctx->copy_ev.push_back(e);
        }
        HIP_TRY(ctx, hipEventRecord(ctx->copy_ev[pieces.size()], stream));
        pieces.push_back(pc);
        return RT_OK;
    };
    // Enqueues the copy of the frame's pixels [off, off + count) from src (which holds them in frame order).
    auto enqueue_copy = [&](const int32_t *src, size_t off, size_t count, hipStream_t stream) -> int {
        for (size_t o = off, len = 0; o < off + count; o += len) {
            // (the frame's last pieces are smaller)
            len = std::min(!direct && o + chunk >= (size_t)total ? std::max<size_t>(chunk / 4, 1) : chunk, off + count - o);
            HIP_TRY(ctx, hipMemcpyAsync(host + o, src + (o - off), len * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
            int r = piece_event(stream, Piece{o, len, len, 1});
            if (r != RT_OK) return r;
        }
        return RT_OK;
    };
    // Waits for the pieces in order and, when staged, moves each on into the caller's array - large frames with
    // helper threads beside the caller's (worker t of T takes the pieces t, t + T, ...; RT_HIP_MOVER_THREADS).
    auto drain_copy = [&]() -> int {
        const int wanted = direct || total < RT_MOVERS_MIN_PIXELS ? 0
                           : std::min(tune.mover_threads >= 0 ? tune.mover_threads : RT_MOVER_THREADS, RT_MAX_MOVER_THREADS);
        if (wanted > 0 && !ctx->movers) ctx->movers = new (std::nothrow) HostMovers();
        const int helpers = wanted > 0 && ctx->movers ? ctx->movers->helpers(wanted) : 0;
        std::atomic<int> failed{(int)hipSuccess};
        const int device = root.device;
        auto work = [&](int worker, int workers) {
            if (worker > 0) (void)hipSetDevice(device);
            for (size_t k = (size_t)worker; k < pieces.size(); k += (size_t)workers) {
                hipEvent_t ev = ctx->copy_ev[k];
                const hipError_t e = wait_until(ctx, [ev] { return hipEventQuery(ev); });
                if (e != hipSuccess) {
                    failed.store((int)e);
                    break;
                }
                if (!direct)
                    for (size_t r = 0; r < pieces[k].runs; r++)
                        copy_streaming(pixels_out + pieces[k].off + r * pieces[k].stride,
                                       ctx->staging + pieces[k].off + r * pieces[k].stride, pieces[k].len * sizeof(int32_t));
            }
            if (!direct) store_fence();   // this thread's streaming stores are visible before the call returns
        };
        if (helpers > 0) ctx->movers->run(work);
        else work(0, 1);
        if ((hipError_t)failed.load() == hipErrorNotReady) {
            char what[256];
            describe_progress(root, what, sizeof what);
            ctx->wedged = true;
            return fail(ctx, RT_ERR_STALLED, "the frame's pixels did not arrive within the time limit (RT_HIP_FRAME_TIMEOUT_MS); %s; "
                        "the context must be destroyed", what);
        }
        HIP_TRY(ctx, (hipError_t)failed.load());
        return RT_OK;
    };

    // One device, a large frame, nobody timing the kernels: the frame is rendered as row blocks one after the
    // other and each finished block is copied while the next one renders.  A block is every parts-th band of
    // RT_BAND_ROWS rows (rows cost very different amounts - the reflecting ceiling fills the top of the reference's
    // view - and with contiguous halves the 3840x2160 frame's first block took 0.45 ms and its second 0.31; a frame
    // costs its blocks' kernels plus the last block's copy), rendered stacked into its part of the device frame and
    // put in place by ONE strided copy to the host (source pitch = a band, destination pitch = parts bands).
    // Blocks in flight side by side, each in streams and ray queues of its own, were tried and are gone: kernels of
    // different hardware queues take the device from one another (two halves 1.35 ms against 1.07 one after the
    // other; the frame is bound by the work it issues, not by idle stretches another block could fill).
    // RT_HIP_FRAME_PARTS / RT_HIP_PART_BANDS=0 (tuning aids): number of blocks, contiguous blocks.
    int parts = 1;
    if (G == 1 && !ctx->profiling) {
        parts = tune.frame_parts > 0 ? tune.frame_parts
                                      : total < RT_MIN_SPLIT_PIXELS ? 1 : std::max(2, (int)(total / RT_PART_PIXELS));
        parts = std::max(1, std::min({parts, RT_MAX_FRAME_PARTS, h}));
    }
    if (parts > 1) {
        const bool bands = tune.part_bands != 0 && h >= 2 * parts * RT_BAND_ROWS;
        const size_t band_px = (size_t)RT_BAND_ROWS * (size_t)w;
        const int nbands = (h + RT_BAND_ROWS - 1) / RT_BAND_ROWS;
        if (!ctx->copy_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        double host_t[RT_MAX_FRAME_PARTS][2] = {};
        size_t first_piece[RT_MAX_FRAME_PARTS + 1] = {};
        size_t done_px = 0;   // pixels of the blocks enqueued so far = where the next block lies in root.out
        for (int p = 0; p < parts; p++) {
            if (!ctx->part_ev[p])
                HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->part_ev[p], tune.handoff_trace ? hipEventDefault : hipEventDisableTiming));
            if (p == 0 && tune.handoff_trace) {
                if (!ctx->trace_ev) HIP_TRY(ctx, hipEventCreate(&ctx->trace_ev));
                HIP_TRY(ctx, hipEventRecord(ctx->trace_ev, root.stream));
            }
            host_t[p][0] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_call).count();
            const int y0 = (int)(((long long)h * p) / parts), y1 = (int)(((long long)h * (p + 1)) / parts);
            const RowSet rs = bands ? RowSet{0, rows_of_part(h, RT_BAND_ROWS, p, parts), RT_BAND_ROWS, parts, p} : contiguous_rows(y0, y1);
            const size_t block = (size_t)rs.rows * (size_t)w;
            int32_t *const src = root.out + done_px;
            RowsCall call;
            call.part = p;
            call.keep_counters = p > 0;
            if ((rc = enqueue_rows(ctx, root, cam, w, h, rs, max_depth, src, root.stream, call)) != RT_OK) return rc;
            HIP_TRY(ctx, hipEventRecord(ctx->part_ev[p], root.stream));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->part_ev[p], 0));
            if (!bands) {
                if ((rc = enqueue_copy(src, (size_t)y0 * (size_t)w, block, ctx->copy_stream)) != RT_OK) return rc;
            } else {
                const int mine = (nbands - p + parts - 1) / parts;                // bands of block p
                const bool short_last = h % RT_BAND_ROWS != 0 && (nbands - 1) % parts == p;
                const int full = short_last ? mine - 1 : mine;                     // of full height
                // (a staged frame's bands go in several strided copies, so that the caller's thread moves the first
                // ones on while the later ones travel)
                const int step = direct ? std::max(1, full) : std::max(1, (int)(chunk / band_px));
                for (int b = 0, n = 0; b < full; b += n) {
                    // (the frame's last pieces are smaller)
                    n = std::min(p == parts - 1 && b + step >= full ? std::max(1, step / 4) : step, full - b);
                    const size_t first = ((size_t)b * (size_t)parts + (size_t)p) * band_px;   // the frame's pixel the copy starts at
                    HIP_TRY(ctx, hipMemcpy2DAsync(host + first, (size_t)parts * band_px * sizeof(int32_t), src + (size_t)b * band_px,
                                                  band_px * sizeof(int32_t), band_px * sizeof(int32_t), (size_t)n,
                                                  hipMemcpyDeviceToHost, ctx->copy_stream));
                    if ((rc = piece_event(ctx->copy_stream, Piece{first, band_px, (size_t)parts * band_px, (size_t)n})) != RT_OK) return rc;
                }
                if (short_last) {
                    const size_t first = (size_t)(nbands - 1) * band_px, len = (size_t)(h - (nbands - 1) * RT_BAND_ROWS) * (size_t)w;
                    HIP_TRY(ctx, hipMemcpyAsync(host + first, src + (size_t)full * band_px, len * sizeof(int32_t), hipMemcpyDeviceToHost,
                                                ctx->copy_stream));
                    if ((rc = piece_event(ctx->copy_stream, Piece{first, len, len, 1})) != RT_OK) return rc;
                }
            }
            done_px += block;
            host_t[p][1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_call).count();
            first_piece[p + 1] = pieces.size();
        }
        if ((rc = drain_copy()) != RT_OK) return rc;
        if ((rc = wait_stream(ctx, root, root.stream)) != RT_OK) return rc;   // (done already: the copies waited for the kernels)
        ctx->gather_ms = 0.0;
        // Here the copy overlaps the kernels: what is reported is the time of the whole call.
        ctx->copy_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
        if (tune.handoff_trace) {
            fprintf(stderr, "[rt_hip] rt_render %dx%d, %d %s blocks: %.1f us\n", w, h, parts, bands ? "interleaved" : "contiguous",
                    ctx->copy_ms * 1e3);
            for (int p = 0; p < parts; p++) {
                float k = -1.f, c = -1.f;
                (void)hipEventElapsedTime(&k, ctx->trace_ev, ctx->part_ev[p]);
                (void)hipEventElapsedTime(&c, ctx->trace_ev, ctx->copy_ev[first_piece[p + 1] - 1]);
                fprintf(stderr, "[rt_hip]   block %d: host enqueue %.1f .. %.1f us; device: kernels done %.1f us, copy done %.1f us\n", p,
                        host_t[p][0], host_t[p][1], k * 1e3, c * 1e3);
            }
            (void)hipGetLastError();
        }
        return check_schedule(ctx, root);
    }

    // Several devices and enough rows: device g takes the bands g, g + G, ... of RT_BAND_ROWS rows
    // (rows cost very different amounts; interleaved, every device gets its share of the
    // expensive ones) and renders them stacked into a block of its own.  ONE strided copy per device
    // (source pitch = a band, destination pitch = G bands) then puts them where they belong: in the host's frame,
    // every device over its own PCIe link at the same time, or (RT_HIP_HOST_GATHER=0) in device 0's frame over xGMI.
    // (Device gather: a device that cannot write device 0's memory takes a contiguous block, which it can hand
    // over through host memory in one piece.)
    const bool interleaved = G > 1 && h >= 2 * G * RT_BAND_ROWS && (host_gather || all_peers);
    if (interleaved) {
        const size_t band_px = (size_t)RT_BAND_ROWS * (size_t)w;
        const int nbands = (h + RT_BAND_ROWS - 1) / RT_BAND_ROWS;
        for (int g = 0; g < G; g++) {
            DeviceState &d = ctx->dev[g];
            const RowSet rs{0, rows_of_part(h, RT_BAND_ROWS, g, G), RT_BAND_ROWS, G, g};
            const size_t block = (size_t)rs.rows * (size_t)w;
            HIP_TRY(ctx, hipSetDevice(d.device));
            if (block > d.cap_bands) {
                if (d.bands) (void)hipFree(d.bands);
                d.bands = nullptr;
                d.cap_bands = 0;
                HIP_TRY(ctx, hipMalloc((void **)&d.bands, block * sizeof(int32_t)));
                d.cap_bands = block;
            }
            if ((rc = enqueue_rows(ctx, d, cam, w, h, rs, max_depth, d.bands, d.stream)) != RT_OK) return rc;
        }
        // A staged frame's bands go in several strided copies per device, so that the caller's thread can move the
        // first ones on while the later ones travel.
        for (int g = 0; g < G; g++) {
            DeviceState &d = ctx->dev[g];
            const int mine = (nbands - g + G - 1) / G;                      // bands of device g
            const bool short_last = h % RT_BAND_ROWS != 0 && (nbands - 1) % G == g;
            const int full = short_last ? mine - 1 : mine;                  // of full height
            int32_t *const target = host_gather ? host : root.out;
            const hipMemcpyKind kind = host_gather ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
            HIP_TRY(ctx, hipSetDevice(d.device));
            const int step = host_gather && !direct ? std::max(1, (int)(chunk / band_px)) : std::max(1, full);
            for (int b = 0; b < full; b += step) {
                const int n = std::min(step, full - b);
                const size_t first = ((size_t)b * (size_t)G + (size_t)g) * band_px;   // the frame's pixel the copy starts at
                HIP_TRY(ctx, hipMemcpy2DAsync(target + first, (size_t)G * band_px * sizeof(int32_t), d.bands + (size_t)b * band_px,
                                              band_px * sizeof(int32_t), band_px * sizeof(int32_t), (size_t)n, kind, d.stream));
                if (host_gather && (rc = piece_event(d.stream, Piece{first, band_px, (size_t)G * band_px, (size_t)n})) != RT_OK) return rc;
            }
            if (short_last) {
                const size_t first = (size_t)(nbands - 1) * band_px, len = (size_t)(h - (nbands - 1) * RT_BAND_ROWS) * (size_t)w;
                HIP_TRY(ctx, hipMemcpyAsync(target + first, d.bands + (size_t)full * band_px, len * sizeof(int32_t), kind, d.stream));
                if (host_gather && (rc = piece_event(d.stream, Piece{first, len, len, 1})) != RT_OK) return rc;
            }
            if (d.timed) HIP_TRY(ctx, hipEventRecord(d.ev[4], d.stream));
        }
    } else {
        // Device g renders its block; device 0 writes straight into the full frame.
        for (int g = 0; g < G; g++) {
            DeviceState &d = ctx->dev[g];
            const int y0 = ctx->last_rows[g], y1 = ctx->last_rows[g + 1];
            const size_t block = (size_t)(y1 - y0) * (size_t)w;
            int32_t *target = host_gather ? nullptr : root.out + (size_t)y0 * (size_t)w;
            if (g > 0 || host_gather) {
                HIP_TRY(ctx, hipSetDevice(d.device));
                if (block > d.cap_out) {
                    if (d.out) (void)hipFree(d.out);
                    d.out = nullptr;
                    d.cap_out = 0;
                    HIP_TRY(ctx, hipMalloc((void **)&d.out, std::max<size_t>(block, 1) * sizeof(int32_t)));
                    d.cap_out = block;
                }
                target = d.out;
            }
            if ((rc = enqueue_rows(ctx, d, cam, w, h, contiguous_rows(y0, y1), max_depth, target, d.stream)) != RT_OK) return rc;
            if (host_gather && block > 0) {
                if ((rc = enqueue_copy(d.out, (size_t)y0 * (size_t)w, block, d.stream)) != RT_OK) return rc;
                if (d.timed) HIP_TRY(ctx, hipEventRecord(d.ev[4], d.stream));
            }
        }
        // Device gather: every other device writes its block into device 0's frame (one copy per
        // device, each over its own link), then one copy to the host.
        for (int g = 1; g < G && !host_gather; g++) {
            DeviceState &d = ctx->dev[g];
            const int y0 = ctx->last_rows[g], y1 = ctx->last_rows[g + 1];
            const size_t bytes = (size_t)(y1 - y0) * (size_t)w * sizeof(int32_t);
            if (bytes == 0) continue;
            HIP_TRY(ctx, hipSetDevice(d.device));
            if (d.peer_to_root) {
                HIP_TRY(ctx, hipMemcpyPeerAsync(root.out + (size_t)y0 * (size_t)w, root.device, d.out, d.device, bytes,
                                                d.stream));
                if (d.timed) HIP_TRY(ctx, hipEventRecord(d.ev[4], d.stream));
            } else {
                // No peer access: through the page-locked staging frame, then up to device 0 on its
                // stream (the frame's final copy to the host follows on the same stream).
                int32_t *via = ctx->staging + (size_t)y0 * (size_t)w;
                HIP_TRY(ctx, hipMemcpyAsync(via, d.out, bytes, hipMemcpyDeviceToHost, d.stream));
                HIP_TRY(ctx, hipEventRecord(d.ev[4], d.stream));
                HIP_TRY(ctx, hipSetDevice(root.device));
                HIP_TRY(ctx, hipStreamWaitEvent(root.stream, d.ev[4], 0));
                HIP_TRY(ctx, hipMemcpyAsync(root.out + (size_t)y0 * (size_t)w, via, bytes, hipMemcpyHostToDevice, root.stream));
            }
        }
    }
    if (host_gather) {
        // The pieces arrive on G streams at once; the caller's thread takes them in the order they were enqueued.
        if ((rc = drain_copy()) != RT_OK) return rc;
    }
    ctx->gather_ms = 0.0;
    for (int g = G - 1; g >= 0; g--) {
        DeviceState &d = ctx->dev[g];
        HIP_TRY(ctx, hipSetDevice(d.device));
        if ((rc = wait_stream(ctx, d, d.stream)) != RT_OK) return rc;
        if ((g > 0 || interleaved || host_gather) && d.timed && d.last_P > 0) {
            float ms = 0.f;   // (host gather: the device's own copy to the host)
            if (hipEventElapsedTime(&ms, d.ev[3], d.ev[4]) == hipSuccess)
                ctx->gather_ms = std::max(ctx->gather_ms, (double)ms);
        }
        if (d.last_P > 0 && (rc = check_schedule(ctx, d)) != RT_OK) return rc;
    }
    if (host_gather) {
        ctx->copy_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
        return RT_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    if ((rc = enqueue_copy(root.out, 0, (size_t)total, root.stream)) != RT_OK) return rc;
    if ((rc = drain_copy()) != RT_OK) return rc;
    ctx->copy_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}

int rt_render(rt_ctx *ctx, const rt_camera *cam, int w, int h, int max_depth, int32_t *pixels_out)
{
    const int rc = check_render_args(ctx, cam, w, h, max_depth);
    if (rc != RT_OK) return rc;
    if (!pixels_out) return fail(ctx, RT_ERR_INVALID_ARG, "pixels_out is NULL");
    return guarded(ctx, [&]() -> int { return render_to_host(ctx, cam, w, h, max_depth, pixels_out); });
}

int rt_frame_status(rt_ctx *ctx)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    return guarded(ctx, [&]() -> int {
        int rc = RT_OK;
        for (DeviceState &d : ctx->dev) {
            if (!d.last_stream) continue;
            HIP_TRY(ctx, hipSetDevice(d.device));
            int r = wait_stream(ctx, d, d.last_stream);
            if (r == RT_OK) r = check_schedule(ctx, d);
            if (r != RT_OK) rc = r;
        }
        return rc;
    });
}

int rt_debug_rays(rt_ctx *ctx, const rt_camera *cam, int w, int h, int row, int x_step, int max_rays,
                  rt_debug_ray *out, int *out_count)
{
    int rc = check_render_args(ctx, cam, w, h, 0);
    if (rc != RT_OK) return rc;
    if (x_step <= 0 || !out || !out_count) return fail(ctx, RT_ERR_INVALID_ARG, "bad debug-ray arguments");
    const int count = (w + x_step - 1) / x_step;
    if (count > max_rays) return fail(ctx, RT_ERR_LIMIT, "%d rays do not fit the caller's %d entries", count, max_rays);
    static_assert(sizeof(rt_debug_ray) == 14 * sizeof(float), "rt_debug_ray is 14 packed words");
    return guarded(ctx, [&]() -> int {
    DeviceState &d = ctx->dev[0];
    HIP_TRY(ctx, hipSetDevice(d.device));
    std::vector<float> host((size_t)count * 16);
    float *dev = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&dev, (size_t)count * 16 * sizeof(float)));
    const DevCamera dc = make_camera(cam, w, h, contiguous_rows(0, h));
    rtk_debug_rays(d.scene, dc, row, x_step, count, dev, d.stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(host.data(), dev, host.size() * sizeof(float), hipMemcpyDeviceToHost, d.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d.stream);
    (void)hipFree(dev);
    if (e != hipSuccess) return fail(ctx, RT_ERR_HIP, "debug rays failed: %s", hipGetErrorString(e));
    for (int i = 0; i < count; i++) memcpy(&out[i], &host[(size_t)i * 16], sizeof(rt_debug_ray));
    *out_count = count;
    return RT_OK;
    });
}

int rt_fxaa_device(rt_ctx *ctx, const int32_t *dev_pixels_in, int32_t *dev_pixels_out, int w, int h, void *hip_stream)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (w <= 0 || h <= 0) return fail(ctx, RT_ERR_INVALID_ARG, "bad frame size %d x %d", w, h);
    if ((long long)w * (long long)h > INT32_MAX) return fail(ctx, RT_ERR_LIMIT, "frame exceeds 2^31-1 pixels");
    if (!dev_pixels_in || !dev_pixels_out) return fail(ctx, RT_ERR_INVALID_ARG, "a frame pointer is NULL");
    const size_t n = (size_t)w * (size_t)h;
    if (dev_pixels_in < dev_pixels_out + n && dev_pixels_out < dev_pixels_in + n)
        return fail(ctx, RT_ERR_INVALID_ARG, "the filter reads neighbours: input and output frames must not overlap");
    return guarded(ctx, [&]() -> int {
    DeviceState &d = ctx->dev[0];
    HIP_TRY(ctx, hipSetDevice(d.device));
    rtk_fxaa(dev_pixels_in, dev_pixels_out, w, h, hip_stream ? static_cast<hipStream_t>(hip_stream) : d.stream);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
    });
}

int rt_unpin_pixels(rt_ctx *ctx)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    return guarded(ctx, [&]() -> int {
    if (ctx->pinned) {
        for (DeviceState &d : ctx->dev) {   // no copy into the range may be in flight
            HIP_TRY(ctx, hipSetDevice(d.device));
            HIP_TRY(ctx, hipStreamSynchronize(d.stream));
        }
        void *p = ctx->pinned_base;
        const bool owned = ctx->pinned_owned;
        ctx->pinned = nullptr;
        ctx->pinned_base = nullptr;
        ctx->pinned_count = 0;
        ctx->pinned_owned = false;
        HIP_TRY(ctx, owned ? hipHostFree(p) : hipHostUnregister(p));
    }
    return RT_OK;
    });
}

int rt_pin_pixels(rt_ctx *ctx, int32_t *pixels, int64_t count)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!pixels || count <= 0) return fail(ctx, RT_ERR_INVALID_ARG, "bad framebuffer range (%p, %lld)", (void *)pixels, (long long)count);
    int rc = rt_unpin_pixels(ctx);
    if (rc != RT_OK) return rc;
    return guarded(ctx, [&]() -> int {
    HIP_TRY(ctx, hipSetDevice(ctx->dev[0].device));
    // Whole pages: every page that holds a byte of the array (all of them are mapped).
    const uintptr_t page = 4096, lo = (uintptr_t)pixels & ~(page - 1),
                    hi = ((uintptr_t)(pixels + count) + page - 1) & ~(page - 1);
    const hipError_t e = hipHostRegister((void *)lo, (size_t)(hi - lo), hipHostRegisterDefault);
    if (e != hipSuccess) {
        // Some systems refuse to page-lock user memory; rt_render keeps using its staged copy.
        (void)hipGetLastError();   // the failure must not linger as the process's last HIP error
        return fail(ctx, RT_ERR_HIP, "hipHostRegister refused the framebuffer (%s); rt_render keeps staging its copy",
                    hipGetErrorString(e));
    }
    ctx->pinned_base = (void *)lo;
    ctx->pinned = pixels;
    ctx->pinned_count = (size_t)count;
    return RT_OK;
    });
}

int rt_alloc_pixels(rt_ctx *ctx, int64_t count, int32_t **pixels_out)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!pixels_out) return fail(ctx, RT_ERR_INVALID_ARG, "pixels_out is NULL");
    *pixels_out = nullptr;
    if (count <= 0 || count > INT32_MAX) return fail(ctx, RT_ERR_INVALID_ARG, "bad framebuffer size %lld", (long long)count);
    int rc = rt_unpin_pixels(ctx);
    if (rc != RT_OK) return rc;
    return guarded(ctx, [&]() -> int {
    HIP_TRY(ctx, hipSetDevice(ctx->dev[0].device));
    void *p = nullptr;
    HIP_TRY(ctx, hipHostMalloc(&p, (size_t)count * sizeof(int32_t), hipHostMallocDefault));
    ctx->pinned_base = p;
    ctx->pinned = static_cast<int32_t *>(p);
    ctx->pinned_count = (size_t)count;
    ctx->pinned_owned = true;
    *pixels_out = ctx->pinned;
    return RT_OK;
    });
}

int rt_get_stats(rt_ctx *ctx, rt_stats *out)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!out) return fail(ctx, RT_ERR_INVALID_ARG, "out is NULL");
    memset(out, 0, sizeof *out);
    return guarded(ctx, [&]() -> int {
    out->num_devices = (int32_t)ctx->dev.size();
    double kernel_ms = 0.0, primary_ms = 0.0, trace_ms = 0.0, shade_ms = 0.0;
    for (DeviceState &d : ctx->dev) {
        if (d.last_P == 0 || !d.last().cnt) continue;
        HIP_TRY(ctx, hipSetDevice(d.device));
        HIP_TRY(ctx, hipDeviceSynchronize());
        std::vector<int> cnt(2 * (size_t)(d.last_depth + 2));
        DevCounters c{};
        if (d.snap_parts > 0) {   // a frame rendered in parts: the levels' record counts of all parts
            for (int p = 0; p < d.snap_parts; p++) {
                const int *part_cnt = reinterpret_cast<const int *>(d.outcome_snap + (size_t)p * FRAME_OUTCOME_BYTES +
                                                                    sizeof(DevSched));
                for (size_t i = 0; i < cnt.size(); i++) cnt[i] += part_cnt[i];
            }
        } else {
            HIP_TRY(ctx, hipMemcpy(cnt.data(), d.last().cnt, cnt.size() * sizeof(int), hipMemcpyDeviceToHost));
        }
        {
            static_assert(sizeof(DevCounters) % sizeof(uint64_t) == 0, "DevCounters is an array of 64-bit counts");
            constexpr size_t words = sizeof(DevCounters) / sizeof(uint64_t);
            std::vector<uint64_t> slots(RT_COUNTER_SLOTS * words);
            HIP_TRY(ctx, hipMemcpy(slots.data(), d.last().counters, slots.size() * sizeof(uint64_t),
                                   hipMemcpyDeviceToHost));
            uint64_t sum[words] = {};
            for (size_t i = 0; i < slots.size(); i++) sum[i % words] += slots[i];
            memcpy(&c, sum, sizeof c);
        }
        int rc = check_schedule(ctx, d);
        if (rc != RT_OK) return rc;
        uint64_t hits = 0, secondary = 0;
        for (int k = 0; k <= d.last_depth; k++) {
            hits += (uint64_t)cnt[2 * k] + (uint64_t)cnt[2 * k + 1];
            secondary += (uint64_t)cnt[2 * k];
        }
        const uint64_t P = (uint64_t)d.last_P;
        out->rays_primary += P;
        out->shaded_hits += hits;
        out->rays_shadow += hits * (uint64_t)d.scene.nl;
        out->rays_secondary += secondary;
        out->sphere_tests += c.sphere_tests + P * (uint64_t)d.scene.ns;
        out->plane_tests += c.plane_tests + P * (uint64_t)d.scene.np;
        out->triangle_tests += c.triangle_tests + P * (uint64_t)d.scene.nt;
        out->exact_candidates += c.exact_candidates;
        out->filter_ray_tests += c.filter_ray_tests;
        out->filter_origin_terms += c.filter_origin_terms;
        out->primary_filter_ray_tests += c.primary_filter_ray_tests;
        out->primary_exact_candidates += c.primary_exact_candidates;
        out->shadow_candidates += c.shadow_candidates;
        if (d.timed) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, d.ev[0], d.ev[3]) == hipSuccess) kernel_ms = std::max(kernel_ms, (double)ms);
            if (hipEventElapsedTime(&ms, d.ev[0], d.ev[1]) == hipSuccess) primary_ms = std::max(primary_ms, (double)ms);
            if (hipEventElapsedTime(&ms, d.ev[1], d.ev[2]) == hipSuccess) trace_ms = std::max(trace_ms, (double)ms);
            if (hipEventElapsedTime(&ms, d.ev[2], d.ev[3]) == hipSuccess) shade_ms = std::max(shade_ms, (double)ms);
            (void)hipGetLastError();  // a failed query must not linger as the process's last error
        }
    }
    out->kernel_ms = kernel_ms;
    out->primary_ms = primary_ms;
    out->trace_ms = trace_ms;
    out->shade_ms = shade_ms;
    out->gather_ms = ctx->gather_ms;
    out->copy_ms = ctx->copy_ms;
    return RT_OK;
    });
}

}  // extern "C"
