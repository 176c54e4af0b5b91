// rt_capi.hip -- C ABI (include/rtrace_hip.h) over the gfx950 kernels.  Host side only orchestrates:
// validation, device copies of the Scene, tile tables, streams, launches, read-back.  No pixel arithmetic
// happens on the host and there is NO CPU fallback: without a device every render entry point fails.
#include "../../include/rtrace_hip.h"
#include "rt_debug.h"
#include "host/hierarchy.hpp"
#include "rt_kernels.hpp"
#include "rt_skip.hpp"
#include "rt_skip_fast.hpp"
#include "rt_skip_fast64.hpp"
#include "rt_skip2_fast.hpp"
#include "rt_skip2.hpp"
#include "rt_flat.hpp"
#include "rt_flat_wf.hpp"
#include "rt_flat_sc.hpp"
#include "rt_flat_f64.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>       // types and prototypes only: librccl.so is loaded with dlopen when the first gang is created
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local char g_err[512] = "";

// What the last render call of this thread launched (rt_last_launch_flags, include/rtrace_hip.h).
thread_local uint32_t g_launch_flags = 0;

// Diagnostic controls (rt_debug.h): process-wide, -1 = default.  The library reads no environment variable.
// They exist only in the build with -DRT_TEST_HOOKS (tests/c/librtrace_hip_test.so: the parity tests and tools/); in the product
// library every control is a compile-time -1, the rt_debug_* entry points do not exist and the loop flavours only a control can
// select are not instantiated.
#ifdef RT_TEST_HOOKS
struct KnobArray {
    std::atomic<long long> v[RT_DEBUG_KEYS];
    KnobArray() { for (auto &k : v) k.store(-1, std::memory_order_relaxed); }
    std::atomic<long long> &operator[](int i) { return v[i]; }
} g_knob;
std::atomic<long long> g_count[RT_DEBUG_COUNTERS] = {};
std::atomic<bool> g_trace_on{ false };
std::mutex g_trace_mu;
std::string g_trace_path;
// (a background builder of dispatch orders works for a caller that had no control set: it must not see one that is set meanwhile)
thread_local bool g_knobs_at_default = false;
inline long long knob(int key) { return g_knobs_at_default ? -1 : g_knob[key].load(std::memory_order_relaxed); }
inline void count_event(int counter, long long n = 1) { g_count[counter].fetch_add(n, std::memory_order_relaxed); }
inline void count_store(int counter, long long v) { g_count[counter].store(v, std::memory_order_relaxed); }
inline void knobs_at_default() { g_knobs_at_default = true; }
#else
constexpr long long knob(int) { return -1; }
inline void count_event(int, long long = 1) {}
inline void count_store(int, long long) {}
inline void knobs_at_default() {}
#endif

rt_status hip_fail(hipError_t e, const char *what, int line, const char *file = __builtin_FILE())      // (file: the caller's, one of rt_capi*.h*)
{
    const char *base = strrchr(file, '/');
    snprintf(g_err, sizeof g_err, "%s failed at %s:%d: %s", what, base ? base + 1 : file, line, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP;
}

#define HIP_TRY(expr)                                                       \
    do {                                                                    \
        hipError_t e__ = (expr);                                            \
        if (e__ != hipSuccess) return hip_fail(e__, #expr, __LINE__);       \
    } while (0)

// RT_DEBUG_PRINT_COSTS: where rt_scene_create and the first use of a tile list spend their host time
struct StageClock {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what)
    {
        if (knob(RT_DEBUG_PRINT_COSTS) <= 0) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[rtrace_hip] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(n - t).count());
        t = n;
    }
};

// Per-call working set: own stream, tile table, counters, staging output.  A scene keeps a pool of these so
// concurrent callers (the reference's pool threads, render.rs:283) never share one.
struct Context {
    hipStream_t stream = nullptr;
    bool owns_stream = true;           // (a scene's first context works on the scene's own stream: a stream is a hardware queue plus 20 MB of
                                       // staging the runtime allocates with it -- 11 - 16 ms of a fresh process, each)
    hipStream_t stream2 = nullptr;     // rt_render_frame_stream: a batch is encoded here while the next one renders on `stream` (made when first needed)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // two upload slots (a flat pass uses two tables); pinned host side: the H2D copy is truly asynchronous
    rt::TileDev *d_tiles[2] = { nullptr, nullptr };
    rt::TileDev *h_tiles[2] = { nullptr, nullptr };
    size_t tiles_cap[2] = { 0, 0 };
    bool tiles_live[2] = { false, false };   // the slot was uploaded through during the current lease: what this context enqueued may still read it (upload_tiles)
    rt::Counters *d_counters = nullptr;
    uint8_t *d_out = nullptr;
    size_t out_cap = 0;
    uint8_t *h_out = nullptr;         // pinned staging for pageable destinations (rt_render_tiles), same size as d_out
    size_t h_out_cap = 0;
    std::vector<hipEvent_t> chunk_ev; // one per staged chunk in flight
    void *d_sample_gdot = nullptr;    // SPLIT path: per-sample n.light [spp*spp][n_px] (REAL)
    uint8_t *d_sample_state = nullptr;
    size_t sample_cap = 0;            // bytes of d_sample_gdot
    void *d_queue1 = nullptr, *d_queue2 = nullptr;   // flat wavefront pipeline: shadow-ray queues, Quad<REAL> per sample
    size_t queue_cap = 0;             // bytes of each queue
    rt::FlatQueues *d_queues = nullptr;
    unsigned flat_first_pass_items = 0;   // items the first shadow pass of the last flat launch covered (rt_stats.tests_executed)
    bool busy = false;       // leased to a caller right now
    bool inflight = false;   // released by an asynchronous caller; reusable once ev1 has completed

    ~Context()
    {
        for (int k = 0; k < 2; ++k) {
            if (d_tiles[k]) (void)hipFree(d_tiles[k]);
            if (h_tiles[k]) (void)hipHostFree(h_tiles[k]);
        }
        if (d_counters) (void)hipFree(d_counters);
        if (d_out) (void)hipFree(d_out);
        if (h_out) (void)hipHostFree(h_out);
        for (hipEvent_t e : chunk_ev) (void)hipEventDestroy(e);
        if (d_sample_gdot) (void)hipFree(d_sample_gdot);
        if (d_sample_state) (void)hipFree(d_sample_state);
        if (d_queue1) (void)hipFree(d_queue1);
        if (d_queue2) (void)hipFree(d_queue2);
        if (d_queues) (void)hipFree(d_queues);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream2) (void)hipStreamDestroy(stream2);
        if (stream && owns_stream) (void)hipStreamDestroy(stream);
    }
};

}  // namespace

struct rt_scene {
    int device = 0;
    double setup_total_ms = 0.0, setup_first_stream_ms = 0.0;      // rt_scene_setup_cost
    rt_precision precision = RT_F32;
    uint32_t n_items = 0, n_bounds = 0;
    void *d_items = nullptr;       // Item<REAL>[n_items], DFS order
    void *d_prim = nullptr, *d_shad = nullptr;   // Node<REAL>[n_nodes + kNodePad]: skip-pointer streams (RT_TRAVERSAL_SKIP)
    void *d_cprim = nullptr, *d_cshad = nullptr;   // fused scenes: the compacted streams of the fused assembly loops
    // f32: FNode copies of the four streams for the filtered loops (rt_skip.hpp) and the compacted stream's own_item table
    void *d_xprim = nullptr, *d_xshad = nullptr, *d_xcprim = nullptr, *d_xcshad = nullptr, *d_xown = nullptr;
    rt::FilterConsts fc{};
    void *d_fc = nullptr;          // device copy of fc
    uint32_t n_nodes = 0, n_fnodes = 0;
    bool fused = false;            // every BOUND is followed by an ITEM with the same centre (rt_skip.hpp, Node)
    // the hierarchy once more in sibling-contiguous order for the lane-cooperative walk (rt_coop.hpp; f64: CNode64); coop.fanout == 0: none
    void *d_coop_prim = nullptr, *d_coop_shad = nullptr;
    rt::CoopView coop{};
    void *d_fprim = nullptr, *d_fprim_rr = nullptr, *d_fshad = nullptr;   // pre-formed per-item terms (RT_TRAVERSAL_FLAT)
    uint32_t n_padded = 0;
    // The flat-scan arrays are derived when RT_TRAVERSAL_FLAT is first asked for (ensure_flat): a caller of the hierarchy walk -- the default,
    // `make image` -- never pays for them.  h_items: the caller's items, kept for that day.
    std::vector<unsigned char> h_items;
    std::mutex flat_mu;
    bool flat_ready = false;
    void *d_f64_pf = nullptr, *d_f64_sf = nullptr, *d_f64_sg = nullptr;        // f64, the filtered LDS scan (rt_flat_f64.hpp)
    double flat_centre64[3] = { 0, 0, 0 };
    void *d_pf = nullptr, *d_pe = nullptr, *d_sg = nullptr, *d_se = nullptr;   // f32, the scalar-fed scan (rt_flat_sc.hpp): filter groups of four
                                                                               // items and exact records, primary / shadow
    uint32_t flat_filter_bytes = 0, flat_shadow_bytes = 0;                     // 128 x number of primary / shadow filter group pairs
    float flat_centre[3] = { 0, 0, 0 };                                        // the shadow filter's reference point (centroid of the item centres)
    double light[3] = { 0, 0, 0 }, eye[3] = { 0, 0, 0 };   // exact copies of the REAL values
    std::mutex mu;
    std::vector<std::unique_ptr<Context>> pool;
    // Immutable device copies of recently used tile tables (a scheduler re-submits the same bucket list every
    // frame): a hit means a pass enqueues nothing but its kernel.
    // dev_order: one descriptor per 16x16 block of the pass, most expensive first (block_order below), or NULL.
    // One dispatch order of a tile list: descriptors (+ optional workgroup offsets, + the holes of a cooperative pass, rt_kernels.hpp BlockList).
    // A list whose pass could use the lane-cooperative walk gets several (none / a few cooperative thresholds) and the library TRIES them: the
    // first launches take turns, timed with a pair of events each, and the fastest is kept -- whether the cooperative walk pays depends on
    // how much of the pass is tail, which no estimate made here predicted as well as three measurements do.
    struct Order { rt::BlockDesc *dev_order = nullptr; uint32_t n_order = 0; uint32_t *dev_wg = nullptr; uint32_t n_wg = 0; uint64_t *dev_holes = nullptr; uint32_t n_holes = 0;
                   // the trial: kOrderTrialSamples timed launches, each with an event pair of its own (they may all be in flight at once)
                   // (slot = count % kOrderTrialSamples; a sample whose events could not be read is LOST and issued again)
                   hipEvent_t e0[3] = { nullptr, nullptr, nullptr }, e1[3] = { nullptr, nullptr, nullptr }; int issued = 0, harvested = 0, good = 0; float best_ms = 1e30f; };
    struct CachedTable { std::vector<rt::TileDev> host; unsigned w = 0, h = 0, passes = 0; rt::TileDev *dev = nullptr; std::vector<Order> orders; int chosen = 0; unsigned turn = 0; bool building = false;
                         void *order_arena = nullptr;      // ONE device allocation holds every order's arrays (allocation calls wait for a busy device)
                         long long coop_key = 0;
                         // a table uploaded asynchronously on its first caller's stream (device_table): until `landed` has completed, launches on
                         // OTHER streams wait for it
                         hipEvent_t landed = nullptr; hipStream_t landed_on = nullptr; };
    std::vector<CachedTable> tables;
    std::vector<std::thread> builders;           // dispatch orders being made in the background (build_orders_async); joined by rt_scene_destroy
    // The scene's own worker thread (started by rt_scene_create next to the cost map): it makes the dispatch orders of new tile lists --
    // handing it a list costs the first frame a few microseconds, starting a thread cost it 75-115 us (`builders` is the fallback)
    std::thread worker;
    std::mutex wmu;
    std::condition_variable wcv;
    std::deque<std::function<void()>> wjobs;
    bool wstop = false;
    // pinned staging for the tile tables of new lists (bump-allocated, never reused: a copy may still be queued behind the caller's kernels)
    char *h_tab_stage = nullptr;
    size_t tab_stage_used = 0;
    // Tests per primary ray (its shadow ray included) on a kCostRes x kCostRes grid over the camera's field of view,
    // rendered once per scene with the counting kernel.  It only ever decides the ORDER in which blocks are dispatched.
    std::once_flag cost_once;
    // pinned staging of what rt_scene_create uploads (upload_words): bump-allocated, released when the scene's streams have been derived
    char *h_up = nullptr; size_t up_cap = 0, up_used = 0;
    std::mutex exact_mu;                   // exact_block_costs: the cost arena is also where a tile list's heaviest blocks are counted again
    hipStream_t cost_stream = nullptr;
    void *d_cost_arena = nullptr, *h_cost = nullptr;
    bool cost_started = false;
    bool main_stream_taken = false;        // cost_stream doubles as the first context's stream (acquire)
    std::vector<uint32_t> cost_map;
    // Concurrent rt_render_region callers (the reference's pool threads, render.rs:283-294) are merged into shared passes:
    // whoever finds no pass running becomes its leader and renders every request that is waiting at that moment.
    struct RegionReq { rt_options o; rt_traversal trav; rt_region region; uint8_t *out; rt_status st = RT_OK; bool taken = false, done = false; char err[256] = "";
                       std::condition_variable cv; };      // signalled when the request is done, or when its owner should lead
    // Frame-ahead for rt_render_region (render.rs:283-294 calls it once per 64x64 bucket): the first request for a bucket of the
    // scheduler's grid renders the WHOLE grid in one pass into pinned staging, and the following requests of that frame are a 16 KB
    // copy each.  A bucket is handed out once per rendered frame: asking for one again means the caller has started its next frame,
    // and the grid is rendered again -- every byte a caller receives was rendered for the frame it belongs to.
    // While the caller copies frame k's buckets out the device is idle: the pass for frame k + 1 is started right away into a second
    // staging buffer (a Scene is immutable, so its bytes are those a pass started later would produce) and is simply waited for when
    // the caller comes back for its next frame.  One pass too many is rendered when the caller stops (RT_DEBUG_FRAME_AHEAD = 1: off).
    struct FrameAhead { rt_options o{}; rt_traversal trav = RT_TRAVERSAL_SKIP; uint8_t *h = nullptr, *h_next = nullptr; size_t cap = 0; std::vector<size_t> off;
                        std::vector<rt_region> grid; std::vector<uint8_t> served; bool valid = false, next_inflight = false;
                        rt_options seen_o{}; rt_traversal seen_trav = RT_TRAVERSAL_SKIP; int seen_idx = -1;      // the last lone request (frame-ahead engages with the second bucket)
                        hipStream_t stream = nullptr; hipEvent_t ev = nullptr; int readers = 0; std::mutex mu; std::condition_variable cv; } ahead;
    std::mutex comb_mu;
    std::vector<RegionReq *> comb_pending;
    int comb_leaders = 0;              // passes being led right now (<= kMaxRegionLeaders)
};

namespace {
#include "rt_capi_scene.hpp"
#include "rt_capi_dispatch.hpp"
#include "rt_capi_launch.hpp"
}  // namespace

extern "C" {
#include "rt_capi_entry.hpp"
#include "rt_capi_gang.hpp"
}  // extern "C"
