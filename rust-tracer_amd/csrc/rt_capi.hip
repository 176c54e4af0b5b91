// rt_capi.hip -- C ABI (include/rtrace_hip.h) over the gfx950 kernels.  Host side only orchestrates:
// validation, device copies of the Scene, tile tables, streams, launches, read-back.  No pixel arithmetic
// happens on the host and there is NO CPU fallback: without a device every render entry point fails.
#include "../../include/rtrace_hip.h"
#include "rt_debug.h"
#include "rt_kernels.hpp"
#include "rt_skip.hpp"
#include "rt_skip_fast.hpp"
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

rt_status hip_fail(hipError_t e, const char *what, int line)
{
    snprintf(g_err, sizeof g_err, "%s failed at rt_capi.hip:%d: %s", what, line, hipGetErrorString(e));
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
    // f32: the hierarchy once more in sibling-contiguous order for the lane-cooperative walk (rt_coop.hpp); coop.fanout == 0: none
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

// `bytes` (a multiple of 4) of host data into device memory on `stream`, through pinned staging and k_upload_words -- no copy engine
// (rt_kernels.hpp says why).  The staging is the scene's arena while rt_scene_create runs (reserve_upload), else a buffer of the caller's.
rt_status upload_words(void *d_dst, const void *h_pinned_src, size_t bytes, hipStream_t stream)
{
    void *alias = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&alias, const_cast<void *>(h_pinned_src), 0));
    const size_t n = bytes / 4;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(rt::k_upload_words, dim3(std::max(1u, blocks)), dim3(256), 0, stream, static_cast<const uint32_t *>(alias), static_cast<uint32_t *>(d_dst), n);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}
rt_status scene_upload(rt_scene *s, void *d_dst, const void *src, size_t bytes)       // rt_scene_create's uploads, on the scene's stream
{
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (s->h_up && s->up_used + need <= s->up_cap) {
        char *h = s->h_up + s->up_used;
        memcpy(h, src, bytes);
        s->up_used += need;
        return upload_words(d_dst, h, bytes, s->cost_stream);
    }
    HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, s->cost_stream));      // (no arena: the copy engine after all; `src` outlives the caller's synchronise)
    return RT_OK;
}

// Contexts released by asynchronous callers and still in flight: enough to keep the device fed; each may hold per-sample
// buffers (GBs at 4096^2 x 16).  Synchronous callers (one per host thread) each hold their own while they run.
constexpr size_t kMaxAsyncContexts = 3;

void release(rt_scene *s, Context *c, bool inflight);

rt_status acquire(rt_scene *s, Context **out)
{
    Context *victim = nullptr;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        for (auto &c : s->pool) {
            if (c->busy) continue;
            if (c->inflight) {
                if (hipEventQuery(c->ev1) != hipSuccess) { (void)hipGetLastError(); continue; }
                c->inflight = false;
            }
            c->busy = true; c->tiles_live[0] = c->tiles_live[1] = false; *out = c.get(); return RT_OK;
        }
        // A caller that keeps enqueuing asynchronous passes without ever synchronising must not grow the pool (and its
        // per-sample buffers) without bound: past kMaxAsyncContexts, take the oldest pass still in flight and wait for it
        // OUTSIDE the lock (other threads of the scene, the one-kernel fast path included, go on meanwhile).
        size_t inflight = 0;
        for (auto &c : s->pool) inflight += (!c->busy && c->inflight) ? 1 : 0;
        if (inflight >= kMaxAsyncContexts)
            for (auto &c : s->pool)
                if (!c->busy && c->inflight) { c->busy = true; victim = c.get(); break; }
    }
    if (victim) {
        hipError_t e = hipEventSynchronize(victim->ev1);
        if (e != hipSuccess) { release(s, victim, true); return hip_fail(e, "hipEventSynchronize(context)", __LINE__); }
        victim->inflight = false;
        victim->tiles_live[0] = victim->tiles_live[1] = false;
        *out = victim;
        return RT_OK;
    }
    std::unique_ptr<Context> c(new (std::nothrow) Context());
    if (!c) return RT_ERR_OUT_OF_MEMORY;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        if (s->cost_stream && !s->main_stream_taken) { s->main_stream_taken = true; c->stream = s->cost_stream; c->owns_stream = false; }
    }
    if (!c->stream) HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    HIP_TRY(hipMalloc(&c->d_counters, sizeof(rt::Counters) * rt::kCounterStripes));
    c->busy = true;
    *out = c.get();
    std::lock_guard<std::mutex> lk(s->mu);
    s->pool.push_back(std::move(c));
    return RT_OK;
}

void release(rt_scene *s, Context *c, bool inflight)
{
    std::lock_guard<std::mutex> lk(s->mu);
    c->busy = false;
    c->inflight = inflight;
}

struct Lease {
    rt_scene *s; Context *c; bool inflight = false;
    ~Lease() { if (c) release(s, c, inflight); }
};

// Validates the regions (ImageRegion invariants, inside the image) and lays out blocks + output offsets.
rt_status build_tile_table(const rt_options *o, const rt_region *tiles, uint32_t n, std::vector<rt::TileDev> &tab,
                           uint64_t *total_px, uint32_t *total_blocks, uint32_t block_w = rt::kBlockW, uint32_t block_h = rt::kBlockH)
{
    uint64_t px = 0, blocks = 0;
    tab.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        const rt_region &t = tiles[i];
        if (!(t.l < t.r && t.b < t.t && t.r <= o->width && t.t <= o->height)) {
            snprintf(g_err, sizeof g_err, "tile %u {l=%u,t=%u,r=%u,b=%u} is empty or outside %ux%u", i, t.l, t.t,
                     t.r, t.b, o->width, o->height);
            return RT_ERR_INVALID_REGION;
        }
        const uint32_t w = t.r - t.l, h = t.t - t.b;
        const uint32_t bxs = (w + block_w - 1) / block_w, bys = (h + block_h - 1) / block_h;
        if (px + (uint64_t)w * h > 0xFFFFFFFFull || blocks + (uint64_t)bxs * bys > 0x7FFFFFFFull) {
            snprintf(g_err, sizeof g_err, "tile list too large for one pass");
            return RT_ERR_INVALID_ARGUMENT;
        }
        tab[i] = rt::TileDev{ t.l, t.t, t.r, t.b, (uint32_t)px, (uint32_t)blocks, bxs };
        px += (uint64_t)w * h;
        blocks += (uint64_t)bxs * bys;
    }
    *total_px = px;
    *total_blocks = (uint32_t)blocks;
    return RT_OK;
}

template <typename T>
rt::FlatView<T> flat_view_of(const rt_scene *s)
{
    rt::FlatView<T> v;
    v.prim = static_cast<const rt::Quad<T> *>(s->d_fprim);
    v.prim_rr = static_cast<const T *>(s->d_fprim_rr);
    v.shad = static_cast<const rt::Quad<T> *>(s->d_fshad);
    v.items = static_cast<const rt::Item<T> *>(s->d_items);
    v.n_items = s->n_items;
    v.n_padded = s->n_padded;
    v.light = { (T)s->light[0], (T)s->light[1], (T)s->light[2] };
    v.eye = { (T)s->eye[0], (T)s->eye[1], (T)s->eye[2] };
    return v;
}

template <typename T>
rt_status upload_flat(rt_scene *s, const void *host_items)
{
    s->n_padded = (s->n_items + 7u) & ~7u;     // the scan consumes 4 items per step, unrolled twice
    // any-hit scan order of the shadow pass: radius descending (stable), see k_build_flat
    std::vector<unsigned> order(s->n_items);
    for (unsigned i = 0; i < s->n_items; ++i) order[i] = i;
    const T *it = static_cast<const T *>(host_items);
    std::stable_sort(order.begin(), order.end(), [it](unsigned a, unsigned b) { return it[4 * a + 3] > it[4 * b + 3]; });
    unsigned *d_order = nullptr;
    HIP_TRY(hipMalloc(&d_order, sizeof(unsigned) * s->n_items));
    struct Free { unsigned *p; ~Free() { (void)hipFree(p); } } free_order{ d_order };
    HIP_TRY(hipMemcpy(d_order, order.data(), sizeof(unsigned) * s->n_items, hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&s->d_fprim, sizeof(rt::Quad<T>) * s->n_padded));
    HIP_TRY(hipMalloc(&s->d_fprim_rr, sizeof(T) * s->n_padded));
    HIP_TRY(hipMalloc(&s->d_fshad, sizeof(rt::Quad<T>) * s->n_padded));
    const rt::V3<T> eye = { (T)s->eye[0], (T)s->eye[1], (T)s->eye[2] };
    hipLaunchKernelGGL((rt::k_build_flat<T>), dim3((s->n_padded + 255) / 256), dim3(256), 0, nullptr,
                       static_cast<const rt::Item<T> *>(s->d_items), d_order, s->n_items, s->n_padded, eye, static_cast<rt::Quad<T> *>(s->d_fprim),
                       static_cast<T *>(s->d_fprim_rr), static_cast<rt::Quad<T> *>(s->d_fshad));
    HIP_TRY(hipGetLastError());
    if constexpr (sizeof(T) == 4) {
        const uint32_t n4 = (s->n_items + rt::kFlatFilterItems - 1) / rt::kFlatFilterItems, fpairs = (n4 + 1) / 2;
        const uint32_t n3 = (s->n_items + rt::kFlatShadowItems - 1) / rt::kFlatShadowItems, spairs = (n3 + 1) / 2;
        const uint32_t n_fgroups = 2 * fpairs + rt::kFlatPadGroups;          // pad groups: never hit; the scans load one pair ahead
        const uint32_t n_sgroups = 2 * spairs + rt::kFlatPadGroups;
        s->flat_filter_bytes = fpairs * 128u;
        s->flat_shadow_bytes = spairs * 128u;
        // the shadow filter takes centres and origins relative to a point inside the scene: the centroid of the item centres
        double m0[3] = { 0, 0, 0 };
        for (unsigned i = 0; i < s->n_items; ++i)
            for (int k = 0; k < 3; ++k) m0[k] += (double)it[4 * i + k];
        for (int k = 0; k < 3; ++k) s->flat_centre[k] = (float)(m0[k] / (double)s->n_items);
        HIP_TRY(hipMalloc(&s->d_pf, sizeof(rt::FGroup) * n_fgroups));
        HIP_TRY(hipMalloc(&s->d_pe, sizeof(rt::FExact) * s->n_items));
        HIP_TRY(hipMalloc(&s->d_sg, sizeof(rt::FGroup) * n_sgroups));
        HIP_TRY(hipMalloc(&s->d_se, sizeof(rt::FExactShadow) * s->n_items));
        const uint32_t n_threads = std::max(n_fgroups * rt::kFlatFilterItems, n_sgroups * rt::kFlatShadowItems);
        hipLaunchKernelGGL(rt::k_build_flat_groups, dim3((n_threads + 255) / 256), dim3(256), 0, nullptr,
                           static_cast<const rt::Item<float> *>(s->d_items), d_order, s->n_items, n_fgroups, n_sgroups,
                           rt::V3<float>{ (float)s->eye[0], (float)s->eye[1], (float)s->eye[2] },
                           rt::V3<float>{ s->flat_centre[0], s->flat_centre[1], s->flat_centre[2] },
                           rt::V3<float>{ -(float)s->light[0], -(float)s->light[1], -(float)s->light[2] }, static_cast<rt::FGroup *>(s->d_pf),
                           static_cast<rt::FExact *>(s->d_pe), static_cast<rt::FGroup *>(s->d_sg), static_cast<rt::FExactShadow *>(s->d_se));
        HIP_TRY(hipGetLastError());
    }
    if constexpr (sizeof(T) == 8) {
        // the filtered f64 scan (rt_flat_f64.hpp): per-item bound terms next to the exact arrays
        double m0[3] = { 0, 0, 0 };
        for (unsigned i = 0; i < s->n_items; ++i)
            for (int k = 0; k < 3; ++k) m0[k] += (double)it[4 * i + k];
        for (int k = 0; k < 3; ++k) s->flat_centre64[k] = m0[k] / (double)s->n_items;
        const size_t n_alloc = (size_t)s->n_padded + rt::kFlatF64Tail;
        HIP_TRY(hipMalloc(&s->d_f64_pf, sizeof(rt::Quad<double>) * n_alloc));
        HIP_TRY(hipMalloc(&s->d_f64_sf, sizeof(rt::Quad<double>) * n_alloc));
        HIP_TRY(hipMalloc(&s->d_f64_sg, sizeof(double) * n_alloc));
        hipLaunchKernelGGL(rt::k_build_flat_f64, dim3((unsigned)((n_alloc + 255) / 256)), dim3(256), 0, nullptr,
                           static_cast<const rt::Item<double> *>(s->d_items), d_order, s->n_items, s->n_padded,
                           rt::V3<double>{ s->eye[0], s->eye[1], s->eye[2] },
                           rt::V3<double>{ s->flat_centre64[0], s->flat_centre64[1], s->flat_centre64[2] },
                           rt::V3<double>{ -s->light[0], -s->light[1], -s->light[2] }, static_cast<rt::Quad<double> *>(s->d_f64_pf),
                           static_cast<rt::Quad<double> *>(s->d_f64_sf), static_cast<double *>(s->d_f64_sg));
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipDeviceSynchronize());
    return RT_OK;
}

rt::FlatF64View flat_f64_view_of(const rt_scene *s)
{
    rt::FlatF64View v;
    v.pf = static_cast<const rt::Quad<double> *>(s->d_f64_pf);
    v.sf = static_cast<const rt::Quad<double> *>(s->d_f64_sf);
    v.sg = static_cast<const double *>(s->d_f64_sg);
    v.centre = { s->flat_centre64[0], s->flat_centre64[1], s->flat_centre64[2] };
    return v;
}

rt::FlatScView flat_sc_view_of(const rt_scene *s)
{
    rt::FlatScView v;
    v.pf = static_cast<const rt::FGroup *>(s->d_pf);
    v.pe = static_cast<const rt::FExact *>(s->d_pe);
    v.sg = static_cast<const rt::FGroup *>(s->d_sg);
    v.se = static_cast<const rt::FExactShadow *>(s->d_se);
    v.n_fbytes = s->flat_filter_bytes;
    v.n_sbytes = s->flat_shadow_bytes;
    v.centre = { s->flat_centre[0], s->flat_centre[1], s->flat_centre[2] };
    v.items = static_cast<const rt::Item<float> *>(s->d_items);
    v.n_items = s->n_items;
    v.light = { (float)s->light[0], (float)s->light[1], (float)s->light[2] };
    v.eye = { (float)s->eye[0], (float)s->eye[1], (float)s->eye[2] };
    return v;
}

template <typename T>
rt::SkipView<T> skip_view_of(const rt_scene *s)
{
    rt::SkipView<T> v;
    v.prim = static_cast<const rt::Node<T> *>(s->d_prim);
    v.shad = static_cast<const rt::Node<T> *>(s->d_shad);
    v.fprim = static_cast<const rt::Node<T> *>(s->d_cprim);
    v.fshad = static_cast<const rt::Node<T> *>(s->d_cshad);
    v.items = static_cast<const rt::Item<T> *>(s->d_items);
    v.n_nodes = s->n_nodes;
    v.n_fnodes = s->n_fnodes;
    v.light = { (T)s->light[0], (T)s->light[1], (T)s->light[2] };
    v.eye = { (T)s->eye[0], (T)s->eye[1], (T)s->eye[2] };
    v.xprim = static_cast<const rt::FNode *>(s->d_xprim);
    v.xshad = static_cast<const rt::FNodeS *>(s->d_xshad);
    v.xfprim = static_cast<const rt::FNode *>(s->d_xcprim);
    v.xfshad = static_cast<const rt::FNodeS *>(s->d_xcshad);
    v.xown = static_cast<const uint32_t *>(s->d_xown);
    v.fc = static_cast<const rt::FilterConsts *>(s->d_fc);
    return v;
}

// The one argument of the render kernels (rt_skip.hpp SkipArgs: what a wave needs first lies first).
template <typename T>
rt::SkipArgs<T> skip_args(const rt_scene *s, const rt::BlockDesc *order, const uint32_t *wg_first, unsigned w, unsigned h, unsigned frame_w, uint8_t *out,
                          const rt::TileDev *tiles, unsigned n_tiles, unsigned spp, rt::Counters *counters, uint32_t *lane_cost, rt::SampleBuf<T> sb,
                          rt::CoopView cv = rt::CoopView{}, const uint64_t *holes = nullptr, unsigned n_holes = 0)
{
    rt::SkipArgs<T> a{};
    a.order = order; a.wg_first = wg_first; a.width = w; a.height = h; a.frame_w = frame_w; a.out = out; a.tiles = tiles; a.n_tiles = n_tiles;
    a.spp_arg = spp; a.sc = skip_view_of<T>(s); a.counters = counters; a.lane_cost = lane_cost; a.holes = holes; a.n_holes = n_holes; a.sb = sb; a.cv = cv;
    return a;
}


// Constants of the filtered loops' shadow bounds (rt_skip.hpp FilterConsts, shadow_filter_bounds; derivation in DESIGN.md 4.1).
// eps = 2^-24, eta = | |l|^2 - 1 | for the f32 shadow direction l, Rc = max |c - m0| over every node centre, Ro = the radius around
// m0 the bounds cover ray origins in (a ray further out gets a NaN: no sure verdict, shadow_filter_origin), S = Rc + Ro (1 + 4 eps);
// a0 = 11 eps S: a = cl - ol >= a0 proves b = dot(centre - origin, l) >= 0 as the reference rounds it.
template <typename T>
void filter_constants(rt_scene *s, const std::vector<rt::RawNode<T>> &raw, const T *items)
{
    double m0[3] = { 0, 0, 0 };
    for (uint32_t i = 0; i < s->n_items; ++i)
        for (int k = 0; k < 3; ++k) m0[k] += (double)items[4 * i + k];
    rt::FilterConsts &fc = s->fc;
    for (int k = 0; k < 3; ++k) { fc.m0[k] = (float)(m0[k] / (double)s->n_items); m0[k] = (double)fc.m0[k]; }
    auto dist = [&](double x, double y, double z) { return std::sqrt((x - m0[0]) * (x - m0[0]) + (y - m0[1]) * (y - m0[1]) + (z - m0[2]) * (z - m0[2])); };
    double rc = 0, rit = 0;
    for (const rt::RawNode<T> &r : raw) rc = std::max(rc, dist((double)r.cx, (double)r.cy, (double)r.cz));
    for (uint32_t i = 0; i < s->n_items; ++i)
        rit = std::max(rit, dist((double)items[4 * i], (double)items[4 * i + 1], (double)items[4 * i + 2]) + (double)items[4 * i + 3]);
    const double eye_d = dist(s->eye[0], s->eye[1], s->eye[2]);
    const double eye_abs = std::fabs(s->eye[0]) + std::fabs(s->eye[1]) + std::fabs(s->eye[2]), m0_abs = std::fabs(m0[0]) + std::fabs(m0[1]) + std::fabs(m0[2]);
    // shadow origins lie on an item's surface, pushed out by hit.distance * sqrt(eps) (render.rs:199): 1 % and a bit of room
    double ro = 1.01 * rit + 1e-3 * (eye_d + rit) + 1e-5 * (eye_abs + m0_abs);
    if (const long long pc = knob(RT_DEBUG_FILTER_RO_PERCENT); pc >= 0) ro *= (double)pc / 100.0;   // tests only
    const double eps = 0x1p-24;
    // plane perpendicular to the shadow direction l = -light (f32 components)
    // plane perpendicular to the shadow direction l = -light: the f32 components an f32 scene's reference uses, the f64 ones for an f64 scene
    const bool f64 = sizeof(T) == 8;
    const double l[3] = { f64 ? -s->light[0] : -(double)(float)s->light[0], f64 ? -s->light[1] : -(double)(float)s->light[1], f64 ? -s->light[2] : -(double)(float)s->light[2] };
    const double l2 = l[0] * l[0] + l[1] * l[1] + l[2] * l[2], ln = std::sqrt(l2);
    const double lh[3] = { l[0] / ln, l[1] / ln, l[2] / ln };
    int ax = 0;
    for (int k = 1; k < 3; ++k) if (std::fabs(lh[k]) < std::fabs(lh[ax])) ax = k;
    double a[3] = { 0, 0, 0 }; a[ax] = 1.0;
    double e1[3] = { lh[1] * a[2] - lh[2] * a[1], lh[2] * a[0] - lh[0] * a[2], lh[0] * a[1] - lh[1] * a[0] };
    const double n1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    for (int k = 0; k < 3; ++k) e1[k] /= n1;
    const double e2[3] = { lh[1] * e1[2] - lh[2] * e1[1], lh[2] * e1[0] - lh[0] * e1[2], lh[0] * e1[1] - lh[1] * e1[0] };
    for (int k = 0; k < 3; ++k) { fc.e1[k] = (float)e1[k]; fc.e2[k] = (float)e2[k]; }
    for (int k = 0; k < 3; ++k) fc.l[k] = (float)l[k];
    fc.eta = std::fabs(l2 - 1.0);
    fc.S = (rc + ro * (1.0 + 4.0 * eps)) * (1.0 + 1e-9);
    auto up = [](double v) { float f = (float)v; if ((double)f < v) f = std::nextafterf(f, INFINITY); return std::nextafterf(f, INFINITY); };
    fc.a0 = up(11.0 * eps * fc.S + 1e-37);
    fc.k1 = up((fc.eta + 10.2 * eps) * (1.0 + fc.eta) * (1.0 + 12.0 * eps));
    { float kc = (float)(1.0 / (1.0 + 4.0 * 0x1p-10)); if ((double)kc > 1.0 / (1.0 + 4.0 * 0x1p-10)) kc = std::nextafterf(kc, 0.0f); fc.kc = std::nextafterf(kc, 0.0f); }
    float ro2 = (float)(ro * ro);
    if ((double)ro2 > ro * ro) ro2 = std::nextafterf(ro2, 0.0f);
    fc.ro2 = ro2;
}

// FNode copies (rt_skip.hpp) of one pair of f32 Node streams, END nodes included.
rt_status derive_fstreams(const rt_scene *s, const void *d_prim, const void *d_shad, size_t n_nodes, bool compacted, void **d_xprim, void **d_xshad,
                          void **d_own)
{
    const size_t total = n_nodes + rt::kNodePad;
    HIP_TRY(hipMalloc(d_xprim, sizeof(rt::FNode) * total));
    HIP_TRY(hipMalloc(d_xshad, sizeof(rt::FNodeS) * total));
    if (d_own) HIP_TRY(hipMalloc(d_own, sizeof(uint32_t) * total));
    hipLaunchKernelGGL(rt::k_build_fstreams, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s->cost_stream, static_cast<const rt::Node<float> *>(d_prim),
                       static_cast<const rt::Node<float> *>(d_shad), (unsigned)total, compacted, s->fc, static_cast<rt::FNode *>(*d_xprim),
                       static_cast<rt::FNodeS *>(*d_xshad), d_own ? static_cast<uint32_t *>(*d_own) : nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->cost_stream));
    return RT_OK;
}

// Merges items and group bounds into the DFS pre-order node stream of rt_skip.hpp.  ranges must form a laminar
// family given in pre-order (outer group before the groups nested in it).  Groups without items are dropped:
// their bound test cannot change any hit.
template <typename T>
rt_status build_raw_stream(const T *items, uint32_t n_items, const T *bounds, const rt_range *ranges, uint32_t n_bounds,
                           std::vector<rt::RawNode<T>> &out)
{
    out.clear();
    out.reserve((size_t)n_items + n_bounds);
    struct Open { uint32_t node; uint32_t end; };
    std::vector<Open> stack;
    uint32_t b = 0;
    for (uint32_t pos = 0; pos <= n_items; ++pos) {
        while (!stack.empty() && stack.back().end == pos) {           // subtree complete: its skip target is here
            out[stack.back().node].skip = (uint32_t)out.size();
            stack.pop_back();
        }
        if (pos == n_items) break;
        while (b < n_bounds && (uint32_t)ranges[b].first == pos) {
            const uint32_t end = pos + (uint32_t)ranges[b].count;
            if (!stack.empty() && end > stack.back().end) {
                snprintf(g_err, sizeof g_err, "rt_scene_create: range %u is not nested inside its enclosing group", b);
                return RT_ERR_INVALID_ARGUMENT;
            }
            if (ranges[b].count > 0) {
                stack.push_back({ (uint32_t)out.size(), end });
                out.push_back({ bounds[4 * b], bounds[4 * b + 1], bounds[4 * b + 2], bounds[4 * b + 3], 0u, 0u, T(0), 0u, 0u });
            }
            ++b;
        }
        if (b < n_bounds && (uint32_t)ranges[b].first < pos) {
            snprintf(g_err, sizeof g_err, "rt_scene_create: ranges are not in DFS pre-order at %u", b);
            return RT_ERR_INVALID_ARGUMENT;
        }
        out.push_back({ items[4 * pos], items[4 * pos + 1], items[4 * pos + 2], items[4 * pos + 3], 0u, pos, T(0), 0u, 0u });
    }
    if (b != n_bounds || !stack.empty()) {
        snprintf(g_err, sizeof g_err, "rt_scene_create: ranges are not a DFS pre-order nesting of the item array");
        return RT_ERR_INVALID_ARGUMENT;
    }
    // a BOUND is marked by skip != 0; skip targets are > the node's own index >= 0, so they are never 0
    return RT_OK;
}

// Device streams (primary + shadow) of one raw stream: nodes [0, n) and kNodePad END nodes behind them.
template <typename T>
rt_status derive_streams(const rt_scene *s, const std::vector<rt::RawNode<T>> &raw, bool compacted, void **d_prim, void **d_shad)
{
    const size_t n = raw.size(), total = n + rt::kNodePad;
    rt::RawNode<T> *d_raw = nullptr;
    HIP_TRY(hipMalloc(&d_raw, sizeof(rt::RawNode<T>) * n));
    hipError_t e = scene_upload(const_cast<rt_scene *>(s), d_raw, raw.data(), sizeof(rt::RawNode<T>) * n) == RT_OK ? hipSuccess : hipErrorUnknown;      // (`raw` outlives the synchronise below)
    if (e == hipSuccess) e = hipMalloc(d_prim, sizeof(rt::Node<T>) * total);
    if (e == hipSuccess) e = hipMalloc(d_shad, sizeof(rt::Node<T>) * total);
    if (e == hipSuccess) {
        const rt::V3<T> eye = { (T)s->eye[0], (T)s->eye[1], (T)s->eye[2] };
        hipLaunchKernelGGL((rt::k_build_streams<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s->cost_stream, d_raw, (unsigned)n, eye, compacted,
                           static_cast<rt::Node<T> *>(*d_prim), static_cast<rt::Node<T> *>(*d_shad));
        e = hipGetLastError();
    }
    const hipError_t se = hipStreamSynchronize(s->cost_stream);
    if (e == hipSuccess) e = se;
    (void)hipFree(d_raw);
    if (e != hipSuccess) return hip_fail(e, "derive_streams", __LINE__);
    return RT_OK;
}

rt_status upload_coop(rt_scene *s, const std::vector<rt::RawNode<float>> &raw);

template <typename T>
rt_status upload_streams(rt_scene *s, const void *items, const void *bounds, const rt_range *ranges)
{
    std::vector<rt::RawNode<T>> raw;
    rt_status st = build_raw_stream<T>(static_cast<const T *>(items), s->n_items, static_cast<const T *>(bounds), ranges, s->n_bounds, raw);
    if (st != RT_OK) return st;
    // the traversal loops address the streams with 32-bit byte offsets (END nodes included) and keep two flag bits in the item word
    if (((uint64_t)raw.size() + rt::kNodePad) * sizeof(rt::Node<T>) > 0xFFFFFFFFull || s->n_items > rt::kNodeIndexMask) {
        snprintf(g_err, sizeof g_err, "rt_scene_create: %zu stream nodes exceed what the traversal streams can address", raw.size());
        return RT_ERR_UNSUPPORTED;
    }
    s->n_nodes = (uint32_t)raw.size();
    // fused: every BOUND directly followed by an ITEM with the same centre, bit for bit (then the values v, b, b*b - vv a
    // ray forms for the two are the same bits).
    bool fused = !raw.empty();
    for (size_t i = 0; fused && i < raw.size(); ++i)
        if (raw[i].skip != 0u)
            fused = i + 1 < raw.size() && raw[i + 1].skip == 0u && memcmp(&raw[i].cx, &raw[i + 1].cx, 3 * sizeof(T)) == 0;
    s->fused = fused;
    StageClock clk;
    if ((st = derive_streams<T>(s, raw, false, &s->d_prim, &s->d_shad)) != RT_OK) return st;
    clk.lap("  streams: plain (first kernel)");
    if (fused) {
        // compacted streams: the ITEM behind every BOUND moves into the BOUND node (it is never a jump target: `skip` points
        // behind a whole subtree, and a subtree never starts with its group's own sphere)
        std::vector<uint32_t> new_index(raw.size() + 1);
        uint32_t k = 0;
        for (size_t i = 0; i < raw.size(); ++i) {
            new_index[i] = k;
            if (!(i > 0 && raw[i - 1].skip != 0u)) ++k;              // dropped: the node directly behind a BOUND
        }
        new_index[raw.size()] = k;
        std::vector<rt::RawNode<T>> compact;
        compact.reserve(k);
        for (size_t i = 0; i < raw.size(); ++i) {
            if (i > 0 && raw[i - 1].skip != 0u) continue;
            rt::RawNode<T> r = raw[i];
            if (r.skip != 0u) {
                r.own_r = raw[i + 1].r; r.own_item = raw[i + 1].item;
                r.skip = new_index[r.skip];
                if (r.skip == 0u) { snprintf(g_err, sizeof g_err, "rt_scene_create: internal: skip target 0"); return RT_ERR_INVALID_ARGUMENT; }
            }
            compact.push_back(r);
        }
        s->n_fnodes = (uint32_t)compact.size();
        if ((st = derive_streams<T>(s, compact, true, &s->d_cprim, &s->d_cshad)) != RT_OK) return st;
        clk.lap("  streams: compacted");
    }
    if constexpr (sizeof(T) == 8) {
        // f64: FNode copies of the primary streams for the filtered primary walk (rt_skip.hpp k_build_fstream64)
        filter_constants<T>(s, raw, static_cast<const T *>(items));
        HIP_TRY(hipMalloc(&s->d_fc, sizeof(rt::FilterConsts)));
        { rt_status ust = scene_upload(s, s->d_fc, &s->fc, sizeof(rt::FilterConsts)); if (ust != RT_OK) return ust; }
        auto derive64 = [&](const void *d_prim, const void *d_shad, size_t n_nodes, bool compacted, void **d_x, void **d_xs, void **d_own) -> rt_status {
            const size_t total = n_nodes + rt::kNodePad;
            HIP_TRY(hipMalloc(d_x, sizeof(rt::FNode) * total));
            HIP_TRY(hipMalloc(d_xs, sizeof(rt::FNodeS) * total));
            if (d_own) HIP_TRY(hipMalloc(d_own, sizeof(uint32_t) * total));
            hipLaunchKernelGGL(rt::k_build_fstream64, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s->cost_stream, static_cast<const rt::Node<double> *>(d_prim),
                               static_cast<const rt::Node<double> *>(d_shad), (unsigned)total, compacted, s->fc, static_cast<rt::FNode *>(*d_x),
                               static_cast<rt::FNodeS *>(*d_xs), d_own ? static_cast<uint32_t *>(*d_own) : nullptr);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(s->cost_stream));
            return RT_OK;
        };
        if ((uint64_t)(s->n_nodes + rt::kNodePad) * sizeof(rt::Node<T>) <= 0xFFFFFFFFull) {
            if ((st = derive64(s->d_prim, s->d_shad, s->n_nodes, false, &s->d_xprim, &s->d_xshad, nullptr)) != RT_OK) return st;
            if (fused && (st = derive64(s->d_cprim, s->d_cshad, s->n_fnodes, true, &s->d_xcprim, &s->d_xcshad, &s->d_xown)) != RT_OK) return st;
        }
    }
    if constexpr (sizeof(T) == 4) {
        filter_constants<T>(s, raw, static_cast<const T *>(items));
        HIP_TRY(hipMalloc(&s->d_fc, sizeof(rt::FilterConsts)));
        { rt_status ust = scene_upload(s, s->d_fc, &s->fc, sizeof(rt::FilterConsts)); if (ust != RT_OK) return ust; }
        clk.lap("  streams: filter constants");
        if ((st = derive_fstreams(s, s->d_prim, s->d_shad, s->n_nodes, false, &s->d_xprim, &s->d_xshad, nullptr)) != RT_OK) return st;
        if (fused && (st = derive_fstreams(s, s->d_cprim, s->d_cshad, s->n_fnodes, true, &s->d_xcprim, &s->d_xcshad, &s->d_xown)) != RT_OK) return st;
        clk.lap("  streams: filtered");
        if ((st = upload_coop(s, raw)) != RT_OK) return st;
        clk.lap("  streams: cooperative copy");
    }
    return RT_OK;
}

// The lane-cooperative walk's copy of the hierarchy (rt_coop.hpp): the nodes of the plain stream in breadth-first order, so that the
// children of a group are consecutive records.  Scenes whose largest child count (or number of top-level nodes) exceeds what a
// work-list word holds simply get none: the cooperative walk is an optimisation of the skip-pointer walk, never a requirement.
rt_status upload_coop(rt_scene *s, const std::vector<rt::RawNode<float>> &raw)
{
    const uint32_t n = (uint32_t)raw.size();
    if (n == 0 || n >= rt::kCoopMaxNodes) return RT_OK;
    auto next_sibling = [&](uint32_t i) { return raw[i].skip ? raw[i].skip : i + 1u; };
    std::vector<uint32_t> perm;                       // breadth-first position -> stream index
    perm.reserve(n);
    for (uint32_t i = 0; i < n; i = next_sibling(i)) perm.push_back(i);
    const uint32_t n_roots = (uint32_t)perm.size();
    uint32_t fanout = n_roots;
    std::vector<uint2> link(n);
    for (uint32_t j = 0; j < perm.size(); ++j) {
        const uint32_t i = perm[j];
        if (raw[i].skip == 0u) { link[j] = make_uint2(raw[i].item, 0u); continue; }
        const uint32_t first = (uint32_t)perm.size();
        for (uint32_t c = i + 1u; c < raw[i].skip; c = next_sibling(c)) perm.push_back(c);
        const uint32_t count = (uint32_t)perm.size() - first;
        if (count == 0u) return RT_OK;                // cannot happen (groups without items are dropped); no copy rather than a wrong one
        link[j] = make_uint2(first, count);
        fanout = std::max(fanout, count);
    }
    if (perm.size() != n || fanout > rt::kCoopMaxFanout) return RT_OK;
    uint32_t *d_perm = nullptr; uint2 *d_link = nullptr;
    hipError_t e = hipMalloc(&d_perm, sizeof(uint32_t) * n);
    if (e == hipSuccess) e = hipMalloc(&d_link, sizeof(uint2) * n);
    if (e == hipSuccess && scene_upload(s, d_perm, perm.data(), sizeof(uint32_t) * n) != RT_OK) e = hipErrorUnknown;
    if (e == hipSuccess && scene_upload(s, d_link, link.data(), sizeof(uint2) * n) != RT_OK) e = hipErrorUnknown;
    if (e == hipSuccess) e = hipMalloc(&s->d_coop_prim, sizeof(rt::CNode) * n);
    if (e == hipSuccess) e = hipMalloc(&s->d_coop_shad, sizeof(rt::CNode) * n);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rt::k_build_coop, dim3((n + 255) / 256), dim3(256), 0, s->cost_stream, s->d_prim, s->d_shad, (unsigned)sizeof(rt::Node<float>), d_perm, d_link, n,
                           static_cast<rt::CNode *>(s->d_coop_prim), static_cast<rt::CNode *>(s->d_coop_shad));
        e = hipGetLastError();
    }
    { const hipError_t se = hipStreamSynchronize(s->cost_stream); if (e == hipSuccess) e = se; }
    if (d_perm) (void)hipFree(d_perm);
    if (d_link) (void)hipFree(d_link);
    if (e != hipSuccess) return hip_fail(e, "upload_coop", __LINE__);
    s->coop.prim = static_cast<const rt::CNode *>(s->d_coop_prim);
    s->coop.shad = static_cast<const rt::CNode *>(s->d_coop_shad);
    s->coop.n_roots = n_roots;
    s->coop.fanout = fanout;
    return RT_OK;
}

// spp > 1 runs sample-parallel (one thread per sample + a resolve pass) unless spp*spp exceeds grid.y's limit.
bool use_split(unsigned spp) { return spp > 1 && (unsigned long long)spp * spp <= 65535ull; }
// spp 2 / 4 / 8: the samples of a pixel fill 4 / 16 / 64 lanes of a wave (rt_skip.hpp, kSkipPacked)
bool packed_samples(unsigned spp)
{
    return (spp == 2 || spp == 4 || spp == 8) && knob(RT_DEBUG_PACKED_SAMPLES) != 0;
}

// Two rays per lane (rt_skip2.hpp) unless csrc/rt_debug.h RT_DEBUG_SKIP_RAYS says otherwise.  A wave of 128 rays walks the union of more
// paths and a frame is as long as its heaviest waves, so the second ray pays once there is enough work to be throughput-bound.
// Measured with both kernels' walks behind their conservative bounds (round 3: tools/skip2_sweep.sh, profiles/r03d_skip2_sweep.log;
// one / two rays per lane, us per launch).  21,845 spheres, spp 1: 1920x1080 45.5 / 57.5, 2304x1296 65.6 / 61.7, 2560x1440 77.8 / 71.1,
// 3840x2160 153.9 / 131.6; sample-packed: 1024x768 spp 2 108.5 / 121.2, 640x480 spp 4 147.8 / 152.6, 800x600 spp 4 202.0 / 192.5,
// 1024x768 spp 4 (`make image`) 263.6 / 244.0, 2048x2048 spp 4 952 / 828.  87,381 spheres, spp 1: 2560x1440 92.4 / 115.8, 3200x1800
// 138.5 / 131.7, 3840x2160 181.5 / 153.8; sample-packed: 1280x720 spp 2 163.3 / 175.1, 640x480 spp 4 187.9 / 194.6, 800x600 spp 4
// 247.7 / 242.5, 1920x1080 spp 4 724 / 625, 4096x4096 spp 4 4206 / 3420.  (Round 2, before the bounds: spp 1 from 3.5 M / 6 M pixels,
// sample-packed modes only on the large scene.)
// End of round 4, both kernels at eight waves per SIMD (the one-ray kernel gained more from its eighth than the two-ray kernel: it was the
// one waiting more).  21,845 spheres, spp 1: 1920x1080 41.4 / 58.4, 2560x1440 69.3 / 70.2, 3200x1800 99.7 / 96.9, 3840x2160 133.4 / 127.6;
// sample-packed: 1920x1080 spp 2 158.6 / 165.8, 640x480 spp 4 125.5 / 140.8, 800x600 spp 4 170.9 / 177.9, 1024x768 spp 4 206.7 / 210.9,
// 1920x1080 spp 4 475 / 465, 2048x2048 spp 4 732 / 700.  87,381 spheres, spp 1: 2560x1440 95.1 / 117.7, 3840x2160 163.3 / 146.4;
// sample-packed: 1024x768 spp 4 271.5 / 257.9, 1920x1080 spp 4 591 / 519, 4096x4096 spp 4 3437 / 2841.
bool skip2_by_default(uint64_t total_px, unsigned spp, uint32_t n_nodes)
{
    const bool large_scene = n_nodes >= 65536u;
    if (spp == 1) return total_px >= (large_scene ? 5000000ull : 4000000ull);
    return total_px * spp * spp >= (large_scene ? 6000000ull : 20000000ull);
}

constexpr size_t kMaxCachedTables = 32;

// Device copy of `tab`: from the scene's cache when seen before (or cacheable now), else through the context.
rt_status device_table(rt_scene *s, Context *c, const std::vector<rt::TileDev> &tab, hipStream_t stream, const rt::TileDev **out,
                       int slot = 0, const rt_options *o = nullptr, rt::BlockList *order_out = nullptr, bool cacheable = true, bool will_be_timed = false);

// Copies the tile table through the context's pinned buffer; truly asynchronous on `stream`.  A table of one or two tiles
// is read by the kernel straight from the pinned copy instead (one PCIe read per workgroup beats a copy operation on the stream).
constexpr size_t kZeroCopyTableTiles = 2;

rt_status upload_tiles(Context *c, const std::vector<rt::TileDev> &tab, hipStream_t stream, int slot, const rt::TileDev **out)
{
    const size_t tab_bytes = tab.size() * sizeof(rt::TileDev);
    // A second table through the same slot within one lease (the batches of rt_render_tiles_stream / rt_render_frame_stream once the scene's
    // table cache is full): the copy queued for the previous batch may not have read the pinned staging yet, and that batch's kernels --
    // on either of the context's streams -- may still be reading the device copy.  Rare and slow on purpose: wait for all of it.
    if (c->tiles_live[slot]) {
        HIP_TRY(hipStreamSynchronize(stream));
        if (c->stream && c->stream != stream) HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->stream2 && c->stream2 != stream) HIP_TRY(hipStreamSynchronize(c->stream2));
    }
    c->tiles_live[slot] = true;
    if (c->tiles_cap[slot] < tab.size()) {
        if (c->d_tiles[slot]) HIP_TRY(hipFree(c->d_tiles[slot]));
        if (c->h_tiles[slot]) HIP_TRY(hipHostFree(c->h_tiles[slot]));
        c->d_tiles[slot] = nullptr; c->h_tiles[slot] = nullptr; c->tiles_cap[slot] = 0;
        HIP_TRY(hipMalloc(&c->d_tiles[slot], tab_bytes));
        HIP_TRY(hipHostMalloc(&c->h_tiles[slot], tab_bytes, hipHostMallocDefault));
        c->tiles_cap[slot] = tab.size();
    }
    memcpy(c->h_tiles[slot], tab.data(), tab_bytes);
    if (tab.size() <= kZeroCopyTableTiles) {
        void *alias = nullptr;
        if (hipHostGetDevicePointer(&alias, c->h_tiles[slot], 0) == hipSuccess) { *out = static_cast<const rt::TileDev *>(alias); return RT_OK; }
        (void)hipGetLastError();
    }
    HIP_TRY(hipMemcpyAsync(c->d_tiles[slot], c->h_tiles[slot], tab_bytes, hipMemcpyHostToDevice, stream));
    *out = c->d_tiles[slot];
    return RT_OK;
}

constexpr unsigned kCostRes = 256;
// the cost arena's tile-table area: one tile for the map itself; up to kExactBlocks 16x16 blocks when a tile list's heaviest blocks are
// counted again at the frame's own resolution (exact_block_costs)
constexpr unsigned kExactBlocks = kCostRes * kCostRes / (rt::kBlockW * rt::kBlockH);       // what the arena's pixel areas hold: 256
constexpr size_t kCostTileBytes = (kExactBlocks * sizeof(rt::TileDev) + 255) & ~(size_t)255;
constexpr size_t kTableStageBytes = 256 * 1024;       // pinned staging for the tile tables of new lists (a 1080p list of 64x64 buckets: 10 KB)

// The scene's cost map: one counting render of a kCostRes^2 image (same camera: x spans the same field of view at every
// width), each lane storing the number of tests its pixel took.
// The pinned host side of the cost map: [map: kCostRes^2 words | tile-table area | staging for the tile tables of new lists].  Made by
// rt_scene_create (0.15 ms); the device side and the counting render wait until a tile list wants dispatch orders (start_cost_map): a
// process that renders ONE frame (`make image`) never pays for them.
rt_status alloc_cost_host(rt_scene *s)
{
    constexpr size_t kPx = (size_t)kCostRes * kCostRes * 4;
    HIP_TRY(hipHostMalloc(&s->h_cost, kPx + kCostTileBytes + kTableStageBytes, hipHostMallocDefault));
    s->h_tab_stage = static_cast<char *>(s->h_cost) + kPx + kCostTileBytes;
    return RT_OK;
}

// Enqueues the counting render of the cost map on the scene's own stream (by whoever first asks for the map: cost_map_of, normally the
// scene's worker thread).  No copy engine: the one tile is read from pinned memory, the lanes store their counts into the pinned map.
template <typename T>
rt_status start_cost_map(rt_scene *s)
{
    constexpr unsigned R = kCostRes;
    const rt::TileDev tile{ 0, (uint16_t)R, (uint16_t)R, 0, 0u, 0u, R / rt::kBlockW };
    if (!s->h_cost) return RT_ERR_OUT_OF_MEMORY;
    // ONE device allocation, kept until the scene goes (hipMalloc / hipFree wait for a busy device): tile | frame | costs | counters
    constexpr size_t kTileBytes = kCostTileBytes, kPx = (size_t)R * R * 4, kCnt = sizeof(rt::Counters) * rt::kCounterStripes;
    HIP_TRY(hipMalloc(&s->d_cost_arena, kTileBytes + 2 * kPx + kCnt));
    hipStream_t stream = s->cost_stream;
    char *base = static_cast<char *>(s->d_cost_arena);
    uint8_t *d_out = reinterpret_cast<uint8_t *>(base + kTileBytes);
    rt::Counters *d_cnt = reinterpret_cast<rt::Counters *>(base + kTileBytes + 2 * kPx);
    memcpy(static_cast<char *>(s->h_cost) + kPx, &tile, sizeof tile);
    void *h_alias = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&h_alias, s->h_cost, 0));
    const rt::TileDev *tile_alias = reinterpret_cast<const rt::TileDev *>(static_cast<char *>(h_alias) + kPx);
    hipLaunchKernelGGL(rt::k_zero_words, dim3(64), dim3(256), 0, stream, reinterpret_cast<uint32_t *>(d_cnt), kCnt / 4);
    rt::SampleBuf<T> sb{ nullptr, nullptr, R * R };
    hipLaunchKernelGGL((rt::k_render_skip<T, true, 1, rt::kSkipLoop>), dim3((R / rt::kBlockW) * (R / rt::kBlockH)), dim3(rt::kBlockThreads), 0, stream,
                       skip_args<T>(s, nullptr, nullptr, R, R, 0u, d_out, tile_alias, 1u, 1u, d_cnt, static_cast<uint32_t *>(h_alias), sb));
    HIP_TRY(hipGetLastError());
    s->cost_started = true;
    return RT_OK;
}

// NULL when the scene has no hierarchy (or the map could not be made: ordering is an optimisation, never an error).
const std::vector<uint32_t> *cost_map_of(rt_scene *s)
{
    std::call_once(s->cost_once, [s] {
        if (s->n_nodes == 0) return;
        if ((s->precision == RT_F32 ? start_cost_map<float>(s) : start_cost_map<double>(s)) != RT_OK) { s->cost_started = false; (void)hipGetLastError(); return; }
        if (hipStreamSynchronize(s->cost_stream) != hipSuccess) { (void)hipGetLastError(); return; }
        const uint32_t *h = static_cast<const uint32_t *>(s->h_cost);
        s->cost_map.assign(h, h + (size_t)kCostRes * kCostRes);
    });
    return s->cost_map.empty() ? nullptr : &s->cost_map;
}

// Tests per pixel at the FRAME's resolution for a few blocks of a tile list (round 6).  The scene's cost map has one cell per 7.5 pixels of
// a 1080p frame, and the rays that meet several hundred nodes follow silhouettes thinner than that: a threshold on the map picks some of a
// heavy pixel's neighbours and misses the pixel, and the wave that keeps it is as long as ever (tools/wave_timeline.py, the cooperative walk
// at 1080p: the quads walked in 14 us, the frame's longest wave still 41).  So the blocks the map ranks highest are counted again, exactly:
// one counting launch over those blocks alone (<= 256 blocks = 65,536 pixels, ~30 us of device time, once per tile list, on the scene's own
// stream), each lane storing the number of tests its pixel took.  px[i * 256 + (y - y0) * 16 + (x - x0)] for block i of `blocks`.
struct ExactCosts { std::vector<uint32_t> block; std::vector<uint32_t> px; uint32_t top = 0; };      // block: raster index of the counted blocks
template <typename T>
rt_status exact_block_costs(rt_scene *s, const std::vector<rt::BlockDesc> &raster, const std::vector<uint32_t> &blocks, unsigned w, unsigned h, ExactCosts &out)
{
    out = ExactCosts{};
    if (blocks.empty() || blocks.size() > kExactBlocks || !s->d_cost_arena || !s->h_cost) return RT_OK;
    std::lock_guard<std::mutex> lk(s->exact_mu);                 // one counting launch at a time through the scene's arena
    constexpr size_t kPx = (size_t)kCostRes * kCostRes * 4, kCnt = sizeof(rt::Counters) * rt::kCounterStripes;
    char *base = static_cast<char *>(s->d_cost_arena);
    uint8_t *d_out = reinterpret_cast<uint8_t *>(base + kCostTileBytes);
    rt::Counters *d_cnt = reinterpret_cast<rt::Counters *>(base + kCostTileBytes + 2 * kPx);
    // (like the map itself: the tile table is read from the pinned arena, the counts are stored into it -- the map was copied out of it
    // when it was collected)
    void *h_alias = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&h_alias, s->h_cost, 0));
    rt::TileDev *h_tiles = reinterpret_cast<rt::TileDev *>(static_cast<char *>(s->h_cost) + kPx);
    const rt::TileDev *tile_alias = reinterpret_cast<const rt::TileDev *>(static_cast<char *>(h_alias) + kPx);
    uint32_t *h_counts = static_cast<uint32_t *>(s->h_cost);
    std::vector<rt::TileDev> tiles(blocks.size());
    for (size_t i = 0; i < blocks.size(); ++i) {
        const rt::BlockDesc &d = raster[blocks[i]];
        // a 16 x 16 tile of its own, clipped like the block; 256 pixels of the tile-major output each
        tiles[i] = rt::TileDev{ d.x0, (uint16_t)std::min<unsigned>(d.y0 + rt::kBlockH, d.t), (uint16_t)std::min<unsigned>(d.x0 + rt::kBlockW, d.r), d.y0,
                                (uint32_t)(i * rt::kBlockW * rt::kBlockH), (uint32_t)i, 1u };
    }
    hipStream_t stream = s->cost_stream;
    memcpy(h_tiles, tiles.data(), tiles.size() * sizeof(rt::TileDev));
    memset(h_counts, 0, blocks.size() * rt::kBlockW * rt::kBlockH * 4);           // (pixels outside a clipped block are not stored)
    hipLaunchKernelGGL(rt::k_zero_words, dim3(64), dim3(256), 0, stream, reinterpret_cast<uint32_t *>(d_cnt), kCnt / 4);
    rt::SampleBuf<T> sb{ nullptr, nullptr, (unsigned)(blocks.size() * rt::kBlockW * rt::kBlockH) };
    hipLaunchKernelGGL((rt::k_render_skip<T, true, 1, rt::kSkipLoop>), dim3((unsigned)blocks.size()), dim3(rt::kBlockThreads), 0, stream,
                       skip_args<T>(s, nullptr, nullptr, w, h, 0u, d_out, tile_alias, (unsigned)tiles.size(), 1u, d_cnt, static_cast<uint32_t *>(h_alias), sb));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    const std::vector<uint32_t> raw(h_counts, h_counts + blocks.size() * rt::kBlockW * rt::kBlockH);
    // the kernel stores tile-major with the tile's own pitch (its clipped width): re-pitch to 16
    out.block = blocks;
    out.px.assign(raw.size(), 0u);
    for (size_t i = 0; i < blocks.size(); ++i) {
        const unsigned tw = (unsigned)tiles[i].r - tiles[i].l, th = (unsigned)tiles[i].t - tiles[i].b;
        for (unsigned y = 0; y < th; ++y)
            for (unsigned x = 0; x < tw; ++x) {
                const uint32_t v = raw[i * 256 + (size_t)y * tw + x];
                out.px[i * 256 + y * 16 + x] = v;
                out.top = std::max(out.top, v);
            }
    }
    return RT_OK;
}

// Dispatch order of a pass's 16x16 blocks: descending estimated cost (the largest cost-map value under the block),
// ties in grid order.  The frame is as long as its last wave's chain of dependent node steps and the chains differ by
// more than 10x across the image, so the long ones have to start first (measured at 1080p: 141 -> 115 us for the
// same one-block tiles in raster vs. descending order).
constexpr uint64_t kFixedBlockCost = 8;      // what a block costs besides its tests (ray set-up, store), in units of the cost map
constexpr size_t kNarrowMax = 64;
constexpr uint64_t kNarrowPercent = 60;
constexpr size_t kNarrowPassBlocks = 16384;
constexpr size_t kNarrowLevel2Blocks = 4096;
// the cooperative walk (rt_coop.hpp): blocks whose estimate reaches kCoopPercent of the pass's largest (and kCoopMinCost tests), at most
// 1 / kCoopMaxShare of the pass
constexpr uint64_t kCoopPercent = 40, kCoopMinCost = 96;
constexpr size_t kCoopMaxShare = 8, kCoopPassBlocks = 4096;
constexpr unsigned kCoopLevel = 2;
constexpr unsigned kCoopRestLevel = 0;      // what is left of a block with holes: 0 = one descriptor (8x8 pixels per wave), 1 = four (4x4 per wave)

// `passes`: how many times the render kernel walks the list in one launch (one per sample in the sample-parallel path).
// Workgroups a launch keeps resident at once: 8 waves per SIMD, 4 waves per workgroup, 256 CUs.
constexpr size_t kResidentWorkgroups = 2048;

// coop (optional): the scene's cooperative copy; holes (with coop): one 64-bit word per descriptor [0, holes->size()) of the list -- the
// 2x2-pixel quads of that 16x16 block (bit (y >> 1) * 8 + (x >> 1)) which cooperative descriptors further down the list render instead.
void block_order(const std::vector<uint32_t> *map, const std::vector<rt::TileDev> &tab, unsigned w, unsigned h, unsigned passes,
                 std::vector<rt::BlockDesc> &descs, std::vector<uint32_t> &wg_first, const rt::CoopView *coop = nullptr, std::vector<uint64_t> *holes = nullptr,
                 int coop_percent = -1,            // 0: no cooperative quads; > 0: from that share of the pass's largest estimate; -1: as rt_debug.h says (default share)
                 const ExactCosts *exact = nullptr, uint32_t exact_thr = 0,      // cooperative quads by EXACT tests per pixel (exact_block_costs) from exact_thr on, instead
                 std::vector<uint32_t> *heaviest = nullptr)                      // out: the raster indices of the heaviest blocks (what exact_block_costs is asked for); descs is not made
{
    wg_first.clear();
    if (holes) holes->clear();
    constexpr int R = (int)kCostRes;
    std::vector<uint32_t> cost;
    std::vector<rt::BlockDesc> raster;
    auto map_col = [&](unsigned x) { return std::clamp((int)((uint64_t)x * R / w), 0, R - 1); };
    auto map_row = [&](unsigned y) { return std::clamp((int)std::floor(((double)y - h / 2.0) * R / w + R / 2.0), 0, R - 1); };
    // the largest map value under the pixels [x0, x1] x [y0, y1], grown by `grow` cells on every side
    auto map_max = [&](unsigned x0, unsigned y0, unsigned x1, unsigned y1, int grow) {
        uint32_t m = 0;
        for (int Y = std::max(0, map_row(y0) - grow); Y <= std::min(R - 1, map_row(y1) + grow); ++Y)
            for (int X = std::max(0, map_col(x0) - grow); X <= std::min(R - 1, map_col(x1) + grow); ++X) m = std::max(m, (*map)[(size_t)Y * R + X]);
        return m;
    };
    for (const rt::TileDev &t : tab) {
        const unsigned bys = ((unsigned)(t.t - t.b) + rt::kBlockH - 1) / rt::kBlockH;
        const uint32_t pitch = (uint32_t)t.r - t.l;
        for (unsigned by = 0; by < bys; ++by)
            for (unsigned bx = 0; bx < t.blks_x; ++bx) {
                const unsigned x0 = t.l + bx * rt::kBlockW, y0 = t.b + by * rt::kBlockH;
                raster.push_back(rt::BlockDesc{ (uint16_t)x0, (uint16_t)y0, t.r, t.t, pitch, t.out_px - t.b * pitch - t.l });
                uint32_t m = 0;
                if (map) m = map_max(x0, y0, std::min<unsigned>(x0 + rt::kBlockW, t.r) - 1, std::min<unsigned>(y0 + rt::kBlockH, t.t) - 1, 0);
                cost.push_back(m);
            }
    }
    std::vector<uint32_t> order(cost.size());
    for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&cost](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
    if (heaviest) {
        heaviest->assign(order.begin(), order.begin() + std::min<size_t>(order.size(), kExactBlocks));
        while (!heaviest->empty() && cost[heaviest->back()] == 0) heaviest->pop_back();        // (nothing to count where the map sees nothing)
        return;
    }
    // The most expensive blocks go out as four narrow workgroups each (rt_kernels.hpp, kBlockNarrow): those whose cost
    // estimate is at least kNarrowPercent of the pass's maximum, at most kNarrowMax and 1/128 of the pass (rt_debug.h can
    // override the cap for A/B runs -- the table is built once per tile list, when it is first seen).
    size_t n_narrow = 0;
    if (knob(RT_DEBUG_PRINT_COSTS) > 0 && !cost.empty()) {
        fprintf(stderr, "[rtrace_hip] block costs, descending:");
        for (size_t i = 0; i < order.size(); i = i < 64 ? i + 4 : i * 2) fprintf(stderr, " #%zu=%u", i, cost[order[i]]);
        fprintf(stderr, "\n");
    }
    // The quads whose rays meet the most nodes are walked COOPERATIVELY (rt_coop.hpp; single-pass f32 launches of scenes that have the
    // cooperative copy, blocks not dealt to workgroups on the host).  A block that holds such quads -- 2x2 pixels whose estimate (the map
    // cells under them) reaches the threshold -- goes out as its
    // ordinary descriptor plus a 64-bit word of HOLES, the quads its own waves leave out, and one cooperative descriptor per 4x4-pixel
    // region that has holes (level 2: a 2x2 quad = 4 rays per wave; rt_debug.h can ask for 16 rays or one): the waves of those trace
    // the holes and nothing else.  RT_DEBUG_COOP: 0 never, 2 every quad of every block (tests).
    size_t n_coop = 0;
    uint64_t coop_thr = 0;
    const long long coop_knob = knob(RT_DEBUG_COOP);
    const bool coop_all = coop_knob == 2;
    if (coop && holes && coop->fanout != 0u && passes == 1 && knob(RT_DEBUG_WG_POLICY) <= 0 && coop_knob != 0 && coop_percent != 0 && !cost.empty() && (map || coop_all)) {
        if (coop_all) n_coop = order.size();
        else {
            const uint64_t top = cost[order[0]];
            const long long t = knob(RT_DEBUG_COOP_THR), m = knob(RT_DEBUG_COOP_MAX);
            coop_thr = t >= 0 ? (uint64_t)t : std::max<uint64_t>(kCoopMinCost, top * (uint64_t)(coop_percent > 0 ? coop_percent : (int)kCoopPercent) / 100);
            // (a pass of more blocks is throughput-bound from its first to its last wave -- 1080p: DESIGN.md 4.1 -- and a cooperative test costs
            // four to five times the vector instructions of a test of the skip-pointer walk: it only pays where waves wait for a few chains)
            const size_t cap = m >= 0 ? (size_t)m : order.size() > kCoopPassBlocks ? 0 : order.size() / kCoopMaxShare;
            while (n_coop < order.size() && n_coop < cap && cost[order[n_coop]] >= coop_thr) ++n_coop;
        }
    }
    // ... or by EXACT tests per pixel (exact_block_costs: the counted blocks, this list's heaviest by the map): a 2x2-pixel quad is walked
    // cooperatively when one of its pixels took exact_thr tests or more.  The blocks that hold such quads move to the front of the order
    // (the holes of a pass are indexed by descriptor position); what is left of them goes out as 4x4-pixel quarters, like the narrow tier.
    std::vector<uint64_t> exact_hole;                    // of order[0 .. n_coop)
    const bool use_exact = exact && exact_thr > 0 && !exact->block.empty() && coop && holes && coop->fanout != 0u && passes == 1 && knob(RT_DEBUG_WG_POLICY) <= 0 &&
                           coop_knob != 0 && !coop_all && !cost.empty();
    const uint32_t top_estimate = cost.empty() ? 0u : cost[order[0]];
    if (use_exact) {
        std::vector<uint64_t> hole_of(cost.size(), 0ull);
        for (size_t k = 0; k < exact->block.size(); ++k) {
            const uint32_t *px = &exact->px[k * 256];
            uint64_t hole = 0;
            for (unsigned qy = 0; qy < 8; ++qy)
                for (unsigned qx = 0; qx < 8; ++qx) {
                    const uint32_t m = std::max(std::max(px[(2 * qy) * 16 + 2 * qx], px[(2 * qy) * 16 + 2 * qx + 1]), std::max(px[(2 * qy + 1) * 16 + 2 * qx], px[(2 * qy + 1) * 16 + 2 * qx + 1]));
                    if (m >= exact_thr) hole |= 1ull << (qy * 8u + qx);
                }
            hole_of[exact->block[k]] = hole;
        }
        std::stable_partition(order.begin(), order.end(), [&hole_of](uint32_t b) { return hole_of[b] != 0ull; });
        n_coop = 0;
        while (n_coop < order.size() && hole_of[order[n_coop]] != 0ull) exact_hole.push_back(hole_of[order[n_coop++]]);
        coop_thr = exact_thr;
    }
    if (map && !cost.empty()) {
        const long long e = knob(RT_DEBUG_NARROW_MAX);
        // a pass of more blocks than kNarrowPassBlocks is throughput-bound: narrowing only adds work there (3840x2160 + 2 %)
        // (and only in single-pass launches: the packed sample-parallel mapping has its own, finer ray packets)
        // (behind a cooperative tier the next blocks are narrowed more generously: tools/coop_sweep.py, 800x600 28.4 -> 26.4 us)
        size_t cap = passes > 1 ? 0 : e >= 0 ? (size_t)e : order.size() > kNarrowPassBlocks ? 0 : std::min<size_t>(kNarrowMax, order.size() / (n_coop && !coop_all && !use_exact ? 32 : 128));
        if (use_exact && e < 0) cap = cap > n_coop ? cap - n_coop : 0;        // (the blocks with exact holes are narrow already: the tier is as large as without them)
        // (with the heaviest blocks walked cooperatively, "expensive" is measured against the cooperative threshold)
        const uint64_t top = use_exact ? top_estimate : n_coop && !coop_all ? coop_thr : cost[order[0]];
        while (n_coop + n_narrow < order.size() && n_narrow < cap && cost[order[n_coop + n_narrow]] > 0 &&
               (uint64_t)cost[order[n_coop + n_narrow]] * 100 >= top * kNarrowPercent)
            ++n_narrow;
    }
    descs.clear();
    descs.reserve(order.size() + 15 * n_narrow + 63 * n_coop);
    std::vector<uint32_t> dcost;                          // cost estimate of every descriptor, descending
    dcost.reserve(order.size() + 15 * n_narrow + 63 * n_coop);
    if (n_coop) {
        // level 1: workgroups of 8x8 pixels (a 4x4 quad = 16 rays per wave); 2: 4x4 (2x2 = 4 rays per wave); 3: 2x2 (one ray per wave)
        const long long lk = knob(RT_DEBUG_COOP_LEVEL);
        const unsigned level = lk >= 1 && lk <= 3 ? (unsigned)lk : kCoopLevel, step = 16u >> level, cnt = 1u << level, quad = step / 2u;
        const unsigned grain = std::max(2u, quad);           // a hole is decided for `grain` x `grain` pixels at once: whole cooperative quads
        // the chains follow silhouettes thinner than a map cell: where cells are small (a few pixels) their neighbours count too
        const int grow = (w + R - 1) / R <= 4 ? 1 : 0;
        const unsigned rest_level = knob(RT_DEBUG_COOP_REST) >= 0 ? (unsigned)std::min(1ll, knob(RT_DEBUG_COOP_REST)) : use_exact ? 1u : kCoopRestLevel;
        std::vector<rt::BlockDesc> cdescs;
        std::vector<uint32_t> ccost;
        for (size_t i = 0; i < n_coop; ++i) {
            const rt::BlockDesc &d = raster[order[i]];
            uint64_t hole = use_exact ? exact_hole[i] : 0ull;
            for (unsigned gy = 0; !use_exact && gy < 16u; gy += grain)
                for (unsigned gx = 0; gx < 16u; gx += grain) {
                    const unsigned px0 = d.x0 + gx, py0 = d.y0 + gy;
                    if (!(px0 < d.r && py0 < d.t)) continue;                    // outside a clipped edge tile
                    if (!(coop_all || map_max(px0, py0, std::min<unsigned>(px0 + grain, d.r) - 1, std::min<unsigned>(py0 + grain, d.t) - 1, grow) >= coop_thr)) continue;
                    for (unsigned sy = 0; sy < grain; sy += 2)
                        for (unsigned sx = 0; sx < grain; sx += 2) hole |= 1ull << (((gy + sy) >> 1) * 8u + ((gx + sx) >> 1));
                }
            if (rest_level == 0u || !hole) { descs.push_back(d); dcost.push_back(cost[order[i]]); holes->push_back(hole); }
            else
                for (unsigned qy = 0; qy < 2; ++qy)             // what is left of the block as four quarters (a 4x4 patch per wave), each with its 4x4 holes
                    for (unsigned qx = 0; qx < 2; ++qx) {
                        rt::BlockDesc n = d;
                        n.x0 = (uint16_t)(d.x0 + qx * 8u); n.y0 = (uint16_t)(d.y0 + qy * 8u);
                        if (!(n.x0 < d.r && n.y0 < d.t)) continue;
                        uint64_t sub = 0;
                        for (unsigned sy = 0; sy < 4; ++sy)
                            for (unsigned sx = 0; sx < 4; ++sx)
                                if ((hole >> ((qy * 4u + sy) * 8u + qx * 4u + sx)) & 1ull) sub |= 1ull << (sy * 4u + sx);
                        if (sub == 0xFFFFull) continue;          // nothing left of this quarter
                        n.pitch |= 1u << rt::kBlockNarrowShift;
                        descs.push_back(n); dcost.push_back(cost[order[i]]); holes->push_back(sub);
                    }
            if (!hole) continue;
            for (unsigned qy = 0; qy < cnt; ++qy)
                for (unsigned qx = 0; qx < cnt; ++qx) {
                    rt::BlockDesc n = d;
                    n.x0 = (uint16_t)(d.x0 + qx * step); n.y0 = (uint16_t)(d.y0 + qy * step);
                    uint32_t mask = 0;
                    for (unsigned wv = 0; wv < 4; ++wv) {
                        const unsigned lx = qx * step + (wv & 1u) * quad, ly = qy * step + (wv >> 1) * quad;      // the wave's quad, inside the block
                        if ((hole >> ((ly >> 1) * 8u + (lx >> 1))) & 1ull) mask |= 1u << wv;
                    }
                    if (!mask) continue;
                    n.pitch |= (level << rt::kBlockNarrowShift) | (mask << rt::kBlockCoopShift);
                    cdescs.push_back(n); ccost.push_back(cost[order[i]]);
                }
        }
        descs.insert(descs.end(), cdescs.begin(), cdescs.end());
        dcost.insert(dcost.end(), ccost.begin(), ccost.end());
    }
    for (size_t i = n_coop; i < order.size(); ++i) {
        const rt::BlockDesc &d = raster[order[i]];
        if (i >= n_coop + n_narrow) { descs.push_back(d); dcost.push_back(cost[order[i]]); continue; }
        // 4x4 pixels per wave; 2x2 in a pass so small that its waves all start at once anyway (800x600: 63 -> 52 us; at 1080p the
        // sixteen-fold wave count of 2x2 costs more throughput than the shorter chains buy)
        const long long l2 = knob(RT_DEBUG_NARROW_L2);
        const unsigned level = (l2 >= 0 ? (long long)(i - n_coop) < l2 : (order.size() <= kNarrowLevel2Blocks && n_coop == 0)) ? 2u : 1u, step = 16u >> level, cnt = 1u << level;
        for (unsigned qy = 0; qy < cnt; ++qy)
            for (unsigned qx = 0; qx < cnt; ++qx) {
                rt::BlockDesc n = d;
                n.x0 = (uint16_t)(d.x0 + qx * step); n.y0 = (uint16_t)(d.y0 + qy * step);
                n.pitch |= level << rt::kBlockNarrowShift;
                if (n.x0 < d.r && n.y0 < d.t) { descs.push_back(n); dcost.push_back(cost[order[i]]); }      // parts outside a clipped edge tile have no pixels
            }
    }
    // A sample-parallel pass has one workgroup per block AND sample: 49,152 for `make image`, 1,048,576 for BASELINE config 5,
    // each living a few microseconds.  Past 32,768 workgroups the blocks are dealt out here instead, about eight to a
    // workgroup (8,192 .. 65,536 workgroups per launch): descriptors in descending cost, each to the workgroup with the least
    // estimated work so far (longest-processing-time-first); a workgroup renders its descriptors in that order, so the long
    // chains still start first.  Measured (tools/knob_sweep.py wg_policy): make image 303 -> 278 us, config 5 4.67 -> 4.26 ms;
    // passes the dispatcher can deal one block at a time (1080p, 4K at spp 1) lose by it -- its dynamic balancing beats
    // a static deal by estimated cost -- and are left alone.
    const long long policy = knob(RT_DEBUG_WG_POLICY);
    const size_t total_wg = (size_t)descs.size() * std::max(1u, passes);
    size_t n_wg = 0;
    if (policy > 0) n_wg = kResidentWorkgroups * (size_t)policy / std::max(1u, passes);
    else if (policy < 0 && total_wg > 32768) n_wg = std::clamp<size_t>(total_wg / 8, 8192, 65536) / std::max(1u, passes);
    if (map && n_wg >= 64 && descs.size() > n_wg && n_coop == 0) {        // (the holes of a cooperative pass are indexed by descriptor position)
        std::vector<std::vector<uint32_t>> lists(n_wg);
        std::vector<std::pair<uint64_t, uint32_t>> heap;                      // (load, workgroup), min-heap
        heap.reserve(n_wg);
        for (uint32_t g = 0; g < n_wg; ++g) heap.emplace_back(0ull, g);
        auto cmp = [](const std::pair<uint64_t, uint32_t> &a, const std::pair<uint64_t, uint32_t> &b) { return a > b; };
        std::make_heap(heap.begin(), heap.end(), cmp);
        for (uint32_t i = 0; i < descs.size(); ++i) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            auto &top = heap.back();
            lists[top.second].push_back(i);
            top.first += (uint64_t)dcost[i] + kFixedBlockCost;
            std::push_heap(heap.begin(), heap.end(), cmp);
        }
        std::vector<rt::BlockDesc> dealt;
        dealt.reserve(descs.size());
        wg_first.reserve(n_wg + 1);
        for (const auto &l : lists) {
            wg_first.push_back((uint32_t)dealt.size());
            for (uint32_t i : l) dealt.push_back(descs[i]);
        }
        wg_first.push_back((uint32_t)dealt.size());
        descs.swap(dealt);
    }
}

bool block_order_enabled() { return knob(RT_DEBUG_BLOCK_ORDER) != 0; }     // read per call: A/B timing interleaves both


void release_order(rt_scene::Order &od)           // (its arrays live in the table's arena)
{
    for (hipEvent_t e : od.e0) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : od.e1) if (e) (void)hipEventDestroy(e);
    od = rt_scene::Order{};
}

// The dispatch order a launch of this table uses (called under the scene's lock).  While a table with several candidates is undecided, its
// launches take turns: every candidate is handed out kOrderTrialSamples times with a pair of events (the caller records them around the
// launch; all of them may be in flight at once -- a caller that enqueues far ahead of the device is not waited for); results are collected
// here as they complete, and once every sample is in, the candidate with the smallest one stays.
constexpr int kOrderTrialSamples = 3;
rt::BlockList pick_order(rt_scene::CachedTable &t, bool will_be_timed)      // will_be_timed: a non-counting hierarchy-walk launch (launch_render records the pair)
{
    if (t.orders.empty()) return rt::BlockList{};
    // (orders[0] never has holes: what a launch that cannot walk cooperatively -- counters, f64, two rays per lane -- falls back to)
    auto list_of = [&t](const rt_scene::Order &od) {
        rt::BlockList l{ od.dev_order, od.n_order, od.dev_wg, od.n_wg, od.dev_holes, od.n_holes };
        l.plain_d = t.orders[0].dev_order; l.plain_n = t.orders[0].n_order; l.plain_wg_first = t.orders[0].dev_wg; l.plain_n_wg = t.orders[0].n_wg;
        return l;
    };
    if (t.chosen >= 0) return list_of(t.orders[(size_t)t.chosen]);
    bool all_done = true;
    for (auto &od : t.orders) {
        while (od.harvested < od.issued && hipEventQuery(od.e1[od.harvested % kOrderTrialSamples]) == hipSuccess) {
            float ms = 0.f;
            const int slot = od.harvested % kOrderTrialSamples;
            if (hipEventElapsedTime(&ms, od.e0[slot], od.e1[slot]) == hipSuccess && ms > 0.f) { od.best_ms = std::min(od.best_ms, ms); ++od.good; }
            ++od.harvested;
        }
        (void)hipGetLastError();                         // hipErrorNotReady is not an error here
        all_done = all_done && od.good >= kOrderTrialSamples;
    }
    auto best_known = [&t] {
        size_t best = 0;
        for (size_t i = 1; i < t.orders.size(); ++i) if (t.orders[i].best_ms < t.orders[best].best_ms) best = i;
        return best;
    };
    if (all_done) {
        const size_t best = best_known();
        t.chosen = (int)best;
        if (knob(RT_DEBUG_PRINT_STEPS) > 0) {
            fprintf(stderr, "[rtrace_hip] dispatch orders of a %zu-tile list, ms:", t.host.size());
            for (const auto &od : t.orders) fprintf(stderr, " %.4f%s", od.best_ms, od.dev_holes ? "c" : "");
            fprintf(stderr, " -> #%zu\n", best);
        }
        return list_of(t.orders[best]);
    }
    for (size_t k = 0; will_be_timed && k < t.orders.size(); ++k) {
        rt_scene::Order &od = t.orders[(t.turn + k) % t.orders.size()];
        const int in_flight = od.issued - od.harvested;
        if (od.good + in_flight >= kOrderTrialSamples) continue;     // (in_flight < kOrderTrialSamples follows: the slot is free)
        t.turn = (unsigned)((t.turn + k + 1) % t.orders.size());
        rt::BlockList l = list_of(od);
        l.ev0 = od.e0[od.issued % kOrderTrialSamples]; l.ev1 = od.e1[od.issued % kOrderTrialSamples];
        ++od.issued;
        return l;
    }
    return list_of(t.orders[best_known()]);              // every sample is in flight: the best known so far, untimed
}

// Something about the dispatch was asked for explicitly (rt_debug.h: tests, A/B tools): then a tile list's orders are made at once, by the
// caller, so that its very first launch already runs what was asked for.
bool order_knobs_set()
{
    for (int k : { RT_DEBUG_COOP, RT_DEBUG_COOP_THR, RT_DEBUG_COOP_MAX, RT_DEBUG_COOP_LEVEL, RT_DEBUG_COOP_REST, RT_DEBUG_NARROW_MAX, RT_DEBUG_NARROW_L2, RT_DEBUG_WG_POLICY,
                   RT_DEBUG_PRINT_COSTS, RT_DEBUG_EXACT_COSTS })
        if (knob(k) >= 0) return true;
    return knob(RT_DEBUG_ASYNC_ORDERS) == 0;
}

// The candidate dispatch orders of one tile list: the plain one first; where the cooperative walk could serve the pass and nothing was
// asked for explicitly, a few thresholds in percent of the pass's largest estimate (pick_order tries them: chosen = -1).
rt_status build_orders(rt_scene *s, const std::vector<uint32_t> *map, const std::vector<rt::TileDev> &tab, unsigned w, unsigned h, unsigned passes,
                       std::vector<rt_scene::Order> &orders, int &chosen, void **arena_out)
{
    const bool coop_pass = s->precision == RT_F32 && s->coop.fanout != 0u && passes == 1;
    uint64_t total_px = 0, total_blocks = 0;
    for (const rt::TileDev &td : tab) {
        total_px += (uint64_t)(td.r - td.l) * (td.t - td.b);
        total_blocks += (uint64_t)td.blks_x * (((unsigned)(td.t - td.b) + rt::kBlockH - 1) / rt::kBlockH);
    }
    const long long rays = knob(RT_DEBUG_SKIP_RAYS);
    const bool two_rays = rays < 0 ? skip2_by_default(total_px, 1, s->fused ? s->n_fnodes : s->n_nodes) : rays == 2;      // k_render_skip2 knows no cooperative quads
    // Candidate 0 is ALWAYS the plain order of a pass that could walk cooperatively (coop_percent 0: no holes, no cooperative descriptors):
    // pick_order and launch_skip_one hand it to every launch that cannot take holes (counters, f64, two rays per lane).
    // Candidates: the plain order first; then cooperative thresholds.  Where the scene's stream is there to count the list's heaviest blocks
    // again at the frame's own resolution (exact_block_costs), the thresholds are shares of the largest EXACT count of tests per pixel and the
    // quads are picked pixel by pixel -- any pass size; without it (or when a control of rt_debug.h asks for the old way) shares of the map's
    // largest estimate, small passes only.
    struct Want { int pc; uint32_t exact_thr; };
    std::vector<Want> wants;
    ExactCosts exact;
    const bool knobs = knob(RT_DEBUG_COOP) >= 0 || knob(RT_DEBUG_COOP_THR) >= 0 || knob(RT_DEBUG_COOP_MAX) >= 0;
    if (coop_pass && map && !two_rays && !knobs && knob(RT_DEBUG_EXACT_COSTS) != 0) {
        std::vector<rt::BlockDesc> none; std::vector<uint32_t> none_wg, heaviest;
        block_order(map, tab, w, h, passes, none, none_wg, nullptr, nullptr, 0, nullptr, 0, &heaviest);
        std::vector<rt::BlockDesc> raster;
        for (const rt::TileDev &t : tab) {          // (block_order's raster enumeration: the indices `heaviest` holds)
            const unsigned bys = ((unsigned)(t.t - t.b) + rt::kBlockH - 1) / rt::kBlockH;
            const uint32_t pitch = (uint32_t)t.r - t.l;
            for (unsigned by = 0; by < bys; ++by)
                for (unsigned bx = 0; bx < t.blks_x; ++bx)
                    raster.push_back(rt::BlockDesc{ (uint16_t)(t.l + bx * rt::kBlockW), (uint16_t)(t.b + by * rt::kBlockH), t.r, t.t, pitch, 0u });
        }
        if (exact_block_costs<float>(s, raster, heaviest, w, h, exact) != RT_OK) { (void)hipGetLastError(); exact = ExactCosts{}; }
    }
    bool asked = false;
    if (!exact.block.empty() && exact.top >= kCoopMinCost) {
        wants.push_back({ 0, 0u });
        for (unsigned pc : { 85u, 70u, 58u, 48u, 40u }) wants.push_back({ 0, std::max<uint32_t>((uint32_t)kCoopMinCost, exact.top * pc / 100u) });
    } else if (coop_pass && map && !two_rays && total_blocks <= kCoopPassBlocks && !knobs)
        wants = { { 0, 0u }, { 28, 0u }, { 34, 0u }, { 40, 0u }, { 48, 0u }, { 58, 0u } };
    else if (coop_pass && !two_rays && knobs && (knob(RT_DEBUG_COOP) > 0 || knob(RT_DEBUG_COOP_THR) >= 0 || knob(RT_DEBUG_COOP_MAX) >= 0)) { wants = { { 0, 0u }, { -1, 0u } }; asked = true; }       // as asked, behind the plain one
    else if (coop_pass) wants = { { 0, 0u } };
    else wants = { { -1, 0u } };
    // every candidate on the host first, then ONE device allocation for all their arrays: hipMalloc / hipFree wait for a busy device,
    // and this may run in the background of a caller who keeps it busy
    struct Host { std::vector<rt::BlockDesc> order; std::vector<uint32_t> wg_first; std::vector<uint64_t> holes; bool any_hole = false; };
    std::vector<Host> cand;
    for (const Want &wt : wants) {
        Host c;
        block_order(map, tab, w, h, passes, c.order, c.wg_first, coop_pass ? &s->coop : nullptr, &c.holes, wt.pc, wt.exact_thr ? &exact : nullptr, wt.exact_thr);
        c.any_hole = std::any_of(c.holes.begin(), c.holes.end(), [](uint64_t v) { return v != 0; });
        if (&wt != &wants[0] && !c.any_hole) continue;     // the same dispatch as the plain one
        if (!cand.empty() && wt.exact_thr && cand.back().any_hole && cand.back().holes == c.holes && cand.back().order.size() == c.order.size()) continue;   // (two thresholds, the same quads)
        cand.push_back(std::move(c));
    }
    auto up = [](size_t n) { return (n + 255) & ~(size_t)255; };
    size_t bytes = 0;
    for (const Host &c : cand)
        bytes += up(c.order.size() * sizeof(rt::BlockDesc)) + (c.any_hole ? up(c.holes.size() * sizeof(uint64_t)) : 0) + (c.wg_first.empty() ? 0 : up(c.wg_first.size() * sizeof(uint32_t)));
    char *arena = nullptr;
    hipError_t e = hipMalloc(&arena, std::max<size_t>(bytes, 256));
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(dispatch orders)", __LINE__);
    auto fail = [&](hipError_t err) { for (auto &od : orders) release_order(od); orders.clear(); (void)hipFree(arena); return hip_fail(err, "dispatch orders", __LINE__); };
    size_t off = 0;
    for (const Host &c : cand) {
        rt_scene::Order od;
        od.dev_order = reinterpret_cast<rt::BlockDesc *>(arena + off); od.n_order = (uint32_t)c.order.size();
        if ((e = hipMemcpyAsync(od.dev_order, c.order.data(), c.order.size() * sizeof(rt::BlockDesc), hipMemcpyHostToDevice, s->cost_stream)) != hipSuccess) return fail(e);
        off += up(c.order.size() * sizeof(rt::BlockDesc));
        if (c.any_hole) {
            od.dev_holes = reinterpret_cast<uint64_t *>(arena + off); od.n_holes = (uint32_t)c.holes.size();
            if ((e = hipMemcpyAsync(od.dev_holes, c.holes.data(), c.holes.size() * sizeof(uint64_t), hipMemcpyHostToDevice, s->cost_stream)) != hipSuccess) return fail(e);
            off += up(c.holes.size() * sizeof(uint64_t));
        }
        if (!c.wg_first.empty()) {
            od.dev_wg = reinterpret_cast<uint32_t *>(arena + off); od.n_wg = (uint32_t)c.wg_first.size() - 1;
            if ((e = hipMemcpyAsync(od.dev_wg, c.wg_first.data(), c.wg_first.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s->cost_stream)) != hipSuccess) return fail(e);
            off += up(c.wg_first.size() * sizeof(uint32_t));
        }
        orders.push_back(od);
    }
    if ((e = hipStreamSynchronize(s->cost_stream)) != hipSuccess) return fail(e);       // (the candidates' host arrays go out of scope; the orders are in device memory from here on)
    if (!orders.empty() && orders[0].dev_holes) {       // (cannot happen: candidate 0 is made with coop_percent 0 wherever holes are possible)
        for (auto &od : orders) release_order(od);
        orders.clear(); (void)hipFree(arena);
        snprintf(g_err, sizeof g_err, "internal: the plain dispatch order has holes");
        return RT_ERR_INVALID_ARGUMENT;
    }
    chosen = 0;
    if (orders.size() > 1 && asked) chosen = 1;      // asked for explicitly
    else if (orders.size() > 1) {
        chosen = -1;                                                // to be decided by measurement
        for (auto &od : orders)
            for (int k = 0; k < kOrderTrialSamples; ++k) {
                e = hipEventCreate(&od.e0[k]);
                if (e == hipSuccess) e = hipEventCreate(&od.e1[k]);
                if (e != hipSuccess) return fail(e);
            }
    }
    *arena_out = arena;
    return RT_OK;
}

// Builder threads must not outlive the HIP runtime: a process that exits without destroying its scenes (a Python interpreter does not run
// every finalizer) still has them joined, by an exit handler registered when the first one is started -- later than the runtime's own
// teardown was registered, hence run before it.
void stop_worker(rt_scene *s);
std::mutex g_live_mu;
std::vector<rt_scene *> g_live_scenes;           // scenes that ever started a builder and are not destroyed yet

void join_builders_at_exit()
{
    std::vector<rt_scene *> live;
    { std::lock_guard<std::mutex> lk(g_live_mu); live.swap(g_live_scenes); }
    for (rt_scene *sc : live) {
        for (std::thread &b : sc->builders) if (b.joinable()) b.join();
        stop_worker(sc);
    }
}

void note_builder(rt_scene *s)
{
    static std::once_flag once;
    std::call_once(once, [] { std::atexit(join_builders_at_exit); });
    std::lock_guard<std::mutex> lk(g_live_mu);
    if (std::find(g_live_scenes.begin(), g_live_scenes.end(), s) == g_live_scenes.end()) g_live_scenes.push_back(s);
}

void forget_scene(rt_scene *s)
{
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live_scenes.erase(std::remove(g_live_scenes.begin(), g_live_scenes.end(), s), g_live_scenes.end());
}

// The scene's worker: runs the jobs handed to it one after the other; on stop, the ones still queued as well (they are finite and somebody
// may be waiting for `building` to clear).
void worker_main(rt_scene *s)
{
    knobs_at_default();                           // it only ever serves lists that were first seen with no dispatch control set
    (void)hipSetDevice(s->device);
    for (;;) {
        std::function<void()> job;
        bool stopping;
        {
            std::unique_lock<std::mutex> lk(s->wmu);
            s->wcv.wait(lk, [s] { return s->wstop || !s->wjobs.empty(); });
            if (s->wjobs.empty()) return;
            job = std::move(s->wjobs.front());
            s->wjobs.pop_front();
            stopping = s->wstop;
        }
        // let the caller's first launch (and whoever waits for it) have the runtime to itself: the orders' allocations and blocking copies
        // took 20-50 us out of a one-shot caller's first frame when they started at once, and nobody misses them for another 0.3 ms
        if (!stopping) std::this_thread::sleep_for(std::chrono::microseconds(300));
        job();
    }
}

void stop_worker(rt_scene *s)
{
    if (!s->worker.joinable()) return;
    { std::lock_guard<std::mutex> lk(s->wmu); s->wstop = true; }
    s->wcv.notify_all();
    s->worker.join();
}

// The same from a thread of its own (see device_table): cost map, orders, uploads -- then the finished orders are handed to table `index`
// under the scene's lock.  Whatever fails here only costs the ordering: the table keeps rendering through the tile table.
void build_orders_async(rt_scene *s, size_t index, std::vector<rt::TileDev> tab, unsigned w, unsigned h, unsigned passes)
{
    knobs_at_default();                           // this thread only exists because no dispatch control was set when the list was first seen
    std::vector<rt_scene::Order> orders;
    int chosen = 0;
    void *arena = nullptr;
    bool ok = hipSetDevice(s->device) == hipSuccess;
    if (ok) {
        const std::vector<uint32_t> *map = cost_map_of(s);
        // (the uploads are blocking copies: the data is in device memory when they return.  No device-wide synchronise here -- the caller's
        // own launches keep the device busy and it would wait for all of them)
        ok = build_orders(s, map, tab, w, h, passes, orders, chosen, &arena) == RT_OK;
    }
    (void)hipGetLastError();
    std::lock_guard<std::mutex> lk(s->mu);
    rt_scene::CachedTable &t = s->tables[index];
    if (ok) { t.orders = std::move(orders); t.chosen = chosen; t.order_arena = arena; }
    t.building = false;
}

rt_status device_table(rt_scene *s, Context *c, const std::vector<rt::TileDev> &tab, hipStream_t stream, const rt::TileDev **out, int slot,
                       const rt_options *o, rt::BlockList *order_out, bool cacheable, bool will_be_timed)
{
    const size_t bytes = tab.size() * sizeof(rt::TileDev);
    const unsigned w = o ? o->width : 0u, h = o ? o->height : 0u;
    const unsigned passes = (o && use_split(o->samples_per_pixel)) ? (unsigned)o->samples_per_pixel * o->samples_per_pixel : 1u;
    if (order_out) *order_out = rt::BlockList{};
    StageClock clk;
    // the cooperative walk's controls (rt_debug.h) are part of a dispatch table's identity: tests render one tile list with and without
    long long coop_key = 0;
    if (o && order_out)
        for (int k : { RT_DEBUG_COOP, RT_DEBUG_COOP_THR, RT_DEBUG_COOP_MAX, RT_DEBUG_COOP_LEVEL, RT_DEBUG_COOP_REST, RT_DEBUG_NARROW_MAX, RT_DEBUG_NARROW_L2, RT_DEBUG_SKIP_RAYS, RT_DEBUG_EXACT_COSTS })
            coop_key = coop_key * 1000003ll + (knob(k) + 2);
    if (cacheable) {
        std::lock_guard<std::mutex> lk(s->mu);
        for (auto &t : s->tables)
            if ((!o || (t.w == w && t.h == h && t.passes == passes && (!order_out || t.coop_key == coop_key))) && t.host.size() == tab.size() &&
                memcmp(t.host.data(), tab.data(), bytes) == 0) {
                *out = t.dev;
                if (t.landed) {                      // uploaded on its first caller's stream: has it arrived?
                    if (hipEventQuery(t.landed) == hipSuccess) { (void)hipEventDestroy(t.landed); t.landed = nullptr; }
                    else {
                        (void)hipGetLastError();
                        if (stream != t.landed_on) HIP_TRY(hipStreamWaitEvent(stream, t.landed, 0));
                    }
                }
                if (order_out && block_order_enabled()) *order_out = pick_order(t, will_be_timed);
                return RT_OK;
            }
        if (s->tables.size() < kMaxCachedTables) {
            rt_scene::CachedTable t;
            // The dispatch orders (and the scene's cost map they are made from) cost the host a few milliseconds: unless something was asked
            // for explicitly (rt_debug.h), they are made by the scene's worker thread while this and the next launches find their blocks through
            // the tile table -- a one-shot caller (`make image`) never waits for them, a scheduler gets them a few frames in.
            const bool want_orders = o && order_out;
            const bool in_background = want_orders && !order_knobs_set();
            HIP_TRY(hipMalloc(&t.dev, bytes));
            auto drop = [&t] { (void)hipFree(t.dev); for (auto &od : t.orders) release_order(od); if (t.order_arena) (void)hipFree(t.order_arena);
                               if (t.landed) (void)hipEventDestroy(t.landed); };
            // The table itself: through the scene's pinned staging on the caller's own stream when the list is new to a caller in a hurry
            // (the launch that follows is behind it on that stream; launches on other streams wait for `landed`) -- a blocking copy and the
            // device-wide synchronise it needs cost the first frame 50 us.
            const size_t staged = (bytes + 255) & ~(size_t)255;
            bool async_copy = false;
            if (in_background && s->h_tab_stage && s->tab_stage_used + staged <= kTableStageBytes &&
                hipEventCreateWithFlags(&t.landed, hipEventDisableTiming) == hipSuccess) {
                char *h = s->h_tab_stage + s->tab_stage_used;
                memcpy(h, tab.data(), bytes);
                if (upload_words(t.dev, h, bytes, stream) == RT_OK && hipEventRecord(t.landed, stream) == hipSuccess) {      // (a kernel, not the copy engine: rt_kernels.hpp k_upload_words)
                    s->tab_stage_used += staged;
                    t.landed_on = stream;
                    async_copy = true;
                } else { (void)hipGetLastError(); (void)hipEventDestroy(t.landed); t.landed = nullptr; }
            } else (void)hipGetLastError();
            hipError_t e = async_copy ? hipSuccess : hipMemcpy(t.dev, tab.data(), bytes, hipMemcpyHostToDevice);    // blocking, once per table
            clk.lap("tile table upload");
            if (e != hipSuccess) { drop(); return hip_fail(e, "hipMemcpy(tile table)", __LINE__); }
            if (want_orders && !in_background) {
                const std::vector<uint32_t> *map = cost_map_of(s);
                clk.lap("cost map (cached after 1st)");
                rt_status bst = build_orders(s, map, tab, w, h, passes, t.orders, t.chosen, &t.order_arena);
                clk.lap("dispatch orders");
                if (bst != RT_OK) { drop(); return bst; }
            }
            // The copies above are blocking for the host, but the render kernel runs on another (non-blocking) stream: make sure
            // the tables have landed in device memory before anything can be launched against them (once per tile list).
            if (!async_copy) {
                e = hipDeviceSynchronize();
                if (e != hipSuccess) { drop(); return hip_fail(e, "hipDeviceSynchronize(tile tables)", __LINE__); }
            }
            clk.lap("device sync");
            t.building = in_background;
            t.host = tab; t.w = w; t.h = h; t.passes = passes; t.coop_key = coop_key;
            *out = t.dev;
            s->tables.push_back(std::move(t));
            if (in_background) {
                const size_t index = s->tables.size() - 1;
                note_builder(s);
                if (s->worker.joinable()) {
                    { std::lock_guard<std::mutex> wl(s->wmu); s->wjobs.emplace_back([s, index, tab, w, h, passes] { build_orders_async(s, index, tab, w, h, passes); }); }
                    s->wcv.notify_one();
                } else s->builders.emplace_back([s, index, tab, w, h, passes] { knobs_at_default(); build_orders_async(s, index, tab, w, h, passes); });
            }
            if (order_out && block_order_enabled()) *order_out = pick_order(s->tables.back(), will_be_timed);
            return RT_OK;
        }
    }
    if (!c) { *out = nullptr; return RT_OK; }                       // cache full and no context to upload through
    return upload_tiles(c, tab, stream, slot, out);
}

// Variant of k_render_skip (rt_skip.hpp VAR bits): the generated assembly loops, fused where the scene allows it.
// rt_debug.h overrides it for A/B runs (read per call so one process can interleave variants, tools/ab.py); the fused bit
// is dropped for scenes that are not fused.
int skip_variant(const rt_scene *s)
{
    int v = 1 | 2 | 4 | 16;
    if (const long long o = knob(RT_DEBUG_SKIP_VARIANT); o >= 0) v = (int)o & 23;
    if (v & 2) v |= 1;                                  // the assembly loops imply the lean sqrt in what C++ remains
    if (!s->fused || !(v & 2)) v &= ~4;
    if (!s->d_xprim || !(v & 2)) v &= ~16;      // the filtered loops (f32: both walks; f64: the primary walk) need their streams
#ifdef RT_TEST_HOOKS
    if ((v & 3) == 3 && g_trace_on.load(std::memory_order_relaxed)) v |= 8;      // diagnostic build of the assembly variants
#endif
    return v;
}


// (Re)allocates the context's per-sample buffers {n.light, state} for `samples` samples of REAL size `esz`.
rt_status ensure_sample_buffers(Context *c, size_t samples, size_t esz)
{
    const size_t need = samples * esz;
    if (c->sample_cap >= need) return RT_OK;
    if (c->d_sample_gdot) HIP_TRY(hipFree(c->d_sample_gdot));
    if (c->d_sample_state) HIP_TRY(hipFree(c->d_sample_state));
    c->d_sample_gdot = nullptr; c->d_sample_state = nullptr; c->sample_cap = 0;
    HIP_TRY(hipMalloc(&c->d_sample_gdot, need));
    HIP_TRY(hipMalloc(&c->d_sample_state, samples));
    c->sample_cap = need;
    return RT_OK;
}

// rt_flat_wf.hpp: primary+shade -> shadow pass over the largest spheres -> shadow pass over the rest -> ordered resolve.
template <typename T, int CHUNK>
rt_status launch_flat_wavefront(const rt_scene *s, Context *c, hipStream_t stream, unsigned w, unsigned h, unsigned spp, const rt::TileDev *d_tab32,
                                unsigned nt, uint32_t blocks32, const rt::TileDev *d_tab16, uint32_t blocks16, uint64_t total_px, uint8_t *d_out,
                                rt::Counters *cnt, unsigned frame_w)
{
    const size_t ns = (size_t)spp * spp, samples = ns * total_px;
    if (ns > 65535 || samples > 0xFFFFFFFFull) {
        snprintf(g_err, sizeof g_err, "flat traversal: too many samples for one pass");
        return RT_ERR_INVALID_ARGUMENT;
    }
    rt_status st = ensure_sample_buffers(c, samples, sizeof(T));
    if (st != RT_OK) return st;
    const size_t qbytes = samples * sizeof(rt::Quad<T>);
    if (c->queue_cap < qbytes) {
        if (c->d_queue1) HIP_TRY(hipFree(c->d_queue1));
        if (c->d_queue2) HIP_TRY(hipFree(c->d_queue2));
        c->d_queue1 = c->d_queue2 = nullptr; c->queue_cap = 0;
        HIP_TRY(hipMalloc(&c->d_queue1, qbytes));
        HIP_TRY(hipMalloc(&c->d_queue2, qbytes));
        c->queue_cap = qbytes;
    }
    if (!c->d_queues) HIP_TRY(hipMalloc(&c->d_queues, sizeof(rt::FlatQueues)));
    HIP_TRY(hipMemsetAsync(c->d_queues, 0, sizeof(rt::FlatQueues), stream));
    rt::SampleBuf<T> sb{ static_cast<T *>(c->d_sample_gdot), c->d_sample_state, (unsigned)total_px };
    rt::Quad<T> *q1 = static_cast<rt::Quad<T> *>(c->d_queue1), *q2 = static_cast<rt::Quad<T> *>(c->d_queue2);
    const dim3 b(rt::kBlockThreads);
    if constexpr (sizeof(T) == 4) {
        if (knob(RT_DEBUG_FLAT_KERNELS) != 0) {
            // f32: the scalar-fed scan (rt_flat_sc.hpp): two rays per lane, a workgroup = two 16x16-pixel blocks of two waves each (the resolve table serves both)
            constexpr unsigned kFirstPassGroups = 342;                  // the 1,026 largest spheres (an even number of groups)
            const rt::FlatScView sv = flat_sc_view_of(s);
            const unsigned first_bytes = std::min(kFirstPassGroups * 64u, sv.n_sbytes);
            c->flat_first_pass_items = first_bytes / 64u * rt::kFlatShadowItems;
            hipLaunchKernelGGL(rt::k_flat_primary_sc, dim3((blocks16 + 1) / 2, (unsigned)ns), dim3(rt::kFlatScPrimaryThreads), 0, stream, sv, w, h, spp,
                               d_tab16, nt, blocks16, sb, q1, c->d_queues, cnt);
            HIP_TRY(hipGetLastError());
            const size_t rays_per_wg = (size_t)rt::kBlockThreads * rt::kFlatScRays;
            const dim3 gsh((unsigned)((samples + rays_per_wg - 1) / rays_per_wg));      // worst case; surplus waves leave at once
            hipLaunchKernelGGL(rt::k_flat_shadow_sc, gsh, b, 0, stream, sv, 0u, first_bytes, q1, &c->d_queues->n1, q2, &c->d_queues->n2, sb, cnt);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(rt::k_flat_shadow_sc, gsh, b, 0, stream, sv, first_bytes, 0xFFFFFF80u, q2, &c->d_queues->n2,
                               (rt::Quad<T> *)nullptr, (unsigned *)nullptr, sb, cnt);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL((rt::k_resolve_samples<T>), dim3(blocks16), b, 0, stream, sb, spp, d_tab16, nt, d_out, frame_w, false);
            return RT_OK;
        }
    }
    c->flat_first_pass_items = (unsigned)CHUNK;
    const rt::FlatView<T> view = flat_view_of<T>(s);
    if constexpr (sizeof(T) == 8) {
        if (knob(RT_DEBUG_FLAT_KERNELS) != 0) {
            // f64: the same pipeline with the conservative bound in front of the exact test (rt_flat_f64.hpp)
            const rt::FlatF64View fx = flat_f64_view_of(s);
            hipLaunchKernelGGL((rt::k_flat_primary_f64<CHUNK>), dim3(blocks32, (unsigned)ns), b, 0, stream, view, fx, w, h, spp, d_tab32, nt, sb, q1, c->d_queues, cnt);
            HIP_TRY(hipGetLastError());
            const unsigned rays_per_wg = rt::kBlockThreads * rt::kFlatR;
            const dim3 gsh((unsigned)((samples + rays_per_wg - 1) / rays_per_wg));
            hipLaunchKernelGGL((rt::k_flat_shadow_f64<CHUNK>), gsh, b, 0, stream, view, fx, 0u, (unsigned)CHUNK, q1, &c->d_queues->n1, q2, &c->d_queues->n2, sb, cnt);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL((rt::k_flat_shadow_f64<CHUNK>), gsh, b, 0, stream, view, fx, (unsigned)CHUNK, 0xFFFFFFFFu, q2, &c->d_queues->n2,
                               (rt::Quad<T> *)nullptr, (unsigned *)nullptr, sb, cnt);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL((rt::k_resolve_samples<T>), dim3(blocks16), b, 0, stream, sb, spp, d_tab16, nt, d_out, frame_w, false);
            return RT_OK;
        }
    }
#ifndef RT_TEST_HOOKS
    // (round 1's unfiltered LDS kernels, rt_flat_wf.hpp: only rt_debug.h's RT_DEBUG_FLAT_KERNELS = 0 selects them)
    (void)d_tab32; (void)blocks32; (void)view;
    snprintf(g_err, sizeof g_err, "internal: no flat-scan kernels for this precision");
    return RT_ERR_UNSUPPORTED;
#else
    hipLaunchKernelGGL((rt::k_flat_primary<T, CHUNK>), dim3(blocks32, (unsigned)ns), b, 0, stream, view, w, h, spp, d_tab32, nt, sb, q1, c->d_queues, cnt);
    HIP_TRY(hipGetLastError());
    const unsigned rays_per_block = rt::kBlockThreads * rt::kFlatR;
    const dim3 gshadow((unsigned)((samples + rays_per_block - 1) / rays_per_block));      // worst case; surplus workgroups leave at once
    hipLaunchKernelGGL((rt::k_flat_shadow<T, CHUNK>), gshadow, b, 0, stream, view, 0u, (unsigned)CHUNK, q1, &c->d_queues->n1, q2, &c->d_queues->n2, sb, cnt);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL((rt::k_flat_shadow<T, CHUNK>), gshadow, b, 0, stream, view, (unsigned)CHUNK, 0xFFFFFFFFu, q2, &c->d_queues->n2,
                       (rt::Quad<T> *)nullptr, (unsigned *)nullptr, sb, cnt);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL((rt::k_resolve_samples<T>), dim3(blocks16), b, 0, stream, sb, spp, d_tab16, nt, d_out, frame_w, false);
    return RT_OK;
#endif
}

// The render kernel of a hierarchy-walk launch: f32 launches that do not count run the build held to 8 waves per SIMD (rt_skip.hpp).
template <typename T, bool COUNT, int VAR, int MODE, bool COOP = false>
constexpr auto skip_kernel()
{
    if constexpr (sizeof(T) == 4 && !COUNT && COOP) return &rt::k_render_skip_f32_coop<COUNT, VAR, MODE>;
    else if constexpr (sizeof(T) == 4 && !COUNT) return &rt::k_render_skip_f32<COUNT, VAR, MODE>;
    else if constexpr (sizeof(T) == 8 && !COUNT && (VAR & 18) == 18) return &rt::k_render_skip_f64<VAR, MODE>;
    else return &rt::k_render_skip<T, COUNT, VAR, MODE, COOP>;
}

template <typename T, bool COUNT, int VAR>
rt_status launch_skip_one(const rt_scene *s, Context *c, dim3 grid, hipStream_t stream, unsigned w, unsigned h, unsigned spp,
                          const rt::TileDev *d_tab, unsigned nt, uint64_t total_px, uint8_t *d_out, rt::Counters *cnt, unsigned frame_w,
                          rt::BlockList order)
{
    // two rays per lane (rt_skip2.hpp): f32, fused assembly loops, launches that neither count nor trace
    bool two_rays = false;
    // (fused scenes: the fused assembly loops, filtered or not; other scenes: the filtered assembly loops over the plain streams)
    constexpr bool kTwoRayFlavour = !COUNT && sizeof(T) == 4 && ((VAR & 15) == 7 || (VAR & 31) == 19);
    if constexpr (kTwoRayFlavour) {
        const long long k = knob(RT_DEBUG_SKIP_RAYS);
        two_rays = k < 0 ? skip2_by_default(total_px, spp, (VAR & 4) ? s->n_fnodes : s->n_nodes) : k == 2;
        two_rays = two_rays && (spp == 1 || (use_split(spp) && packed_samples(spp)));
    }
    // an order with cooperative quads needs the COOP flavour of k_render_skip: everything else renders the plain order of the same list
    constexpr bool kCoopFlavour = !COUNT && sizeof(T) == 4 && (VAR == 19 || VAR == 23 || VAR == 31);
    if (order.holes && !(kCoopFlavour && spp == 1 && !two_rays && !order.wg_first)) {
        order.d = order.plain_d; order.n = order.plain_n; order.wg_first = order.plain_wg_first; order.n_wg = order.plain_n_wg;
        order.holes = nullptr; order.n_holes = 0;
    }
    const dim3 b(rt::kBlockThreads);
    const unsigned lds = (unsigned)std::max(0ll, knob(RT_DEBUG_LDS_BYTES));
    uint32_t *no_cost = nullptr;
    // rt_debug_wave_trace(<file>) (diagnostic, tools/wave_timeline.py): the launch records every wave's start / end /
    // placement and the records are written to <file> -- synchronous, one file per launch (overwritten).
    std::string trace_file;
#ifdef RT_TEST_HOOKS
    if (VAR & 8) { std::lock_guard<std::mutex> lk(g_trace_mu); trace_file = g_trace_path; }
#endif
    const char *trace_path = trace_file.empty() ? nullptr : trace_file.c_str();
    const dim3 rgrid(order.d ? (order.wg_first ? order.n_wg : order.n) : grid.x);      // render workgroups: one per descriptor, or dealt
    const size_t trace_words = (size_t)(order.d ? order.n : grid.x) * 4 * 8 * (use_split(spp) ? (size_t)spp * spp : 1);
    struct Trace {
        uint32_t *d = nullptr; const char *path; size_t words; hipStream_t stream;
        ~Trace()
        {
            if (!d) return;
            std::vector<uint32_t> h(words);
            if (hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(h.data(), d, words * 4, hipMemcpyDeviceToHost) == hipSuccess) {
                if (FILE *f = fopen(path, "wb")) { fwrite(h.data(), 4, words, f); fclose(f); }
            }
            (void)hipFree(d);
        }
    } tr{ nullptr, trace_path, trace_words, stream };
    if (trace_path && hipMalloc(&tr.d, trace_words * 4) == hipSuccess) {
        (void)hipMemsetAsync(tr.d, 0, trace_words * 4, stream);
        no_cost = tr.d;
    }
    rt::SampleBuf<T> sb{ nullptr, nullptr, (unsigned)total_px };
    const dim3 b2(rt::kSkip2Threads);
    if (!use_split(spp)) {
        if constexpr (kTwoRayFlavour) {
            if (two_rays) {
                count_event(RT_DEBUG_COUNT_TWO_RAY_LAUNCHES); g_launch_flags |= RT_LAUNCH_TWO_RAYS;
                hipLaunchKernelGGL((rt::k_render_skip2<rt::kSkipOne, (VAR & 16) != 0, (VAR & 4) != 0>), rgrid, b2, 0, stream, skip_view_of<float>(s), w, h, spp, d_tab, nt, d_out, sb, frame_w,
                                   order.d, order.wg_first);
                return RT_OK;
            }
        }
        // Steady-state frames -- f32, one sample per pixel, a dispatch list, the filtered assembly loops -- run the kernel that was written
        // around a wave's fixed costs (rt_skip_fast.hpp), with or without cooperative quads; everything else the generic one.
        if constexpr (!COUNT && sizeof(T) == 4 && ((VAR & ~8) == 19 || (VAR & ~8) == 23)) {
            if (spp == 1 && order.d && !order.wg_first && lds == 0 && knob(RT_DEBUG_FAST_KERNEL) != 0) {
                rt::FastArgs fa{};
                const rt::SkipView<float> sv = skip_view_of<float>(s);
                constexpr bool kFused = (VAR & 4) != 0;
                fa.order = order.d;
                fa.walk_prim = kFused ? sv.xfprim : sv.xprim;
                fa.width = w; fa.height = h; fa.nb = (kFused ? sv.n_fnodes : sv.n_nodes) * (unsigned)sizeof(rt::Node<float>); fa.frame_w = frame_w; fa.out = d_out;
                fa.eye[0] = sv.eye.x; fa.eye[1] = sv.eye.y; fa.eye[2] = sv.eye.z; fa.light[0] = sv.light.x; fa.light[1] = sv.light.y; fa.light[2] = sv.light.z;
                fa.items = sv.items; fa.own = sv.xown; fa.walk_shad = kFused ? sv.xfshad : sv.xshad; fa.exact_shad = kFused ? sv.fshad : sv.shad;
                memcpy(fa.fc, &s->fc, sizeof fa.fc);
                fa.trace = no_cost;
                g_launch_flags |= RT_LAUNCH_FAST_KERNEL;
                if (order.holes) {
                    fa.holes = order.holes; fa.n_holes = order.n_holes; fa.cv = s->coop;
                    count_event(RT_DEBUG_COUNT_COOP_LAUNCHES); g_launch_flags |= RT_LAUNCH_COOPERATIVE;
                    hipLaunchKernelGGL((rt::k_render_skip_fast_coop<(VAR & ~8), (VAR & 8) != 0>), rgrid, b, 0, stream, fa);
                } else hipLaunchKernelGGL((rt::k_render_skip_fast<(VAR & ~8), (VAR & 8) != 0>), rgrid, b, 0, stream, fa);
                return RT_OK;
            }
        }
        if constexpr (!COUNT && sizeof(T) == 4 && (VAR == 19 || VAR == 23 || VAR == 31)) {
            if (spp == 1 && order.d && order.holes && !order.wg_first) {        // some quads of the pass are walked cooperatively (rt_coop.hpp)
                count_event(RT_DEBUG_COUNT_COOP_LAUNCHES); g_launch_flags |= RT_LAUNCH_COOPERATIVE;
                hipLaunchKernelGGL((skip_kernel<T, COUNT, VAR, rt::kSkipOne, true>()), rgrid, b, lds, stream, 
                                   skip_args<T>(s, order.d, order.wg_first, w, h, frame_w, d_out, d_tab, nt, spp, cnt, no_cost, sb, s->coop, order.holes, order.n_holes));
                return RT_OK;
            }
        }
        if (spp == 1)
            hipLaunchKernelGGL((skip_kernel<T, COUNT, VAR, rt::kSkipOne>()), rgrid, b, lds, stream, 
                               skip_args<T>(s, order.d, order.wg_first, w, h, frame_w, d_out, d_tab, nt, spp, cnt, no_cost, sb));
        else
            hipLaunchKernelGGL((skip_kernel<T, COUNT, VAR, rt::kSkipLoop>()), rgrid, b, lds, stream, 
                               skip_args<T>(s, order.d, order.wg_first, w, h, frame_w, d_out, d_tab, nt, spp, cnt, no_cost, sb));
        return RT_OK;
    }
    const size_t ns = (size_t)spp * spp;
    {
        rt_status bst = ensure_sample_buffers(c, ns * total_px, sizeof(T));
        if (bst != RT_OK) return bst;
    }
    sb.gdot = static_cast<T *>(c->d_sample_gdot);
    sb.state = c->d_sample_state;
    const bool packed = packed_samples(spp);
    bool done2 = false;
    if constexpr (kTwoRayFlavour) {
        if (two_rays) {
            count_event(RT_DEBUG_COUNT_TWO_RAY_LAUNCHES); g_launch_flags |= RT_LAUNCH_TWO_RAYS;
            hipLaunchKernelGGL((rt::k_render_skip2<rt::kSkipPacked, (VAR & 16) != 0, (VAR & 4) != 0>), dim3(rgrid.x, (unsigned)ns), b2, 0, stream, skip_view_of<float>(s), w, h, spp, d_tab, nt,
                               d_out, sb, frame_w, order.d, order.wg_first);
            done2 = true;
        }
    }
    if (done2) {
    } else if (packed)
        hipLaunchKernelGGL((skip_kernel<T, COUNT, VAR, rt::kSkipPacked>()), dim3(rgrid.x, (unsigned)ns), b, lds, stream, 
                           skip_args<T>(s, order.d, order.wg_first, w, h, frame_w, d_out, d_tab, nt, spp, cnt, no_cost, sb));
    else
        hipLaunchKernelGGL((skip_kernel<T, COUNT, VAR, rt::kSkipSplit>()), dim3(rgrid.x, (unsigned)ns), b, lds, stream, 
                           skip_args<T>(s, order.d, order.wg_first, w, h, frame_w, d_out, d_tab, nt, spp, cnt, no_cost, sb));
    HIP_TRY(hipGetLastError());
    if constexpr (sizeof(T) == 4) {
        if (packed) {        // one word per sample, [pixel][sample] (rt_kernels.hpp sample_word)
            const uint4 *words = reinterpret_cast<const uint4 *>(sb.gdot);
            if (ns == 4) hipLaunchKernelGGL((rt::k_resolve_words<4>), grid, b, 0, stream, words, d_tab, nt, d_out, frame_w);
            else if (ns == 16) hipLaunchKernelGGL((rt::k_resolve_words<16>), grid, b, 0, stream, words, d_tab, nt, d_out, frame_w);
            else hipLaunchKernelGGL((rt::k_resolve_words<64>), grid, b, 0, stream, words, d_tab, nt, d_out, frame_w);
            return RT_OK;
        }
    }
    hipLaunchKernelGGL((rt::k_resolve_samples<T>), grid, b, 0, stream, sb, spp, d_tab, nt, d_out, frame_w, packed);
    return RT_OK;
}

template <typename T, bool COUNT>
rt_status launch_skip_var(const rt_scene *s, Context *c, dim3 grid, hipStream_t stream, unsigned w, unsigned h, unsigned spp,
                          const rt::TileDev *d_tab, unsigned nt, uint64_t total_px, uint8_t *d_out, rt::Counters *cnt, unsigned frame_w,
                          rt::BlockList order)
{
    // a counting launch always runs the C++ loops: the assembly bits would only duplicate kernels
    const int v = skip_variant(s);
    if constexpr (COUNT) {
#ifdef RT_TEST_HOOKS
        if (!(v & 1)) return launch_skip_one<T, true, 0>(s, c, grid, stream, w, h, spp, d_tab, nt, total_px, d_out, cnt, frame_w, order);
#endif
        return launch_skip_one<T, true, 1>(s, c, grid, stream, w, h, spp, d_tab, nt, total_px, d_out, cnt, frame_w, order);
    } else {
        switch (v) {
        // what a scene gets by itself: the filtered assembly loops, fused where the scene is concentric (f64 scenes too large for the
        // filter streams' 32-bit offsets: the unfiltered ones)
        case 19: return launch_skip_one<T, false, 19>(s, c, grid, stream, w, h, spp, d_tab, nt, total_px, d_out, cnt, frame_w, order);
        case 23: return launch_skip_one<T, false, 23>(s, c, grid, stream, w, h, spp, d_tab, nt, total_px, d_out, cnt, frame_w, order);
        case 3: if constexpr (sizeof(T) == 8) return launch_skip_one<T, false, 3>(s, c, grid, stream, w, h, spp, d_tab, nt, total_px, d_out, cnt, frame_w, order); else break;
        case 7: if constexpr (sizeof(T) == 8) return launch_skip_one<T, false, 7>(s, c, grid, stream, w, h, spp, d_tab, nt, total_px, d_out, cnt, frame_w, order); else break;
#ifdef RT_TEST_HOOKS
        // flavours only rt_debug.h's RT_DEBUG_SKIP_VARIANT / rt_debug_wave_trace can ask for
        case 0: return launch_skip_one<T, false, 0>(s, c, grid, stream, w, h, spp, d_tab, nt, total_px, d_out, cnt, frame_w, order);
        case 27: if constexpr (sizeof(T) == 4) return launch_skip_one<T, false, 27>(s, c, grid, stream, w, h, spp, d_tab, nt, total_px, d_out, cnt, frame_w, order); else break;
        case 31: if constexpr (sizeof(T) == 4) return launch_skip_one<T, false, 31>(s, c, grid, stream, w, h, spp, d_tab, nt, total_px, d_out, cnt, frame_w, order); else break;
        case 11: return launch_skip_one<T, false, 11>(s, c, grid, stream, w, h, spp, d_tab, nt, total_px, d_out, cnt, frame_w, order);
        case 15: return launch_skip_one<T, false, 15>(s, c, grid, stream, w, h, spp, d_tab, nt, total_px, d_out, cnt, frame_w, order);
#endif
        default: break;
        }
#ifdef RT_TEST_HOOKS
        if constexpr (sizeof(T) == 4) {
            if (v == 3) return launch_skip_one<T, false, 3>(s, c, grid, stream, w, h, spp, d_tab, nt, total_px, d_out, cnt, frame_w, order);
            if (v == 7) return launch_skip_one<T, false, 7>(s, c, grid, stream, w, h, spp, d_tab, nt, total_px, d_out, cnt, frame_w, order);
        }
        return launch_skip_one<T, false, 1>(s, c, grid, stream, w, h, spp, d_tab, nt, total_px, d_out, cnt, frame_w, order);
#else
        snprintf(g_err, sizeof g_err, "internal: no traversal loops for variant %d", v);      // (skip_variant cannot return anything else without a control)
        return RT_ERR_UNSUPPORTED;
#endif
    }
}

rt_status launch_skip(const rt_scene *s, Context *c, dim3 grid, hipStream_t stream, unsigned w, unsigned h, unsigned spp,
                      const rt::TileDev *d_tab, unsigned nt, uint64_t total_px, uint8_t *d_out, rt::Counters *cnt, unsigned frame_w,
                      rt::BlockList order)
{
    if (s->precision == RT_F32)
        return cnt ? launch_skip_var<float, true>(s, c, grid, stream, w, h, spp, d_tab, nt, total_px, d_out, cnt, frame_w, order)
                   : launch_skip_var<float, false>(s, c, grid, stream, w, h, spp, d_tab, nt, total_px, d_out, cnt, frame_w, order);
    return cnt ? launch_skip_var<double, true>(s, c, grid, stream, w, h, spp, d_tab, nt, total_px, d_out, cnt, frame_w, order)
               : launch_skip_var<double, false>(s, c, grid, stream, w, h, spp, d_tab, nt, total_px, d_out, cnt, frame_w, order);
}

rt_status ensure_flat(rt_scene *s)
{
    std::lock_guard<std::mutex> lk(s->flat_mu);
    if (s->flat_ready) return RT_OK;
    HIP_TRY(hipSetDevice(s->device));
    rt_status st = s->precision == RT_F32 ? upload_flat<float>(s, s->h_items.data()) : upload_flat<double>(s, s->h_items.data());
    if (st != RT_OK) {
        // a later call tries again from nothing: what this attempt had already allocated goes back (nothing was launched against it
        // that has not been waited for: the failing call was an allocation, a launch or the synchronise itself)
        (void)hipDeviceSynchronize(); (void)hipGetLastError();
        for (void **p : { &s->d_fprim, &s->d_fprim_rr, &s->d_fshad, &s->d_pf, &s->d_pe, &s->d_sg, &s->d_se, &s->d_f64_pf, &s->d_f64_sf, &s->d_f64_sg })
            if (*p) { (void)hipFree(*p); *p = nullptr; }
        return st;
    }
    s->flat_ready = true;
    std::vector<unsigned char>().swap(s->h_items);
    return RT_OK;
}

rt_status check_traversal(rt_scene *s, rt_traversal trav)
{
    if (trav != RT_TRAVERSAL_FLAT && trav != RT_TRAVERSAL_SKIP) {
        snprintf(g_err, sizeof g_err, "unknown traversal %d", (int)trav);
        return RT_ERR_INVALID_ARGUMENT;
    }
    if (trav == RT_TRAVERSAL_FLAT) return ensure_flat(s);
    if (trav == RT_TRAVERSAL_SKIP && s->n_nodes == 0) {
        snprintf(g_err, sizeof g_err, "the hierarchy (skip) traversal needs a scene created with subtree bounds");
        return RT_ERR_UNSUPPORTED;
    }
    return RT_OK;
}

// The render kernels of one pass.  c may be NULL when the pass needs no per-call device state (no counters, no
// sample buffers): then nothing but the kernel itself is enqueued.
rt_status launch_render(rt_scene *s, Context *c, const rt_options *o, rt_traversal trav, const rt::TileDev *d_tab, unsigned nt,
                        uint32_t total_blocks, uint64_t total_px, uint8_t *d_out, unsigned frame_w, hipStream_t stream, rt::Counters *cnt,
                        const rt::TileDev *d_tab16 = nullptr, uint32_t blocks16 = 0, rt::BlockList order = rt::BlockList{})
{
    const dim3 grid(total_blocks);
    const unsigned w = o->width, h = o->height, spp = o->samples_per_pixel;
    if (spp == 0) {
        // render.rs:219-250 with no sample to take: 0 * inf = NaN in every channel, and `NaN as u8` is 0 (set_pixel_from_vector, render.rs:96-108)
        g_launch_flags = cnt ? RT_LAUNCH_COUNTING : 0u;
        if (frame_w == 0) HIP_TRY(hipMemsetAsync(d_out, 0, (size_t)total_px * 4, stream));
        else hipLaunchKernelGGL(rt::k_zero_tiles, dim3(nt), dim3(rt::kBlockThreads), 0, stream, frame_w, d_tab, reinterpret_cast<unsigned *>(d_out));
        HIP_TRY(hipGetLastError());
        return RT_OK;
    }
    g_launch_flags = (trav == RT_TRAVERSAL_FLAT ? RT_LAUNCH_FLAT_PIPELINE : 0u) | (order.d ? RT_LAUNCH_ORDERED : 0u) |
                     (trav == RT_TRAVERSAL_SKIP && use_split(spp) ? RT_LAUNCH_SAMPLE_PARALLEL : 0u) | (cnt ? RT_LAUNCH_COUNTING : 0u);
    // a dispatch order that is being timed against others (pick_order); never a counting launch: its loops are different ones
    const bool timed = order.ev0 && order.ev1 && !cnt && trav == RT_TRAVERSAL_SKIP;
    if (timed) HIP_TRY(hipEventRecord(order.ev0, stream));
    struct Stop { hipEvent_t e; hipStream_t s; ~Stop() { if (e) (void)hipEventRecord(e, s); } } stop{ timed ? order.ev1 : nullptr, stream };
    if (trav == RT_TRAVERSAL_FLAT && d_tab16) {                     // wavefront pipeline (needs a context and the 16x16 table)
        rt_status fst = s->precision == RT_F32
            ? launch_flat_wavefront<float, 1024>(s, c, stream, w, h, spp, d_tab, nt, total_blocks, d_tab16, blocks16, total_px, d_out, cnt, frame_w)
            : launch_flat_wavefront<double, 512>(s, c, stream, w, h, spp, d_tab, nt, total_blocks, d_tab16, blocks16, total_px, d_out, cnt, frame_w);
        if (fst != RT_OK) return fst;
    } else if (trav == RT_TRAVERSAL_FLAT) {
        snprintf(g_err, sizeof g_err, "flat traversal launched without its resolve table");
        return RT_ERR_INVALID_ARGUMENT;
    } else {
        rt_status lst = launch_skip(s, c, grid, stream, w, h, spp, d_tab, nt, total_px, d_out, cnt, frame_w, order);
        if (lst != RT_OK) return lst;
    }
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

// Enqueues every kernel of one pass on `stream` through a leased context.  d_out must hold 4 * total_px bytes
// (tile-major) or the whole frame (frame_w != 0).
rt_status enqueue_pass(rt_scene *s, Context *c, const rt_options *o, rt_traversal trav, const std::vector<rt::TileDev> &tab,
                       uint32_t total_blocks, uint64_t total_px, uint8_t *d_out, unsigned frame_w, hipStream_t stream, bool want_counters,
                       const std::vector<rt::TileDev> *tab16 = nullptr, uint32_t blocks16 = 0, bool cacheable = true,
                       const rt::TileDev **d_tab_out = nullptr)       // the device copy of `tab` the pass was launched with
{
    const rt::TileDev *d_tab = nullptr, *d_tab16 = nullptr;
    rt::BlockList order;
    {
        // (will_be_timed: launch_render records the trial's event pair -- a pass without samples launches nothing and records none)
        rt_status ust = trav == RT_TRAVERSAL_SKIP ? device_table(s, c, tab, stream, &d_tab, 0, o, &order, cacheable, !want_counters && o->samples_per_pixel != 0)
                                                  : device_table(s, c, tab, stream, &d_tab, 0, nullptr, nullptr, cacheable);
        if (ust != RT_OK) return ust;
        if (tab16) {
            if ((ust = device_table(s, c, *tab16, stream, &d_tab16, 1, nullptr, nullptr, cacheable)) != RT_OK) return ust;
        }
        if (d_tab_out) *d_tab_out = d_tab;
    }
    if (want_counters) {
        HIP_TRY(hipMemsetAsync(c->d_counters, 0, sizeof(rt::Counters) * rt::kCounterStripes, stream));
        HIP_TRY(hipEventRecord(c->ev0, stream));
    }
    rt_status st = launch_render(s, c, o, trav, d_tab, (unsigned)tab.size(), total_blocks, total_px, d_out, frame_w, stream,
                                 want_counters ? c->d_counters : nullptr, d_tab16, blocks16, order);
    if (st != RT_OK) return st;
    HIP_TRY(hipEventRecord(c->ev1, stream));
    return RT_OK;
}

rt_status read_stats(rt_scene *s, Context *c, hipStream_t stream, rt_traversal trav, rt_stats *st)
{
    std::vector<rt::Counters> stripes(rt::kCounterStripes);
    HIP_TRY(hipMemcpyAsync(stripes.data(), c->d_counters, sizeof(rt::Counters) * rt::kCounterStripes, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    rt::Counters h{};
    for (const rt::Counters &k : stripes) {
        h.primary += k.primary; h.hits += k.hits; h.shadow += k.shadow; h.occluded += k.occluded;
        h.sphere_tests += k.sphere_tests; h.bound_tests += k.bound_tests; h.wave_steps += k.wave_steps;
        h.max_wave_steps = std::max(h.max_wave_steps, k.max_wave_steps);
        h.max_wave_cycles = std::max(h.max_wave_cycles, k.max_wave_cycles);
        h.max_wave_ref100mhz = std::max(h.max_wave_ref100mhz, k.max_wave_ref100mhz);
        h.wave_item_steps += k.wave_item_steps;
        h.filter_pass += k.filter_pass; h.filter_violations += k.filter_violations; h.primary_tests += k.primary_tests;
    }
    count_event(RT_DEBUG_COUNT_FILTER_PASS, (long long)h.filter_pass);
    count_event(RT_DEBUG_COUNT_FILTER_VIOLATIONS, (long long)h.filter_violations);
    count_store(RT_DEBUG_COUNT_PRIMARY_TESTS, (long long)h.primary_tests);
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    st->primary = h.primary; st->hits = h.hits; st->shadow = h.shadow; st->occluded = h.occluded;
    st->primary_tests = trav == RT_TRAVERSAL_FLAT ? h.primary * (uint64_t)s->n_items : h.primary_tests;
    if (trav == RT_TRAVERSAL_FLAT) {
        st->sphere_tests = (h.primary + h.shadow) * (uint64_t)s->n_items;
        st->bound_tests = 0;
        st->tests_executed = st->sphere_tests;
        if (c->d_queues) {                                          // the shadow queues' lengths say what actually ran
            rt::FlatQueues q{};
            HIP_TRY(hipMemcpy(&q, c->d_queues, sizeof q, hipMemcpyDeviceToHost));
            const uint64_t chunk = c->flat_first_pass_items, n = s->n_items;
            st->tests_executed = h.primary * n + (uint64_t)q.n1 * std::min<uint64_t>(chunk, n) + (uint64_t)q.n2 * (n > chunk ? n - chunk : 0);
        }
    } else {
        st->sphere_tests = h.sphere_tests; st->bound_tests = h.bound_tests;
        st->tests_executed = h.sphere_tests + h.bound_tests;
    }
    if (knob(RT_DEBUG_PRINT_STEPS) > 0)
        fprintf(stderr, "[rtrace_hip] wave_steps %llu (%llu at ITEM nodes) max_wave_steps %llu longest wave: %llu cycles, %.2f us, %.0f MHz\n",
                h.wave_steps, h.wave_item_steps, h.max_wave_steps, h.max_wave_cycles, h.max_wave_ref100mhz / 100.0,
                h.max_wave_ref100mhz ? 100.0 * h.max_wave_cycles / h.max_wave_ref100mhz : 0.0);
    st->device_ms = ms;
    st->longest_wave_cycles = trav == RT_TRAVERSAL_FLAT ? 0 : h.max_wave_cycles; st->longest_wave_ref100mhz = trav == RT_TRAVERSAL_FLAT ? 0 : h.max_wave_ref100mhz;
    return RT_OK;
}

bool check_common(rt_scene *s, const rt_options *o, const rt_region *tiles, uint32_t n, const void *out)
{
    if (!s || !o || !tiles || !out || n == 0) { snprintf(g_err, sizeof g_err, "NULL argument or n_tiles == 0"); return false; }
    if (o->width == 0 || o->height == 0) {       // (samples_per_pixel == 0 is the reference's black frame: launch_render)
        snprintf(g_err, sizeof g_err, "width and height must be >= 1");
        return false;
    }
    return true;
}

template <typename T>
bool items_valid(const void *p, uint32_t n, bool need_positive_radius)
{
    const T *v = static_cast<const T *>(p);
    for (uint64_t i = 0; i < (uint64_t)n * 4; ++i) {
        if (!std::isfinite(v[i]) || std::fabs((double)v[i]) > 1e15) return false;
        if (need_positive_radius && (i & 3) == 3 && !(v[i] > T(0))) return false;
    }
    return true;
}

}  // namespace

extern "C" {

int rt_abi_version(void) { return RTRACE_HIP_ABI_VERSION; }

#ifdef RT_TEST_HOOKS
rt_status rt_debug_set(int key, long long value)
{
    if (key < 0 || key >= RT_DEBUG_KEYS) { snprintf(g_err, sizeof g_err, "rt_debug_set: unknown key %d", key); return RT_ERR_INVALID_ARGUMENT; }
    g_knob[key].store(value < 0 ? -1 : value, std::memory_order_relaxed);
    return RT_OK;
}

long long rt_debug_count(int counter)
{
    return counter >= 0 && counter < RT_DEBUG_COUNTERS ? g_count[counter].load(std::memory_order_relaxed) : -1;
}

// Test infrastructure (csrc/rt_debug.h): the flat scan's conservative filter against the exact discriminant, for every primary
// ray of a width x height x spp frame and every item.  counts: {disc >= 0, bound >= 0, disc >= 0 && bound < 0} primary, then shadow.
rt_status rt_debug_flat_filter_check(rt_scene *s, uint32_t width, uint32_t height, uint32_t spp, unsigned long long counts[6])
{
    if (!s || !counts || !width || !height || !spp) return RT_ERR_INVALID_ARGUMENT;
    if (rt_status fst = ensure_flat(s); fst != RT_OK) return fst;
    HIP_TRY(hipSetDevice(s->device));
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc(&d, 6 * sizeof(unsigned long long)));
    struct Free { unsigned long long *p; ~Free() { (void)hipFree(p); } } fr{ d };
    HIP_TRY(hipMemset(d, 0, 6 * sizeof(unsigned long long)));
    const uint64_t px = (uint64_t)width * height;
    if (s->precision != RT_F32) {
        hipLaunchKernelGGL(rt::k_flat_filter_check_f64, dim3((unsigned)((px + rt::kBlockThreads - 1) / rt::kBlockThreads), spp * spp), dim3(rt::kBlockThreads), 0,
                           nullptr, flat_view_of<double>(s), flat_f64_view_of(s), (const unsigned *)nullptr, width, height, spp, d);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(counts, d, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        return RT_OK;
    }
    hipLaunchKernelGGL(rt::k_flat_filter_check, dim3((unsigned)((px + rt::kBlockThreads - 1) / rt::kBlockThreads), spp * spp), dim3(rt::kBlockThreads), 0,
                       nullptr, flat_sc_view_of(s), width, height, spp, d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(counts, d, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RT_OK;
}

rt_status rt_debug_wave_trace(const char *path)
{
    std::lock_guard<std::mutex> lk(g_trace_mu);
    g_trace_path = path ? path : "";
    g_trace_on.store(!g_trace_path.empty(), std::memory_order_relaxed);
    return RT_OK;
}

#endif  // RT_TEST_HOOKS

const char *rt_last_error_message(void) { return g_err; }

uint32_t rt_last_launch_flags(void) { return g_launch_flags; }

#ifndef RT_BUILD_INFO
#define RT_BUILD_INFO "unknown toolchain (built without csrc/Makefile)"
#endif
const char *rt_build_info(void)
{
#ifdef RT_TEST_HOOKS
    return RT_BUILD_INFO " | RT_TEST_HOOKS";
#else
    return RT_BUILD_INFO;
#endif
}

const char *rt_strerror(rt_status st)
{
    switch (st) {
    case RT_OK: return "ok";
    case RT_ERR_INVALID_ARGUMENT: return "invalid argument";
    case RT_ERR_INVALID_REGION: return "image region empty or outside the image";
    case RT_ERR_NO_DEVICE: return "no usable gfx950 device";
    case RT_ERR_HIP: return "HIP runtime or kernel failure";
    case RT_ERR_OUT_OF_MEMORY: return "out of memory";
    case RT_ERR_UNSUPPORTED: return "unsupported request";
    }
    return "unknown status";
}

rt_status rt_device_count(int *n)
{
    if (!n) return RT_ERR_INVALID_ARGUMENT;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess || c <= 0) {
        *n = 0;
        snprintf(g_err, sizeof g_err, "hipGetDeviceCount: %s", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        (void)hipGetLastError();
        return RT_ERR_NO_DEVICE;
    }
    *n = c;
    return RT_OK;
}

uint64_t rt_tiles_rgba_bytes(const rt_region *tiles, uint32_t n_tiles)
{
    if (!tiles) return 0;
    uint64_t px = 0;
    for (uint32_t i = 0; i < n_tiles; ++i) {
        if (!(tiles[i].l < tiles[i].r && tiles[i].b < tiles[i].t)) return 0;
        px += (uint64_t)(tiles[i].r - tiles[i].l) * (tiles[i].t - tiles[i].b);
    }
    return px * 4;
}

rt_status rt_scene_create(int device, rt_precision precision, const void *dfs_items, uint32_t n_items,
                          const void *light_unit, const void *eye, const void *bounds, const rt_range *ranges,
                          uint32_t n_bounds, rt_scene **out)
{
    if (out) *out = nullptr;
    if (!out || !dfs_items || !light_unit || !eye || n_items == 0 || (precision != RT_F32 && precision != RT_F64)) {
        snprintf(g_err, sizeof g_err, "rt_scene_create: NULL argument, n_items == 0 or bad precision");
        return RT_ERR_INVALID_ARGUMENT;
    }
    if ((n_bounds != 0) != (bounds != nullptr && ranges != nullptr)) {
        snprintf(g_err, sizeof g_err, "rt_scene_create: bounds, ranges and n_bounds must be given together");
        return RT_ERR_INVALID_ARGUMENT;
    }
    const bool f32 = precision == RT_F32;
    const bool ok = f32 ? (items_valid<float>(dfs_items, n_items, true) && (!n_bounds || items_valid<float>(bounds, n_bounds, false)))
                        : (items_valid<double>(dfs_items, n_items, true) && (!n_bounds || items_valid<double>(bounds, n_bounds, false)));
    if (!ok) {
        snprintf(g_err, sizeof g_err, "rt_scene_create: items must be finite, |v| <= 1e15, radius > 0");
        return RT_ERR_INVALID_ARGUMENT;
    }
    for (uint32_t i = 0; i < n_bounds; ++i) {
        if (ranges[i].first < 0 || ranges[i].count < 0 || (uint64_t)ranges[i].first + (uint64_t)ranges[i].count > n_items) {
            snprintf(g_err, sizeof g_err, "rt_scene_create: range %u outside the item array", i);
            return RT_ERR_INVALID_ARGUMENT;
        }
    }
    int ndev = 0;
    rt_status st = rt_device_count(&ndev);
    if (st != RT_OK) return st;
    if (device < 0 || device >= ndev) {
        snprintf(g_err, sizeof g_err, "device %d out of range (%d visible)", device, ndev);
        return RT_ERR_NO_DEVICE;
    }
    StageClock clk;
    HIP_TRY(hipSetDevice(device));
    clk.lap("scene: hipSetDevice");

    std::unique_ptr<rt_scene> s(new (std::nothrow) rt_scene());
    if (!s) return RT_ERR_OUT_OF_MEMORY;
    s->device = device; s->precision = precision; s->n_items = n_items; s->n_bounds = n_bounds;
    const size_t esz = f32 ? sizeof(float) : sizeof(double);
    for (int k = 0; k < 3; ++k) {
        s->light[k] = f32 ? (double)static_cast<const float *>(light_unit)[k] : static_cast<const double *>(light_unit)[k];
        s->eye[k] = f32 ? (double)static_cast<const float *>(eye)[k] : static_cast<const double *>(eye)[k];
        // Bounds that keep every intermediate of primitive.rs:55-72 finite in f32 (squares of sums of coordinates stay below
        // 2e33), so no inf - inf and no NaN can arise anywhere on the path: |eye| <= 1e15 like the items, and
        // light_unit is a unit vector by contract (|component| <= 2 leaves room for rounding).
        if (!std::isfinite(s->light[k]) || !std::isfinite(s->eye[k]) || std::fabs(s->eye[k]) > 1e15 || std::fabs(s->light[k]) > 2.0) {
            snprintf(g_err, sizeof g_err, "rt_scene_create: eye must be finite with |coordinate| <= 1e15, light_unit a unit vector");
            return RT_ERR_INVALID_ARGUMENT;
        }
    }
    {
        // a unit vector as the host's `normalized` leaves it: the flat scan's shadow filter (rt_flat_sc.hpp) bounds its rounding
        // errors with |light_unit| <= 1 + 1e-3
        const double l2 = s->light[0] * s->light[0] + s->light[1] * s->light[1] + s->light[2] * s->light[2];
        if (std::fabs(l2 - 1.0) > 2e-3) {
            snprintf(g_err, sizeof g_err, "rt_scene_create: light_unit must be a unit vector (its squared length is %.6g)", l2);
            return RT_ERR_INVALID_ARGUMENT;
        }
    }
    auto fail = [&](rt_status code) { rt_scene_destroy(s.release()); return code; };
    hipError_t e;
    s->h_items.assign(static_cast<const unsigned char *>(dfs_items), static_cast<const unsigned char *>(dfs_items) + esz * 4 * n_items);
    // ONE stream carries everything this call enqueues (uploads, the kernels that derive the streams, the cost map's counting render) and is
    // the first context's stream afterwards: the null stream is never touched
    if ((e = hipStreamCreateWithFlags(&s->cost_stream, hipStreamNonBlocking)) != hipSuccess) return fail(hip_fail(e, "hipStreamCreate(scene)", __LINE__));
    clk.lap("scene: stream");
    if ((e = hipMalloc(&s->d_items, esz * 4 * n_items)) != hipSuccess) return fail(hip_fail(e, "hipMalloc(items)", __LINE__));
    {
        // everything this call uploads goes through ONE pinned arena and k_upload_words: items, the raw streams (plain + compacted: a node per
        // item and per bound, twice), the cooperative copy's tables, the filter's constants -- no hipMemcpy on the way to the first frame
        const size_t nodes = (size_t)n_items + n_bounds;
        const size_t raw_sz = f32 ? sizeof(rt::RawNode<float>) : sizeof(rt::RawNode<double>);
        const size_t want = esz * 4 * n_items + 2 * nodes * raw_sz + nodes * (sizeof(uint32_t) + sizeof(uint2)) + 64 * 1024;
        if (want <= ((size_t)1 << 30) && hipHostMalloc(reinterpret_cast<void **>(&s->h_up), want, hipHostMallocDefault) == hipSuccess) { s->up_cap = want; s->up_used = 0; }
        else { (void)hipGetLastError(); s->h_up = nullptr; }
    }
    { rt_status ust = scene_upload(s.get(), s->d_items, s->h_items.data(), esz * 4 * n_items); if (ust != RT_OK) return fail(ust); }      // (the scene's own copy of the items)
    if ((e = hipStreamSynchronize(s->cost_stream)) != hipSuccess) return fail(hip_fail(e, "upload(items)", __LINE__));
    clk.lap("scene: items");
    if (n_bounds) {
        rt_status sst = f32 ? upload_streams<float>(s.get(), dfs_items, bounds, ranges) : upload_streams<double>(s.get(), dfs_items, bounds, ranges);
        if (sst != RT_OK) return fail(sst);
        clk.lap("scene: streams (total)");
        // the cost map the dispatch orders are made from is rendered when a tile list first wants orders (cost_map_of, from the scene's worker
        // thread); here only its pinned host side, which is also where new lists' tile tables are staged (failing only costs the ordering)
        if (alloc_cost_host(s.get()) != RT_OK) { (void)hipGetLastError(); s->h_cost = nullptr; s->h_tab_stage = nullptr; }
        try { s->worker = std::thread(worker_main, s.get()); note_builder(s.get()); } catch (...) {}      // (without it a new list starts a thread of its own)
        clk.lap("scene: pinned cost arena + worker");
    }
    // (every upload has been consumed: derive_streams and upload_coop synchronise the stream behind their kernels)
    if (s->h_up) { (void)hipStreamSynchronize(s->cost_stream); (void)hipHostFree(s->h_up); s->h_up = nullptr; s->up_cap = s->up_used = 0; }
    *out = s.release();
    return RT_OK;
}

rt_status rt_scene_destroy(rt_scene *s)
{
    if (!s) return RT_OK;
    (void)hipSetDevice(s->device);
    for (std::thread &b : s->builders) if (b.joinable()) b.join();          // dispatch orders still being made in the background
    stop_worker(s);
    forget_scene(s);
    if (s->ahead.stream) { (void)hipStreamSynchronize(s->ahead.stream); (void)hipStreamDestroy(s->ahead.stream); }      // a pass rendered ahead may still be running
    s->pool.clear();
    for (auto &t : s->tables) { (void)hipFree(t.dev); for (auto &od : t.orders) release_order(od); if (t.order_arena) (void)hipFree(t.order_arena); if (t.landed) (void)hipEventDestroy(t.landed); }
    if (s->d_items) (void)hipFree(s->d_items);
    if (s->d_prim) (void)hipFree(s->d_prim);
    if (s->d_shad) (void)hipFree(s->d_shad);
    if (s->d_cprim) (void)hipFree(s->d_cprim);
    if (s->d_cshad) (void)hipFree(s->d_cshad);
    for (void *p : { s->d_xprim, s->d_xshad, s->d_xcprim, s->d_xcshad, s->d_xown, s->d_fc, s->d_coop_prim, s->d_coop_shad, s->d_cost_arena }) if (p) (void)hipFree(p);
    if (s->cost_stream) { (void)hipStreamSynchronize(s->cost_stream); (void)hipStreamDestroy(s->cost_stream); }
    if (s->h_cost) (void)hipHostFree(s->h_cost);
    if (s->h_up) (void)hipHostFree(s->h_up);
    if (s->ahead.ev) (void)hipEventDestroy(s->ahead.ev);
    if (s->ahead.h) (void)rt_host_free(s->ahead.h);
    if (s->ahead.h_next) (void)rt_host_free(s->ahead.h_next);
    if (s->d_fprim) (void)hipFree(s->d_fprim);
    if (s->d_fprim_rr) (void)hipFree(s->d_fprim_rr);
    if (s->d_fshad) (void)hipFree(s->d_fshad);
    for (void *p : { s->d_f64_pf, s->d_f64_sf, s->d_f64_sg }) if (p) (void)hipFree(p);
    if (s->d_pf) (void)hipFree(s->d_pf);
    if (s->d_pe) (void)hipFree(s->d_pe);
    if (s->d_sg) (void)hipFree(s->d_sg);
    if (s->d_se) (void)hipFree(s->d_se);
    delete s;
    return RT_OK;
}

// Shared body of rt_render_tiles_device (frame_w == 0, tile-major output) and rt_render_frame_device (row-major frame).
static rt_status render_device(rt_scene *s, const rt_options *o, rt_traversal trav, const rt_region *tiles, uint32_t n, void *out_device,
                               unsigned frame_w, void *hip_stream, rt_stats *stats)
{
    if (!check_common(s, o, tiles, n, out_device)) return RT_ERR_INVALID_ARGUMENT;
    if ((reinterpret_cast<uintptr_t>(out_device) & 3u) != 0) {
        snprintf(g_err, sizeof g_err, "the device output buffer must be 4-byte aligned");
        return RT_ERR_INVALID_ARGUMENT;
    }
    rt_status st = check_traversal(s, trav);
    if (st != RT_OK) return st;
    std::vector<rt::TileDev> tab;
    uint64_t total_px = 0; uint32_t total_blocks = 0;
    const bool flat2 = trav == RT_TRAVERSAL_FLAT;
    st = build_tile_table(o, tiles, n, tab, &total_px, &total_blocks, flat2 ? rt::kFlatBlockW : rt::kBlockW, flat2 ? rt::kFlatBlockH : rt::kBlockH);
    if (st != RT_OK) return st;
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    uint8_t *out = static_cast<uint8_t *>(out_device);
    // the flat wavefront pipeline resolves per pixel with the 16x16-block table
    const bool wavefront = flat2;
    std::vector<rt::TileDev> tab16;
    uint32_t blocks16 = 0;
    if (wavefront) {
        uint64_t px16 = 0;
        if ((st = build_tile_table(o, tiles, n, tab16, &px16, &blocks16)) != RT_OK) return st;
    }
    const bool split = trav == RT_TRAVERSAL_SKIP && use_split(o->samples_per_pixel);
    if (!stats && !split && !wavefront) {
        // Fast path: a cached tile table and no per-call device state -> the call enqueues exactly one kernel.
        const rt::TileDev *d_tab = nullptr;
        rt::BlockList order;
        if ((st = device_table(s, nullptr, tab, stream, &d_tab, 0, o, &order, true, trav == RT_TRAVERSAL_SKIP && o->samples_per_pixel != 0)) != RT_OK) return st;
        if (d_tab)
            return launch_render(s, nullptr, o, trav, d_tab, (unsigned)tab.size(), total_blocks, total_px, out, frame_w, stream, nullptr, nullptr, 0, order);
    }
    Context *c = nullptr;
    if ((st = acquire(s, &c)) != RT_OK) return st;
    Lease lease{ s, c };
    st = enqueue_pass(s, c, o, trav, tab, total_blocks, total_px, out, frame_w, stream, stats != nullptr, wavefront ? &tab16 : nullptr, blocks16);
    if (st != RT_OK) {
        // some kernels of the pass may already be enqueued and using the context's buffers: it goes back to the pool marked
        // in flight behind everything that is on the stream now
        (void)hipEventRecord(c->ev1, stream);
        (void)hipGetLastError();
        lease.inflight = true;
        return st;
    }
    if (stats) return read_stats(s, c, stream, trav, stats);
    // Asynchronous return: the context's buffers are still in use by the enqueued work, so it goes back to the pool
    // marked in-flight and is only reused once its end event has completed.
    lease.inflight = true;
    return RT_OK;
}

rt_status rt_render_tiles_device(rt_scene *s, const rt_options *o, rt_traversal trav, const rt_region *tiles, uint32_t n,
                                 void *rgba_out_device, void *hip_stream, rt_stats *stats)
{
    return render_device(s, o, trav, tiles, n, rgba_out_device, 0u, hip_stream, stats);
}

rt_status rt_render_frame_device(rt_scene *s, const rt_options *o, rt_traversal trav, const rt_region *tiles, uint32_t n,
                                 void *frame_rgba_device, void *hip_stream, rt_stats *stats)
{
    return render_device(s, o, trav, tiles, n, frame_rgba_device, o ? (unsigned)o->width : 0u, hip_stream, stats);
}

// How the bytes of a pass get into the caller's HOST buffer.
//   pinned   (rt_host_alloc / rt_host_register memory, or any hipHostMalloc'd / registered range): the render kernel stores
//            straight into it over PCIe (no device copy of the frame, no separate D2H) or, as an alternative, renders
//            into device memory followed by ONE asynchronous D2H at full link speed;
//   pageable (Vec<u8>, malloc): the runtime has to bounce through pinned memory and a CPU copy whatever we do; one
//            hipMemcpyAsync to the caller's pointer, or our own pinned staging in 1 MiB chunks with the CPU copy of chunk k
//            overlapping the DMA of chunk k+1.
//   scattered (the merged rt_render_region passes: every tile has its own destination): the kernel stores into the context's
//            pinned staging and the CPU hands each caller its 16 KB.
enum HostCopy { kCopyAuto = 0, kCopyDirect = 1, kCopyStaged = 2, kCopyZero = 3, kCopyZeroStaged = 4 };

struct HostDest { bool pinned = false; uint8_t *dev_alias = nullptr; bool bad = false; size_t room = 0; };

// Host ranges this library pinned itself (rt_host_alloc / rt_host_register), base -> {bytes, device alias}.  Only these are
// written by the render kernel directly.  Asking the runtime instead (hipPointerGetAttributes) is not safe: it also reports
// ranges it locked on its own for an earlier pageable copy, and such a record can outlive the caller's buffer -- a kernel
// store to it is a GPU memory fault (seen as an intermittent fault on freshly allocated numpy buffers).
struct PinnedRange { size_t bytes; uint8_t *alias; };
static std::mutex g_pinned_mu;
static std::map<uintptr_t, PinnedRange> g_pinned;

static HostDest classify_host_pointer(const void *p)
{
    HostDest d;
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    {
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        auto it = g_pinned.upper_bound(a);
        if (it != g_pinned.begin()) {
            --it;
            if (a - it->first < it->second.bytes) {
                d.pinned = true;
                d.dev_alias = it->second.alias ? it->second.alias + (a - it->first) : nullptr;
                d.room = it->second.bytes - (a - it->first);
                return d;
            }
        }
    }
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return d; }      // plain pageable memory
    if (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeArray) d.bad = true;            // a device pointer is a caller error here
    return d;
}

constexpr size_t kStageChunk = 1u << 20;

// rt_render_tiles for a list of tiles whose bytes go to host memory.  `scatter` (optional, n entries): tile i's bytes go to
// scatter[i] instead of lying back to back at rgba_out (the coalesced rt_render_region path).
static rt_status render_tiles_host(rt_scene *s, const rt_options *o, rt_traversal trav, const rt_region *tiles, uint32_t n,
                                   uint8_t *rgba_out, uint8_t *const *scatter, rt_stats *stats, bool cacheable)
{
    rt_status st = check_traversal(s, trav);
    if (st != RT_OK) return st;
    std::vector<rt::TileDev> tab;
    uint64_t total_px = 0; uint32_t total_blocks = 0;
    const bool flat2 = trav == RT_TRAVERSAL_FLAT;
    st = build_tile_table(o, tiles, n, tab, &total_px, &total_blocks, flat2 ? rt::kFlatBlockW : rt::kBlockW,
                                    flat2 ? rt::kFlatBlockH : rt::kBlockH);
    if (st != RT_OK) return st;
    HIP_TRY(hipSetDevice(s->device));
    HostDest dest;
    if (!scatter) {
        dest = classify_host_pointer(rgba_out);
        if (dest.bad) {
            snprintf(g_err, sizeof g_err, "rt_render_tiles: rgba_out is device memory; use rt_render_tiles_device");
            return RT_ERR_INVALID_ARGUMENT;
        }
    }
    long long mode = knob(RT_DEBUG_HOST_COPY);
    if (mode <= 0) mode = dest.pinned ? (dest.dev_alias ? kCopyZero : kCopyDirect) : kCopyDirect;
    if (scatter) mode = kCopyZeroStaged;
    if (mode == kCopyZero && (!dest.dev_alias || dest.room < (size_t)total_px * 4)) mode = kCopyDirect;
    if (mode == kCopyStaged && dest.pinned) mode = kCopyDirect;          // staging a pinned destination is pointless

    Context *c = nullptr;
    if ((st = acquire(s, &c)) != RT_OK) return st;
    Lease lease{ s, c };
    const size_t bytes = (size_t)total_px * 4;
    uint8_t *d_target = nullptr;
    if (mode == kCopyStaged || mode == kCopyZeroStaged) {
        if (c->h_out_cap < bytes) {
            if (c->h_out) HIP_TRY(hipHostFree(c->h_out));
            c->h_out = nullptr; c->h_out_cap = 0;
            HIP_TRY(hipHostMalloc(&c->h_out, std::max(bytes, (size_t)1 << 20), hipHostMallocDefault));
            c->h_out_cap = std::max(bytes, (size_t)1 << 20);
        }
    }
    if (mode == kCopyZero) {
        d_target = dest.dev_alias;
    } else if (mode == kCopyZeroStaged) {
        void *alias = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&alias, c->h_out, 0));
        d_target = static_cast<uint8_t *>(alias);
    } else {
        if (c->out_cap < bytes) {
            if (c->d_out) HIP_TRY(hipFree(c->d_out));
            c->d_out = nullptr; c->out_cap = 0;
            HIP_TRY(hipMalloc(&c->d_out, bytes));
            c->out_cap = bytes;
        }
        d_target = c->d_out;
    }
    std::vector<rt::TileDev> tab16;
    uint32_t blocks16 = 0;
    const bool wavefront = flat2;
    if (wavefront) {
        uint64_t px16 = 0;
        if ((st = build_tile_table(o, tiles, n, tab16, &px16, &blocks16)) != RT_OK) return st;
    }
    st = enqueue_pass(s, c, o, trav, tab, total_blocks, total_px, d_target, 0u, c->stream, stats != nullptr, wavefront ? &tab16 : nullptr, blocks16,
                      cacheable);
    // from here on kernels of this pass may be running: an error return first waits for them (they write d_out / h_out or the caller's
    // pinned buffer), so that the context is not handed to the next caller with work in flight
#define HIP_DRAIN(expr)                                                                                                   \
    do {                                                                                                                  \
        hipError_t e__ = (expr);                                                                                          \
        if (e__ != hipSuccess) { (void)hipStreamSynchronize(c->stream); return hip_fail(e__, #expr, __LINE__); }          \
    } while (0)
    if (st != RT_OK) { (void)hipStreamSynchronize(c->stream); (void)hipGetLastError(); return st; }
    if (mode == kCopyDirect) {
        HIP_DRAIN(hipMemcpyAsync(rgba_out, c->d_out, bytes, hipMemcpyDeviceToHost, c->stream));
    } else if (mode == kCopyStaged) {
        const size_t chunks = (bytes + kStageChunk - 1) / kStageChunk;
        while (c->chunk_ev.size() < chunks) {
            hipEvent_t e = nullptr;
            HIP_DRAIN(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            c->chunk_ev.push_back(e);
        }
        for (size_t k = 0; k < chunks; ++k) {
            const size_t off = k * kStageChunk, len = std::min(kStageChunk, bytes - off);
            HIP_DRAIN(hipMemcpyAsync(c->h_out + off, c->d_out + off, len, hipMemcpyDeviceToHost, c->stream));
            HIP_DRAIN(hipEventRecord(c->chunk_ev[k], c->stream));
        }
        // the CPU copy of chunk k runs while the DMA engine moves chunk k + 1
        size_t tile = 0, tile_off = 0;                    // scatter cursor: current tile and bytes of it already delivered
        for (size_t k = 0; k < chunks; ++k) {
            HIP_DRAIN(hipEventSynchronize(c->chunk_ev[k]));
            const size_t off = k * kStageChunk, len = std::min(kStageChunk, bytes - off);
            if (!scatter) { memcpy(rgba_out + off, c->h_out + off, len); continue; }
            size_t pos = off;
            while (pos < off + len) {
                const size_t tbytes = (size_t)(tiles[tile].r - tiles[tile].l) * (tiles[tile].t - tiles[tile].b) * 4;
                const size_t take = std::min(tbytes - tile_off, off + len - pos);
                memcpy(scatter[tile] + tile_off, c->h_out + pos, take);
                pos += take; tile_off += take;
                if (tile_off == tbytes) { ++tile; tile_off = 0; }
            }
        }
    }
    rt_status rst = RT_OK;
    if (stats) rst = read_stats(s, c, c->stream, trav, stats);          // synchronises the stream
    else HIP_DRAIN(hipStreamSynchronize(c->stream));
#undef HIP_DRAIN
    if (rst == RT_OK && mode == kCopyZeroStaged) {
        size_t off = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const size_t tbytes = (size_t)(tiles[i].r - tiles[i].l) * (tiles[i].t - tiles[i].b) * 4;
            memcpy(scatter ? scatter[i] : rgba_out + off, c->h_out + off, tbytes);
            off += tbytes;
        }
    }
    return rst;
}

rt_status rt_render_tiles(rt_scene *s, const rt_options *o, rt_traversal trav, const rt_region *tiles, uint32_t n,
                          uint8_t *rgba_out, rt_stats *stats)
{
    if (!check_common(s, o, tiles, n, rgba_out)) return RT_ERR_INVALID_ARGUMENT;
    return render_tiles_host(s, o, trav, tiles, n, rgba_out, nullptr, stats, true);
}


// rt_render_tiles with delivery in completion order: the list is cut into batches that are ALL enqueued at once (kernels storing
// into pinned staging), and each batch's buckets are handed to the callback as soon as that batch's event has fired -- while the
// later batches are still rendering.  What render.rs:301-307 does with its channel, without serialising launches behind host calls.
rt_status rt_render_tiles_stream(rt_scene *s, const rt_options *o, rt_traversal trav, const rt_region *tiles, uint32_t n,
                                 rt_tile_callback callback, void *user)
{
    if (!callback) { snprintf(g_err, sizeof g_err, "rt_render_tiles_stream: NULL callback"); return RT_ERR_INVALID_ARGUMENT; }
    if (!check_common(s, o, tiles, n, tiles)) return RT_ERR_INVALID_ARGUMENT;
    rt_status st = check_traversal(s, trav);
    if (st != RT_OK) return st;
    const bool flat2 = trav == RT_TRAVERSAL_FLAT;
    // batches: at least kStreamBatch buckets (enough workgroups to fill the device), at most kStreamMaxBatches of them (each batch keeps
    // a cached tile table on the device)
    constexpr uint32_t kStreamBatch = 64, kStreamMaxBatches = 16;
    const uint32_t per = std::max(kStreamBatch, (n + kStreamMaxBatches - 1) / kStreamMaxBatches), n_batches = (n + per - 1) / per;
    struct Batch { std::vector<rt::TileDev> tab, tab16; uint64_t px = 0; uint32_t blocks = 0, blocks16 = 0; size_t byte_off = 0; };
    std::vector<Batch> batches(n_batches);
    size_t total_bytes = 0;
    for (uint32_t k = 0; k < n_batches; ++k) {
        Batch &b = batches[k];
        const uint32_t first = k * per, cnt = std::min(per, n - first);
        if ((st = build_tile_table(o, tiles + first, cnt, b.tab, &b.px, &b.blocks, flat2 ? rt::kFlatBlockW : rt::kBlockW, flat2 ? rt::kFlatBlockH : rt::kBlockH)) != RT_OK) return st;
        if (flat2) { uint64_t px16 = 0; if ((st = build_tile_table(o, tiles + first, cnt, b.tab16, &px16, &b.blocks16)) != RT_OK) return st; }
        b.byte_off = total_bytes;
        total_bytes += (size_t)b.px * 4;
    }
    HIP_TRY(hipSetDevice(s->device));
    Context *c = nullptr;
    if ((st = acquire(s, &c)) != RT_OK) return st;
    Lease lease{ s, c };
    if (c->h_out_cap < total_bytes) {
        if (c->h_out) HIP_TRY(hipHostFree(c->h_out));
        c->h_out = nullptr; c->h_out_cap = 0;
        HIP_TRY(hipHostMalloc(&c->h_out, std::max(total_bytes, (size_t)1 << 20), hipHostMallocDefault));
        c->h_out_cap = std::max(total_bytes, (size_t)1 << 20);
    }
    void *alias = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&alias, c->h_out, 0));
    while (c->chunk_ev.size() < n_batches) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->chunk_ev.push_back(e);
    }
    auto drain = [&](rt_status code) { (void)hipStreamSynchronize(c->stream); (void)hipGetLastError(); return code; };   // nothing may still be writing the staging
    for (uint32_t k = 0; k < n_batches; ++k) {
        Batch &b = batches[k];
        st = enqueue_pass(s, c, o, trav, b.tab, b.blocks, b.px, static_cast<uint8_t *>(alias) + b.byte_off, 0u, c->stream, false, flat2 ? &b.tab16 : nullptr,
                          b.blocks16, true);
        if (st != RT_OK) return drain(st);
        hipError_t e = hipEventRecord(c->chunk_ev[k], c->stream);
        if (e != hipSuccess) return drain(hip_fail(e, "hipEventRecord(stream batch)", __LINE__));
    }
    for (uint32_t k = 0; k < n_batches; ++k) {
        hipError_t e = hipEventSynchronize(c->chunk_ev[k]);
        if (e != hipSuccess) return drain(hip_fail(e, "hipEventSynchronize(stream batch)", __LINE__));
        const uint32_t first = k * per, cnt = std::min(per, n - first);
        size_t off = batches[k].byte_off;
        for (uint32_t i = first; i < first + cnt; ++i) {
            callback(user, i, &tiles[i], c->h_out + off);
            off += (size_t)(tiles[i].r - tiles[i].l) * (tiles[i].t - tiles[i].b) * 4;
        }
    }
    return RT_OK;
}

// The same streaming pass for a writer that keeps its image in the FILE's pixel format (render.rs:373-401): the buckets of a batch are
// rendered tile-major into device memory and k_encode_tiles puts them -- converted -- into their place in the caller's row-major frame;
// memory this library pinned is written by that kernel itself, anything else through pinned staging and a CPU copy of the batch's rows.
rt_status rt_render_frame_stream(rt_scene *s, const rt_options *o, rt_traversal trav, const rt_region *tiles, uint32_t n, rt_frame_format format,
                                 uint8_t *frame_out, rt_batch_callback callback, void *user)
{
    if (!check_common(s, o, tiles, n, frame_out)) return RT_ERR_INVALID_ARGUMENT;
    if (format != RT_FRAME_RGBA && format != RT_FRAME_RGB && format != RT_FRAME_GREY) { snprintf(g_err, sizeof g_err, "rt_render_frame_stream: unknown frame format %d", (int)format); return RT_ERR_INVALID_ARGUMENT; }
    if ((reinterpret_cast<uintptr_t>(frame_out) & 3u) != 0) { snprintf(g_err, sizeof g_err, "rt_render_frame_stream: frame_out must be 4-byte aligned"); return RT_ERR_INVALID_ARGUMENT; }
    rt_status st = check_traversal(s, trav);
    if (st != RT_OK) return st;
    const unsigned bpp = format == RT_FRAME_RGBA ? 4u : format == RT_FRAME_RGB ? 3u : 1u;
    const size_t frame_bytes = (size_t)o->width * o->height * bpp;
    const bool flat2 = trav == RT_TRAVERSAL_FLAT;
    // Batches: at most kStreamMaxBatches, each at least a million samples (a 1080p frame at one sample per pixel: two batches -- progress
    // reports matter for renders that take long, and a short batch leaves most of the chip idle), and -- where the list is the scheduler's
    // row-major grid (render.rs:273-298) -- whole bucket ROWS, so that what a batch delivers is complete rows of the image.
    constexpr uint32_t kStreamMaxBatches = 16;
    const uint64_t ns = (uint64_t)o->samples_per_pixel * o->samples_per_pixel;
    uint32_t per = std::max<uint32_t>((n + kStreamMaxBatches - 1) / kStreamMaxBatches, (uint32_t)std::clamp<uint64_t>((1ull << 20) / (4096ull * std::max<uint64_t>(ns, 1)), 16, 256));
    {
        uint32_t row = 1;
        while (row < n && tiles[row].b == tiles[0].b) ++row;
        bool grid = n % row == 0;
        for (uint32_t i = 0; grid && i < n; ++i) grid = tiles[i].b == tiles[i - i % row].b && tiles[i].t == tiles[i - i % row].t && tiles[i].l == tiles[i % row].l && tiles[i].r == tiles[i % row].r;
        if (grid) per = (per + row - 1) / row * row;
    }
    const uint32_t n_batches = (n + per - 1) / per;
    struct Batch { std::vector<rt::TileDev> tab, tab16; uint64_t px = 0; uint32_t blocks = 0, blocks16 = 0; size_t byte_off = 0; };
    std::vector<Batch> batches(n_batches);
    size_t total_bytes = 0;
    for (uint32_t k = 0; k < n_batches; ++k) {
        Batch &b = batches[k];
        const uint32_t first = k * per, cnt = std::min(per, n - first);
        if ((st = build_tile_table(o, tiles + first, cnt, b.tab, &b.px, &b.blocks, flat2 ? rt::kFlatBlockW : rt::kBlockW, flat2 ? rt::kFlatBlockH : rt::kBlockH)) != RT_OK) return st;
        if (flat2) { uint64_t px16 = 0; if ((st = build_tile_table(o, tiles + first, cnt, b.tab16, &px16, &b.blocks16)) != RT_OK) return st; }
        b.byte_off = total_bytes;
        total_bytes += (size_t)b.px * 4;
    }
    HIP_TRY(hipSetDevice(s->device));
    const HostDest dest = classify_host_pointer(frame_out);
    if (dest.bad) { snprintf(g_err, sizeof g_err, "rt_render_frame_stream: frame_out is device memory"); return RT_ERR_INVALID_ARGUMENT; }
    const bool direct = dest.pinned && dest.dev_alias && dest.room >= frame_bytes;
    Context *c = nullptr;
    if ((st = acquire(s, &c)) != RT_OK) return st;
    Lease lease{ s, c };
    if (c->out_cap < total_bytes) {
        if (c->d_out) HIP_TRY(hipFree(c->d_out));
        c->d_out = nullptr; c->out_cap = 0;
        HIP_TRY(hipMalloc(&c->d_out, total_bytes));
        c->out_cap = total_bytes;
    }
    uint8_t *target = dest.dev_alias;
    if (!direct) {
        if (c->h_out_cap < frame_bytes) {
            if (c->h_out) HIP_TRY(hipHostFree(c->h_out));
            c->h_out = nullptr; c->h_out_cap = 0;
            HIP_TRY(hipHostMalloc(&c->h_out, std::max(frame_bytes, (size_t)1 << 20), hipHostMallocDefault));
            c->h_out_cap = std::max(frame_bytes, (size_t)1 << 20);
        }
        void *alias = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&alias, c->h_out, 0));
        target = static_cast<uint8_t *>(alias);
    }
    while (c->chunk_ev.size() < n_batches) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->chunk_ev.push_back(e);
    }
    // several batches: a batch is encoded (PCIe-bound, a handful of waves) on a second stream while the next one renders
    hipStream_t enc = c->stream;
    if (n_batches >= 4) {                                            // (a stream is a hardware queue: 5 - 9 ms to create, once per context)
        if (!c->stream2 && hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); c->stream2 = nullptr; }
        if (c->stream2) enc = c->stream2;
    }
    auto drain = [&](rt_status code) { (void)hipStreamSynchronize(c->stream); if (c->stream2) (void)hipStreamSynchronize(c->stream2); (void)hipGetLastError(); return code; };   // nothing may still be writing the frame
    for (uint32_t k = 0; k < n_batches; ++k) {
        Batch &b = batches[k];
        // the encode reads the very table the batch was rendered with (looking it up again could upload it a second time, unordered with
        // the encode: ADVICE r5); a table that is NOT the scene's immutable cached copy -- the cache is full, it went through the context's
        // one upload slot -- is only safe in stream order, so such a batch is encoded on the render stream
        const rt::TileDev *d_tab = nullptr;
        st = enqueue_pass(s, c, o, trav, b.tab, b.blocks, b.px, c->d_out + b.byte_off, 0u, c->stream, false, flat2 ? &b.tab16 : nullptr, b.blocks16, true, &d_tab);
        if (st != RT_OK) return drain(st);
        if (c->tiles_live[0]) enc = c->stream;
        if (enc != c->stream) {                                      // (enqueue_pass recorded ev1 behind the batch's kernels)
            const hipError_t we = hipStreamWaitEvent(enc, c->ev1, 0);
            if (we != hipSuccess) return drain(hip_fail(we, "rt_render_frame_stream(wait)", __LINE__));
        }
        const unsigned *src = reinterpret_cast<const unsigned *>(c->d_out + b.byte_off);
        const dim3 grid((unsigned)b.tab.size()), blk(rt::kBlockThreads);
        if (bpp == 4) hipLaunchKernelGGL((rt::k_encode_tiles<4>), grid, blk, 0, enc, (unsigned)o->width, d_tab, (unsigned)b.tab.size(), src, target);
        else if (bpp == 3) hipLaunchKernelGGL((rt::k_encode_tiles<3>), grid, blk, 0, enc, (unsigned)o->width, d_tab, (unsigned)b.tab.size(), src, target);
        else hipLaunchKernelGGL((rt::k_encode_tiles<1>), grid, blk, 0, enc, (unsigned)o->width, d_tab, (unsigned)b.tab.size(), src, target);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(c->chunk_ev[k], enc);
        if (e != hipSuccess) return drain(hip_fail(e, "rt_render_frame_stream(encode)", __LINE__));
    }
    for (uint32_t k = 0; k < n_batches; ++k) {
        hipError_t e = hipEventSynchronize(c->chunk_ev[k]);
        if (e != hipSuccess) return drain(hip_fail(e, "hipEventSynchronize(stream batch)", __LINE__));
        const uint32_t first = k * per, cnt = std::min(per, n - first);
        if (!direct)
            for (uint32_t i = first; i < first + cnt; ++i) {
                const size_t seg = (size_t)(tiles[i].r - tiles[i].l) * bpp;
                for (unsigned y = tiles[i].b; y < tiles[i].t; ++y) {
                    const size_t off = ((size_t)y * o->width + tiles[i].l) * bpp;
                    memcpy(frame_out + off, c->h_out + off, seg);
                }
            }
        if (callback) callback(user, first, cnt);
    }
    // (the last batch's event is behind everything on both streams: the context goes back idle)
    return RT_OK;
}

rt_status rt_host_alloc(size_t bytes, void **out)
{
    if (!out || bytes == 0) { snprintf(g_err, sizeof g_err, "rt_host_alloc: NULL argument or 0 bytes"); return RT_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    int ndev = 0;
    rt_status st = rt_device_count(&ndev);
    if (st != RT_OK) return st;
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocPortable | hipHostMallocMapped));
    void *alias = nullptr;
    if (hipHostGetDevicePointer(&alias, *out, 0) != hipSuccess) { (void)hipGetLastError(); alias = nullptr; }
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    g_pinned[reinterpret_cast<uintptr_t>(*out)] = PinnedRange{ bytes, static_cast<uint8_t *>(alias) };
    return RT_OK;
}

rt_status rt_host_free(void *p)
{
    if (!p) return RT_OK;
    {
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        if (g_pinned.erase(reinterpret_cast<uintptr_t>(p)) == 0) {
            snprintf(g_err, sizeof g_err, "rt_host_free: not a pointer rt_host_alloc returned");
            return RT_ERR_INVALID_ARGUMENT;
        }
    }
    HIP_TRY(hipHostFree(p));
    return RT_OK;
}

rt_status rt_host_register(void *p, size_t bytes)
{
    if (!p || bytes == 0) { snprintf(g_err, sizeof g_err, "rt_host_register: NULL argument or 0 bytes"); return RT_ERR_INVALID_ARGUMENT; }
    int ndev = 0;
    rt_status st = rt_device_count(&ndev);
    if (st != RT_OK) return st;
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped));
    void *alias = nullptr;
    if (hipHostGetDevicePointer(&alias, p, 0) != hipSuccess) { (void)hipGetLastError(); alias = nullptr; }
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    g_pinned[reinterpret_cast<uintptr_t>(p)] = PinnedRange{ bytes, static_cast<uint8_t *>(alias) };
    return RT_OK;
}

rt_status rt_host_unregister(void *p)
{
    if (!p) return RT_OK;
    {
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        if (g_pinned.erase(reinterpret_cast<uintptr_t>(p)) == 0) {
            snprintf(g_err, sizeof g_err, "rt_host_unregister: not a pointer rt_host_register was given");
            return RT_ERR_INVALID_ARGUMENT;
        }
    }
    HIP_TRY(hipHostUnregister(p));
    return RT_OK;
}

rt_status rt_blit_tiles_device(rt_scene *s, const rt_options *o, const rt_region *tiles, uint32_t n, const uint32_t *src_px_offset,
                               const void *src, void *frame, void *hip_stream)
{
    if (!check_common(s, o, tiles, n, frame) || !src) { snprintf(g_err, sizeof g_err, "rt_blit_tiles_device: NULL argument"); return RT_ERR_INVALID_ARGUMENT; }
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(frame)) & 3u) != 0) {
        snprintf(g_err, sizeof g_err, "rt_blit_tiles_device: buffers must be 4-byte aligned");
        return RT_ERR_INVALID_ARGUMENT;
    }
    std::vector<rt::TileDev> tab;
    uint64_t total_px = 0; uint32_t total_blocks = 0;
    rt_status st = build_tile_table(o, tiles, n, tab, &total_px, &total_blocks);
    if (st != RT_OK) return st;
    if (src_px_offset)
        for (uint32_t i = 0; i < n; ++i) tab[i].out_px = src_px_offset[i];
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const rt::TileDev *d_tab = nullptr;
    if ((st = device_table(s, nullptr, tab, stream, &d_tab)) != RT_OK) return st;
    if (d_tab) {                                                    // cached table: one kernel, nothing else
        hipLaunchKernelGGL(rt::k_blit_tiles, dim3(total_blocks), dim3(rt::kBlockThreads), 0, stream, (unsigned)o->width, d_tab, (unsigned)n,
                           static_cast<const unsigned *>(src), static_cast<unsigned *>(frame));
        HIP_TRY(hipGetLastError());
        return RT_OK;
    }
    Context *c = nullptr;
    if ((st = acquire(s, &c)) != RT_OK) return st;
    Lease lease{ s, c };
    if ((st = device_table(s, c, tab, stream, &d_tab)) != RT_OK) return st;
    hipLaunchKernelGGL(rt::k_blit_tiles, dim3(total_blocks), dim3(rt::kBlockThreads), 0, stream, (unsigned)o->width, d_tab,
                       (unsigned)n, static_cast<const unsigned *>(src), static_cast<unsigned *>(frame));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev1, stream));
    lease.inflight = true;
    return RT_OK;
}

rt_status rt_scene_traits(const rt_scene *s, uint32_t *traits)
{
    if (!s || !traits) { snprintf(g_err, sizeof g_err, "NULL argument"); return RT_ERR_INVALID_ARGUMENT; }
    *traits = (s->n_nodes ? RT_SCENE_HAS_BOUNDS : 0u) | (s->fused ? RT_SCENE_CONCENTRIC : 0u);
    return RT_OK;
}

typedef void (*selftest_kernel)(unsigned, unsigned long long, unsigned long long *, unsigned *);
static rt_status selftest_all_f32(int device, selftest_kernel kernel, const char *what, uint64_t *mismatches, uint32_t *first_bad_bits)
{
    if (!mismatches || !first_bad_bits) return RT_ERR_INVALID_ARGUMENT;
    int ndev = 0;
    rt_status st = rt_device_count(&ndev);
    if (st != RT_OK) return st;
    if (device < 0 || device >= ndev) return RT_ERR_NO_DEVICE;
    HIP_TRY(hipSetDevice(device));
    unsigned long long *d_bad = nullptr;
    unsigned *d_first = nullptr;
    HIP_TRY(hipMalloc(&d_bad, sizeof *d_bad));
    HIP_TRY(hipMalloc(&d_first, sizeof *d_first));
    HIP_TRY(hipMemset(d_bad, 0, sizeof *d_bad));
    HIP_TRY(hipMemset(d_first, 0xFF, sizeof *d_first));
    // all 2^32 bit patterns: non-negative values, negatives, infinities and NaNs
    hipLaunchKernelGGL(kernel, dim3(256 * 32), dim3(256), 0, nullptr, 0u, 1ull << 32, d_bad, d_first);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    unsigned long long bad = 0; unsigned first = 0;
    if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&first, d_first, sizeof first, hipMemcpyDeviceToHost);
    (void)hipFree(d_bad); (void)hipFree(d_first);
    if (e != hipSuccess) return hip_fail(e, what, __LINE__);
    *mismatches = bad; *first_bad_bits = first;
    return RT_OK;
}

rt_status rt_selftest_sqrt(int device, uint64_t *mismatches, uint32_t *first_bad_bits)
{
    return selftest_all_f32(device, rt::k_selftest_sqrt, "rt_selftest_sqrt", mismatches, first_bad_bits);
}

rt_status rt_selftest_rcp(int device, uint64_t *mismatches, uint32_t *first_bad_bits)
{
    return selftest_all_f32(device, rt::k_selftest_rcp, "rt_selftest_rcp", mismatches, first_bad_bits);
}

constexpr int kMaxRegionLeaders = 2;

// Renders every request of `batch` in ONE pass and delivers each tile to its caller's buffer.
static void run_region_batch(rt_scene *s, const std::vector<rt_scene::RegionReq *> &batch)
{
    std::vector<rt_region> regs(batch.size());
    std::vector<uint8_t *> outs(batch.size());
    for (size_t i = 0; i < batch.size(); ++i) { regs[i] = batch[i]->region; outs[i] = batch[i]->out; }
    count_event(RT_DEBUG_COUNT_REGION_CALLS, (long long)batch.size());
    count_event(RT_DEBUG_COUNT_REGION_PASSES);
    g_err[0] = '\0';
    rt_status st = RT_OK;
    if (batch.size() == 1) {
        st = render_tiles_host(s, &batch[0]->o, batch[0]->trav, regs.data(), 1, outs[0], outs.data(), nullptr, false);
    } else {
        st = render_tiles_host(s, &batch[0]->o, batch[0]->trav, regs.data(), (uint32_t)regs.size(), outs[0], outs.data(), nullptr, false);
        if (st == RT_ERR_INVALID_REGION) {
            // one caller's bad region must not fail its neighbours: everyone on their own
            for (rt_scene::RegionReq *r : batch) {
                g_err[0] = '\0';
                r->st = render_tiles_host(s, &r->o, r->trav, &r->region, 1, r->out, &r->out, nullptr, false);
                snprintf(r->err, sizeof r->err, "%s", g_err);
            }
            return;
        }
    }
    for (rt_scene::RegionReq *r : batch) { r->st = st; snprintf(r->err, sizeof r->err, "%s", g_err); }
}

// rt_render_region through the scene's frame-ahead (rt_scene::FrameAhead).  false: the request is not a bucket of the scheduler's
// grid (render.rs:273-298: 64x64, edge buckets clipped) or the frame is too large to keep -- the caller renders it on its own.
constexpr unsigned kBucket = 64;
constexpr size_t kFrameAheadMaxBytes = (size_t)1 << 28;
constexpr uint64_t kFrameAheadMaxSampleBytes = 1ull << 30;
static bool region_from_frame_ahead(rt_scene *s, const rt_options *o, rt_traversal trav, const rt_region *region, uint8_t *out, rt_status *st)
{
    const unsigned w = o->width, h = o->height;
    if (region->l % kBucket || region->b % kBucket || region->r != std::min<unsigned>(region->l + kBucket, w) ||
        region->t != std::min<unsigned>(region->b + kBucket, h) || region->l >= w || region->b >= h)
        return false;
    const size_t frame_bytes = (size_t)w * h * 4;
    // a whole-grid pass of a sample-parallel frame also needs its per-sample buffers (spp >= 2: a word or more per sample), twice with the
    // pass rendered ahead: the bucket on its own needs a few MB -- leave large frames to the per-bucket path
    const uint64_t ns = (uint64_t)o->samples_per_pixel * o->samples_per_pixel;
    const uint64_t sample_bytes = use_split(o->samples_per_pixel) ? (uint64_t)w * h * ns * (s->precision == RT_F32 ? 5u : 9u) : 0u;
    if (frame_bytes > kFrameAheadMaxBytes || sample_bytes > kFrameAheadMaxSampleBytes || check_traversal(s, trav) != RT_OK) return false;
    const unsigned nbx = (w + kBucket - 1) / kBucket, idx = (region->b / kBucket) * nbx + region->l / kBucket;
    rt_scene::FrameAhead &a = s->ahead;
    std::unique_lock<std::mutex> lk(a.mu);
    auto drain_next = [&] { if (a.next_inflight) { (void)hipEventSynchronize(a.ev); (void)hipGetLastError(); a.next_inflight = false; } };
    // The copies out of the staging run OUTSIDE the lock (the reference's pool threads call this concurrently, render.rs:283-294), so
    // a new pass -- which replaces the staging the readers copy from -- waits until the last of them is done, and whoever was waiting
    // looks again afterwards: another caller may have brought the new pass in meanwhile.
    for (;;) {
        const bool same = a.valid && a.trav == trav && a.o.width == o->width && a.o.height == o->height && a.o.samples_per_pixel == o->samples_per_pixel;
        if (same && !a.served[idx]) break;
        if (!same) {
            // Whole-grid passes are for a caller that walks the grid (the scheduler, render.rs:273-298).  A lone request -- a partial redraw,
            // a tool, a test -- is rendered on its own: the frame-ahead engages with the SECOND distinct bucket asked for with the same options.
            const bool seen = a.seen_idx >= 0 && a.seen_trav == trav && a.seen_o.width == o->width && a.seen_o.height == o->height &&
                              a.seen_o.samples_per_pixel == o->samples_per_pixel;
            if (!seen || a.seen_idx == (int)idx) { a.seen_o = *o; a.seen_trav = trav; a.seen_idx = (int)idx; return false; }
        }
        if (a.readers != 0) { a.cv.wait(lk); continue; }
        bool have = false;
        if (!same) {
            drain_next();                                   // a pass for other options may still be writing h_next
            a.valid = false;
            a.grid.clear(); a.off.clear();
            size_t off = 0;
            for (unsigned y = 0; y < h; y += kBucket)
                for (unsigned x = 0; x < w; x += kBucket) {
                    const rt_region r{ (uint16_t)x, (uint16_t)std::min(y + kBucket, h), (uint16_t)std::min(x + kBucket, w), (uint16_t)y };
                    a.grid.push_back(r);
                    a.off.push_back(off);
                    off += (size_t)(r.r - r.l) * (r.t - r.b) * 4;
                }
            if (a.cap < frame_bytes) {
                if (a.h) (void)rt_host_free(a.h);
                if (a.h_next) (void)rt_host_free(a.h_next);
                a.h = a.h_next = nullptr; a.cap = 0;
                void *p = nullptr, *q = nullptr;
                // whatever fails in here: the caller renders its bucket on its own (the per-bucket path needs a few MB, not two pinned frames)
                if (rt_host_alloc(frame_bytes, &p) != RT_OK) { a.valid = false; return false; }
                if (rt_host_alloc(frame_bytes, &q) != RT_OK) { (void)rt_host_free(p); a.valid = false; return false; }
                a.h = static_cast<uint8_t *>(p); a.h_next = static_cast<uint8_t *>(q); a.cap = frame_bytes;
            }
            a.o = *o; a.trav = trav;
        } else if (a.next_inflight) {
            // the pass that was started when the previous frame was first asked for
            const hipError_t e = hipEventSynchronize(a.ev);
            a.next_inflight = false;
            if (e == hipSuccess) { std::swap(a.h, a.h_next); have = true; } else (void)hipGetLastError();
        }
        if (!have) {
            // the whole grid in one pass, the kernel storing into the pinned staging (rt_host_alloc'd memory is recognised by address)
            if (render_tiles_host(s, o, trav, a.grid.data(), (uint32_t)a.grid.size(), a.h, nullptr, nullptr, true) != RT_OK) {
                a.valid = false; a.seen_idx = -1;          // e.g. out of memory for the whole grid: the bucket alone may still fit
                return false;
            }
        }
        a.served.assign(a.grid.size(), 0);
        a.valid = true;
        count_event(RT_DEBUG_COUNT_FRAME_AHEAD_PASSES);
        if (knob(RT_DEBUG_FRAME_AHEAD) != 1 && sample_bytes <= kFrameAheadMaxSampleBytes / 4) {
            // the next frame's pass, asynchronously, on a stream of its own; whatever fails here only costs the overlap
            hipError_t e = hipSuccess;
            if (!a.stream) e = hipStreamCreateWithFlags(&a.stream, hipStreamNonBlocking);
            if (e == hipSuccess && !a.ev) e = hipEventCreateWithFlags(&a.ev, hipEventDisableTiming);
            const HostDest next = classify_host_pointer(a.h_next);
            if (e == hipSuccess && next.pinned && next.dev_alias && next.room >= frame_bytes &&
                rt_render_tiles_device(s, o, trav, a.grid.data(), (uint32_t)a.grid.size(), next.dev_alias, a.stream, nullptr) == RT_OK &&
                hipEventRecord(a.ev, a.stream) == hipSuccess)
                a.next_inflight = true;
            else if (a.stream) { (void)hipStreamSynchronize(a.stream); (void)hipGetLastError(); }
        }
        break;
    }
    const uint8_t *src = a.h + a.off[idx];
    a.served[idx] = 1;
    ++a.readers;
    lk.unlock();
    memcpy(out, src, (size_t)(region->r - region->l) * (region->t - region->b) * 4);
    lk.lock();
    if (--a.readers == 0) a.cv.notify_all();
    *st = RT_OK;
    return true;
}

rt_status rt_render_region(rt_scene *s, const rt_options *o, rt_traversal trav, const rt_region *region, uint8_t *rgba_out,
                           rt_stats *stats)
{
    if (!check_common(s, o, region, 1, rgba_out)) return RT_ERR_INVALID_ARGUMENT;
    if (!stats && knob(RT_DEBUG_FRAME_AHEAD) != 0) {
        if (classify_host_pointer(rgba_out).bad) {              // the frame-ahead path copies with the CPU: same answer as rt_render_tiles gives
            snprintf(g_err, sizeof g_err, "rt_render_region: rgba_out is device memory; use rt_render_tiles_device");
            return RT_ERR_INVALID_ARGUMENT;
        }
        rt_status fst = RT_OK;
        if (region_from_frame_ahead(s, o, trav, region, rgba_out, &fst)) return fst;
    }
    if (stats || knob(RT_DEBUG_COALESCE) == 0) return render_tiles_host(s, o, trav, region, 1, rgba_out, &rgba_out, stats, false);
    // Group commit: the reference calls this from up to RTRACEMAXPROCS pool threads at once (render.rs:283-294), and one
    // 64x64 bucket per device pass would leave 255 of 256 CUs idle.  A caller that finds no pass running leads the next
    // one and renders every request waiting at that moment (same options and traversal) together; the others sleep until
    // their bytes are in their buffer.  A lone caller degenerates to one pass per call.
    rt_scene::RegionReq me;
    me.o = *o; me.trav = trav; me.region = *region; me.out = rgba_out;
    const long long k = knob(RT_DEBUG_COALESCE);
    const int max_leaders = k > 0 ? (int)std::min<long long>(k, 8) : kMaxRegionLeaders;
    std::unique_lock<std::mutex> lk(s->comb_mu);
    s->comb_pending.push_back(&me);
    while (!me.done) {
        // up to max_leaders passes at once: while one leader waits for its kernel or hands out bytes, the next batch is
        // already being set up and rendered on another stream.  Sleepers are woken one by one (their request is done, or it
        // is their turn to lead), never all at once.
        if (me.taken || s->comb_leaders >= max_leaders) { me.cv.wait(lk); continue; }
        ++s->comb_leaders;
        std::vector<rt_scene::RegionReq *> batch, rest;
        const rt_scene::RegionReq *head = s->comb_pending.front();
        for (rt_scene::RegionReq *r : s->comb_pending) {
            const bool same = r->trav == head->trav && r->o.width == head->o.width && r->o.height == head->o.height &&
                              r->o.samples_per_pixel == head->o.samples_per_pixel;
            (same ? batch : rest).push_back(r);
            if (same) r->taken = true;
        }
        s->comb_pending.swap(rest);
        lk.unlock();
        run_region_batch(s, batch);
        lk.lock();
        --s->comb_leaders;
        if (!s->comb_pending.empty()) s->comb_pending.front()->cv.notify_one();      // someone whose request is still waiting leads next
        for (rt_scene::RegionReq *r : batch) {
            r->done = true;
            if (r != &me) r->cv.notify_one();
        }
    }
    lk.unlock();
    if (me.st != RT_OK) snprintf(g_err, sizeof g_err, "%s", me.err);
    return me.st;
}


// ---------------------------------------------------------------------------------------------------------------------
// Gang: the buckets of one frame dealt over several GPUs of this node by ONE process, shards brought to the root GPU by one
// RCCL gather over xGMI (replaces the channel of render.rs:271,293,301 for the multi-GPU case).
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string error;
};

static void load_rccl(Rccl &r, std::initializer_list<const char *> names)
{
    for (const char *name : names) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) {
        const char *why = dlerror();                          // (a second call returns NULL: the message is handed out once)
        r.error = std::string("dlopen(") + *names.begin() + "): " + (why ? why : "not found");
        return;
    }
    auto sym = [&](const char *n) { void *p = dlsym(r.lib, n); if (!p) r.error = std::string(*names.begin()) + " lacks " + n; return p; };
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.Gather = reinterpret_cast<decltype(r.Gather)>(sym("ncclGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
}

// librccl.so is ~0.5 GB: it is loaded on first use, never for single-GPU renders.
static Rccl *real_rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { load_rccl(r, { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }); });
    return &r;
}

// Test infrastructure (rt_debug.h rt_debug_rccl_library): a stand-in library given by path takes the place of librccl.so for the gangs
// created while it is set, and such gangs may put several ranks on ONE device -- how the N > 1 code is executed on a one-GPU box.
#ifdef RT_TEST_HOOKS
static std::mutex g_standin_mu;
static std::shared_ptr<Rccl> g_standin;

static std::shared_ptr<Rccl> current_standin()
{
    std::lock_guard<std::mutex> lk(g_standin_mu);
    return g_standin;
}
#else
static std::shared_ptr<Rccl> current_standin() { return nullptr; }      // the product only ever talks to librccl.so
#endif

static thread_local const Rccl *g_err_rccl = nullptr;       // whose error strings rccl_fail prints

static rt_status rccl_fail(ncclResult_t e, const char *what, int line)
{
    snprintf(g_err, sizeof g_err, "%s failed at rt_capi.hip:%d: %s", what, line, g_err_rccl && g_err_rccl->GetErrorString ? g_err_rccl->GetErrorString(e) : "RCCL error");
    return RT_ERR_HIP;
}

#define RCCL_TRY(expr)                                                      \
    do {                                                                    \
        ncclResult_t e__ = (expr);                                          \
        if (e__ != ncclSuccess) return rccl_fail(e__, #expr, __LINE__);     \
    } while (0)

}  // namespace

// Where bucket i of a frame goes when its buckets are dealt over nd devices: device i % nd in the caller's (the
// scheduler's row-major, render.rs:273-298) order, tile-major inside the device's shard; shards padded to the longest one so the
// gather moves equal counts.  Pure arithmetic (no device needed): rt_debug_gang_layout exposes it to the CPU tests, which hold it
// against dist.shard_layout.
struct GangLayout {
    std::vector<std::vector<rt_region>> shard;      // per device: its buckets
    std::vector<uint64_t> shard_px;                 // per device: pixels of its shard (before padding)
    uint64_t max_px = 0;                            // padded shard length in pixels
    std::vector<rt_region> gathered_regs;           // every bucket, in gathered order (device-major)
    std::vector<uint32_t> gathered_off;             // its first pixel in the gathered [nd][max_px] buffer
    std::vector<uint32_t> device_of, px_offset;     // per input bucket: its device and its first pixel inside that device's shard
};

static void gang_layout(const rt_region *tiles, uint32_t n, size_t nd, GangLayout &L)
{
    L = GangLayout{};
    L.shard.resize(nd); L.shard_px.assign(nd, 0); L.device_of.resize(n); L.px_offset.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        const size_t d = i % nd;
        L.device_of[i] = (uint32_t)d;
        L.px_offset[i] = (uint32_t)L.shard_px[d];
        L.shard[d].push_back(tiles[i]);
        L.shard_px[d] += (uint64_t)(tiles[i].r - tiles[i].l) * (tiles[i].t - tiles[i].b);
    }
    for (size_t d = 0; d < nd; ++d) L.max_px = std::max(L.max_px, L.shard_px[d]);
    for (size_t d = 0; d < nd; ++d) {
        uint64_t px = 0;
        for (const rt_region &t : L.shard[d]) {
            L.gathered_regs.push_back(t);
            L.gathered_off.push_back((uint32_t)(d * L.max_px + px));
            px += (uint64_t)(t.r - t.l) * (t.t - t.b);
        }
    }
}

struct rt_gang {
    std::shared_ptr<Rccl> standin;            // set: this gang talks to a stand-in library (tests), else to librccl.so
    const Rccl *nccl = nullptr;
    std::vector<int> devices;
    std::vector<rt_scene *> scenes;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;         // per device: renders
    std::vector<hipStream_t> comm_streams;    // per device: the gather (and on the root the blit and the copy to the host) -- a frame's
                                              // gather runs under the next frame's render (rt_gang_render_frames)
    std::vector<hipEvent_t> ev_rendered[2], ev_gathered[2];      // per shard-buffer parity and device
    std::vector<uint8_t *> d_shard[2];        // per device: its tile-major shard, double-buffered
    size_t shard_cap = 0;                     // bytes of each d_shard
    uint8_t *d_gathered[2] = { nullptr, nullptr };   // root: [n_devices][shard bytes]
    size_t gathered_cap = 0;
    uint8_t *d_frame = nullptr;               // root: row-major RGBA frame (pageable destinations)
    size_t frame_cap = 0;
    // the layout of the last tile list (a scheduler submits the same bucket list every frame)
    std::vector<rt_region> last_tiles;
    GangLayout layout;
    std::mutex mu;                            // one call at a time per gang
};

rt_status rt_gang_create(const int *devices, int n_devices, rt_precision precision, const void *dfs_items, uint32_t n_items,
                         const void *light_unit, const void *eye, const void *bounds, const rt_range *ranges, uint32_t n_bounds,
                         rt_gang **out)
{
    if (out) *out = nullptr;
    if (!out || !devices || n_devices < 1 || n_devices > 64) {
        snprintf(g_err, sizeof g_err, "rt_gang_create: NULL argument or n_devices outside 1..64");
        return RT_ERR_INVALID_ARGUMENT;
    }
    const std::shared_ptr<Rccl> standin = current_standin();
    if (!standin)                                       // RCCL wants one GPU per rank; only a stand-in library (tests) takes several ranks on one
        for (int a = 0; a < n_devices; ++a)
            for (int b = a + 1; b < n_devices; ++b)
                if (devices[a] == devices[b]) { snprintf(g_err, sizeof g_err, "rt_gang_create: device %d listed twice", devices[a]); return RT_ERR_INVALID_ARGUMENT; }
    const Rccl *r = standin ? standin.get() : real_rccl();
    g_err_rccl = r;
    if (!r->error.empty() || !r->Gather) { snprintf(g_err, sizeof g_err, "rt_gang_create: %s", r->error.c_str()); return RT_ERR_UNSUPPORTED; }
    std::unique_ptr<rt_gang> g(new (std::nothrow) rt_gang());
    if (!g) return RT_ERR_OUT_OF_MEMORY;
    g->standin = standin; g->nccl = r;
    auto fail = [&](rt_status st) { rt_gang_destroy(g.release()); return st; };
    g->devices.assign(devices, devices + n_devices);
    for (int d = 0; d < n_devices; ++d) {
        rt_scene *s = nullptr;
        rt_status st = rt_scene_create(devices[d], precision, dfs_items, n_items, light_unit, eye, bounds, ranges, n_bounds, &s);
        if (st != RT_OK) return fail(st);
        g->scenes.push_back(s);
        hipStream_t stream = nullptr;
        hipError_t e = hipSetDevice(devices[d]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
        if (e != hipSuccess) return fail(hip_fail(e, "hipStreamCreate(gang)", __LINE__));
        g->streams.push_back(stream);
        hipStream_t cs = nullptr;
        if ((e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking)) != hipSuccess) return fail(hip_fail(e, "hipStreamCreate(gang)", __LINE__));
        g->comm_streams.push_back(cs);
        for (int p = 0; p < 2; ++p) {
            hipEvent_t a = nullptr, b2 = nullptr;
            if ((e = hipEventCreateWithFlags(&a, hipEventDisableTiming)) != hipSuccess) return fail(hip_fail(e, "hipEventCreate(gang)", __LINE__));
            g->ev_rendered[p].push_back(a);
            if ((e = hipEventCreateWithFlags(&b2, hipEventDisableTiming)) != hipSuccess) return fail(hip_fail(e, "hipEventCreate(gang)", __LINE__));
            g->ev_gathered[p].push_back(b2);
            g->d_shard[p].push_back(nullptr);
        }
    }
    g->comms.assign((size_t)n_devices, nullptr);
    ncclResult_t ne = r->CommInitAll(g->comms.data(), n_devices, g->devices.data());       // one communicator per device, this process
    if (ne != ncclSuccess) { g->comms.clear(); return fail(rccl_fail(ne, "ncclCommInitAll", __LINE__)); }
    *out = g.release();
    return RT_OK;
}

rt_status rt_gang_destroy(rt_gang *g)
{
    if (!g) return RT_OK;
    for (ncclComm_t c : g->comms)
        if (c && g->nccl) (void)g->nccl->CommDestroy(c);
    for (size_t d = 0; d < g->devices.size(); ++d) {
        (void)hipSetDevice(g->devices[d]);
        for (int p = 0; p < 2; ++p) {
            if (d < g->d_shard[p].size() && g->d_shard[p][d]) (void)hipFree(g->d_shard[p][d]);
            if (d < g->ev_rendered[p].size() && g->ev_rendered[p][d]) (void)hipEventDestroy(g->ev_rendered[p][d]);
            if (d < g->ev_gathered[p].size() && g->ev_gathered[p][d]) (void)hipEventDestroy(g->ev_gathered[p][d]);
        }
        if (d < g->streams.size() && g->streams[d]) (void)hipStreamDestroy(g->streams[d]);
        if (d < g->comm_streams.size() && g->comm_streams[d]) (void)hipStreamDestroy(g->comm_streams[d]);
        if (d == 0) {
            for (int p = 0; p < 2; ++p) if (g->d_gathered[p]) (void)hipFree(g->d_gathered[p]);
            if (g->d_frame) (void)hipFree(g->d_frame);
        }
    }
    for (rt_scene *s : g->scenes) rt_scene_destroy(s);
    delete g;
    return RT_OK;
}

rt_status rt_gang_size(const rt_gang *g, int *n_devices)
{
    if (!g || !n_devices) { snprintf(g_err, sizeof g_err, "NULL argument"); return RT_ERR_INVALID_ARGUMENT; }
    *n_devices = (int)g->devices.size();
    return RT_OK;
}

// The gang's frames: `k` frames of the same tile list, frame f to frames_host[f].  Per device a render stream and a communication
// stream: render(f) -> [event] -> gather(f) on the communication streams -> blit(f) (+ copy to the host) on the root's, while
// render(f + 1) already runs into the other shard buffer (it waits for gather(f - 1), the last reader of that buffer).
static rt_status gang_sync_all(rt_gang *g)
{
    for (size_t d = 0; d < g->devices.size(); ++d) {
        (void)hipSetDevice(g->devices[d]);
        (void)hipStreamSynchronize(g->streams[d]);
        (void)hipStreamSynchronize(g->comm_streams[d]);
    }
    (void)hipGetLastError();
    return RT_OK;
}

static rt_status gang_render(rt_gang *g, const rt_options *o, rt_traversal trav, const rt_region *tiles, uint32_t n, uint8_t *const *frames_host,
                             uint32_t k, rt_stats *stats)
{
    const Rccl *r = g->nccl;
    g_err_rccl = r;
    const size_t nd = g->devices.size();
    // the layout of this tile list (cached: a scheduler submits the same list every frame)
    bool new_list = false;
    if (g->last_tiles.size() != n || memcmp(g->last_tiles.data(), tiles, sizeof(rt_region) * n) != 0) {
        gang_layout(tiles, n, nd, g->layout);
        g->last_tiles.assign(tiles, tiles + n);
        new_list = true;
    }
    const GangLayout &L = g->layout;
    if (L.max_px * nd > 0xFFFFFFFFull) { snprintf(g_err, sizeof g_err, "rt_gang_render_frame: tile list too large for one pass"); return RT_ERR_INVALID_ARGUMENT; }
    const size_t shard_bytes = (size_t)L.max_px * 4, frame_bytes = (size_t)o->width * o->height * 4;
    // buffers
    if (g->shard_cap < shard_bytes) {
        for (size_t d = 0; d < nd; ++d) {
            HIP_TRY(hipSetDevice(g->devices[d]));
            for (int p = 0; p < 2; ++p) {
                if (g->d_shard[p][d]) HIP_TRY(hipFree(g->d_shard[p][d]));
                g->d_shard[p][d] = nullptr;
                HIP_TRY(hipMalloc(&g->d_shard[p][d], shard_bytes));
                // the padding behind a short shard travels too.  On the stream that renders into the buffer: the device's streams do not
                // synchronise with the null stream, where a plain hipMemset would run
                HIP_TRY(hipMemsetAsync(g->d_shard[p][d], 0, shard_bytes, g->streams[d]));
            }
        }
        g->shard_cap = shard_bytes;
    }
    HIP_TRY(hipSetDevice(g->devices[0]));
    if (g->gathered_cap < shard_bytes * nd) {
        for (int p = 0; p < 2; ++p) {
            if (g->d_gathered[p]) HIP_TRY(hipFree(g->d_gathered[p]));
            g->d_gathered[p] = nullptr;
        }
        g->gathered_cap = 0;
        for (int p = 0; p < 2; ++p) HIP_TRY(hipMalloc(&g->d_gathered[p], shard_bytes * nd));
        g->gathered_cap = shard_bytes * nd;
    }
    // destinations: memory this library pinned is written by the root's blit kernel itself (no device copy of the frame, no D2H)
    std::vector<uint8_t *> alias(k, nullptr);
    bool need_dev_frame = false;
    for (uint32_t f = 0; f < k; ++f) {
        const HostDest dest = classify_host_pointer(frames_host[f]);
        if (dest.bad) { snprintf(g_err, sizeof g_err, "rt_gang_render_frame: the frame pointer is device memory"); return RT_ERR_INVALID_ARGUMENT; }
        if (dest.pinned && dest.dev_alias && dest.room >= frame_bytes && knob(RT_DEBUG_HOST_COPY) != kCopyDirect) alias[f] = dest.dev_alias;
        else need_dev_frame = true;
    }
    if (need_dev_frame && g->frame_cap < frame_bytes) {
        if (g->d_frame) HIP_TRY(hipFree(g->d_frame));
        g->d_frame = nullptr; g->frame_cap = 0;
        HIP_TRY(hipMalloc(&g->d_frame, frame_bytes));
        // pixels outside the listed buckets: zero, never stale device memory (on the stream of the blit that writes the frame)
        HIP_TRY(hipMemsetAsync(g->d_frame, 0, frame_bytes, g->comm_streams[0]));
        g->frame_cap = frame_bytes;
    } else if (need_dev_frame && new_list) {
        HIP_TRY(hipMemsetAsync(g->d_frame, 0, g->frame_cap, g->comm_streams[0]));      // ... nor what an earlier tile list left there
    }
    rt_stats total{};
    auto fail = [&](rt_status st) { gang_sync_all(g); return st; };      // nothing of this gang may still be running when an error returns
    for (uint32_t f = 0; f < k; ++f) {
        const int p = (int)(f & 1u);
        // 1. every device renders its shard (asynchronous unless counters are wanted)
        for (size_t d = 0; d < nd; ++d) {
            hipError_t e = hipSetDevice(g->devices[d]);
            if (e == hipSuccess && f >= 2) e = hipStreamWaitEvent(g->streams[d], g->ev_gathered[p][d], 0);      // the buffer's last reader
            if (e != hipSuccess) return fail(hip_fail(e, "gang render", __LINE__));
            if (!L.shard[d].empty()) {
                rt_stats st{};
                rt_status rs = rt_render_tiles_device(g->scenes[d], o, trav, L.shard[d].data(), (uint32_t)L.shard[d].size(), g->d_shard[p][d], g->streams[d],
                                                      (stats && f == 0) ? &st : nullptr);
                if (rs != RT_OK) return fail(rs);
                if (stats && f == 0) {
                    total.primary += st.primary; total.hits += st.hits; total.shadow += st.shadow; total.occluded += st.occluded;
                    total.sphere_tests += st.sphere_tests; total.bound_tests += st.bound_tests; total.tests_executed += st.tests_executed;
                    total.primary_tests += st.primary_tests;
                    total.device_ms = std::max(total.device_ms, st.device_ms);
                    if (st.longest_wave_cycles > total.longest_wave_cycles) { total.longest_wave_cycles = st.longest_wave_cycles; total.longest_wave_ref100mhz = st.longest_wave_ref100mhz; }
                }
            }
            if ((e = hipEventRecord(g->ev_rendered[p][d], g->streams[d])) != hipSuccess) return fail(hip_fail(e, "gang render", __LINE__));
            if ((e = hipStreamWaitEvent(g->comm_streams[d], g->ev_rendered[p][d], 0)) != hipSuccess) return fail(hip_fail(e, "gang render", __LINE__));
        }
        // 2. the one collective on the data path: equal-length u8 shards to the root GPU
        ncclResult_t ne = r->GroupStart();
        if (ne != ncclSuccess) return fail(rccl_fail(ne, "ncclGroupStart", __LINE__));
        for (size_t d = 0; d < nd; ++d) {
            ne = r->Gather(g->d_shard[p][d], d == 0 ? g->d_gathered[p] : nullptr, shard_bytes, ncclUint8, 0, g->comms[d], g->comm_streams[d]);
            if (ne != ncclSuccess) { (void)r->GroupEnd(); return fail(rccl_fail(ne, "ncclGather", __LINE__)); }
        }
        if ((ne = r->GroupEnd()) != ncclSuccess) return fail(rccl_fail(ne, "ncclGroupEnd", __LINE__));
        for (size_t d = 0; d < nd; ++d) {
            hipError_t e = hipSetDevice(g->devices[d]);
            if (e == hipSuccess) e = hipEventRecord(g->ev_gathered[p][d], g->comm_streams[d]);
            if (e != hipSuccess) return fail(hip_fail(e, "gang gather", __LINE__));
        }
        // 3. root: set_pixels_from_buffer for every bucket (render.rs:112-126, 422-424) -- straight into the caller's frame when it is pinned
        hipError_t e = hipSetDevice(g->devices[0]);
        if (e != hipSuccess) return fail(hip_fail(e, "gang blit", __LINE__));
        uint8_t *target = alias[f] ? alias[f] : g->d_frame;
        rt_status bs = rt_blit_tiles_device(g->scenes[0], o, L.gathered_regs.data(), (uint32_t)L.gathered_regs.size(), L.gathered_off.data(), g->d_gathered[p],
                                            target, g->comm_streams[0]);
        if (bs != RT_OK) return fail(bs);
        if (!alias[f]) {
            if ((e = hipMemcpyAsync(frames_host[f], g->d_frame, frame_bytes, hipMemcpyDeviceToHost, g->comm_streams[0])) != hipSuccess)
                return fail(hip_fail(e, "gang copy", __LINE__));
        }
    }
    for (size_t d = 0; d < nd; ++d) {
        hipError_t e = hipSetDevice(g->devices[d]);
        if (e == hipSuccess) e = hipStreamSynchronize(g->streams[d]);
        if (e == hipSuccess) e = hipStreamSynchronize(g->comm_streams[d]);
        if (e != hipSuccess) return fail(hip_fail(e, "gang synchronize", __LINE__));
    }
    if (stats) *stats = total;
    return RT_OK;
}

rt_status rt_gang_render_frame(rt_gang *g, const rt_options *o, rt_traversal trav, const rt_region *tiles, uint32_t n,
                               uint8_t *frame_rgba_host, rt_stats *stats)
{
    if (!g || !check_common(g->scenes.empty() ? nullptr : g->scenes[0], o, tiles, n, frame_rgba_host)) {
        if (!g) snprintf(g_err, sizeof g_err, "NULL gang");
        return RT_ERR_INVALID_ARGUMENT;
    }
    if (rt_tiles_rgba_bytes(tiles, n) == 0) { snprintf(g_err, sizeof g_err, "rt_gang_render_frame: empty region in the tile list"); return RT_ERR_INVALID_REGION; }
    std::lock_guard<std::mutex> lk(g->mu);
    return gang_render(g, o, trav, tiles, n, &frame_rgba_host, 1, stats);
}

rt_status rt_gang_render_frames(rt_gang *g, const rt_options *o, rt_traversal trav, const rt_region *tiles, uint32_t n,
                                uint8_t *const *frames_rgba_host, uint32_t n_frames, rt_stats *stats)
{
    if (!g || !frames_rgba_host || n_frames == 0 || !check_common(g->scenes.empty() ? nullptr : g->scenes[0], o, tiles, n, frames_rgba_host[0])) {
        if (!g || !frames_rgba_host || n_frames == 0) snprintf(g_err, sizeof g_err, "rt_gang_render_frames: NULL argument or no frames");
        return RT_ERR_INVALID_ARGUMENT;
    }
    for (uint32_t f = 0; f < n_frames; ++f)
        if (!frames_rgba_host[f]) { snprintf(g_err, sizeof g_err, "rt_gang_render_frames: frame %u is NULL", f); return RT_ERR_INVALID_ARGUMENT; }
    if (rt_tiles_rgba_bytes(tiles, n) == 0) { snprintf(g_err, sizeof g_err, "rt_gang_render_frames: empty region in the tile list"); return RT_ERR_INVALID_REGION; }
    std::lock_guard<std::mutex> lk(g->mu);
    return gang_render(g, o, trav, tiles, n, frames_rgba_host, n_frames, stats);
}

#ifdef RT_TEST_HOOKS
// Test infrastructure (rt_debug.h): a stand-in for librccl.so, by path; NULL: the real library again.  Gangs keep the one they were made with.
rt_status rt_debug_rccl_library(const char *path)
{
    std::shared_ptr<Rccl> r;
    if (path && *path) {
        r = std::make_shared<Rccl>();
        load_rccl(*r, { path });
        if (!r->error.empty() || !r->Gather) { snprintf(g_err, sizeof g_err, "rt_debug_rccl_library: %s", r->error.c_str()); return RT_ERR_INVALID_ARGUMENT; }
    }
    std::lock_guard<std::mutex> lk(g_standin_mu);
    g_standin = r;
    return RT_OK;
}

// Test infrastructure (rt_debug.h): the gang's sharding arithmetic without a device.
rt_status rt_debug_gang_layout(const rt_region *tiles, uint32_t n, uint32_t n_devices, uint32_t *device_of, uint32_t *px_offset, uint64_t *shard_px,
                               uint64_t *padded_px)
{
    if (!tiles || n == 0 || n_devices == 0 || !device_of || !px_offset || !shard_px || !padded_px) {
        snprintf(g_err, sizeof g_err, "rt_debug_gang_layout: NULL argument");
        return RT_ERR_INVALID_ARGUMENT;
    }
    GangLayout L;
    gang_layout(tiles, n, n_devices, L);
    memcpy(device_of, L.device_of.data(), sizeof(uint32_t) * n);
    memcpy(px_offset, L.px_offset.data(), sizeof(uint32_t) * n);
    memcpy(shard_px, L.shard_px.data(), sizeof(uint64_t) * n_devices);
    *padded_px = L.max_px;
    return RT_OK;
}

// Test infrastructure (rt_debug.h): what the scene's cost map (tests per primary ray, its shadow ray included) predicts for the
// shards of a frame dealt over n_devices: cost[d] = sum over device d's buckets of the map's value under every 4th pixel x 16.
rt_status rt_debug_shard_costs(rt_scene *s, const rt_options *o, const rt_region *tiles, uint32_t n, uint32_t n_devices, double *cost)
{
    if (!s || !o || !tiles || !cost || n_devices == 0) { snprintf(g_err, sizeof g_err, "rt_debug_shard_costs: NULL argument"); return RT_ERR_INVALID_ARGUMENT; }
    HIP_TRY(hipSetDevice(s->device));
    const std::vector<uint32_t> *map = cost_map_of(s);
    if (!map) { snprintf(g_err, sizeof g_err, "rt_debug_shard_costs: the scene has no cost map (no hierarchy)"); return RT_ERR_UNSUPPORTED; }
    constexpr int R = (int)kCostRes;
    const unsigned w = o->width, h = o->height;
    for (uint32_t d = 0; d < n_devices; ++d) cost[d] = 0.0;
    for (uint32_t i = 0; i < n; ++i) {
        double c = 0.0;
        for (unsigned y = tiles[i].b; y < tiles[i].t; y += 4)
            for (unsigned x = tiles[i].l; x < tiles[i].r; x += 4) {
                const int X = std::clamp((int)((uint64_t)x * R / w), 0, R - 1);
                const int Y = std::clamp((int)std::floor(((double)y - h / 2.0) * R / w + R / 2.0), 0, R - 1);
                c += 16.0 * ((*map)[(size_t)Y * R + X] + kFixedBlockCost / 256.0);
            }
        cost[i % n_devices] += c;
    }
    return RT_OK;
}
#endif  // RT_TEST_HOOKS

}  // extern "C"
