// rt_capi_gang.hpp -- part of rt_capi.hip: rt_gang_* -- one frame over several GPUs of ONE process, one RCCL gather.
// (included by rt_capi.hip where its text used to stand: nothing here is a header of its own)
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

