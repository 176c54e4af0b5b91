// rt_capi_entry.hpp -- part of rt_capi.hip: the entry points of include/rtrace_hip.h (and, with -DRT_TEST_HOOKS, csrc/rt_debug.h) for one device.
// (included by rt_capi.hip where its text used to stand: nothing here is a header of its own)

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

// The automatic hierarchy for an arbitrary sphere list: csrc/host/hierarchy.hpp, exported so that the Python host runs the same code as the C++ one.
rt_status rt_build_hierarchy(const double *spheres, uint32_t n, uint32_t leaf_size, const double *eye, rt_precision precision,
                             void *items_out, void *bounds_out, rt_range *ranges_out, uint64_t *order_out, uint32_t *n_groups_out)
{
    if (!spheres || n == 0 || !items_out || !bounds_out || !ranges_out || !n_groups_out || (precision != RT_F32 && precision != RT_F64)) {
        snprintf(g_err, sizeof g_err, "rt_build_hierarchy: NULL argument, n == 0 or bad precision");
        return RT_ERR_INVALID_ARGUMENT;
    }
    for (size_t i = 0; i < (size_t)n * 4; ++i)
        if (!std::isfinite(spheres[i])) { snprintf(g_err, sizeof g_err, "rt_build_hierarchy: sphere %zu is not finite", i / 4); return RT_ERR_INVALID_ARGUMENT; }
    try {
        const rt_host::FlatHierarchy h = rt_host::build_hierarchy(spheres, n, leaf_size, eye);
        const size_t g = h.ranges.size() / 2;
        if (precision == RT_F32) {
            for (size_t i = 0; i < h.items.size(); ++i) static_cast<float *>(items_out)[i] = (float)h.items[i];
            for (size_t i = 0; i < h.bounds.size(); ++i) static_cast<float *>(bounds_out)[i] = (float)h.bounds[i];
        } else {
            memcpy(items_out, h.items.data(), h.items.size() * sizeof(double));
            memcpy(bounds_out, h.bounds.data(), h.bounds.size() * sizeof(double));
        }
        for (size_t i = 0; i < g; ++i) ranges_out[i] = rt_range{ h.ranges[2 * i], h.ranges[2 * i + 1] };
        if (order_out) memcpy(order_out, h.order.data(), h.order.size() * sizeof(uint64_t));
        *n_groups_out = (uint32_t)g;
    } catch (const std::exception &e) {
        snprintf(g_err, sizeof g_err, "rt_build_hierarchy: %s", e.what());
        return RT_ERR_OUT_OF_MEMORY;
    }
    return RT_OK;
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
    const auto t_setup = std::chrono::steady_clock::now();
    auto ms_since_setup = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_setup).count(); };
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
    // (the first stream of a process is ~19 ms -- the runtime makes its first hardware queue, tools/init_probe.hip --, and the library's code
    // object another ~3.5 ms at its first launch: a helper asks for a kernel's attributes meanwhile, which is what loads the code object)
    std::thread code_loader;
    try {
        code_loader = std::thread([device] {
            if (hipSetDevice(device) != hipSuccess) return;
            hipFuncAttributes fa;
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(rt::k_upload_words));
            (void)hipGetLastError();
        });
    } catch (...) {}                              // (no thread to be had: the first launch loads the code object, as it always did)
    const double t_before_stream = ms_since_setup();
    e = hipStreamCreateWithFlags(&s->cost_stream, hipStreamNonBlocking);
    s->setup_first_stream_ms = ms_since_setup() - t_before_stream;
    clk.lap("scene: stream");
    if (code_loader.joinable()) code_loader.join();
    clk.lap("scene: code object (helper)");
    if (e != hipSuccess) return fail(hip_fail(e, "hipStreamCreate(scene)", __LINE__));
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
    s->setup_total_ms = ms_since_setup();
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

rt_status rt_scene_setup_cost(const rt_scene *s, double *total_ms, double *stream_ms)
{
    if (!s || !total_ms || !stream_ms) { snprintf(g_err, sizeof g_err, "NULL argument"); return RT_ERR_INVALID_ARGUMENT; }
    *total_ms = s->setup_total_ms;
    *stream_ms = s->setup_first_stream_ms;
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


