// rt_capi_scene.hpp -- part of rt_capi.hip (one translation unit, split for reading): leasing of per-call contexts, tile tables,
// the scene's uploads and derived streams (skip-pointer, filtered, compacted, cooperative copy, flat-scan arrays).
// (included by rt_capi.hip where its text used to stand: nothing here is a header of its own)

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

template <typename T> rt_status upload_coop(rt_scene *s, const std::vector<rt::RawNode<T>> &raw);

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
            if ((st = upload_coop<T>(s, raw)) != RT_OK) return st;         // (with the filtered streams: the cooperative flavour is built for those loops)
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
        if ((st = upload_coop<T>(s, raw)) != RT_OK) return st;
        clk.lap("  streams: cooperative copy");
    }
    return RT_OK;
}

// The lane-cooperative walk's copy of the hierarchy (rt_coop.hpp): the nodes of the plain stream in breadth-first order, so that the
// children of a group are consecutive records.  Scenes whose largest child count (or number of top-level nodes) exceeds what a
// work-list word holds simply get none: the cooperative walk is an optimisation of the skip-pointer walk, never a requirement.
template <typename T>
rt_status upload_coop(rt_scene *s, const std::vector<rt::RawNode<T>> &raw)
{
    typedef typename rt::CNodeOf<T>::type Rec;
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
    if (e == hipSuccess) e = hipMalloc(&s->d_coop_prim, sizeof(Rec) * n);
    if (e == hipSuccess) e = hipMalloc(&s->d_coop_shad, sizeof(Rec) * n);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rt::k_build_coop<T>, dim3((n + 255) / 256), dim3(256), 0, s->cost_stream, s->d_prim, s->d_shad, (unsigned)sizeof(rt::Node<T>), d_perm, d_link, n,
                           static_cast<Rec *>(s->d_coop_prim), static_cast<Rec *>(s->d_coop_shad));
        e = hipGetLastError();
    }
    { const hipError_t se = hipStreamSynchronize(s->cost_stream); if (e == hipSuccess) e = se; }
    if (d_perm) (void)hipFree(d_perm);
    if (d_link) (void)hipFree(d_link);
    if (e != hipSuccess) return hip_fail(e, "upload_coop", __LINE__);
    s->coop.prim = s->d_coop_prim;
    s->coop.shad = s->d_coop_shad;
    s->coop.n_roots = n_roots;
    s->coop.fanout = fanout;
    return RT_OK;
}

