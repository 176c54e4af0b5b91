// rt_capi_launch.hpp -- part of rt_capi.hip: which kernel a pass runs (loop flavours, k_render_skip_fast, two rays per lane, the flat
// pipeline), enqueueing a pass, reading its counters.
// (included by rt_capi.hip where its text used to stand: nothing here is a header of its own)
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
    else if constexpr (sizeof(T) == 8 && !COUNT && (VAR & 18) == 18 && COOP) return &rt::k_render_skip_f64_coop<VAR, MODE>;
    else if constexpr (sizeof(T) == 8 && !COUNT && (VAR & 18) == 18) return &rt::k_render_skip_f64<VAR, MODE>;
    else return &rt::k_render_skip<T, COUNT, VAR, MODE, COOP>;
}

// The arguments of k_render_skip2_fast (rt_skip2_fast.hpp): the two-ray kernel's, in the order it reads them
template <bool FUSED>
rt::Fast2Args fast2_args(const rt_scene *s, const rt::BlockList &order, unsigned w, unsigned h, unsigned spp, unsigned frame_w, void *dst)
{
    const rt::SkipView<float> sv = skip_view_of<float>(s);
    rt::Fast2Args a{};
    a.order = order.d; a.wg_first = order.wg_first;
    a.width = w; a.height = h; a.spp = spp;
    a.nb = (FUSED ? sv.n_fnodes : sv.n_nodes) * (unsigned)sizeof(rt::Node<float>);
    a.walk_prim = FUSED ? static_cast<const void *>(sv.xfprim) : static_cast<const void *>(sv.xprim);
    a.frame_w = frame_w;
    a.items = sv.items; a.own = sv.xown;
    a.eye[0] = sv.eye.x; a.eye[1] = sv.eye.y; a.eye[2] = sv.eye.z; a.light[0] = sv.light.x; a.light[1] = sv.light.y; a.light[2] = sv.light.z;
    a.walk_shad = FUSED ? static_cast<const void *>(sv.xfshad) : static_cast<const void *>(sv.xshad);
    a.exact_shad = FUSED ? static_cast<const void *>(sv.fshad) : static_cast<const void *>(sv.shad);
    a.fc = sv.fc;
    a.dst = dst;
    return a;
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
    constexpr bool kCoopFlavour = !COUNT && (VAR == 19 || VAR == 23 || VAR == 31);       // (f32, and since round 6 f64: the filtered assembly loops)
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
                if constexpr ((VAR & 16) != 0) {
                    if (order.d && knob(RT_DEBUG_FAST_KERNEL) != 0) {      // the same kernel with its arguments fetched where they are needed (rt_skip2_fast.hpp; spp 1: lists are not dealt)
                        g_launch_flags |= RT_LAUNCH_FAST_KERNEL;
                        hipLaunchKernelGGL((rt::k_render_skip2_fast<rt::kSkipOne, (VAR & 4) != 0>), rgrid, b2, 0, stream, fast2_args<(VAR & 4) != 0>(s, order, w, h, spp, frame_w, d_out));
                        return RT_OK;
                    }
                }
                hipLaunchKernelGGL((rt::k_render_skip2<rt::kSkipOne, (VAR & 16) != 0, (VAR & 4) != 0>), rgrid, b2, 0, stream, skip_view_of<float>(s), w, h, spp, d_tab, nt, d_out, sb, frame_w,
                                   order.d, order.wg_first);
                return RT_OK;
            }
        }
        // Steady-state frames -- f32, one sample per pixel, a dispatch list, the filtered assembly loops -- run the kernel that was written
        // around a wave's fixed costs (rt_skip_fast.hpp), with or without cooperative quads; everything else the generic one.
        if constexpr (!COUNT && sizeof(T) == 4 && ((VAR & ~8) == 19 || (VAR & ~8) == 23)) {
            // (measured, one box, interleaved -- profiles/r06_ab_fast_vs_generic.log: WITHOUT cooperative quads the lean kernel is the generic one's
            // equal, 1 - 2 % behind it at 1080p, 1 % ahead at 2560x1440 -- a wave's shorter start does not shorten a frame that is as long as its
            // longest chain, and the late batch is a round trip the generic kernel's parked values do not make --; WITH them it is 39.8 against
            // 43.5 us at 1080p and 38.4 against 40.5 at 1600x900: eight workgroups per CU where the generic cooperative flavour has seven.  So it runs
            // where a list has holes; RT_DEBUG_FAST_KERNEL = 2 forces it for every ordered f32 spp-1 launch (tests, A/B))
            const long long fk = knob(RT_DEBUG_FAST_KERNEL);
            if (spp == 1 && order.d && !order.wg_first && lds == 0 && fk != 0 && (order.holes || fk == 2)) {
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
        // ... and the f64 twin of the lean kernel (rt_skip_fast64.hpp): 80 scalar / 61 vector registers and no scratch -- EIGHT waves per SIMD -- where
        // k_render_skip_f64 has 96 / 72 and 12 - 36 bytes (seven) and k_render_skip_f64_coop 96 vector registers (five): measured ahead of the generic
        // kernels with AND without cooperative quads (profiles/r06_f64_lean_kernel.log: 2560x1440 91.6 -> 82 us without, 1280x720 42.0 -> 34.5 with,
        // 1080p 55.2 -> 50.2), so every ordered f64 spp-1 launch takes it (RT_DEBUG_FAST_KERNEL = 0: never)
        if constexpr (!COUNT && sizeof(T) == 8 && ((VAR & ~8) == 19 || (VAR & ~8) == 23)) {
            const long long fk = knob(RT_DEBUG_FAST_KERNEL);
            if (spp == 1 && order.d && !order.wg_first && lds == 0 && fk != 0 && (!order.holes || s->coop.fanout != 0u)) {
                rt::FastArgs64 fa{};
                const rt::SkipView<double> sv = skip_view_of<double>(s);
                constexpr bool kFused = (VAR & 4) != 0;
                fa.order = order.d;
                fa.walk_prim = kFused ? sv.xfprim : sv.xprim;
                fa.exact_prim = kFused ? sv.fprim : sv.prim;
                fa.width = w; fa.height = h; fa.nbf = (kFused ? sv.n_fnodes : sv.n_nodes) * (unsigned)sizeof(rt::FNode); fa.frame_w = frame_w; fa.out = d_out;
                fa.items = sv.items; fa.own = sv.xown;
                fa.eye[0] = sv.eye.x; fa.eye[1] = sv.eye.y; fa.eye[2] = sv.eye.z; fa.light[0] = sv.light.x; fa.light[1] = sv.light.y; fa.light[2] = sv.light.z;
                fa.walk_shad = kFused ? sv.xfshad : sv.xshad; fa.exact_shad = kFused ? sv.fshad : sv.shad;
                memcpy(fa.fc, &s->fc, sizeof fa.fc);
                fa.trace = no_cost;
                fa.holes = order.holes; fa.n_holes = order.holes ? order.n_holes : 0u; fa.cv = s->coop;
                g_launch_flags |= RT_LAUNCH_FAST_KERNEL;
                if (order.holes) { count_event(RT_DEBUG_COUNT_COOP_LAUNCHES); g_launch_flags |= RT_LAUNCH_COOPERATIVE; }
                hipLaunchKernelGGL((rt::k_render_skip_fast64_coop<(VAR & ~8), (VAR & 8) != 0>), rgrid, b, 0, stream, fa);
                return RT_OK;
            }
        }
        if constexpr (!COUNT && (VAR == 19 || VAR == 23 || VAR == 31)) {
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
            bool lean2 = false;
            if constexpr ((VAR & 16) != 0) {
                // (measured, profiles/r06_two_ray_lean_kernel.log: ahead by 2 - 8 % on the pyramid scenes -- 3840x2160 127 -> 117 us, 1080p spp 4 473 -> 458,
                // 4096^2 spp 4 L8 2.656 -> 2.608 ms; config 5 itself the same, 2.873 / 2.881 --, behind by 1.4 % on a plain-stream scene whose list is
                // dealt to workgroups on the host, several descriptors each (100,000 arbitrary spheres, 15.87 / 16.09 ms: k_render_skip2 parks its
                // arguments once per workgroup, this kernel fetches them once per descriptor): that case keeps k_render_skip2)
                if (order.d && knob(RT_DEBUG_FAST_KERNEL) != 0 && ((VAR & 4) != 0 || !order.wg_first || knob(RT_DEBUG_FAST_KERNEL) == 2)) {
                    g_launch_flags |= RT_LAUNCH_FAST_KERNEL;
                    hipLaunchKernelGGL((rt::k_render_skip2_fast<rt::kSkipPacked, (VAR & 4) != 0>), dim3(rgrid.x, (unsigned)ns), b2, 0, stream,
                                       fast2_args<(VAR & 4) != 0>(s, order, w, h, spp, frame_w, sb.gdot));
                    lean2 = true;
                }
            }
            if (!lean2)
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

