// rt_flat_f64.hpp -- RT_TRAVERSAL_FLAT for the f64 instantiation (the reference's type-alias swap, vec.rs:6): rt_flat_wf.hpp's
// LDS-staged wavefront pipeline with the f32 scan's conservative FILTER in front of the exact test (rt_flat_sc.hpp).
//
// f64 has no packed arithmetic and a 64-byte scalar record holds two items instead of four, so the scalar-fed two-rays-per-lane
// design of the f32 scan does not carry over; what does is the filter: almost every ray x item test ends in `disc < 0`, and a
// rejection needs disc's sign, not its bits.  Per item and ray the loops form a BOUND of disc with fused multiply-adds --
//      primary:  b' = fma(vz, dz, fma(vy, dy, vx*dx)) ; bound = fma(b', b', K)                              4 instead of 8 operations
//      shadow:   u = fma(cz', 2oz', fma(cy', 2oy', fma(cx', 2ox', -Pm))) ; b' = CL - OL ; bound = fma(b', b', u) + G     6 instead of 16
// -- with K / G per item and OL, Pm per ray as in rt_flat_sc.hpp (same derivation with eps = 2^-53: disc >= 0 implies bound >= 0; the
// constants are formed in f64 itself, whose few-eps roundings disappear in the margins' factor of eight / three; absolute term
// 2^-1000 for results near the subnormal range).  Four items are rejected by one branch; a group with a candidate gets the reference's
// eight / sixteen individually rounded operations (primitive.rs:55-72) from the exact arrays (global memory: rare), in item order,
// with the exact root path behind it -- what is computed for a survivor is rt_flat_wf.hpp's, bit for bit.  k_flat_filter_check_f64
// evaluates both sides for every ray x item pair of a frame (rt_debug_flat_filter_check).
#pragma once
#include "rt_flat_wf.hpp"

namespace rt {

constexpr unsigned kFlatF64Tail = 4;       // pad records behind n_padded: a group's successor is fetched before the group is evaluated

// The high word of a bound, kept apart from its low word (left alone the compiler ANDs whole 64-bit values: twice the instructions).
__device__ __forceinline__ int bound_sign_word(double bound)
{
    int hi = __double2hiint(bound);
    asm("" : "+v"(hi));
    return hi;
}

struct FlatF64View {
    const Quad<double> *pf;     // primary filter: {vx, vy, vz, K} per item, DFS order, padded like FlatView (pad: K = -inf)
    const Quad<double> *sf;     // shadow filter: {cx', cy', cz', CL} per item of the shadow array (radius descending)
    const double *sg;           // shadow filter: G per item (pad: -inf)
    V3<double> centre;          // the point c' and o' are taken relative to (centroid of the item centres)
};

__device__ __forceinline__ double flat_filter_constant_f64(double vv, double rr)
{
    // rr - vv + 2^-46 (vv + rr): 128 eps against the 17 eps the proof needs; + 2^-50 (vv + rr) for this expression's own roundings
    return (rr - vv) + (vv + rr) * (0x1p-46 + 0x1p-50) + 0x1p-1000;
}
__device__ __forceinline__ double flat_shadow_filter_g_f64(double cc, double rr)
{
    return (rr - cc) + (cc + rr) * (0x1p-45 + 0x1p-50) + 0x1p-1000;
}
// Per ray: -Pm = -(|o'|^2 (1 - m)), m = 2^-45, one step towards zero for the product's rounding.
__device__ __forceinline__ double flat_shadow_filter_npm_f64(double p)
{
    double pm = p * (1.0 - 0x1p-45);
    if (pm > 0.0) pm = __longlong_as_double(__double_as_longlong(pm) - 1ll);
    return -pm;
}

__global__ void k_build_flat_f64(const Item<double> *__restrict__ items, const unsigned *__restrict__ shadow_order, unsigned n, unsigned n_padded,
                                 V3<double> eye, V3<double> centre, V3<double> sdir, Quad<double> *__restrict__ pf, Quad<double> *__restrict__ sf,
                                 double *__restrict__ sg)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_padded + kFlatF64Tail) return;                                            // the scans fetch one group ahead
    if (i < n) {
        const Item<double> it = items[i];
        const V3<double> v = { it.cx - eye.x, it.cy - eye.y, it.cz - eye.z };            // primitive.rs:56, the values k_build_flat stores
        pf[i] = { v.x, v.y, v.z, flat_filter_constant_f64(dot(v, v), it.r * it.r) };
        const Item<double> sh = items[shadow_order[i]];
        const double cx = sh.cx - centre.x, cy = sh.cy - centre.y, cz = sh.cz - centre.z;
        const double cl = __builtin_fma(cz, sdir.z, __builtin_fma(cy, sdir.y, cx * sdir.x));
        const double cc = __builtin_fma(cz, cz, __builtin_fma(cy, cy, cx * cx));
        sf[i] = { cx, cy, cz, cl };
        sg[i] = flat_shadow_filter_g_f64(cc, sh.r * sh.r);
    } else {
        pf[i] = { 0.0, 0.0, 0.0, -inf<double>() };
        sf[i] = { 0.0, 0.0, 0.0, 0.0 };
        sg[i] = -inf<double>();
    }
}

// Item i of a pair-interleaved array (rt_flat.hpp): half (i & 1) of quads 2 (i / 2) and 2 (i / 2) + 1.
__device__ __forceinline__ void flat_exact_terms(const Quad<double> *__restrict__ arr, unsigned i, double &a, double &b, double &c, double &d)
{
    const double *q = reinterpret_cast<const double *>(arr + 2 * (i >> 1));
    const unsigned h = i & 1u;
    a = q[0 + h]; b = q[2 + h]; c = q[4 + h]; d = q[6 + h];
}

template <int CHUNK>
__global__ __launch_bounds__(kBlockThreads) void k_flat_primary_f64(FlatView<double> sc, FlatF64View fx, unsigned width, unsigned height, unsigned spp,
                                                                   const TileDev *__restrict__ tiles, unsigned n_tiles, SampleBuf<double> sb,
                                                                   Quad<double> *__restrict__ queue1, FlatQueues *__restrict__ queues,
                                                                   Counters *__restrict__ counters)
{
    typedef double T;

    unsigned lo = 0, hi = n_tiles - 1;
    while (lo < hi) {
        unsigned mid = (lo + hi + 1) >> 1;
        if (tiles[mid].blk_first <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const TileDev tile = tiles[lo];
    const unsigned lb = blockIdx.x - tile.blk_first;
    const unsigned bx = lb % tile.blks_x, by = lb / tile.blks_x;
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned x = tile.l + bx * kFlatBlockW + (wave & 1) * 8 + (lane & 7);
    const unsigned y0 = tile.b + by * kFlatBlockH + (wave >> 1) * 8 + (lane >> 3);
    unsigned ys[kFlatR];
    bool inside[kFlatR];
#pragma unroll
    for (int r = 0; r < kFlatR; ++r) {
        ys[r] = y0 + 16u * r;
        inside[r] = x < tile.r && ys[r] < tile.t;
    }

    const unsigned ssx = blockIdx.y / spp, ssy = blockIdx.y % spp;            // one sample per thread slot (grid.y = spp*spp)
    const T ssf = T(spp);
    const T fw = T(width), fh = T(height);
    const T half_w = fw / T(2.0), half_h = fh / T(2.0);
    const V3<T> eye = sc.eye, light = sc.light;
    const unsigned n = sc.n_padded;

    V3<T> dir[kFlatR];
#pragma unroll
    for (int r = 0; r < kFlatR; ++r) {
        const T xres = T(x) + T(ssx) / ssf;                                    // render.rs:238-243
        const T yres = T(ys[r]) + T(ssy) / ssf;
        dir[r] = normalized(V3<T>{ xres - half_w, (fh - yres) - half_h, fw });
    }

    // ---------------- primary rays: nearest hit, strict `<`, first item in DFS order wins ties ----------------
    T best[kFlatR];
    unsigned best_i[kFlatR];
#pragma unroll
    for (int r = 0; r < kFlatR; ++r) { best[r] = inf<T>(); best_i[r] = 0; }

    // The scan is wave-uniform (every ray of a wave looks at the same item at the same moment), so an item's four terms are scalars:
    // they arrive through the scalar cache and feed the FMAs as SGPR operands -- no LDS staging, no barrier.  A bound is a candidate
    // iff it is >= 0, i.e. iff its sign bit is clear (a bound is never -0: it is a sum with a positive term), so the group's
    // verdict is the AND of the bounds' high words: one 32-bit v_and per bound instead of a 64-bit maximum.
    // Two register sets, used in turn (no copies): while group A is evaluated, group B's records are on their way, and vice versa.
    auto group = [&](const Quad<T> (&f4)[4], unsigned j) {
        int all_negative = -1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const Quad<T> f = f4[k];
#pragma unroll
            for (int r = 0; r < kFlatR; ++r) {
                const T bp = __builtin_fma(f.z, dir[r].z, __builtin_fma(f.y, dir[r].y, f.x * dir[r].x));
                all_negative &= bound_sign_word(__builtin_fma(bp, bp, f.w));
            }
        }
        if (__builtin_expect(all_negative >= 0, 0)) {                       // rare: the bound cannot rule out one of the 4 items for some lane
#pragma unroll
            for (int k = 0; k < 4; ++k) {                                   // item order: first in DFS order wins ties
                T vx, vy, vz, vv;
                flat_exact_terms(sc.prim, j + k, vx, vy, vz, vv);
                const T rr = sc.prim_rr[j + k];
#pragma unroll
                for (int r = 0; r < kFlatR; ++r) {
                    const T b = (vx * dir[r].x + vy * dir[r].y) + vz * dir[r].z;          // primitive.rs:57
                    const T disc = (b * b - vv) + rr;                                      // primitive.rs:58
                    if (!(disc < T(0.0))) {
                        const T s = sqrt_rn_lean(disc);
                        const T t2 = b + s;
                        if (!(t2 < T(0.0))) {
                            const T t1 = b - s;
                            const T d = t1 > T(0.0) ? t1 : t2;
                            if (!(d >= best[r])) { best[r] = d; best_i[r] = j + k; }
                        }
                    }
                }
            }
        }
    };
    Quad<T> ga[4], gb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ga[k] = fx.pf[k];
    for (unsigned j = 0; j < n; j += 8) {                                    // n is a multiple of 8; the arrays end in four more pad records
#pragma unroll
        for (int k = 0; k < 4; ++k) gb[k] = fx.pf[j + 4 + k];
        group(ga, j);
#pragma unroll
        for (int k = 0; k < 4; ++k) ga[k] = fx.pf[j + 8 + k];
        group(gb, j + 4);
    }

    // ---------------- shade (render.rs:190-199), store the sample, queue the shadow ray ----------------
    unsigned c_hits = 0, c_shadow = 0;
#pragma unroll
    for (int r = 0; r < kFlatR; ++r) {
        bool need_shadow = false;
        T gdot = T(0.0);
        V3<T> sp = { T(0.0), T(0.0), T(0.0) };
        uint8_t state = kMiss;
        if (inside[r] && !(best[r] == inf<T>())) {
            ++c_hits;
            const Item<T> it = sc.items[best_i[r]];
            const V3<T> c = { it.cx, it.cy, it.cz };
            const V3<T> nrm = normalized(add(eye, sub(mulf(dir[r], best[r]), c)));     // primitive.rs:83
            gdot = dot(nrm, light);
            if (gdot >= T(0.0)) {
                state = kAmbient;
            } else {
                need_shadow = true;
                ++c_shadow;
                state = kLit;                                                   // until a shadow pass finds an occluder
                const V3<T> ns = mulf(nrm, best[r] * rsqrt_exact(eps<T>()));
                sp = add(add(eye, mulf(dir[r], best[r])), ns);                  // render.rs:199
            }
        }
        const unsigned q = blockIdx.y * sb.n_px + (unsigned)out_index(tile, x, ys[r], 0);      // sample slot (tile-major pixel)
        if (inside[r]) { sb.state[q] = state; sb.gdot[q] = gdot; }
        wave_append(need_shadow, Quad<T>{ sp.x, sp.y, sp.z, owner_to_real<T>(q) }, queue1, &queues->n1);
    }

    if (counters) {
        counters += (blockIdx.x + blockIdx.y) % kCounterStripes;
        unsigned c_prim = 0;
#pragma unroll
        for (int r = 0; r < kFlatR; ++r) c_prim += inside[r] ? 1u : 0u;
        const unsigned long long p = wave_sum(c_prim), hh = wave_sum(c_hits), s2 = wave_sum(c_shadow);
        if (lane == 0) {
            atomicAdd(&counters->primary, p);
            atomicAdd(&counters->hits, hh);
            atomicAdd(&counters->shadow, s2);
        }
    }
}

// One shadow pass: rays of `queue_in` against the shadow items [item_begin, item_end) (multiples of CHUNK), any hit.
template <int CHUNK>
__global__ __launch_bounds__(kBlockThreads) void k_flat_shadow_f64(FlatView<double> sc, FlatF64View fx, unsigned item_begin, unsigned item_end,
                                                                  const Quad<double> *__restrict__ queue_in, const unsigned *__restrict__ n_in,
                                                                  Quad<double> *__restrict__ queue_out, unsigned *__restrict__ n_out, SampleBuf<double> sb,
                                                                  Counters *__restrict__ counters)
{
    typedef double T;
    const unsigned n_rays = *n_in;
    const unsigned first = blockIdx.x * (kBlockThreads * kFlatR);
    if (first >= n_rays) return;
    const unsigned lane = threadIdx.x & 63;

    V3<T> sp[kFlatR], o2[kFlatR];
    T ol[kFlatR], npm[kFlatR];
    unsigned owner[kFlatR];
    bool have[kFlatR], pending[kFlatR], occluded[kFlatR];
    const V3<T> sdir = mulf(sc.light, T(-1.0));                                // render.rs:206
#pragma unroll
    for (int r = 0; r < kFlatR; ++r) {
        const unsigned idx = first + threadIdx.x * kFlatR + r;                 // adjacent rays share a lane: dense waves
        have[r] = idx < n_rays;
        pending[r] = have[r];
        occluded[r] = false;
        owner[r] = 0;
        sp[r] = { T(0.0), T(0.0), T(0.0) };
        if (have[r]) {
            const Quad<T> e = queue_in[idx];
            sp[r] = { e.x, e.y, e.z };
            owner[r] = real_to_owner(e.w);
        }
        const V3<T> oc = sub(sp[r], fx.centre);                                // o'
        o2[r] = mulf(oc, T(2.0));
        ol[r] = __builtin_fma(sdir.z, oc.z, __builtin_fma(sdir.y, oc.y, sdir.x * oc.x));
        npm[r] = flat_shadow_filter_npm_f64(__builtin_fma(oc.z, oc.z, __builtin_fma(oc.y, oc.y, oc.x * oc.x)));
    }
    const unsigned n = min(item_end, sc.n_padded);

    // a settled (or absent) ray must not send the wave into the exact path again: -Pm = -inf makes every bound of its -inf
#pragma unroll
    for (int r = 0; r < kFlatR; ++r) if (!pending[r]) npm[r] = -inf<T>();
    bool any_pending = false;
#pragma unroll
    for (int r = 0; r < kFlatR; ++r) any_pending = any_pending || pending[r];
    if (__ballot(any_pending) != 0) {
        for (unsigned j = item_begin; j < n; j += 4) {                      // items as scalars (see k_flat_primary_f64); waves are independent
            int all_negative = -1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const Quad<T> f = fx.sf[j + k];
                const T g = fx.sg[j + k];
#pragma unroll
                for (int r = 0; r < kFlatR; ++r) {
                    const T u = __builtin_fma(f.z, o2[r].z, __builtin_fma(f.y, o2[r].y, __builtin_fma(f.x, o2[r].x, npm[r])));
                    const T bp = f.w - ol[r];
                    all_negative &= bound_sign_word(__builtin_fma(bp, bp, u) + g);
                }
            }
            const bool candidate = all_negative >= 0;
            if (__builtin_expect(__ballot(candidate) == 0, 1)) continue;    // wave-uniform: the scan's position stays a scalar
            if (candidate) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    T cx, cy, cz, rr;
                    flat_exact_terms(sc.shad, j + k, cx, cy, cz, rr);
#pragma unroll
                    for (int r = 0; r < kFlatR; ++r) {
                        if (!pending[r]) continue;
                        const V3<T> v = { cx - sp[r].x, cy - sp[r].y, cz - sp[r].z };      // primitive.rs:56
                        const T b = dot(v, sdir);
                        const T disc = (b * b - dot(v, v)) + rr;
                        if (!(disc < T(0.0))) {
                            const T t2 = b + sqrt_rn_lean(disc);
                            if (!(t2 < T(0.0))) { occluded[r] = true; pending[r] = false; npm[r] = -inf<T>(); }
                        }
                    }
                }
            }
            bool any = false;
#pragma unroll
            for (int r = 0; r < kFlatR; ++r) any = any || pending[r];
            if (__ballot(any) == 0) break;                                  // every ray of the wave is settled
        }
    }

    unsigned c_occ = 0;
#pragma unroll
    for (int r = 0; r < kFlatR; ++r) {
        if (have[r] && occluded[r]) { sb.state[owner[r]] = kShadowed; ++c_occ; }          // render.rs:211-213
        if (queue_out) wave_append(have[r] && !occluded[r], Quad<T>{ sp[r].x, sp[r].y, sp[r].z, owner_to_real<T>(owner[r]) }, queue_out, n_out);
    }
    if (counters) {
        counters += blockIdx.x % kCounterStripes;
        const unsigned long long oc = wave_sum(c_occ);
        if (lane == 0) atomicAdd(&counters->occluded, oc);
    }
}

// rt_debug_flat_filter_check for f64 scenes: what k_flat_filter_check (rt_flat_sc.hpp) counts, on the f64 arrays.
__global__ __launch_bounds__(kBlockThreads) void k_flat_filter_check_f64(FlatView<double> sc, FlatF64View fx, const unsigned *__restrict__ shadow_order,
                                                                        unsigned width, unsigned height, unsigned spp, unsigned long long *__restrict__ counts)
{
    typedef double T;
    const unsigned px = blockIdx.x * kBlockThreads + threadIdx.x;
    const unsigned x = px % width, y = px / width;
    if (y >= height) return;
    const T ssf = T(spp), fw = T(width), fh = T(height);
    const T half_w = fw / 2.0, half_h = fh / 2.0;
    const unsigned ssx = blockIdx.y / spp, ssy = blockIdx.y % spp;
    const T xres = T(x) + T(ssx) / ssf, yres = T(y) + T(ssy) / ssf;
    const V3<T> d = normalized(V3<T>{ xres - half_w, (fh - yres) - half_h, fw });
    unsigned exact = 0, bound = 0, bad = 0, sexact = 0, sbound = 0, sbad = 0;
    // shadow-type rays: from a point on the primary ray (0.75 .. 1 times the eye's distance from the scene centre along it) towards the light
    const V3<T> ec = sub(sc.eye, fx.centre);
    const V3<T> o = add(sc.eye, mulf(d, sqrt(dot(ec, ec)) * (0.75 + 0.0625 * T(blockIdx.y % 5u))));
    const V3<T> l = mulf(sc.light, -1.0);
    const V3<T> oc = sub(o, fx.centre);
    const V3<T> o2 = mulf(oc, 2.0);
    const T ol = __builtin_fma(l.z, oc.z, __builtin_fma(l.y, oc.y, l.x * oc.x));
    const T npm = flat_shadow_filter_npm_f64(__builtin_fma(oc.z, oc.z, __builtin_fma(oc.y, oc.y, oc.x * oc.x)));
    (void)shadow_order;
    for (unsigned i = 0; i < sc.n_items; ++i) {
        T vx, vy, vz, vv;
        flat_exact_terms(sc.prim, i, vx, vy, vz, vv);
        const T b = (vx * d.x + vy * d.y) + vz * d.z;
        const T disc = (b * b - vv) + sc.prim_rr[i];
        const Quad<T> f = fx.pf[i];
        const T bf = __builtin_fma(f.z, d.z, __builtin_fma(f.y, d.y, f.x * d.x));
        const T bnd = __builtin_fma(bf, bf, f.w);
        const bool ce = disc >= 0.0, cb = bnd >= 0.0;
        exact += ce; bound += cb; bad += ce && !cb;
        T cx, cy, cz, rr;
        flat_exact_terms(sc.shad, i, cx, cy, cz, rr);
        const V3<T> v = { cx - o.x, cy - o.y, cz - o.z };
        const T sb = dot(v, l);
        const T sdisc = (sb * sb - dot(v, v)) + rr;
        const Quad<T> g = fx.sf[i];
        const T u = __builtin_fma(g.z, o2.z, __builtin_fma(g.y, o2.y, __builtin_fma(g.x, o2.x, npm)));
        const T bp = g.w - ol;
        const T sbnd = __builtin_fma(bp, bp, u) + fx.sg[i];
        const bool se = sdisc >= 0.0, sbb = sbnd >= 0.0;
        sexact += se; sbound += sbb; sbad += se && !sbb;
    }
    const unsigned long long e = wave_sum(exact), bo = wave_sum(bound), ba = wave_sum(bad), se = wave_sum(sexact), sbo = wave_sum(sbound), sba = wave_sum(sbad);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&counts[0], e); atomicAdd(&counts[1], bo); atomicAdd(&counts[2], ba);
        atomicAdd(&counts[3], se); atomicAdd(&counts[4], sbo); atomicAdd(&counts[5], sba);
    }
}

}  // namespace rt
