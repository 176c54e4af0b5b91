// rt_flat_sc.hpp -- RT_TRAVERSAL_FLAT for f32, the kernels the product runs: every item of the scene against every ray, with
// the items fed through the SCALAR path, two rays per lane on packed math, and a conservative filter in front of the exact test.
//
// A linear scan is wave-uniform by construction: all rays of a wave test the same item at the same moment.  On CDNA that makes
// the item a scalar: records arrive by s_load (scalar cache, the scene's < 1 MB in L2 behind it) and feed the vector ALU as SGPR
// operands -- no LDS staging, no ds_read, no barrier.  tools/valu_issue_probe.hip measured what the operand costs: a VOP2 that
// reads a different SGPR than its predecessor occupies the SIMD for 4.1 cycles instead of 2.2, exactly what a packed
// v_pk_mul/add/fma_f32 takes with its scalar operand (half of an SGPR pair, broadcast by op_sel) for free.  So a lane carries TWO
// rays in VGPR pairs and every instruction of the scan is packed.  The scan loops are generated assembly (tools/gen_flat_asm.py
// -> rt_flat_rot.hpp), which also documents the filter: almost every test ends in `disc < 0`, and for that only the sign of disc
// is needed -- per item the loops form a bound of disc with 4 (primary) / 6 (shadow) packed instructions whose margin covers every
// rounding of both computations (disc >= 0 implies bound >= 0; flat_filter_constant / flat_shadow_filter_g below, checked
// exhaustively by k_flat_filter_check), reject a group of four items with one branch, and run the reference's eight / sixteen
// individually rounded operations (primitive.rs:55-72) only for the items whose bound is >= 0 for some ray.  What is computed
// for those is the reference's, bit for bit, in item order; the filter only decides what is looked at.
// round 1's LDS + packed-math kernels (rt_flat_wf.hpp) remain what f64 runs and are selectable for f32 with csrc/rt_debug.h
// RT_DEBUG_FLAT_KERNELS = 0.
//
//   k_flat_primary_sc  one lane = two pixels of one sample (a wave = a 16x8 patch): primary rays, nearest-hit scan of all items in
//                      DFS order (strict `<`: the first item in DFS order wins ties, primitive.rs:79), shade; the sample's {state,
//                      n.light} is stored and rays that need a shadow test are appended to queue 1 (one atomic per wave and half)
//   k_flat_shadow_sc   one lane = two queued shadow rays, any hit over a range of the RADIUS-SORTED item array (pass A: the 1,024
//                      largest spheres, which settle 89 % of the occluded rays; survivors re-packed into queue 2; pass B: the
//                      rest); a wave leaves as soon as all its rays are settled
#pragma once
#include "rt_flat_wf.hpp"
#include "rt_flat_rot.hpp"

namespace rt {

// Filter groups -- primary: FOUR items {vx[4], vy[4], vz[4], K[4]}; shadow: THREE items {cx'[3], cy'[3], cz'[3], CL[3], G[3], -} with
// c' = centre - scene centre -- and one exact record per item: primary {vx, vy, vz, vv, rr, 0, 0, 0}, shadow {cx, cy, cz, rr}.
struct alignas(64) FGroup { float f[16]; };
struct alignas(32) FExact { float f[8]; };
struct alignas(16) FExactShadow { float f[4]; };
constexpr unsigned kFlatFilterItems = 4;        // primary filter groups
constexpr unsigned kFlatShadowItems = 3;        // shadow filter groups
constexpr unsigned kFlatPadGroups = 2;          // behind the last pair: the scans load one pair ahead

struct FlatScView {
    const FGroup *pf;       // primary filter groups, DFS order
    const FExact *pe;       // primary exact records, DFS order
    const FGroup *sg;       // shadow filter groups, radius descending
    const FExactShadow *se; // shadow exact records, radius descending
    const Item<float> *items;   // centres for the normal of the winning item
    uint32_t n_items;
    uint32_t n_fbytes;      // primary: 128 * number of filter group pairs
    uint32_t n_sbytes;      // shadow: 128 * number of filter group pairs
    V3<float> centre;       // shadow filter: the point the centres and origins are taken relative to
    V3<float> light, eye;
};

// The filter constant of an item (the proof is in tools/gen_flat_asm.py): the exact test forms, each operation rounded,
//      b = (vx*dx + vy*dy) + vz*dz ; disc = (b*b - vv) + rr                              (primitive.rs:57-58)
// and the filter b' = fma(vz, dz, fma(vy, dy, vx*dx)) ; bound = fma(b', b', K).  With eps = 2^-24 and |d| <= 1 + 2 eps, both b and
// b' lie within 3.1 eps |v| of the exact dot product, so |b*b - b'*b'| <= 12.4 eps |v|^2; the three roundings of disc add at most
// 1.1 eps |v|^2 + 2 eps rr, and the stored vv is within 4 eps of |v|^2.  Hence disc >= 0 implies b'*b' - vv + rr >= -(14 eps vv +
// 3 eps rr) -- K = rr - vv + 2^-17 (vv + rr) + 2^-140, rounded UP, leaves a factor of eight on the relative term, and the absolute
// term covers the results that are subnormal (scenes scaled to 1e-20: errors there are absolute, <= 2^-149 each).
__device__ __forceinline__ float flat_filter_constant(float vv, float rr)
{
    const double k = ((double)rr - (double)vv) + ((double)vv + (double)rr) * 0x1p-17 + 0x1p-140;
    float f = (float)k;
    if ((double)f < k) f = __uint_as_float(__float_as_uint(f) + (f >= 0.f ? 1u : 0xFFFFFFFFu));      // next float up (f is finite and != -0)
    return f;
}

// Shadow rays have their own origin o, so nothing of vv can be pre-formed from v = c - o.  Relative to a point m0 inside the scene,
// c' = fl(c - m0) and o' = fl(o - m0):  |c' - o'|^2 = |c'|^2 + |o'|^2 - 2 c'.o'  and  b = c'.l - o'.l, so per item only
//      u = fma(cz', 2oz', fma(cy', 2oy', fma(cx', 2ox', -Pm))) ; b' = CL - OL ; bound = fma(b', b', u) + G
// remain (six packed instructions), with CL = c'.l and G per item, OL = o'.l and Pm = |o'|^2 (1 - m) per ray, m = 2^-16.  With
// S = |c'|^2 + |o'|^2: b' and the exact b are within 4.5 eps and 5.3 eps (|c'| + |o'|) of c'.l - o'.l (the re-centring costs
// eps (|c'| + |o'|)), so b b and b' b' differ by <= 39.6 eps S; the exact vv and |c' - o'|^2 by <= 14.4 eps S; the exact test's roundings
// add 2.2 eps S + 2 eps rr and the filter's own 21 eps S + eps rr: disc >= 0 implies bound >= -(77.2 eps S + 3 eps rr) + margin, and
// the margin m (S + rr) = 256 eps (S + rr) (+ 2^-140 for subnormal results) is 3.3 times that.  Because S is taken about a point
// inside the scene, the bound stays tight when the scene is far from the coordinate origin.
__device__ __forceinline__ float flat_shadow_filter_g(double cc, float rr)
{
    const double k = ((double)rr - cc) + (cc + (double)rr) * 0x1p-16 + 0x1p-140;
    float f = (float)k;
    if ((double)f < k) f = __uint_as_float(__float_as_uint(f) + (f >= 0.f ? 1u : 0xFFFFFFFFu));      // next float up (finite, != -0)
    return f;
}

// Per ray: -Pm = -(|o'|^2 (1 - m)) rounded towards zero (so that -Pm >= -|o'|^2 (1 - m)); p = the f32 |o'|^2 the kernel formed.
__device__ __forceinline__ float flat_shadow_filter_npm(float p)
{
    float pm = p * (1.0f - 0x1p-16f);
    if (pm > 0.f) pm = __uint_as_float(__float_as_uint(pm) - 1u);           // one step down covers the rounding of the product
    return -pm;
}

// shadow_order[i] = index of the item at position i of the shadow array.  Pad items (i >= n) can never be hit: rr = -inf makes
// disc = (b*b - vv) + rr = -inf whatever the ray is, K = -inf the filter's bound.
__global__ void k_build_flat_groups(const Item<float> *__restrict__ items, const unsigned *__restrict__ shadow_order, unsigned n, unsigned n_fgroups,
                                    unsigned n_sgroups, V3<float> eye, V3<float> centre, V3<float> sdir, FGroup *__restrict__ pf, FExact *__restrict__ pe,
                                    FGroup *__restrict__ sg, FExactShadow *__restrict__ se)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_fgroups * kFlatFilterItems) {
        const unsigned g = i / kFlatFilterItems, k = i % kFlatFilterItems;
        float *p = pf[g].f;
        if (i < n) {
            const Item<float> it = items[i];
            const V3<float> v = { it.cx - eye.x, it.cy - eye.y, it.cz - eye.z };      // primitive.rs:56
            const float vv = dot(v, v), rr = it.r * it.r;                             // primitive.rs:58
            p[0 + k] = v.x; p[4 + k] = v.y; p[8 + k] = v.z; p[12 + k] = flat_filter_constant(vv, rr);
            float *e = pe[i].f;
            e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = vv; e[4] = rr; e[5] = e[6] = e[7] = 0.f;
        } else {
            p[0 + k] = 0.f; p[4 + k] = 0.f; p[8 + k] = 0.f; p[12 + k] = -inf<float>();
        }
    }
    if (i < n_sgroups * kFlatShadowItems) {
        const unsigned g = i / kFlatShadowItems, k = i % kFlatShadowItems;
        float *s = sg[g].f;
        if (i < n) {
            const Item<float> sh = items[shadow_order[i]];
            const float srr = sh.r * sh.r;
            const float cx = sh.cx - centre.x, cy = sh.cy - centre.y, cz = sh.cz - centre.z;          // c', rounded once per component
            const double cl = (double)cx * sdir.x + (double)cy * sdir.y + (double)cz * sdir.z;
            const double cc = (double)cx * cx + (double)cy * cy + (double)cz * cz;
            s[0 + k] = cx; s[3 + k] = cy; s[6 + k] = cz; s[9 + k] = (float)cl; s[12 + k] = flat_shadow_filter_g(cc, srr);
            float *x = se[i].f;
            x[0] = sh.cx; x[1] = sh.cy; x[2] = sh.cz; x[3] = srr;
        } else {
            s[0 + k] = 0.f; s[3 + k] = 0.f; s[6 + k] = 0.f; s[9 + k] = 0.f; s[12 + k] = -inf<float>();
        }
        if (k == 0) s[15] = 0.f;
    }
}

// Test infrastructure of the filter's proof (rt_debug_flat_filter_check): for the primary rays of one frame and every item, the
// exact discriminant and the filter's bound; counts the pairs with disc >= 0, with bound >= 0, and with disc >= 0 but bound < 0
// (which must not exist); counts[3..5]: the same for the shadow filter on rays from a point of each primary ray towards the light.
__global__ __launch_bounds__(kBlockThreads) void k_flat_filter_check(FlatScView sc, unsigned width, unsigned height, unsigned spp,
                                                                    unsigned long long *__restrict__ counts)
{
    const unsigned px = blockIdx.x * kBlockThreads + threadIdx.x;
    const unsigned x = px % width, y = px / width;
    if (y >= height) return;
    const float ssf = float(spp), fw = float(width), fh = float(height);
    const float half_w = fw / 2.0f, half_h = fh / 2.0f;
    const unsigned ssx = blockIdx.y / spp, ssy = blockIdx.y % spp;
    const float xres = float(x) + float(ssx) / ssf, yres = float(y) + float(ssy) / ssf;
    const V3<float> d = normalized(V3<float>{ xres - half_w, (fh - yres) - half_h, fw });
    unsigned long long exact = 0, bound = 0, bad = 0, sexact = 0, sbound = 0, sbad = 0;
    // shadow-type rays: from a point on the primary ray (0.75 .. 1 times the eye's distance from the scene centre along it) towards the light
    const V3<float> ec = sub(sc.eye, sc.centre);
    const V3<float> o = add(sc.eye, mulf(d, sqrtf(dot(ec, ec)) * (0.75f + 0.0625f * float(blockIdx.y % 5u))));
    const V3<float> l = mulf(sc.light, -1.0f);
    const V3<float> oc = sub(o, sc.centre);
    const V3<float> o2 = mulf(oc, 2.0f);
    const float ol = __builtin_fmaf(l.z, oc.z, __builtin_fmaf(l.y, oc.y, l.x * oc.x));
    const float npm = flat_shadow_filter_npm(__builtin_fmaf(oc.z, oc.z, __builtin_fmaf(oc.y, oc.y, oc.x * oc.x)));
    for (unsigned i = 0; i < sc.n_items; ++i) {
        const float *e = sc.pe[i].f;
        const float b = (e[0] * d.x + e[1] * d.y) + e[2] * d.z;
        const float disc = (b * b - e[3]) + e[4];
        const float k = sc.pf[i / kFlatFilterItems].f[12 + i % kFlatFilterItems];
        const float bf = __builtin_fmaf(e[2], d.z, __builtin_fmaf(e[1], d.y, e[0] * d.x));
        const float bnd = __builtin_fmaf(bf, bf, k);
        const bool ce = disc >= 0.f, cb = bnd >= 0.f;
        exact += ce; bound += cb; bad += ce && !cb;
        const float *x = sc.se[i].f;
        const V3<float> v = { x[0] - o.x, x[1] - o.y, x[2] - o.z };
        const float sb = dot(v, l);
        const float sdisc = (sb * sb - dot(v, v)) + x[3];
        const float *g = sc.sg[i / kFlatShadowItems].f;
        const unsigned gk = i % kFlatShadowItems;
        const float u = __builtin_fmaf(g[6 + gk], o2.z, __builtin_fmaf(g[3 + gk], o2.y, __builtin_fmaf(g[0 + gk], o2.x, npm)));
        const float bp = g[9 + gk] - ol;
        const float sbnd = __builtin_fmaf(bp, bp, u) + g[12 + gk];
        const bool se = sdisc >= 0.f, sbb = sbnd >= 0.f;
        sexact += se; sbound += sbb; sbad += se && !sbb;
    }
    exact = wave_sum((unsigned)exact); bound = wave_sum((unsigned)bound); bad = wave_sum((unsigned)bad);
    sexact = wave_sum((unsigned)sexact); sbound = wave_sum((unsigned)sbound); sbad = wave_sum((unsigned)sbad);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&counts[3], sexact); atomicAdd(&counts[4], sbound); atomicAdd(&counts[5], sbad); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&counts[0], exact); atomicAdd(&counts[1], bound); atomicAdd(&counts[2], bad); }
}

constexpr unsigned kFlatScRays = 2;                       // rays per lane (one VGPR pair per quantity)
constexpr unsigned kFlatScPrimaryThreads = kBlockThreads;     // 4 waves = two 16x16 pixel blocks, one 16x8 patch per wave

__global__ __launch_bounds__(kFlatScPrimaryThreads) void k_flat_primary_sc(FlatScView sc, unsigned width, unsigned height, unsigned spp,
                                                                          const TileDev *__restrict__ tiles, unsigned n_tiles, unsigned n_blocks,
                                                                          SampleBuf<float> sb, Quad<float> *__restrict__ queue1, FlatQueues *__restrict__ queues,
                                                                          Counters *__restrict__ counters)
{
    typedef float T;
    const unsigned lane = threadIdx.x & 63;
    const unsigned blk = blockIdx.x * 2 + (threadIdx.x >> 7), wave = (threadIdx.x >> 6) & 1;      // waves are independent: no LDS, no barrier
    if (blk >= n_blocks) return;
    unsigned lo = 0, hi = n_tiles - 1;
    while (lo < hi) {
        unsigned mid = (lo + hi + 1) >> 1;
        if (tiles[mid].blk_first <= blk) lo = mid; else hi = mid - 1;
    }
    const TileDev tile = tiles[lo];
    const unsigned lb = blk - tile.blk_first;
    // a wave = a 16x8 pixel patch: ray h of a lane is the pixel 8h to the right of the lane's own
    const unsigned x0 = tile.l + (lb % tile.blks_x) * kBlockW + (lane & 7);
    const unsigned y = tile.b + (lb / tile.blks_x) * kBlockH + wave * 8 + (lane >> 3);
    bool inside[kFlatScRays];
    for (unsigned h = 0; h < kFlatScRays; ++h) inside[h] = x0 + 8 * h < tile.r && y < tile.t;
    if (__ballot(inside[0]) == 0) return;

    const unsigned ssx = blockIdx.y / spp, ssy = blockIdx.y % spp;            // one sample per ray (grid.y = spp*spp)
    const T ssf = T(spp);
    const T fw = T(width), fh = T(height);
    const T half_w = fw / T(2.0), half_h = fh / T(2.0);
    const V3<T> eye = sc.eye, light = sc.light;
    const T yres = T(y) + T(ssy) / ssf;                                        // render.rs:238-243
    V3<T> dir[kFlatScRays];
    T dx[kFlatScRays], dy[kFlatScRays], dz[kFlatScRays];
    for (unsigned h = 0; h < kFlatScRays; ++h) {
        const T xres = T(x0 + 8 * h) + T(ssx) / ssf;
        dir[h] = normalized(V3<T>{ xres - half_w, (fh - yres) - half_h, fw });
        dx[h] = dir[h].x; dy[h] = dir[h].y; dz[h] = dir[h].z;
    }

    // ---------------- primary rays: nearest hit, strict `<`, first item in DFS order wins ties ----------------
    T best[kFlatScRays];
    unsigned best_i[kFlatScRays];
    flat_primary_scan(sc.pf, sc.n_fbytes, sc.pe, dx, dy, dz, best, best_i);

    // ---------------- shade (render.rs:190-199), store the samples, queue the shadow rays ----------------
    unsigned c_prim = 0, c_hits = 0, c_shadow = 0;
    for (unsigned h = 0; h < kFlatScRays; ++h) {
        const unsigned x = x0 + 8 * h;
        bool need_shadow = false;
        T gdot = T(0.0);
        V3<T> sp = { T(0.0), T(0.0), T(0.0) };
        uint8_t state = kMiss;
        if (inside[h] && !(best[h] == inf<T>())) {
            ++c_hits;
            const Item<T> it = sc.items[best_i[h]];
            const V3<T> c = { it.cx, it.cy, it.cz };
            const V3<T> nrm = normalized(add(eye, sub(mulf(dir[h], best[h]), c)));     // primitive.rs:83
            gdot = dot(nrm, light);
            if (gdot >= T(0.0)) {
                state = kAmbient;
            } else {
                need_shadow = true;
                ++c_shadow;
                state = kLit;                                                   // until a shadow pass finds an occluder
                const V3<T> ns = mulf(nrm, best[h] * rsqrt_exact(eps<T>()));
                sp = add(add(eye, mulf(dir[h], best[h])), ns);                  // render.rs:199
            }
        }
        const unsigned q = blockIdx.y * sb.n_px + (unsigned)out_index(tile, x, y, 0);  // sample slot (tile-major pixel)
        if (inside[h]) { sb.state[q] = state; sb.gdot[q] = gdot; ++c_prim; }
        wave_append(need_shadow, Quad<T>{ sp.x, sp.y, sp.z, owner_to_real<T>(q) }, queue1, &queues->n1);
    }

    if (counters) {
        Counters *const stripe = counters + (blk + blockIdx.y) % kCounterStripes;
        const unsigned long long prim = wave_sum(c_prim), hits = wave_sum(c_hits), sh = wave_sum(c_shadow);
        if (lane == 0) {
            atomicAdd(&stripe->primary, prim);
            atomicAdd(&stripe->hits, hits);
            atomicAdd(&stripe->shadow, sh);
        }
    }
}

// One shadow pass: the rays of `queue_in` against the shadow groups [begin_bytes / 64, end_bytes / 64), any hit; a wave takes
// 128 consecutive queue entries (ray h of a lane is entry 64h + lane of them).  Occluded rays mark their sample kShadowed; the
// others go to queue_out (or, in the last pass, stay kLit).
__global__ __launch_bounds__(kBlockThreads) void k_flat_shadow_sc(FlatScView sc, unsigned begin_bytes, unsigned end_bytes,
                                                                 const Quad<float> *__restrict__ queue_in, const unsigned *__restrict__ n_in,
                                                                 Quad<float> *__restrict__ queue_out, unsigned *__restrict__ n_out, SampleBuf<float> sb,
                                                                 Counters *__restrict__ counters)
{
    typedef float T;
    const unsigned n_rays = *n_in;
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned first = (blockIdx.x * (kBlockThreads / 64) + wave) * (64 * kFlatScRays) + lane;
    if (first - lane >= n_rays) return;                                        // wave-uniform: surplus waves leave at once
    T ox[kFlatScRays], oy[kFlatScRays], oz[kFlatScRays];
    unsigned owner[kFlatScRays], have[kFlatScRays], occluded[kFlatScRays];
    for (unsigned h = 0; h < kFlatScRays; ++h) {
        const unsigned idx = first + 64 * h;
        have[h] = idx < n_rays ? 1u : 0u;
        ox[h] = oy[h] = oz[h] = T(0.0);
        owner[h] = 0;
        occluded[h] = 0;
        if (have[h]) {
            const Quad<T> e = queue_in[idx];
            ox[h] = e.x; oy[h] = e.y; oz[h] = e.z;
            owner[h] = real_to_owner(e.w);
        }
    }
    const V3<T> sdir = mulf(sc.light, T(-1.0));                                // render.rs:206
    // the filter's per-ray terms (flat_shadow_filter_g's comment): origin relative to the scene centre, FMAs welcome
    T o2x[kFlatScRays], o2y[kFlatScRays], o2z[kFlatScRays], ol[kFlatScRays], npm[kFlatScRays];
    for (unsigned h = 0; h < kFlatScRays; ++h) {
        const V3<T> oc = { ox[h] - sc.centre.x, oy[h] - sc.centre.y, oz[h] - sc.centre.z };
        o2x[h] = oc.x * T(2.0); o2y[h] = oc.y * T(2.0); o2z[h] = oc.z * T(2.0);
        ol[h] = __builtin_fmaf(sdir.z, oc.z, __builtin_fmaf(sdir.y, oc.y, sdir.x * oc.x));
        npm[h] = flat_shadow_filter_npm(__builtin_fmaf(oc.z, oc.z, __builtin_fmaf(oc.y, oc.y, oc.x * oc.x)));
    }
    const unsigned end = min(end_bytes, sc.n_sbytes);
    if (begin_bytes < end)
        flat_shadow_scan(sc.sg, begin_bytes, end, sc.se, ox, oy, oz, o2x, o2y, o2z, ol, npm, sdir.x, sdir.y, sdir.z, have, occluded);
    unsigned c_occ = 0;
    for (unsigned h = 0; h < kFlatScRays; ++h) {
        const bool occ = have[h] && occluded[h];
        if (occ) { sb.state[owner[h]] = kShadowed; ++c_occ; }                  // render.rs:211-213
        if (queue_out) wave_append(have[h] && !occ, Quad<T>{ ox[h], oy[h], oz[h], owner_to_real<T>(owner[h]) }, queue_out, n_out);
    }
    if (counters) {
        Counters *const stripe = counters + blockIdx.x % kCounterStripes;
        const unsigned long long oc = wave_sum(c_occ);
        if (lane == 0) atomicAdd(&stripe->occluded, oc);
    }
}

}  // namespace rt
