// rt_flat_sc.hpp -- RT_TRAVERSAL_FLAT, an alternative pair of scan kernels with the items fed through the SCALAR path
// (selected with csrc/rt_debug.h RT_DEBUG_FLAT_KERNELS = 1; the default remains the LDS-staged packed-math pipeline of
// rt_flat_wf.hpp).
//
// A linear scan is wave-uniform by construction: all 64 rays of a wave test the same item at the same moment.  On CDNA that
// makes the item a scalar: records arrive by s_load (scalar cache, 437 KB for the 21,845 spheres in L2 behind it) and feed the
// vector ALU as SGPR operands of plain 4-byte VOP2 instructions -- no LDS staging, no ds_read, no barrier, 8 waves per SIMD.
// tools/valu_issue_probe.hip says the arithmetic itself would be on par with the packed kernels (a v_pk_mul/add_f32 issues in
// 2.55 cycles per wave at 8 waves per SIMD against 1.46 for a VOP2: 1.15x the lane-ops per cycle, not 2x).  As compiled from
// this C++ it is NOT faster (1080p: 7.7 + 3.0 ms against 7.2 + 2.6 ms): hipcc spends 19 scalar / branch instructions per group
// of four items on 64-bit address arithmetic and waits (profiles/, r02 notes) beside the 31 vector ones, and scalar
// instructions compete for the same issue slots.  Kept because it is the shape a hand-scheduled scan would take.
//
//   k_flat_primary_sc  one thread per pixel x one sample: primary ray, nearest-hit scan of all items in DFS order (strict `<`:
//                      the first item in DFS order wins ties, primitive.rs:79), shade; the sample's {state, n.light} is stored
//                      and rays that need a shadow test are appended to queue 1 (one atomic per wave)
//   k_flat_shadow_sc   one thread per queued shadow ray, any hit over a range of the RADIUS-SORTED item array (pass A: the 1,024
//                      largest spheres, which settle 89 % of the occluded rays; survivors re-packed into queue 2; pass B: the rest)
//
// Items are consumed four at a time (one 80- or 64-byte group = one or two scalar loads): the four discriminants are reduced
// with v_max3 and ONE branch rejects the group -- a ray's line meets a handful of the 21,845 spheres -- and the exact sqrt
// path runs per item, in item order, only inside that rarely taken branch.  Per-item terms that do not depend on the ray are
// pre-formed once per scene with the same individually rounded operations (v = c - eye, vv, rr): 8 VALU operations per
// primary test, 16 per shadow test, every one of them the reference's (primitive.rs:55-72).
#pragma once
#include "rt_flat_wf.hpp"

namespace rt {

template <typename T> struct alignas(sizeof(T) * 4) PGroup { T vx[4], vy[4], vz[4], vv[4], rr[4]; };     // primary: pre-formed terms of 4 items
template <typename T> struct alignas(sizeof(T) * 16) SGroup { T cx[4], cy[4], cz[4], rr[4]; };           // shadow: centres + rr of 4 items

template <typename T> struct FlatScView {
    const PGroup<T> *pg;    // DFS order
    const SGroup<T> *sg;    // radius descending
    const Item<T> *items;   // centres for the normal of the winning item
    uint32_t n_items, n_groups;
    V3<T> light, eye;
};

// shadow_order[i] = index of the item at position i of the shadow array.  Pad items (i >= n) can never be hit: rr = -inf
// makes disc = (b*b - vv) + rr = -inf whatever the ray is.
template <typename T>
__global__ void k_build_flat_groups(const Item<T> *__restrict__ items, const unsigned *__restrict__ shadow_order, unsigned n, unsigned n_groups,
                                    V3<T> eye, PGroup<T> *__restrict__ pg, SGroup<T> *__restrict__ sg)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_groups * 4u) return;
    const unsigned g = i >> 2, k = i & 3u;
    if (i < n) {
        const Item<T> it = items[i];
        const V3<T> v = { it.cx - eye.x, it.cy - eye.y, it.cz - eye.z };      // primitive.rs:56
        pg[g].vx[k] = v.x; pg[g].vy[k] = v.y; pg[g].vz[k] = v.z; pg[g].vv[k] = dot(v, v);
        pg[g].rr[k] = it.r * it.r;                                            // primitive.rs:58
        const Item<T> sh = items[shadow_order[i]];
        sg[g].cx[k] = sh.cx; sg[g].cy[k] = sh.cy; sg[g].cz[k] = sh.cz; sg[g].rr[k] = sh.r * sh.r;
    } else {
        pg[g].vx[k] = T(0); pg[g].vy[k] = T(0); pg[g].vz[k] = T(0); pg[g].vv[k] = T(0); pg[g].rr[k] = -inf<T>();
        sg[g].cx[k] = T(0); sg[g].cy[k] = T(0); sg[g].cz[k] = T(0); sg[g].rr[k] = -inf<T>();
    }
}

template <typename T>
__global__ __launch_bounds__(kBlockThreads) void k_flat_primary_sc(FlatScView<T> sc, unsigned width, unsigned height, unsigned spp,
                                                                  const TileDev *__restrict__ tiles, unsigned n_tiles, SampleBuf<T> sb,
                                                                  Quad<T> *__restrict__ queue1, FlatQueues *__restrict__ queues,
                                                                  Counters *__restrict__ counters)
{
    unsigned lo = 0, hi = n_tiles - 1;
    while (lo < hi) {
        unsigned mid = (lo + hi + 1) >> 1;
        if (tiles[mid].blk_first <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const TileDev tile = tiles[lo];
    const unsigned lb = blockIdx.x - tile.blk_first;
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned x = tile.l + (lb % tile.blks_x) * kBlockW + (wave & 1) * 8 + (lane & 7);       // a wave = an 8x8 pixel patch
    const unsigned y = tile.b + (lb / tile.blks_x) * kBlockH + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = x < tile.r && y < tile.t;
    if (__ballot(inside) == 0) return;                                         // waves are independent: no LDS, no barrier

    const unsigned ssx = blockIdx.y / spp, ssy = blockIdx.y % spp;            // one sample per thread (grid.y = spp*spp)
    const T ssf = T(spp);
    const T fw = T(width), fh = T(height);
    const T half_w = fw / T(2.0), half_h = fh / T(2.0);
    const V3<T> eye = sc.eye, light = sc.light;
    const T xres = T(x) + T(ssx) / ssf;                                        // render.rs:238-243
    const T yres = T(y) + T(ssy) / ssf;
    const V3<T> dir = normalized(V3<T>{ xres - half_w, (fh - yres) - half_h, fw });

    // ---------------- primary ray: nearest hit, strict `<`, first item in DFS order wins ties ----------------
    T best = inf<T>();
    unsigned best_i = 0;
    const PGroup<T> *__restrict__ pg = sc.pg;
    const unsigned ng = sc.n_groups;
#pragma unroll 2
    for (unsigned g = 0; g < ng; ++g) {
        const PGroup<T> G = pg[g];                                             // wave-uniform: scalar loads, SGPR operands
        T b[4], disc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            b[k] = (G.vx[k] * dir.x + G.vy[k] * dir.y) + G.vz[k] * dir.z;      // primitive.rs:57
            disc[k] = (b[k] * b[k] - G.vv[k]) + G.rr[k];                       // primitive.rs:58
        }
        const T m = fmax(max3(disc[0], disc[1], disc[2]), disc[3]);
        if (__ballot(!(m < T(0.0))) != 0) {                                     // rare: some lane's line meets one of the 4 items
#pragma unroll
            for (int k = 0; k < 4; ++k) {                                       // in item order
                if (!(disc[k] < T(0.0))) {
                    const T s = sqrt_rn_lean(disc[k]);
                    const T t2 = b[k] + s;
                    if (!(t2 < T(0.0))) {
                        const T t1 = b[k] - s;
                        const T d = t1 > T(0.0) ? t1 : t2;
                        if (!(d >= best)) { best = d; best_i = g * 4u + k; }
                    }
                }
            }
        }
    }

    // ---------------- shade (render.rs:190-199), store the sample, queue the shadow ray ----------------
    bool need_shadow = false;
    T gdot = T(0.0);
    V3<T> sp = { T(0.0), T(0.0), T(0.0) };
    uint8_t state = kMiss;
    unsigned c_hits = 0, c_shadow = 0;
    if (inside && !(best == inf<T>())) {
        ++c_hits;
        const Item<T> it = sc.items[best_i];
        const V3<T> c = { it.cx, it.cy, it.cz };
        const V3<T> nrm = normalized(add(eye, sub(mulf(dir, best), c)));       // primitive.rs:83
        gdot = dot(nrm, light);
        if (gdot >= T(0.0)) {
            state = kAmbient;
        } else {
            need_shadow = true;
            ++c_shadow;
            state = kLit;                                                       // until a shadow pass finds an occluder
            const V3<T> ns = mulf(nrm, best * rsqrt_exact(eps<T>()));
            sp = add(add(eye, mulf(dir, best)), ns);                            // render.rs:199
        }
    }
    const unsigned q = blockIdx.y * sb.n_px + (unsigned)out_index(tile, x, y, 0);      // sample slot (tile-major pixel)
    if (inside) { sb.state[q] = state; sb.gdot[q] = gdot; }
    wave_append(need_shadow, Quad<T>{ sp.x, sp.y, sp.z, owner_to_real<T>(q) }, queue1, &queues->n1);

    if (counters) {
        Counters *const stripe = counters + (blockIdx.x + blockIdx.y) % kCounterStripes;
        const unsigned long long prim = wave_sum(inside ? 1u : 0u), hits = wave_sum(c_hits), sh = wave_sum(c_shadow);
        if (lane == 0) {
            atomicAdd(&stripe->primary, prim);
            atomicAdd(&stripe->hits, hits);
            atomicAdd(&stripe->shadow, sh);
        }
    }
}

// One shadow pass: the rays of `queue_in` against the shadow groups [group_begin, group_end), any hit.  Occluded rays mark their
// sample kShadowed; the others go to queue_out (or, in the last pass, stay kLit).  A wave leaves as soon as all its rays are settled.
template <typename T>
__global__ __launch_bounds__(kBlockThreads) void k_flat_shadow_sc(FlatScView<T> sc, unsigned group_begin, unsigned group_end,
                                                                 const Quad<T> *__restrict__ queue_in, const unsigned *__restrict__ n_in,
                                                                 Quad<T> *__restrict__ queue_out, unsigned *__restrict__ n_out, SampleBuf<T> sb,
                                                                 Counters *__restrict__ counters)
{
    const unsigned n_rays = *n_in;
    const unsigned idx = blockIdx.x * kBlockThreads + threadIdx.x;
    const unsigned lane = threadIdx.x & 63;
    const bool have = idx < n_rays;
    if (__ballot(have) == 0) return;
    V3<T> sp = { T(0.0), T(0.0), T(0.0) };
    unsigned owner = 0;
    if (have) {
        const Quad<T> e = queue_in[idx];
        sp = { e.x, e.y, e.z };
        owner = real_to_owner(e.w);
    }
    bool pending = have, occluded = false;
    const V3<T> sdir = mulf(sc.light, T(-1.0));                                // render.rs:206
    const SGroup<T> *__restrict__ sg = sc.sg;
    const unsigned ge = min(group_end, sc.n_groups);
#pragma unroll 2
    for (unsigned g = group_begin; g < ge; ++g) {
        const SGroup<T> G = sg[g];                                             // wave-uniform: one 64-byte scalar load (f32)
        T b[4], disc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const V3<T> v = { G.cx[k] - sp.x, G.cy[k] - sp.y, G.cz[k] - sp.z };        // primitive.rs:56
            b[k] = dot(v, sdir);
            disc[k] = (b[k] * b[k] - dot(v, v)) + G.rr[k];
        }
        T m = fmax(max3(disc[0], disc[1], disc[2]), disc[3]);
        if (!pending) m = T(-1.0);                                              // a settled (or absent) ray must not re-enter the slow path
        if (__ballot(!(m < T(0.0))) != 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (pending && !(disc[k] < T(0.0))) {
                    const T t2 = b[k] + sqrt_rn_lean(disc[k]);
                    if (!(t2 < T(0.0))) { occluded = true; pending = false; }
                }
            }
            if (__ballot(pending) == 0) break;
        }
    }
    unsigned c_occ = 0;
    if (have && occluded) { sb.state[owner] = kShadowed; ++c_occ; }            // render.rs:211-213
    if (queue_out) wave_append(have && !occluded, Quad<T>{ sp.x, sp.y, sp.z, owner_to_real<T>(owner) }, queue_out, n_out);
    if (counters) {
        Counters *const stripe = counters + blockIdx.x % kCounterStripes;
        const unsigned long long oc = wave_sum(c_occ);
        if (lane == 0) atomicAdd(&stripe->occluded, oc);
    }
}

}  // namespace rt
