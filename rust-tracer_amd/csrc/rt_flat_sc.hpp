// rt_flat_sc.hpp -- RT_TRAVERSAL_FLAT for f32, the kernels the product runs: every item of the scene against every ray, with
// the items fed through the SCALAR path.
//
// A linear scan is wave-uniform by construction: all 64 rays of a wave test the same item at the same moment.  On CDNA that
// makes the item a scalar: records arrive by s_load (scalar cache, 466 KB for the 21,845 spheres in L2 behind it) and feed the
// vector ALU as SGPR operands of plain 4-byte VOP2 instructions -- no LDS staging, no ds_read, no barrier, 8 waves per SIMD.
// tools/valu_issue_probe.hip measured why this is the better shape here than the LDS + packed-math kernels of rt_flat_wf.hpp
// (round 1; still what f64 runs, and selectable for f32 with csrc/rt_debug.h RT_DEBUG_FLAT_KERNELS = 0): a packed v_pk_mul/add_f32
// issues in 2.55 cycles per wave at 8 waves per SIMD against 1.46 for a VOP2 -- 1.15x the lane-ops per cycle, not 2x -- while the
// pair-interleaved operands cost registers (5.5 waves per SIMD) and five ds_read_b128 per 32 packed ops whose latency the
// kernel waits for 65 % of the time.  The scan loops themselves are generated assembly (tools/gen_flat_asm.py ->
// rt_flat_rot.hpp): as C++ the same scan compiled to 19 scalar / branch instructions per four items beside the 31 vector
// ones and was slower than the LDS kernels.
//
//   k_flat_primary_sc  one thread per pixel x one sample: primary ray, nearest-hit scan of all items in DFS order (strict `<`:
//                      the first item in DFS order wins ties, primitive.rs:79), shade; the sample's {state, n.light} is stored
//                      and rays that need a shadow test are appended to queue 1 (one atomic per wave)
//   k_flat_shadow_sc   one thread per queued shadow ray, any hit over a range of the RADIUS-SORTED item array (pass A: the 1,026
//                      largest spheres, which settle 89 % of the occluded rays; survivors re-packed into queue 2; pass B: the
//                      rest); a wave leaves as soon as all its rays are settled
//
// Items are consumed three at a time (one 64-byte group = one s_load_dwordx16): the three discriminants are reduced with ONE
// v_max3 and one branch rejects the group -- a ray's line meets a handful of the 21,845 spheres -- and the exact sqrt path
// runs per item, in item order, only inside that rarely taken branch.  Per-item terms that do not depend on the ray are
// pre-formed once per scene with the same individually rounded operations (v = c - eye, vv, rr): 8 VALU operations per
// primary test, 16 per shadow test, every one of them the reference's (primitive.rs:55-72).
#pragma once
#include "rt_flat_wf.hpp"
#include "rt_flat_rot.hpp"

namespace rt {

// Three items: primary {vx[3], vy[3], vz[3], vv[3], rr[3], pad}, shadow {cx[3], cy[3], cz[3], rr[3], pad[4]}.
struct alignas(64) FGroup { float f[16]; };
constexpr unsigned kFlatGroupItems = 3;
constexpr unsigned kFlatPadGroups = 2;          // behind the last pair: the scan loads one pair ahead

struct FlatScView {
    const FGroup *pg;       // DFS order
    const FGroup *sg;       // radius descending
    const Item<float> *items;   // centres for the normal of the winning item
    uint32_t n_items, n_bytes;  // n_bytes: 128 * number of group pairs
    V3<float> light, eye;
};

// shadow_order[i] = index of the item at position i of the shadow array.  Pad items (i >= n) can never be hit: rr = -inf
// makes disc = (b*b - vv) + rr = -inf whatever the ray is.
__global__ void k_build_flat_groups(const Item<float> *__restrict__ items, const unsigned *__restrict__ shadow_order, unsigned n, unsigned n_groups,
                                    V3<float> eye, FGroup *__restrict__ pg, FGroup *__restrict__ sg)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_groups * kFlatGroupItems) return;
    const unsigned g = i / kFlatGroupItems, k = i % kFlatGroupItems;
    float *p = pg[g].f, *s = sg[g].f;
    if (i < n) {
        const Item<float> it = items[i];
        const V3<float> v = { it.cx - eye.x, it.cy - eye.y, it.cz - eye.z };      // primitive.rs:56
        p[0 + k] = v.x; p[3 + k] = v.y; p[6 + k] = v.z; p[9 + k] = dot(v, v);
        p[12 + k] = it.r * it.r;                                                // primitive.rs:58
        const Item<float> sh = items[shadow_order[i]];
        s[0 + k] = sh.cx; s[3 + k] = sh.cy; s[6 + k] = sh.cz; s[9 + k] = sh.r * sh.r;
    } else {
        p[0 + k] = 0.f; p[3 + k] = 0.f; p[6 + k] = 0.f; p[9 + k] = 0.f; p[12 + k] = -inf<float>();
        s[0 + k] = 0.f; s[3 + k] = 0.f; s[6 + k] = 0.f; s[9 + k] = -inf<float>();
    }
    if (k == 0) { p[15] = 0.f; s[12] = s[13] = s[14] = s[15] = 0.f; }
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
    flat_primary_scan(sc.pg, sc.n_bytes, dx, dy, dz, best, best_i);

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
    const unsigned end = min(end_bytes, sc.n_bytes);
    if (begin_bytes < end)
        flat_shadow_scan(sc.sg, begin_bytes, end, ox, oy, oz, sdir.x, sdir.y, sdir.z, have, occluded);
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
