// rt_flat_wf.hpp -- RT_TRAVERSAL_FLAT as a wavefront pipeline.
//
// A single kernel that does primary scan, shade and shadow scan per pixel keeps a whole wave scanning all 21,845 items for
// as long as ONE of its lanes still has a shadow ray pending, although only 64 % of the pixels cast a shadow ray and 55 % of
// those rays are occluded -- most of them by one of the largest spheres (that first version took 15.2 ms at 1080p).  Because
// an any-hit query may visit the items in any order, the shadow array is sorted by radius (89 % of the occluded rays are
// settled by its first 1,024 items) and the work is cut into passes with the rays re-packed densely in between, so every
// lane of every wave carries a ray that still needs work:
//
//   k_flat_primary   one thread per pixel pair x one sample: primary-ray generation, nearest-hit scan through LDS, shade;
//                    stores the sample's {state, n.light}; rays that need a shadow test are appended to queue 1 (one
//                    atomic per wave: __ballot + popcount + mbcnt)
//   k_flat_shadow    pass A: queue 1 against the first LDS chunk (the 1,024 largest spheres); occluded rays record
//                    kShadowed, survivors are appended to queue 2.   pass B: queue 2 against the remaining chunks.
//   k_resolve_samples (rt_skip.hpp) accumulates each pixel's samples in the reference's order and quantises.
//
// Results are bit-identical to the reference: every ray performs the same individually rounded arithmetic against every
// item it needs (nearest hit: all items in DFS order; any-hit: a boolean OR over the items, order-free).
#pragma once
#include "rt_flat.hpp"
#include "rt_skip.hpp"

namespace rt {

struct FlatQueues {
    unsigned n1, n2;      // rays in queue 1 / queue 2 (device counters, zeroed per pass)
};

template <typename T> __device__ __forceinline__ T owner_to_real(unsigned o);
template <> __device__ __forceinline__ float owner_to_real<float>(unsigned o) { return __uint_as_float(o); }
template <> __device__ __forceinline__ double owner_to_real<double>(unsigned o) { return (double)o; }
__device__ __forceinline__ unsigned real_to_owner(float w) { return __float_as_uint(w); }
__device__ __forceinline__ unsigned real_to_owner(double w) { return (unsigned)w; }

// Appends `item` for every lane with `want` to a global queue: one atomicAdd per wave.
template <typename T>
__device__ __forceinline__ void wave_append(bool want, const Quad<T> &item, Quad<T> *__restrict__ queue, unsigned *__restrict__ counter)
{
    const unsigned long long mask = __ballot(want);
    if (mask == 0) return;
    const unsigned cnt = __popcll(mask);
    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    unsigned base = 0;
    if (rank == 0 && want) base = atomicAdd(counter, cnt);                      // the first wanting lane reserves the slots
    base = (unsigned)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(mask));
    if (want) queue[base + rank] = item;
}

template <typename T, int CHUNK>
__global__ __launch_bounds__(kBlockThreads) void k_flat_primary(FlatView<T> sc, unsigned width, unsigned height, unsigned spp,
                                                               const TileDev *__restrict__ tiles, unsigned n_tiles, SampleBuf<T> sb,
                                                               Quad<T> *__restrict__ queue1, FlatQueues *__restrict__ queues,
                                                               Counters *__restrict__ counters)
{
    __shared__ Quad<T> s_q[CHUNK + CHUNK / 4];
    T *s_rr = reinterpret_cast<T *>(&s_q[CHUNK]);

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

    for (unsigned base = 0; base < n; base += CHUNK) {
        const unsigned cnt = min((unsigned)CHUNK, n - base);                  // multiple of 8
        __syncthreads();
        for (unsigned j = threadIdx.x; j < cnt; j += kBlockThreads) s_q[j] = sc.prim[base + j];
        for (unsigned j = threadIdx.x; j < cnt; j += kBlockThreads) s_rr[j] = sc.prim_rr[base + j];
        __syncthreads();
        __builtin_assume(cnt % 8 == 0);
#pragma unroll 2
        for (unsigned j = 0; j < cnt; j += 4) {
            const Quad<T> qa = s_q[j], qb = s_q[j + 1], qc = s_q[j + 2], qd = s_q[j + 3];
            const Quad<T> rr = *reinterpret_cast<const Quad<T> *>(&s_rr[j]);
            const P2<T> x01(qa.x, qa.y), y01(qa.z, qa.w), z01(qb.x, qb.y), w01(qb.z, qb.w), r01(rr.x, rr.y);
            const P2<T> x23(qc.x, qc.y), y23(qc.z, qc.w), z23(qd.x, qd.y), w23(qd.z, qd.w), r23(rr.z, rr.w);
            T b[kFlatR][4], disc[kFlatR][4];
#pragma unroll
            for (int r = 0; r < kFlatR; ++r) {
                const P2<T> dx(dir[r].x), dy(dir[r].y), dz(dir[r].z);
                const P2<T> b01 = (x01 * dx + y01 * dy) + z01 * dz;            // primitive.rs:57, two items at once
                const P2<T> b23 = (x23 * dx + y23 * dy) + z23 * dz;
                const P2<T> d01 = (b01 * b01 - w01) + r01;                     // primitive.rs:58
                const P2<T> d23 = (b23 * b23 - w23) + r23;
                b[r][0] = b01.lo(); b[r][1] = b01.hi(); b[r][2] = b23.lo(); b[r][3] = b23.hi();
                disc[r][0] = d01.lo(); disc[r][1] = d01.hi(); disc[r][2] = d23.lo(); disc[r][3] = d23.hi();
            }
            T m = fmax(max3(disc[0][0], disc[0][1], disc[0][2]), disc[0][3]);
#pragma unroll
            for (int r = 1; r < kFlatR; ++r) m = max3(max3(m, disc[r][0], disc[r][1]), disc[r][2], disc[r][3]);
            if (!(m < T(0.0))) {                                                // rare: some lane's line meets one of the 4 items
#pragma unroll
                for (int k = 0; k < 4; ++k) {                                   // item order: first in DFS order wins ties
#pragma unroll
                    for (int r = 0; r < kFlatR; ++r) {
                        if (!(disc[r][k] < T(0.0))) {
                            const T s = sqrt_rn_lean(disc[r][k]);
                            const T t2 = b[r][k] + s;
                            if (!(t2 < T(0.0))) {
                                const T t1 = b[r][k] - s;
                                const T d = t1 > T(0.0) ? t1 : t2;
                                if (!(d >= best[r])) { best[r] = d; best_i[r] = base + j + k; }
                            }
                        }
                    }
                }
            }
        }
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
        unsigned n_inside = 0;
#pragma unroll
        for (int r = 0; r < kFlatR; ++r) n_inside += inside[r] ? 1u : 0u;
        const unsigned long long prim = wave_sum(n_inside), hits = wave_sum(c_hits), sh = wave_sum(c_shadow);
        if (lane == 0) {
            atomicAdd(&counters->primary, prim);
            atomicAdd(&counters->hits, hits);
            atomicAdd(&counters->shadow, sh);
        }
    }
}

// One shadow pass: rays of `queue_in` against the shadow items [item_begin, item_end) (multiples of CHUNK), any hit.
// Occluded rays mark their sample kShadowed; the others go to queue_out (or, in the last pass, stay kLit).
template <typename T, int CHUNK>
__global__ __launch_bounds__(kBlockThreads) void k_flat_shadow(FlatView<T> sc, unsigned item_begin, unsigned item_end,
                                                              const Quad<T> *__restrict__ queue_in, const unsigned *__restrict__ n_in,
                                                              Quad<T> *__restrict__ queue_out, unsigned *__restrict__ n_out, SampleBuf<T> sb,
                                                              Counters *__restrict__ counters)
{
    __shared__ Quad<T> s_q[CHUNK];
    const unsigned n_rays = *n_in;
    const unsigned first = blockIdx.x * (kBlockThreads * kFlatR);
    if (first >= n_rays) return;                                               // uniform for the workgroup: before any barrier
    const unsigned lane = threadIdx.x & 63;

    V3<T> sp[kFlatR];
    unsigned owner[kFlatR];
    bool have[kFlatR], pending[kFlatR], occluded[kFlatR];
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
    }
    const V3<T> sdir = mulf(sc.light, T(-1.0));                                // render.rs:206
    const unsigned n = min(item_end, sc.n_padded);

    for (unsigned base = item_begin; base < n; base += CHUNK) {
        bool any_pending = false;
#pragma unroll
        for (int r = 0; r < kFlatR; ++r) any_pending = any_pending || pending[r];
        if (!__syncthreads_or(any_pending ? 1 : 0)) break;
        const unsigned cnt = min((unsigned)CHUNK, n - base);
        for (unsigned j = threadIdx.x; j < cnt; j += kBlockThreads) s_q[j] = sc.shad[base + j];
        __syncthreads();
        if (any_pending) {
            __builtin_assume(cnt % 8 == 0);
#pragma unroll 2
            for (unsigned j = 0; j < cnt; j += 4) {
                const Quad<T> qa = s_q[j], qb = s_q[j + 1], qc = s_q[j + 2], qd = s_q[j + 3];
                const P2<T> x01(qa.x, qa.y), y01(qa.z, qa.w), z01(qb.x, qb.y), r01(qb.z, qb.w);
                const P2<T> x23(qc.x, qc.y), y23(qc.z, qc.w), z23(qd.x, qd.y), r23(qd.z, qd.w);
                const P2<T> lx(sdir.x), ly(sdir.y), lz(sdir.z);
                T b[kFlatR][4], disc[kFlatR][4];
#pragma unroll
                for (int r = 0; r < kFlatR; ++r) {
                    const P2<T> ox(sp[r].x), oy(sp[r].y), oz(sp[r].z);
                    const P2<T> vx01 = x01 - ox, vy01 = y01 - oy, vz01 = z01 - oz;       // primitive.rs:56
                    const P2<T> vx23 = x23 - ox, vy23 = y23 - oy, vz23 = z23 - oz;
                    const P2<T> b01 = (vx01 * lx + vy01 * ly) + vz01 * lz;
                    const P2<T> b23 = (vx23 * lx + vy23 * ly) + vz23 * lz;
                    const P2<T> vv01 = (vx01 * vx01 + vy01 * vy01) + vz01 * vz01;
                    const P2<T> vv23 = (vx23 * vx23 + vy23 * vy23) + vz23 * vz23;
                    const P2<T> d01 = (b01 * b01 - vv01) + r01;
                    const P2<T> d23 = (b23 * b23 - vv23) + r23;
                    b[r][0] = b01.lo(); b[r][1] = b01.hi(); b[r][2] = b23.lo(); b[r][3] = b23.hi();
                    disc[r][0] = d01.lo(); disc[r][1] = d01.hi(); disc[r][2] = d23.lo(); disc[r][3] = d23.hi();
                }
                T m = T(-1.0);                                                  // a settled (or absent) ray must not re-enter the slow path
#pragma unroll
                for (int r = 0; r < kFlatR; ++r)
                    if (pending[r]) m = max3(max3(m, disc[r][0], disc[r][1]), disc[r][2], disc[r][3]);
                if (!(m < T(0.0))) {
#pragma unroll
                    for (int r = 0; r < kFlatR; ++r) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (pending[r] && !(disc[r][k] < T(0.0))) {
                                const T t2 = b[r][k] + sqrt_rn_lean(disc[r][k]);
                                if (!(t2 < T(0.0))) { occluded[r] = true; pending[r] = false; }
                            }
                        }
                    }
                    any_pending = false;
#pragma unroll
                    for (int r = 0; r < kFlatR; ++r) any_pending = any_pending || pending[r];
                    if (!any_pending) break;
                }
            }
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

}  // namespace rt
