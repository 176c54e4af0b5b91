// rt_flat.hpp -- RT_TRAVERSAL_FLAT, second generation: the north-star "linear scan through LDS" kernel tuned for
// the un-fused f32 VALU roofline (FMA is forbidden by parity, so a test costs 8 ops for a primary ray, 16 for a
// shadow ray -- nothing else should issue).
//
// What changed against k_render_fused (rt_kernels.hpp), which spent 19 VALU + 9 SALU instructions per test:
//  * primary rays share Scene::eye, so the ray-independent terms of primitive.rs:56-58 are pre-formed per item
//    (v = c - eye, vv = dot(v, v), rr = r*r; same individually rounded ops): 8 VALU per test instead of 17;
//  * items are consumed four at a time: the 4 x R discriminants are reduced with v_max3 and ONE branch rejects the
//    whole group (a ray's line meets only a handful of the 21,845 spheres); the exact sqrt path runs per item only
//    inside that rarely-taken branch, in item order, so strict-`<` / first-in-DFS-order tie-breaking is unchanged;
//  * every lane carries R = 2 pixels (rows y and y + 16 of a 16x32 block): each ds_read_b128 broadcast feeds two
//    rays, which keeps the LDS pipe (4 cycles per wave-read) below the VALU time of the group.
#pragma once
#include "rt_kernels.hpp"

namespace rt {

constexpr int kFlatBlockW = 16, kFlatBlockH = 32;      // pixels per 256-thread workgroup (2 per lane)
constexpr int kFlatR = 2;

template <typename T> struct alignas(sizeof(T) * 4) Quad { T x, y, z, w; };

template <typename T> struct FlatView {
    const Quad<T> *prim;    // {vx, vy, vz, vv} per item, DFS order, padded to a multiple of 4 items
    const T *prim_rr;       // rr per item, padded (pad items have vv = +big, rr = 0: disc < 0, never a hit)
    const Quad<T> *shad;    // {cx, cy, cz, rr}, padded (pad items carry rr = -1e30: disc < 0 for every ray)
    const Item<T> *items;   // centres for the normal of the winning item
    uint32_t n_items, n_padded;
    V3<T> light, eye;
};

// Pre-forms the per-item terms on the device (exact IEEE ops, the products the CPU path forms per ray).
template <typename T>
__global__ void k_build_flat(const Item<T> *__restrict__ items, unsigned n, unsigned n_padded, V3<T> eye, Quad<T> *__restrict__ prim,
                             T *__restrict__ prim_rr, Quad<T> *__restrict__ shad)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_padded) return;
    if (i < n) {
        const Item<T> it = items[i];
        const V3<T> v = { it.cx - eye.x, it.cy - eye.y, it.cz - eye.z };
        const T rr = it.r * it.r;
        prim[i] = { v.x, v.y, v.z, dot(v, v) };
        prim_rr[i] = rr;
        shad[i] = { it.cx, it.cy, it.cz, rr };
    } else {
        // padding: b = 0, disc = (0 - 1) + 0 < 0 for every ray -> can never hit, never NaN
        prim[i] = { T(0), T(0), T(0), T(1) };
        prim_rr[i] = T(0);
        // shadow pad: rr = -1e30.  |b*b - vv| is a rounding residue of vv <= 3e30 (validated scene), so disc < 0 always
        shad[i] = { T(0), T(0), T(0), T(-1e30) };
    }
}

template <typename T> __device__ __forceinline__ T max3(T a, T b, T c) { return fmax(fmax(a, b), c); }
template <> __device__ __forceinline__ float max3<float>(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

template <typename T, int CHUNK>
__global__ __launch_bounds__(kBlockThreads) void k_render_flat2(FlatView<T> sc, unsigned width, unsigned height, unsigned spp,
                                                               const TileDev *__restrict__ tiles, unsigned n_tiles,
                                                               uint8_t *__restrict__ out, Counters *__restrict__ counters)
{
    // one LDS array (16-B aligned): [0, CHUNK) quads, then CHUNK scalars of rr (primary pass only)
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
    // wave w owns the 8x8 patch (w&1, w>>1) of the upper 16x16 half and the same patch of the lower half
    const unsigned x = tile.l + bx * kFlatBlockW + (wave & 1) * 8 + (lane & 7);
    const unsigned y0 = tile.b + by * kFlatBlockH + (wave >> 1) * 8 + (lane >> 3);
    unsigned ys[kFlatR];
    bool inside[kFlatR];
#pragma unroll
    for (int r = 0; r < kFlatR; ++r) {
        ys[r] = y0 + 16u * r;
        inside[r] = x < tile.r && ys[r] < tile.t;
    }

    const T ssf = T(spp);
    const T total_recip = T(1.0) / (ssf * ssf);
    const T fw = T(width), fh = T(height);
    const T half_w = fw / T(2.0), half_h = fh / T(2.0);
    const V3<T> eye = sc.eye, light = sc.light;
    const unsigned n = sc.n_padded;

    const V3<T> OBJECT = { T(0xae) / T(255.0), T(0x31) / T(255.0), T(0x31) / T(255.0) };
    const V3<T> BACKGROUND = { T(0x22) / T(255.0), T(0x0a) / T(255.0), T(0x0a) / T(255.0) };
    const V3<T> AMBIENT = { BACKGROUND.x * T(0.8), BACKGROUND.y * T(0.8), BACKGROUND.z * T(0.8) };
    const V3<T> sdir = mulf(light, T(-1.0));

    V3<T> g[kFlatR];
    T alpha[kFlatR];
#pragma unroll
    for (int r = 0; r < kFlatR; ++r) { g[r] = { T(0.0), T(0.0), T(0.0) }; alpha[r] = T(0.0); }
    unsigned c_hits = 0, c_shadow = 0, c_occ = 0;

    for (unsigned ssx = 0; ssx < spp; ++ssx) {
        for (unsigned ssy = 0; ssy < spp; ++ssy) {
            V3<T> dir[kFlatR];
#pragma unroll
            for (int r = 0; r < kFlatR; ++r) {
                const T xres = T(x) + T(ssx) / ssf;
                const T yres = T(ys[r]) + T(ssy) / ssf;
                dir[r] = normalized(V3<T>{ xres - half_w, (fh - yres) - half_h, fw });
            }

            // ---------------- primary rays ----------------
            T best[kFlatR];
            unsigned best_i[kFlatR];
#pragma unroll
            for (int r = 0; r < kFlatR; ++r) { best[r] = inf<T>(); best_i[r] = 0; }

            for (unsigned base = 0; base < n; base += CHUNK) {
                const unsigned cnt = min((unsigned)CHUNK, n - base);          // multiple of 4
                __syncthreads();
                for (unsigned j = threadIdx.x; j < cnt; j += kBlockThreads) s_q[j] = sc.prim[base + j];
                for (unsigned j = threadIdx.x; j < cnt; j += kBlockThreads) s_rr[j] = sc.prim_rr[base + j];
                __syncthreads();
                for (unsigned j = 0; j < cnt; j += 4) {
                    const Quad<T> a0 = s_q[j], a1 = s_q[j + 1], a2 = s_q[j + 2], a3 = s_q[j + 3];
                    const Quad<T> rr = *reinterpret_cast<const Quad<T> *>(&s_rr[j]);
                    T b[kFlatR][4], disc[kFlatR][4];
#pragma unroll
                    for (int r = 0; r < kFlatR; ++r) {
                        b[r][0] = (a0.x * dir[r].x + a0.y * dir[r].y) + a0.z * dir[r].z;
                        b[r][1] = (a1.x * dir[r].x + a1.y * dir[r].y) + a1.z * dir[r].z;
                        b[r][2] = (a2.x * dir[r].x + a2.y * dir[r].y) + a2.z * dir[r].z;
                        b[r][3] = (a3.x * dir[r].x + a3.y * dir[r].y) + a3.z * dir[r].z;
                        disc[r][0] = (b[r][0] * b[r][0] - a0.w) + rr.x;
                        disc[r][1] = (b[r][1] * b[r][1] - a1.w) + rr.y;
                        disc[r][2] = (b[r][2] * b[r][2] - a2.w) + rr.z;
                        disc[r][3] = (b[r][3] * b[r][3] - a3.w) + rr.w;
                    }
                    T m = max3(disc[0][0], disc[0][1], disc[0][2]);
                    m = max3(m, disc[0][3], disc[1][0]);
                    m = max3(m, disc[1][1], disc[1][2]);
                    m = fmax(m, disc[1][3]);
                    if (!(m < T(0.0))) {                                      // rare: some lane's line meets one of the 4 items
#pragma unroll
                        for (int k = 0; k < 4; ++k) {                         // item order: first in DFS order wins ties
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

            // ---------------- shade ----------------
            bool need_shadow[kFlatR];
            T gdot[kFlatR];
            V3<T> sp[kFlatR];
#pragma unroll
            for (int r = 0; r < kFlatR; ++r) {
                need_shadow[r] = false; gdot[r] = T(0.0); sp[r] = { T(0.0), T(0.0), T(0.0) };
                if (inside[r]) {
                    if (best[r] == inf<T>()) {
                        g[r] = add(g[r], BACKGROUND);
                    } else {
                        ++c_hits;
                        const Item<T> it = sc.items[best_i[r]];
                        const V3<T> c = { it.cx, it.cy, it.cz };
                        const V3<T> nrm = normalized(add(eye, sub(mulf(dir[r], best[r]), c)));
                        gdot[r] = dot(nrm, light);
                        if (gdot[r] >= T(0.0)) {
                            g[r] = add(g[r], AMBIENT);
                        } else {
                            need_shadow[r] = true;
                            ++c_shadow;
                            const V3<T> ns = mulf(nrm, best[r] * rsqrt_exact(eps<T>()));
                            sp[r] = add(add(eye, mulf(dir[r], best[r])), ns);
                        }
                    }
                }
            }

            // ---------------- shadow rays: any hit, workgroup-level early out ----------------
            bool pending[kFlatR], occluded[kFlatR];
#pragma unroll
            for (int r = 0; r < kFlatR; ++r) { pending[r] = need_shadow[r]; occluded[r] = false; }
            for (unsigned base = 0; base < n; base += CHUNK) {
                if (!__syncthreads_or((pending[0] || pending[1]) ? 1 : 0)) break;
                const unsigned cnt = min((unsigned)CHUNK, n - base);
                for (unsigned j = threadIdx.x; j < cnt; j += kBlockThreads) s_q[j] = sc.shad[base + j];
                __syncthreads();
                if (pending[0] || pending[1]) {
                    for (unsigned j = 0; j < cnt; j += 4) {
                        const Quad<T> a0 = s_q[j], a1 = s_q[j + 1], a2 = s_q[j + 2], a3 = s_q[j + 3];
                        T b[kFlatR][4], disc[kFlatR][4];
#pragma unroll
                        for (int r = 0; r < kFlatR; ++r) {
                            const V3<T> v0 = { a0.x - sp[r].x, a0.y - sp[r].y, a0.z - sp[r].z };
                            const V3<T> v1 = { a1.x - sp[r].x, a1.y - sp[r].y, a1.z - sp[r].z };
                            const V3<T> v2 = { a2.x - sp[r].x, a2.y - sp[r].y, a2.z - sp[r].z };
                            const V3<T> v3 = { a3.x - sp[r].x, a3.y - sp[r].y, a3.z - sp[r].z };
                            b[r][0] = dot(v0, sdir); disc[r][0] = (b[r][0] * b[r][0] - dot(v0, v0)) + a0.w;
                            b[r][1] = dot(v1, sdir); disc[r][1] = (b[r][1] * b[r][1] - dot(v1, v1)) + a1.w;
                            b[r][2] = dot(v2, sdir); disc[r][2] = (b[r][2] * b[r][2] - dot(v2, v2)) + a2.w;
                            b[r][3] = dot(v3, sdir); disc[r][3] = (b[r][3] * b[r][3] - dot(v3, v3)) + a3.w;
                        }
                        // a finished (or absent) ray must not keep re-entering the slow path
                        const T m0 = pending[0] ? fmax(max3(disc[0][0], disc[0][1], disc[0][2]), disc[0][3]) : T(-1.0);
                        const T m1 = pending[1] ? fmax(max3(disc[1][0], disc[1][1], disc[1][2]), disc[1][3]) : T(-1.0);
                        if (!(fmax(m0, m1) < T(0.0))) {
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
                            if (!(pending[0] || pending[1])) break;
                        }
                    }
                }
            }

#pragma unroll
            for (int r = 0; r < kFlatR; ++r) {
                if (need_shadow[r]) {
                    if (!occluded[r]) {
                        g[r] = add(add(g[r], mulf(OBJECT, -gdot[r])), AMBIENT);
                        alpha[r] += T(1.0);
                    } else {
                        ++c_occ;
                        g[r] = add(add(g[r], BACKGROUND), mulf(AMBIENT, -gdot[r]));
                    }
                }
            }
        }
    }

#pragma unroll
    for (int r = 0; r < kFlatR; ++r) {
        if (inside[r]) {
            const V3<T> c = mulf(g[r], total_recip);
            const T a = alpha[r] * total_recip;
            const unsigned tw = tile.r - tile.l;
            const size_t px = (size_t)tile.out_px + (size_t)(ys[r] - tile.b) * tw + (x - tile.l);
            reinterpret_cast<unsigned *>(out)[px] = scale_u8(c.x) | (scale_u8(c.y) << 8) | (scale_u8(c.z) << 16) | (scale_u8(a) << 24);
        }
    }

    if (counters) {
        counters += blockIdx.x % kCounterStripes;
        const unsigned long long prim = wave_sum(((inside[0] ? 1u : 0u) + (inside[1] ? 1u : 0u)) * spp * spp);
        const unsigned long long hits = wave_sum(c_hits), sh = wave_sum(c_shadow), oc = wave_sum(c_occ);
        if (lane == 0) {
            atomicAdd(&counters->primary, prim);
            atomicAdd(&counters->hits, hits);
            atomicAdd(&counters->shadow, sh);
            atomicAdd(&counters->occluded, oc);
        }
    }
}

}  // namespace rt
