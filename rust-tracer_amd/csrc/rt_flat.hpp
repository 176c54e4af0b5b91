// rt_flat.hpp -- RT_TRAVERSAL_FLAT, second generation: the north-star "linear scan through LDS" kernel tuned for
// the un-fused f32 VALU roofline (FMA is forbidden by parity, so a test costs 8 ops for a primary ray, 16 for a
// shadow ray -- nothing else should issue).
//
// One thread per pixel pair does everything Renderer::render_region does for its pixels (render.rs:218-255):
// supersample loop (ssx outer, ssy inner), primary-ray generation, nearest-hit scan over the item array staged
// through LDS in chunks, shade, shadow any-hit scan with a workgroup-level early-out, sequential f32 accumulation,
// f32 -> u8 quantisation.  What it does to stay off everything but the VALU (a first version that tested one item at a
// time with the full 17-op expression spent 19 VALU + 9 SALU instructions per test and ran 2.1x slower):
//  * primary rays share Scene::eye, so the ray-independent terms of primitive.rs:56-58 are pre-formed per item
//    (v = c - eye, vv = dot(v, v), rr = r*r; same individually rounded ops): 8 VALU per test instead of 17;
//  * items are consumed four at a time: the 4 x R discriminants are reduced with v_max3 and ONE branch rejects the
//    whole group (a ray's line meets only a handful of the 21,845 spheres); the exact sqrt path runs per item only
//    inside that rarely-taken branch, in item order, so strict-`<` / first-in-DFS-order tie-breaking is unchanged;
//  * the arithmetic is written on 2-vectors over ITEM PAIRS (v_pk_mul_f32 / v_pk_add_f32): each half is rounded on
//    its own, so results are bit-identical to scalar code, at half the VALU issue slots (a plain f32 VALU op costs
//    4 cycles per wave64 on CDNA4 -- measured: instruction count x 4 / 1024 SIMDs was exactly the kernel's duration --
//    and the 157 TF vector peak is only reachable with packed ops).  Items are therefore stored pair-interleaved:
//    quad 2p = {x0,x1,y0,y1}, quad 2p+1 = {z0,z1,w0,w1} for items 2p, 2p+1;
//  * every lane carries R = 2 pixels (rows y and y + 16 of a 16x32 block): each ds_read_b128 broadcast feeds two
//    rays, which keeps the LDS pipe (4 cycles per wave-read) below the VALU time of the group.
#pragma once
#include "rt_kernels.hpp"

namespace rt {

#ifndef RT_FLAT_R
#define RT_FLAT_R 2
#endif
constexpr int kFlatR = RT_FLAT_R;                              // pixels (rays) per lane: rows y, y+16, ...
constexpr int kFlatBlockW = 16, kFlatBlockH = 16 * kFlatR;     // pixels per 256-thread workgroup

template <typename T> struct alignas(sizeof(T) * 4) Quad { T x, y, z, w; };

// Two values processed by one instruction where the hardware can (f32: v_pk_*; f64 has no packed form).
template <typename T> struct P2;
template <> struct P2<float> {
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 v;
    __device__ __forceinline__ P2() {}
    __device__ __forceinline__ P2(float lo, float hi) { v.x = lo; v.y = hi; }
    __device__ __forceinline__ explicit P2(float both) { v.x = both; v.y = both; }
    __device__ __forceinline__ explicit P2(v2 a) : v(a) {}
    __device__ __forceinline__ P2 operator+(P2 o) const { return P2(v + o.v); }
    __device__ __forceinline__ P2 operator-(P2 o) const { return P2(v - o.v); }
    __device__ __forceinline__ P2 operator*(P2 o) const { return P2(v * o.v); }
    __device__ __forceinline__ float lo() const { return v.x; }
    __device__ __forceinline__ float hi() const { return v.y; }
};
template <> struct P2<double> {
    double a, b;
    __device__ __forceinline__ P2() {}
    __device__ __forceinline__ P2(double lo, double hi) : a(lo), b(hi) {}
    __device__ __forceinline__ explicit P2(double both) : a(both), b(both) {}
    __device__ __forceinline__ P2 operator+(P2 o) const { return P2(a + o.a, b + o.b); }
    __device__ __forceinline__ P2 operator-(P2 o) const { return P2(a - o.a, b - o.b); }
    __device__ __forceinline__ P2 operator*(P2 o) const { return P2(a * o.a, b * o.b); }
    __device__ __forceinline__ double lo() const { return a; }
    __device__ __forceinline__ double hi() const { return b; }
};

template <typename T> struct FlatView {
    const Quad<T> *prim;    // pair-interleaved {vx0,vx1,vy0,vy1},{vz0,vz1,vv0,vv1}; DFS order, padded to a multiple of 8 items
    const T *prim_rr;       // rr per item, padded (pad items have vv = +big, rr = 0: disc < 0, never a hit)
    const Quad<T> *shad;    // pair-interleaved {cx0,cx1,cy0,cy1},{cz0,cz1,rr0,rr1} (pad items carry rr = -1e30: disc < 0)
    const Item<T> *items;   // centres for the normal of the winning item
    uint32_t n_items, n_padded;
    V3<T> light, eye;
};

// Pre-forms the per-item terms on the device (exact IEEE ops, the products the CPU path forms per ray).
// shadow_order[i] = index of the item that sits at position i of the SHADOW array: shadow rays are any-hit, so their scan
// order is free, and the host puts the largest spheres first (they occlude most rays, so fully shadowed waves leave early).
template <typename T>
__global__ void k_build_flat(const Item<T> *__restrict__ items, const unsigned *__restrict__ shadow_order, unsigned n, unsigned n_padded,
                             V3<T> eye, Quad<T> *__restrict__ prim, T *__restrict__ prim_rr, Quad<T> *__restrict__ shad)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_padded) return;
    T vx, vy, vz, vv, rr, cx, cy, cz, srr;
    if (i < n) {
        const Item<T> it = items[i];
        const V3<T> v = { it.cx - eye.x, it.cy - eye.y, it.cz - eye.z };
        vx = v.x; vy = v.y; vz = v.z; vv = dot(v, v);
        rr = it.r * it.r;
        const Item<T> sh = items[shadow_order[i]];
        cx = sh.cx; cy = sh.cy; cz = sh.cz; srr = sh.r * sh.r;
    } else {
        // primary pad: b = 0, disc = (0 - 1) + 0 < 0 for every ray -> can never hit, never NaN
        vx = T(0); vy = T(0); vz = T(0); vv = T(1); rr = T(0);
        // shadow pad: rr = -1e30.  |b*b - vv| is a rounding residue of vv <= 3e30 (validated scene), so disc < 0 always
        cx = T(0); cy = T(0); cz = T(0); srr = T(-1e30);
    }
    // scalar view of the pair-interleaved quads: item i is half (i & 1) of quads 2*(i/2) and 2*(i/2)+1
    T *pq = reinterpret_cast<T *>(prim + 2 * (i >> 1));
    T *sq = reinterpret_cast<T *>(shad + 2 * (i >> 1));
    const unsigned h = i & 1u;
    pq[0 + h] = vx; pq[2 + h] = vy; pq[4 + h] = vz; pq[6 + h] = vv;
    sq[0 + h] = cx; sq[2 + h] = cy; sq[4 + h] = cz; sq[6 + h] = srr;
    prim_rr[i] = rr;
}

template <typename T> __device__ __forceinline__ T max3(T a, T b, T c) { return fmax(fmax(a, b), c); }
template <> __device__ __forceinline__ float max3<float>(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

template <typename T, int CHUNK>
__global__ __launch_bounds__(kBlockThreads) void k_render_flat2(FlatView<T> sc, unsigned width, unsigned height, unsigned spp,
                                                               const TileDev *__restrict__ tiles, unsigned n_tiles,
                                                               uint8_t *__restrict__ out, Counters *__restrict__ counters,
                                                               unsigned frame_w)
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
                const unsigned cnt = min((unsigned)CHUNK, n - base);          // multiple of 8
                __syncthreads();
                for (unsigned j = threadIdx.x; j < cnt; j += kBlockThreads) s_q[j] = sc.prim[base + j];
                for (unsigned j = threadIdx.x; j < cnt; j += kBlockThreads) s_rr[j] = sc.prim_rr[base + j];
                __syncthreads();
                __builtin_assume(cnt % 8 == 0);
#pragma unroll 2
                for (unsigned j = 0; j < cnt; j += 4) {
                    // items j, j+1 in quads qa/qb; items j+2, j+3 in qc/qd; rr of all four in one quad
                    const Quad<T> qa = s_q[j], qb = s_q[j + 1], qc = s_q[j + 2], qd = s_q[j + 3];
                    const Quad<T> rr = *reinterpret_cast<const Quad<T> *>(&s_rr[j]);
                    const P2<T> x01(qa.x, qa.y), y01(qa.z, qa.w), z01(qb.x, qb.y), w01(qb.z, qb.w), r01(rr.x, rr.y);
                    const P2<T> x23(qc.x, qc.y), y23(qc.z, qc.w), z23(qd.x, qd.y), w23(qd.z, qd.w), r23(rr.z, rr.w);
                    T b[kFlatR][4], disc[kFlatR][4];
#pragma unroll
                    for (int r = 0; r < kFlatR; ++r) {
                        const P2<T> dx(dir[r].x), dy(dir[r].y), dz(dir[r].z);
                        const P2<T> b01 = (x01 * dx + y01 * dy) + z01 * dz;          // primitive.rs:57, two items at once
                        const P2<T> b23 = (x23 * dx + y23 * dy) + z23 * dz;
                        const P2<T> d01 = (b01 * b01 - w01) + r01;                   // primitive.rs:58
                        const P2<T> d23 = (b23 * b23 - w23) + r23;
                        b[r][0] = b01.lo(); b[r][1] = b01.hi(); b[r][2] = b23.lo(); b[r][3] = b23.hi();
                        disc[r][0] = d01.lo(); disc[r][1] = d01.hi(); disc[r][2] = d23.lo(); disc[r][3] = d23.hi();
                    }
                    T m = fmax(max3(disc[0][0], disc[0][1], disc[0][2]), disc[0][3]);
#pragma unroll
                    for (int r = 1; r < kFlatR; ++r) m = max3(max3(m, disc[r][0], disc[r][1]), disc[r][2], disc[r][3]);
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
                            const P2<T> vx01 = x01 - ox, vy01 = y01 - oy, vz01 = z01 - oz;           // primitive.rs:56
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
                        // a finished (or absent) ray must not keep re-entering the slow path
                        T m = T(-1.0);
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
            const size_t px = out_index(tile, x, ys[r], frame_w);
            reinterpret_cast<unsigned *>(out)[px] = scale_u8(c.x) | (scale_u8(c.y) << 8) | (scale_u8(c.z) << 16) | (scale_u8(a) << 24);
        }
    }

    if (counters) {
        counters += blockIdx.x % kCounterStripes;
        unsigned n_inside = 0;
#pragma unroll
        for (int r = 0; r < kFlatR; ++r) n_inside += inside[r] ? 1u : 0u;
        const unsigned long long prim = wave_sum(n_inside * spp * spp);
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
