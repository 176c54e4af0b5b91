// rt_flat.hpp -- RT_TRAVERSAL_FLAT, shared pieces: the pre-formed per-item arrays and the packed-pair arithmetic type.
// The kernels that scan them are in rt_flat_wf.hpp.
//
// The flat scan is tuned for the un-fused f32 VALU roofline (FMA is forbidden by parity, so a test costs 8 ops for a
// primary ray, 16 for a shadow ray -- nothing else should issue):
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
//  * every lane carries R = 2 rays (two pixel rows in the primary pass, two queue entries in the shadow passes): each
//    ds_read_b128 broadcast feeds two rays, which keeps the LDS pipe (4 cycles per wave-read) below the VALU time.
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

}  // namespace rt
