// rt_skip.hpp -- RT_TRAVERSAL_SKIP: the reference's bounding-sphere hierarchy walked as a skip-pointer stream.
//
// TypedGroup::intersect (group.rs:72-83) is
//      if bound.distance_from_ray(ray) >= hit.distance { return }  for child in children { intersect(child) }
// Flattened in DFS pre-order this is a linear walk over a node stream: a BOUND node either falls through to the
// next node or jumps to `skip` (the first node after its subtree); an ITEM node runs Sphere::intersect
// (primitive.rs:77-84).  No post-order work exists, so the walk is exactly the recursion.
//
// Wave64 execution: the stream index i is wave-uniform (node records come in through the scalar cache as SGPR
// operands); each lane carries its own ray, its own hit.distance and one `resume` index -- the lane sleeps while
// i < resume.  A lane whose own bound test culls a subtree sets resume = skip; the wave jumps only when no lane
// wants to enter (ballot), otherwise culled lanes are dragged through the subtree with their updates masked.
// Because subtrees nest, one resume register per lane is enough, and each lane performs exactly the tests the
// reference performs for its ray -- including the reference's behaviour when a ray starts inside a bound.
#pragma once
#include "rt_kernels.hpp"
#include "rt_skip_rot.hpp"
#include "rt_coop.hpp"

namespace rt {

// One stream node.  PRIMARY stream (all primary rays share Scene::eye):  a = {vx, vy, vz, vv, rr} with
// v = centre - eye, vv = dot(v, v), rr = radius * radius -- the ray-independent sub-expressions of
// primitive.rs:56-58, computed on the device with the same individually rounded operations.
// SHADOW stream (per-lane origin): a = {cx, cy, cz, rr, -}.
// item word: ITEM nodes carry their index into the DFS items | kNodeItem; the END node behind the last node carries kNodeEnd
// (disc = +inf for every ray: the assembly loops leave when they hit it); BOUND nodes carry no flag.
// skip_off: byte offset of the node the walk continues at when nobody enters -- the first node after a BOUND's subtree, the
// node behind an ITEM.
// A scene is FUSED when every BOUND is directly followed by an ITEM with the same centre, bit for bit (the reference's
// pyramid: a group's first child is the sphere its bound is built around, group.rs:37-41).  Such a scene also gets a COMPACTED
// pair of streams for the fused assembly loops: those ITEM nodes are dropped and each BOUND node carries its sphere's rr and
// item index, so the BOUND step tests it for the lanes that enter (rt_skip_rot.hpp).
constexpr uint32_t kNodeItem = 0x80000000u, kNodeEnd = 0x40000000u, kNodeIndexMask = 0x3FFFFFFFu;
constexpr unsigned kNodePad = 3;            // END + two more END copies: the loops prefetch up to two nodes past END
template <typename T> struct alignas(sizeof(T) * 8) Node {
    T a0, a1, a2, a3, a4;
    uint32_t item;          // see above.  Compacted BOUND: the index of the group's own sphere.
    uint32_t skip_off;
    T own_rr;               // rr of the group's own sphere (compacted streams, BOUND nodes)
    __device__ __forceinline__ bool is_bound() const { return (item & (kNodeItem | kNodeEnd)) == 0u; }
    __device__ __forceinline__ unsigned skip() const { return skip_off / (unsigned)sizeof(Node); }   // as a node index
    __device__ __forceinline__ unsigned index() const { return item & kNodeIndexMask; }
};
static_assert(sizeof(Node<float>) == 32 && sizeof(Node<double>) == 64, "node records are one aligned scalar-load unit");

// Host-built raw stream entry (before the device derives the two streams above).  skip: node index, 0 for an ITEM.
template <typename T> struct RawNode {
    T cx, cy, cz, r;
    uint32_t skip, item;
    T own_r;                // compacted stream, BOUND: radius / index of the sphere the bound is built around
    uint32_t own_item, pad;
};

// FILTERED streams (f32; the *_filt loops of rt_skip_rot.hpp).  Same nodes, same order, same skip offsets as the
// Node streams they are derived from, plus the terms of a conservative bound that is asked before the reference's test:
//   primary  {vx, vy, vz, vv, rr, T, skip_off, tag}     the test can only return a finite distance if fma(vz,dz, fma(vy,dy, vx*dx)) >= T
//   shadow   {w1, w2, cl, R2o, R2i, R2o_own, R2i_own, skip_off}   (FNodeS) TWO-sided: with P2 = (w1 - q1)^2 + (w2 - q2)^2 the squared distance of
//            the centre from the ray in a plane perpendicular to the light ((w1, w2) / (q1, q2): centre / origin in that plane, relative to
//            m0) and a = cl - ol the centre's coordinate along the ray: the reference's test says MISS if P2 > R2o and HIT if
//            P2 + k1 (P2 + a^2) <= R2i and (a >= a0 or P2 + a^2 <= R2i); only a lane in between runs the reference's arithmetic (terms from
//            the Node stream).  R2*_own: the
//            same for the group's own sphere (compacted stream).  Sign bit of R2o: ITEM (or END); sign bit of R2o_own: END.
// primary tag: 0 for a BOUND (compacted stream: rr of the group's own sphere, a non-negative float), item | kNodeItem for an ITEM,
// kNodeItem | kNodeEnd for END.  The own sphere's item index of a compacted BOUND lives in the stream's own_item table.
struct alignas(32) FNode {
    float a0, a1, a2, a3, a4, f5;
    uint32_t skip_off, tag;
};
static_assert(sizeof(FNode) == 32, "one aligned s_load_dwordx8");
struct alignas(32) FNodeS {
    float w1, w2, cl, r2o, r2i, r2o_own, r2i_own;
    uint32_t skip_off;
};
static_assert(sizeof(FNodeS) == 32, "one aligned s_load_dwordx8");

// Constants of the shadow filter (derivation: DESIGN.md 4.1; computed by rt_capi.hip, filter_constants).
struct FilterConsts {
    float m0[3];            // reference point inside the scene (centroid of the item centres)
    float e1[3], e2[3];     // f32 roundings of an orthonormal basis of the plane perpendicular to the light
    float l[3];             // the shadow rays' direction (-light, the f32 values the reference uses)
    float a0;               // a >= a0 proves b >= 0
    float k1;               // inner bound: P2 + k1 * (P2 + a^2) <= R2i (the reference's rounding of disc grows with |centre - origin|^2)
    float kc;               // 1 / (1 + 4 tau): (P2 + a^2) kc >= R2o proves the origin clearly outside the sphere (then b < 0 means miss)
    float ro2;              // a ray whose origin is further than sqrt(ro2) from m0 is not covered: it fails every sure test (NaN)
    double S, eta;          // Rc + Ro: bound of |c - m0| + |o - m0|;  | |l|^2 - 1 |
};

// Outer / inner bound of P2 for a sphere with squared radius rr (as the reference rounds it): DESIGN.md 4.1.
__device__ __forceinline__ float next_f32_above(float f)      // finite f
{
    const uint32_t u = __float_as_uint(f);
    if ((u << 1) == 0u) return __uint_as_float(0x00000001u);
    return __uint_as_float((u >> 31) ? u - 1u : u + 1u);
}
__device__ __forceinline__ float next_f32_below(float f);
__device__ __forceinline__ void shadow_filter_bounds(const FilterConsts &fc, float rr_f, float &r2o, float &r2i)
{
    const double eps = 0x1p-24, rr = rr_f, S2 = fc.S * fc.S, a9 = 9.7 * eps * fc.S;
    if (!(rr < 1e300)) { r2o = r2i = __builtin_huge_valf(); return; }       // END: rr = +inf
    const double so = __builtin_sqrt(rr * (1.0 + 1.01 * eps) + (fc.eta + 12.0 * eps) * S2) + a9;
    const double o = so * so * (1.0 + 2.01 * eps) + 1e-36;
    float of = (float)o;
    if ((double)of < o) of = next_f32_above(of);
    r2o = next_f32_above(of);
    // inner bound, for the test  P2 + k1 (P2 + a^2) <= R2i  (FilterConsts::k1 carries the part that grows with the ray's own distance)
    const double tau = 0x1p-10, es2 = eps * fc.S * eps * fc.S;
    const double c2 = 102.7 * (1.0 + 1.0 / tau) * es2 * (1.0 + fc.eta);
    const double X = (1.0 + fc.eta + 10.2 * eps) * c2 * (1.0 + 6.0 * eps) + 94.1 * es2 * (1.0 + 1.0 / tau);
    const double i = (rr * (1.0 - eps) - X) / ((1.0 + tau) * (1.0 + 2.1 * eps) * (1.0 + fc.eta) * (1.0 + 40.0 * eps)) * (1.0 - 8.0 * eps) - 1e-36;
    r2i = -1.0f;
    if (i > 0.0) {
        float inf_ = (float)i;
        if ((double)inf_ > i) inf_ = next_f32_below(inf_);
        inf_ = next_f32_below(inf_);
        r2i = inf_ > 0.0f ? inf_ : -1.0f;
    }
}

template <typename T> struct SkipView {
    const Node<T> *prim;    // primary-ray stream
    const Node<T> *shad;    // shadow-ray stream
    const Node<T> *fprim;   // compacted streams of a fused scene (else NULL)
    const Node<T> *fshad;
    const Item<T> *items;   // DFS items (centre of the winning item for the normal)
    uint32_t n_nodes, n_fnodes;
    V3<T> light, eye;
    // filtered copies of the streams and the compacted stream's own_item table (f32: all four; f64: the two primary ones; else NULL)
    const FNode *xprim, *xfprim;
    const FNodeS *xshad, *xfshad;
    const uint32_t *xown;
    const FilterConsts *fc;  // device copy (read where a shadow walk starts: kernel arguments that stay live cost SGPRs, and occupancy)
};

// The primary filter's threshold for a node with the stored terms vv = dot(v, v), rr (both as the reference rounds them): any ray
// direction d (a normalised f32 vector, | |d|^2 - 1 | <= 16 eps) for which Sphere::distance_from_ray returns a finite distance has
// fma(vz, dz, fma(vy, dy, vx*dx)) >= T.  -inf (every ray passes) when the eye is not clearly outside the sphere or the squares are
// near the subnormal range.  DESIGN.md 4.1 carries the derivation.
__device__ __forceinline__ float next_f32_below(float f)      // finite f
{
    const uint32_t u = __float_as_uint(f);
    if ((u << 1) == 0u) return __uint_as_float(0x80000001u);
    return __uint_as_float((u >> 31) ? u + 1u : u - 1u);
}
__device__ __forceinline__ float primary_filter_threshold(float vv_f, float rr_f)
{
    const double eps = 0x1p-24, vv = vv_f, rr = rr_f;
    // (vv >= 1e-14: |v| >= 1e-7, so that b >= sqrt(60 eps vv) keeps the squares of the BOUND step's root-free decision -- differences of
    // values of b's magnitude -- in the normal range: bound_shortcut_verdict)
    if (!(vv >= 1e-14) || !(vv - rr >= 64.0 * eps * (vv + rr))) return -__builtin_huge_valf();
    const double m0 = vv - rr * (1.0 + 2.0 * eps) - 1e-44;
    const double t = __builtin_sqrt(m0 * (1.0 - 2.0 * eps)) - 8.0 * eps * __builtin_sqrt(vv);
    float tf = (float)t;
    if ((double)tf > t) tf = next_f32_below(tf);
    return next_f32_below(tf);
}

// The same threshold for an f64 scene's PRIMARY walk (rt_skip_rot.hpp, the double overloads of skip_primary_rot_filt*): the f64 test
// returns a finite distance only if b >= sqrt(vv - rr) (1 - 1e-15) (disc >= 0 and, the eye clearly outside, b > 0); the filter's
// b' = fma(vz, dz, fma(vy, dy, vx*dx)) is formed in f32 from f32 roundings of v and of the ray direction (relative error 2^-24 each) with
// three more roundings: | b' - b | <= 5.3 * 2^-24 * |v| (|d| <= 1 + 2^-23).  T = sqrt(vv - rr) - 6 * 2^-24 * sqrt(vv), rounded down twice.
__device__ __forceinline__ float primary_filter_threshold64(double vv, double rr)
{
    const double eps = 0x1p-24;
    if (!(vv >= 1e-30) || !(vv < 1e300) || !(vv - rr >= 64.0 * eps * (vv + rr))) return -__builtin_huge_valf();
    const double t = __builtin_sqrt(vv - rr) * (1.0 - 1e-14) - 6.0 * eps * __builtin_sqrt(vv);
    float tf = (float)t;
    if ((double)tf > t) tf = next_f32_below(tf);
    return next_f32_below(tf);
}

// The other side of that bound (round 5, FNode::a3 of an f64 scene's BOUND nodes): b' >= T_in proves that the f64 test returns a FINITE distance
// d = t1 with 0 < t1 <= b.  T_in = sqrt(vv - rr) (1 + 1e-14) + 8 * 2^-24 sqrt(vv), rounded up twice (+inf where T is -inf): with | b' - b | <=
// 5.3 * 2^-24 |v| (above) the reference's b is >= sqrt(vv - rr) + 2.7 * 2^-24 |v|, so b^2 - (vv - rr) >= 2 sqrt(vv - rr) * 2.7 * 2^-24 |v| >= 2^-31 vv
// (sqrt(vv - rr) >= sqrt(64 * 2^-24) |v| = 2^-9 |v| by the condition below) -- eight orders of magnitude above the f64 discriminant's own rounding
// error (a few 2^-53 (b^2 + vv + rr)): disc > 0, s = sqrt(disc) < b because vv > rr, t2 > 0 and t1 = RN(b - s) > 0 (b - s >= (vv - rr) / 2b, far
// from the subnormals for vv >= 1e-30).  With it the walk decides "the lane enters" as b' (1 + 2^-11) < hit.distance: d = t1 <= b <= b' + 5.3 * 2^-24
// |v| <= b' (1 + 5.3 * 2^-15) for b' >= T_in >= 2^-9 |v| (the product's own f32 rounding included in 2^-11).
__device__ __forceinline__ float primary_sure_threshold64(double vv, double rr)
{
    const double eps = 0x1p-24;
    if (!(vv >= 1e-30) || !(vv < 1e300) || !(vv - rr >= 64.0 * eps * (vv + rr))) return __builtin_huge_valf();
    const double t = __builtin_sqrt(vv - rr) * (1.0 + 1e-14) + 8.0 * eps * __builtin_sqrt(vv);
    float tf = (float)t;
    if ((double)tf < t) tf = next_f32_above(tf);
    return next_f32_above(tf);
}

// FNode copy of an f64 scene's primary stream (plain or compacted; END nodes included) for the filtered f64 primary walk: f32 roundings of
// v, the threshold above, skip_off in FNode units (half the Node<double> offset), the tag as in the f32 streams (a compacted BOUND's is 0:
// its own sphere's rr is read from the exact record).
__global__ void k_build_fstream64(const Node<double> *__restrict__ prim, const Node<double> *__restrict__ shad, unsigned n_total, bool compacted, FilterConsts fc,
                                  FNode *__restrict__ xprim, FNodeS *__restrict__ xshad, uint32_t *__restrict__ own_item)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    const Node<double> p = prim[i];
    {
        // shadow walk: the f32 walk's TWO-SIDED bound (rt_skip_rot.hpp F64FS) -- the centre in the plane perpendicular to the light and along
        // it (formed in double, rounded once), R2o from the f32 walk's formula with rr rounded UP, R2i with rr rounded DOWN: their margins
        // cover an f32 reference's roundings, a superset of what the f64 reference needs
        const Node<double> sn = shad[i];
        const bool s_end = (sn.item & kNodeEnd) != 0u, s_item = (sn.item & kNodeItem) != 0u;
        const double cx = sn.a0 - (double)fc.m0[0], cy = sn.a1 - (double)fc.m0[1], cz = sn.a2 - (double)fc.m0[2];
        FNodeS fs;
        fs.w1 = s_end ? 0.0f : (float)((cx * (double)fc.e1[0] + cy * (double)fc.e1[1]) + cz * (double)fc.e1[2]);
        fs.w2 = s_end ? 0.0f : (float)((cx * (double)fc.e2[0] + cy * (double)fc.e2[1]) + cz * (double)fc.e2[2]);
        fs.cl = s_end ? 3e38f : (float)((cx * (double)fc.l[0] + cy * (double)fc.l[1]) + cz * (double)fc.l[2]);
        auto up = [](double v) { float f = (float)v; if ((double)f < v) f = next_f32_above(f); return f; };
        auto down = [](double v) { float f = (float)v; if ((double)f > v) f = next_f32_below(f); return f; };
        float unused;
        shadow_filter_bounds(fc, s_end ? __builtin_huge_valf() : up(sn.a3), fs.r2o, unused);
        shadow_filter_bounds(fc, s_end ? __builtin_huge_valf() : down(sn.a3), unused, fs.r2i);
        const bool known = __builtin_fabsf(fs.w1) < 3e38f && __builtin_fabsf(fs.w2) < 3e38f && __builtin_fabsf(fs.cl) < 3e38f;
        if (!s_end && !known) { fs.r2o = __builtin_huge_valf(); fs.r2i = -1.0f; }      // nothing known: never "beyond", never "inside"
        if (s_end || s_item) fs.r2o = -fs.r2o;                                   // sign bit: ITEM or END
        fs.r2o_own = 0.0f; fs.r2i_own = -1.0f;
        if (compacted && !s_end && !s_item) {
            shadow_filter_bounds(fc, up(sn.own_rr), fs.r2o_own, unused);
            shadow_filter_bounds(fc, down(sn.own_rr), unused, fs.r2i_own);
            if (!known) { fs.r2o_own = __builtin_huge_valf(); fs.r2i_own = -1.0f; }
        }
        if (s_end) fs.r2o_own = -0.0f;                                           // sign bit: END
        fs.skip_off = sn.skip_off / 2u;
        xshad[i] = fs;
    }
    const bool end = (p.item & kNodeEnd) != 0u, item = (p.item & kNodeItem) != 0u;
    FNode fp;
    fp.a0 = (float)p.a0; fp.a1 = (float)p.a1; fp.a2 = (float)p.a2;
    fp.f5 = end ? -__builtin_huge_valf() : primary_filter_threshold64(p.a3, p.a4);
    // BOUND nodes: a3 = T_in (above; +inf: never sure), a4 = T_own -- the filter threshold of the group's own sphere in a compacted stream
    // (b' < T_own: it returns INF, as an ITEM node's f5 says of an item), +inf where the BOUND carries no sphere.  (rt_skip_rot.hpp, F64F.sure_enter_path)
    const bool bound = !end && !item;
    fp.a3 = bound && fp.f5 > -__builtin_huge_valf() ? primary_sure_threshold64(p.a3, p.a4) : __builtin_huge_valf();
    fp.a4 = bound && compacted ? primary_filter_threshold64(p.a3, p.own_rr) : __builtin_huge_valf();
    // an f32 rounding that overflowed (|v| > 3.4e38 cannot happen: coordinates are <= 1e15) or lost everything (|v| < 1e-45) says nothing
    if (!end && !(__builtin_fabsf(fp.a0) < 3e38f && __builtin_fabsf(fp.a1) < 3e38f && __builtin_fabsf(fp.a2) < 3e38f)) {
        fp.f5 = -__builtin_huge_valf(); fp.a3 = __builtin_huge_valf(); fp.a4 = -__builtin_huge_valf();
    }
    fp.skip_off = p.skip_off / 2u;
    fp.tag = end ? (kNodeItem | kNodeEnd) : item ? (kNodeItem | (p.item & kNodeIndexMask)) : 0u;
    xprim[i] = fp;
    if (own_item) own_item[i] = (compacted && !end && !item) ? p.item : 0u;
}

// Filtered copies of a pair of Node<float> streams (plain or compacted; END nodes included).
__global__ void k_build_fstreams(const Node<float> *__restrict__ prim, const Node<float> *__restrict__ shad, unsigned n_total, bool compacted,
                                 FilterConsts fc, FNode *__restrict__ xprim, FNodeS *__restrict__ xshad, uint32_t *__restrict__ own_item)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    const Node<float> p = prim[i], s = shad[i];
    const bool end = (p.item & kNodeEnd) != 0u, item = (p.item & kNodeItem) != 0u;
    const uint32_t tag = end ? (kNodeItem | kNodeEnd) : item ? (kNodeItem | (p.item & kNodeIndexMask)) : compacted ? __float_as_uint(p.own_rr) : 0u;
    FNode fp;
    FNodeS fs;
    fp.a0 = p.a0; fp.a1 = p.a1; fp.a2 = p.a2; fp.a3 = p.a3; fp.a4 = p.a4;
    fp.f5 = end ? -__builtin_huge_valf() : primary_filter_threshold(p.a3, p.a4);
    fp.skip_off = p.skip_off; fp.tag = tag;
    const double cx = (double)s.a0 - (double)fc.m0[0], cy = (double)s.a1 - (double)fc.m0[1], cz = (double)s.a2 - (double)fc.m0[2];
    fs.w1 = end ? 0.0f : (float)((cx * (double)fc.e1[0] + cy * (double)fc.e1[1]) + cz * (double)fc.e1[2]);
    fs.w2 = end ? 0.0f : (float)((cx * (double)fc.e2[0] + cy * (double)fc.e2[1]) + cz * (double)fc.e2[2]);
    fs.cl = end ? 3e38f : (float)((cx * (double)fc.l[0] + cy * (double)fc.l[1]) + cz * (double)fc.l[2]);
    shadow_filter_bounds(fc, s.a3, fs.r2o, fs.r2i);                         // END: rr = +inf -> both +inf
    if (end || item) fs.r2o = -fs.r2o;                                      // sign bit: ITEM or END
    fs.r2o_own = 0.0f; fs.r2i_own = -1.0f;
    if (compacted && !end && !item) shadow_filter_bounds(fc, s.own_rr, fs.r2o_own, fs.r2i_own);
    if (end) fs.r2o_own = -0.0f;                                            // sign bit: END
    fs.skip_off = s.skip_off;
    xprim[i] = fp;
    xshad[i] = fs;
    if (own_item) own_item[i] = (compacted && !end && !item) ? p.item : 0u;
}

// Derives both streams from the raw one: exact IEEE ops, no contraction (same products the CPU path forms).  Threads
// n .. n + kNodePad - 1 write the END nodes.
template <typename T>
__global__ void k_build_streams(const RawNode<T> *__restrict__ raw, unsigned n, V3<T> eye, bool compacted, Node<T> *__restrict__ prim,
                                Node<T> *__restrict__ shad)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n + kNodePad) return;
    constexpr unsigned kStride = (unsigned)sizeof(Node<T>);
    Node<T> p, s;
    if (i >= n) {
        // END: v = 0 and rr = +inf make disc = (b*b - vv) + rr = +inf for every ray of either kind
        p.a0 = p.a1 = p.a2 = p.a3 = T(0); p.a4 = inf<T>();
        s.a0 = s.a1 = s.a2 = T(0); s.a3 = inf<T>(); s.a4 = T(0);
        p.own_rr = s.own_rr = T(0);
        p.item = s.item = kNodeEnd;
        p.skip_off = s.skip_off = n * kStride;
    } else {
        const RawNode<T> r = raw[i];
        const V3<T> v = { r.cx - eye.x, r.cy - eye.y, r.cz - eye.z };      // primitive.rs:56
        const T rr = r.r * r.r;                                           // primitive.rs:58
        p.a0 = v.x; p.a1 = v.y; p.a2 = v.z; p.a3 = dot(v, v); p.a4 = rr;
        s.a0 = r.cx; s.a1 = r.cy; s.a2 = r.cz; s.a3 = rr; s.a4 = T(0);
        p.own_rr = s.own_rr = T(0);
        if (r.skip == 0u) {                                               // ITEM
            p.item = s.item = r.item | kNodeItem;
            p.skip_off = s.skip_off = (i + 1u) * kStride;
        } else {                                                          // BOUND
            p.item = s.item = 0u;
            p.skip_off = s.skip_off = r.skip * kStride;
            if (compacted) {
                p.own_rr = s.own_rr = r.own_r * r.own_r;                  // primitive.rs:58
                p.item = s.item = r.own_item;
            }
        }
    }
    prim[i] = p;
    shad[i] = s;
}

// Minimum over the wave's 64 lanes, in every lane's return value (wave-uniform).  DPP row shifts + the two row broadcasts:
// a dozen VALU ops.  (__shfl_xor compiles to ds_bpermute_b32 -- six dependent LDS round trips sat on the shadow walk's
// critical path every time a ray retired.)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    const int id = -1;      // 0xFFFFFFFF: the identity of an unsigned min
#define RT_DPP_MIN(CTRL, ROWS) v = min(v, (unsigned)__builtin_amdgcn_update_dpp(id, (int)v, CTRL, ROWS, 0xf, false))
    RT_DPP_MIN(0x111, 0xf);     // row_shr:1
    RT_DPP_MIN(0x112, 0xf);     // row_shr:2
    RT_DPP_MIN(0x114, 0xf);     // row_shr:4
    RT_DPP_MIN(0x118, 0xf);     // row_shr:8   -> lane 15 of each row holds the row's minimum
    RT_DPP_MIN(0x142, 0xa);     // row_bcast:15 into rows 1 and 3
    RT_DPP_MIN(0x143, 0xc);     // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's minimum
#undef RT_DPP_MIN
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

constexpr unsigned kNever = 0xFFFFFFFFu;

// A shadow ray's origin in the filter's plane: q = ((o - m0) . e1, (o - m0) . e2).  An origin the constants do not cover (further
// than sqrt(ro2) from m0; none is on a scene the library built the constants for, but the bound must not depend on that) gets a NaN,
// and a NaN passes every bound (v_cmp_ngt): such a ray simply runs the reference's test at every node.
__device__ __forceinline__ void shadow_filter_origin(const FilterConsts &fc, float ox, float oy, float oz, float &q1, float &q2, float &ol)
{
    const float x = ox - fc.m0[0], y = oy - fc.m0[1], z = oz - fc.m0[2];
    q1 = __builtin_fmaf(z, fc.e1[2], __builtin_fmaf(y, fc.e1[1], x * fc.e1[0]));
    q2 = __builtin_fmaf(z, fc.e2[2], __builtin_fmaf(y, fc.e2[1], x * fc.e2[0]));
    ol = __builtin_fmaf(z, fc.l[2], __builtin_fmaf(y, fc.l[1], x * fc.l[0]));
    const float d2 = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
    if (!(d2 <= fc.ro2)) q1 = __builtin_nanf("");
}
// ... of an f64 scene's shadow ray: the origin relative to m0 formed in double and rounded once, then the same f32 chains.
__device__ __forceinline__ void shadow_filter_origin64(const FilterConsts &fc, double ox, double oy, double oz, float &q1, float &q2, float &ol)
{
    const float x = (float)(ox - (double)fc.m0[0]), y = (float)(oy - (double)fc.m0[1]), z = (float)(oz - (double)fc.m0[2]);
    q1 = __builtin_fmaf(z, fc.e1[2], __builtin_fmaf(y, fc.e1[1], x * fc.e1[0]));
    q2 = __builtin_fmaf(z, fc.e2[2], __builtin_fmaf(y, fc.e2[1], x * fc.e2[0]));
    ol = __builtin_fmaf(z, fc.l[2], __builtin_fmaf(y, fc.l[1], x * fc.l[0]));
    const float d2 = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
    if (!(d2 <= fc.ro2)) q1 = __builtin_nanf("");
}
// The two bounds as the *_filt loops evaluate them (the counting launches check them against the reference's test).
__device__ __forceinline__ bool primary_filter_pass(const FNode &f, float dx, float dy, float dz)
{
    return f.f5 <= __builtin_fmaf(f.a2, dz, __builtin_fmaf(f.a1, dy, f.a0 * dx));
}
// The root-free decision of a primary BOUND step, as the filtered loops evaluate it (tools/gen_skip_asm.py bound_shortcut): for a
// node the eye is clearly outside of (FNode::f5 > -inf) and a lane with disc >= 0 -- 0: the lane does not enter, 2: it enters, 1: the root decides.
__device__ __forceinline__ int bound_shortcut_verdict(float b, float disc, float best)
{
    if (!(0.0f < b)) return 0;                          // behind the eye: t2 < 0
    const float w = b - best;
    if (w < 0.0f) return 2;                             // RN(b - s) <= b < hit.distance
    const float k = 0x1.00001p+0f;                      // 1 + 2^-20
    return disc * k <= w * w ? 0 : 1;                   // s <= sqrt(disc)(1 + 2^-24) <= b - hit.distance
}

// The squared in-plane distance as the shadow loops form it.  SUM: the one-ray loops (tools/gen_skip_asm.py packed_p2 -- both differences and
// both squares are one packed instruction each, then the sum: three roundings); otherwise the two-ray loops (rt_skip2_rot.hpp: the second
// square is fused into the sum).  Either is within a factor (1 +- 2.01 eps) of the exact value for the rounded differences, which is what
// the bounds' margins assume (NOTES.md, "Shadow rays, two-sided").
template <bool SUM>
__device__ __forceinline__ float shadow_p2(float w1, float w2, float q1, float q2)
{
    const float t0 = w1 - q1, t1 = w2 - q2;
    if constexpr (SUM) return t0 * t0 + t1 * t1;                     // -ffp-contract=off: never fused
    else return __builtin_fmaf(t1, t1, t0 * t0);
}
// 0: the bounds say MISS, 1: they cannot tell, 2: they say HIT
template <bool SUM>
__device__ __forceinline__ int shadow_filter_verdict(const FNodeS &f, const FilterConsts &fc, float q1, float q2, float ol)
{
    const float p2 = shadow_p2<SUM>(f.w1, f.w2, q1, q2);
    if (p2 > __builtin_fabsf(f.r2o)) return 0;
    const float av = f.cl - ol, inn = __builtin_fmaf(av, av, p2);
    if (__builtin_fmaf(inn, fc.k1, p2) <= f.r2i && (fc.a0 <= av || inn <= f.r2i)) return 2;
    return (av <= -fc.a0 && __builtin_fabsf(f.r2o) <= fc.kc * inn) ? 0 : 1;      // behind the origin, origin outside: t2 < 0
}

// ---- The kernels' one argument (round 6) ----------------------------------------------------------------------------------------------
// What a wave needs first lies first: the dispatch list, the frame, then the SkipView -- the compiler fetches kernel arguments in
// instalments, in the order of their first uses, and every instalment is a scalar-memory round trip of a wave that has not started yet
// (tools/wave_timeline.py "prologue").  Steady-state frames do not run these kernels at all but k_render_skip_fast (rt_skip_fast.hpp), which
// was written around that prologue; these serve every other mode.
template <typename T> struct SkipArgs {
    const BlockDesc *order; const uint32_t *wg_first;
    unsigned width, height, frame_w, spp_arg;
    uint8_t *out; const TileDev *tiles;
    unsigned n_tiles, pad_;
    SkipView<T> sc;
    Counters *counters; uint32_t *lane_cost; const uint64_t *holes; unsigned n_holes; SampleBuf<T> sb; CoopView cv;
};
typedef unsigned rt_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned rt_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned rt_u32x8 __attribute__((ext_vector_type(8)));
typedef unsigned rt_u32x16 __attribute__((ext_vector_type(16)));

// VAR bits (all bit-identical in output and counters):
//   1 = sqrt_rn_lean in the C++ loops (same value as the IEEE sqrt for every input, about half the instructions)
//   2 = (launches that do not count tests) the generated assembly traversal loops, rt_skip_rot.hpp
//   4 = (with 2; fused scenes only) their fused flavour
//   8 = wave trace (diagnostic, RT_WAVE_TRACE): every wave records when and where it ran into `lane_cost`
//  16 = (with 2; f32) the FILTERED assembly loops over the FNode streams: a conservative bound in front of every test
//       (launches that count tests: the C++ loops additionally evaluate the bounds and count any test they would have wrongly ruled out)
//
// SPLIT = false: one thread renders its pixel completely (all spp*spp samples in the reference's order) -- used for
//   spp == 1, where it is a single pass.
// SPLIT = true : one thread traces ONE sample (blockIdx.y = ssx * spp + ssy) and stores the sample's outcome
//   {state, n.light}; k_resolve_samples then accumulates each pixel's samples in the reference's order.  A frame's
//   run time is bounded by its slowest wave, and a wave that walks 16 samples one after the other is 16x slower.
//
// MODE: kSkipLoop = SPLIT false, any spp; kSkipSplit = SPLIT true; kSkipOne = SPLIT false with spp == 1 known at compile
//   time (x + 0/1 == x, v * (1/(1*1)) == v and alpha * 1 == alpha bit for bit, so three IEEE divisions and the multiplies
//   by 1.0 leave the per-wave prologue / epilogue -- a fifth of all VALU work of a 1080p frame is outside the loops).
//   kSkipPacked = sample-parallel like kSkipSplit, for spp*spp in {4, 16, 64}: a wave's lanes enumerate the SAMPLES of a
//   few neighbouring pixels (spp 4: 2x2 pixels x 16 samples) instead of one sample of 8x8 pixels; blockIdx.y picks the
//   sub-block of the 16x16 block.  A wave walks the union of its rays' nodes, and 64 rays through four pixels share far more
//   of their walk than 64 rays spread over 64 pixels (`make image`: 0.39 -> see DESIGN.md); samples are stored
//   [pixel][sample], so a wave's stores are one contiguous run.
enum { kSkipLoop = 0, kSkipSplit = 1, kSkipOne = 2, kSkipPacked = 3 };
// COOP (spp 1, the assembly loops; f64: the filtered ones): some quads of the pass are traced by the lane-cooperative walk (rt_coop.hpp):
//   a narrow descriptor (level 1: 8x8 pixels, a 4x4 quad per wave; level 2: 4x4, 2x2 per wave; level 3: 2x2, one pixel per wave) with a
//   4-bit mask in pitch bits 20..23 is cooperative -- wave w of the workgroup traces its quad cooperatively when bit w is set and leaves
//   otherwise; the rays the cooperative walk hands back (rt_coop.hpp: `failed`) are walked by the loops.  The 16x16 block those quads
//   belong to is descriptor di < n_holes of the same list (or its four quarters, level 1), and holes[di] tells its own waves which 2x2-pixel quads to leave out.
template <typename T, bool COUNT, int VAR, int MODE, bool COOP>
__device__ __forceinline__ void render_skip_body(const BlockDesc *__restrict__ order, const uint32_t *__restrict__ wg_first,
                                                 unsigned width, unsigned height, unsigned frame_w, uint8_t *__restrict__ out,
                                                 const TileDev *__restrict__ tiles, unsigned n_tiles, unsigned spp_arg,
                                                 Counters *__restrict__ counters, uint32_t *__restrict__ lane_cost,
                                                 const uint64_t *__restrict__ holes, unsigned n_holes, SampleBuf<T> sb, SkipView<T> sc, CoopView cv,
                                                 [[maybe_unused]] unsigned long long r_entry)      // (wave trace: the wave's very first instruction)
{
    constexpr bool PACKED = MODE == kSkipPacked, SPLIT = MODE == kSkipSplit || PACKED, ONE = MODE == kSkipOne;
    static_assert(!COOP || (!COUNT && ONE && (VAR & 2) != 0 && (sizeof(T) == 4 || (VAR & 16) != 0)), "the cooperative walk serves spp-1 passes of the assembly loops (f64: the filtered ones)");
    [[maybe_unused]] __shared__ typename CoopLdsOf<T>::type coop_lds[COOP ? kBlockThreads / 64 : 1];
    const unsigned spp = ONE ? 1u : spp_arg;
    // `order` (optional): block descriptors in dispatch order, most expensive block first -- a pass is as long as its last
    // wave, so the long chains must not be the ones dispatched last (rt_capi.hip, block_order).  Without it the workgroup
    // finds its block in the tile table.  `wg_first` (optional, with `order`): workgroup w renders the descriptors
    // [wg_first[w], wg_first[w + 1]) one after the other: sample-parallel passes of more than 32,768 workgroups are dealt out
    // on the host, about eight descriptors to a workgroup, longest first to the least loaded one (rt_capi.hip, block_order).
    unsigned d_first = blockIdx.x, d_last = blockIdx.x + 1;
    if (order && wg_first) { d_first = wg_first[blockIdx.x]; d_last = wg_first[blockIdx.x + 1]; }
    for (unsigned di = d_first; di < d_last; ++di) {
    unsigned bx0, by0, tile_r, tile_t, pitch, base;
    unsigned level = 0;             // 0: 8x8 pixels per wave; 1: 4x4 (16 live lanes); 2: 2x2 (4 live lanes); 3: one pixel
    [[maybe_unused]] unsigned coop_mask = 0;
    if (order) {
        const BlockDesc bd = load_block_desc(order, di);
        bx0 = bd.x0; by0 = bd.y0; tile_r = bd.r; tile_t = bd.t; pitch = bd.pitch & 0xFFFFu; base = bd.base;
        level = (bd.pitch >> kBlockNarrowShift) & 3u;
        if constexpr (COOP) coop_mask = (bd.pitch >> kBlockCoopShift) & 15u;
    } else {
        unsigned lo = 0, hi = n_tiles - 1;
        while (lo < hi) {
            unsigned mid = (lo + hi + 1) >> 1;
            if (tiles[mid].blk_first <= blockIdx.x) lo = mid; else hi = mid - 1;
        }
        const TileDev tile = tiles[lo];
        const unsigned lb = blockIdx.x - tile.blk_first;
        bx0 = tile.l + (lb % tile.blks_x) * kBlockW; by0 = tile.b + (lb / tile.blks_x) * kBlockH;
        tile_r = tile.r; tile_t = tile.t;
        pitch = (unsigned)tile.r - tile.l;
        base = tile.out_px - tile.b * pitch - tile.l;
    }
    const unsigned gblock = di;
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    [[maybe_unused]] const bool coop_wave = COOP && __builtin_amdgcn_readfirstlane((int)((coop_mask >> wave) & 1u)) != 0;        // wave-uniform
    [[maybe_unused]] const unsigned coop_rays = (8u >> level) * (8u >> level);
    unsigned x, y, sample = 0;
    bool inside;
    if (PACKED) {
        // spp 2 / 4 / 8: 4 / 16 / 64 samples per pixel, so a wave holds 4x4 / 2x2 / 1 pixels and the workgroup a sub-block of
        // 8x8 / 4x4 / 2x2 pixels -- spp x spp sub-blocks per 16x16 block, picked by blockIdx.y
        const unsigned ns = spp * spp, lg = 31u - (unsigned)__builtin_clz(ns);
        const unsigned ppw = 8u >> (lg >> 1), pb = 3u - (lg >> 1);          // wave patch: ppw x ppw pixels
        const unsigned pi = lane >> lg;
        sample = lane & (ns - 1u);
        x = bx0 + (blockIdx.y % spp) * 2u * ppw + (wave & 1) * ppw + (pi & (ppw - 1u));
        y = by0 + (blockIdx.y / spp) * 2u * ppw + (wave >> 1) * ppw + (pi >> pb);
        inside = x < tile_r && y < tile_t;
    } else {
        const unsigned pw = 8u >> level, pbits = 3u - level;      // wave patch: pw x pw pixels in the first pw*pw lanes
        x = bx0 + (wave & 1) * pw + (lane & (pw - 1));
        y = by0 + (wave >> 1) * pw + ((lane >> pbits) & (pw - 1));
        inside = x < tile_r && y < tile_t && lane < pw * pw;
        sample = blockIdx.y;
        if constexpr (COOP) {
            // a cooperative descriptor's other waves have nothing to do; an ordinary block leaves its holes to the cooperative descriptors
            if (coop_mask != 0u) inside = inside && coop_wave;
            else if (di < n_holes) {          // level 0: a 16x16 block, 8x8 quads; level 1: one of its 8x8 quarters, 4x4 quads
                const unsigned long long hole = holes[di];
                inside = inside && ((hole >> (((y - by0) >> 1) * (8u >> level) + ((x - bx0) >> 1))) & 1ull) == 0ull;
            }
        }
    }
    if (__ballot(inside) == 0) continue;        // waves are independent here: no LDS, no barrier

    unsigned long long t_start = 0, r_start = 0;
    if (COUNT) { t_start = __builtin_amdgcn_s_memtime(); r_start = __builtin_amdgcn_s_memrealtime(); }
    // diagnostic (RT_WAVE_TRACE, tools/wave_timeline.py): launches that do not count may be handed a trace buffer instead of
    // the cost map -- every wave records when it ran (100 MHz clock) and where
    // (a template bit, not a run-time test: hipcc treats s_memrealtime as a memory clobber and then fetches the wave-uniform
    // node records of the C++ loops below with vector loads -- the f64 frame went from 151 to 248 us)
    constexpr bool trace = !COUNT && (VAR & 8) != 0;
    if (trace) r_start = __builtin_amdgcn_s_memrealtime();

    const T ssf = T(spp);
    const T total_recip = T(1.0) / (ssf * ssf);
    const T fw = T(width), fh = T(height);
    const T half_w = fw / T(2.0), half_h = fh / T(2.0);
    const V3<T> eye = sc.eye, light = sc.light;
    const unsigned n = sc.n_nodes;

    const V3<T> OBJECT = { T(0xae) / T(255.0), T(0x31) / T(255.0), T(0x31) / T(255.0) };
    const V3<T> BACKGROUND = { T(0x22) / T(255.0), T(0x0a) / T(255.0), T(0x0a) / T(255.0) };
    const V3<T> AMBIENT = { BACKGROUND.x * T(0.8), BACKGROUND.y * T(0.8), BACKGROUND.z * T(0.8) };
    const V3<T> sdir = mulf(light, T(-1.0));                        // render.rs:206

    V3<T> g = { T(0.0), T(0.0), T(0.0) };
    T alpha = T(0.0);
    unsigned c_hits = 0, c_shadow = 0, c_occ = 0, c_items = 0, c_bounds = 0, c_steps = 0, c_isteps = 0;
    [[maybe_unused]] unsigned c_fpass = 0, c_fviol = 0, c_ptotal = 0;

    const unsigned ss_first = SPLIT ? sample / spp : 0u, ss_last = SPLIT ? ss_first + 1 : spp;
    for (unsigned ssx = ss_first; ssx < ss_last; ++ssx) {
        for (unsigned ssy = SPLIT ? sample % spp : 0u; ssy < (SPLIT ? sample % spp + 1 : spp); ++ssy) {
            const T xres = ONE ? T(x) : T(x) + T(ssx) / ssf;
            const T yres = ONE ? T(y) : T(y) + T(ssy) / ssf;
            V3<T> dir = { xres - half_w, (fh - yres) - half_h, fw };
            dir = normalized(dir);
            // (the filtered assembly loops ask their filter for every lane and look at `resume` only where somebody passes: a lane without
            // a ray carries a direction no bound lets through -- b' = NaN fails `T <= b'`, T = -inf included; nothing else reads it)
            // (f32 walks only: the f64 loops still compare `resume` at the top of every step, tools/gen_skip_asm.py lazy_wake)
            if constexpr ((VAR & 2) != 0 && (VAR & 16) != 0 && !COUNT && sizeof(T) == 4) { if (!inside) dir.x = T(__builtin_nanf("")); }

            [[maybe_unused]] const unsigned t_before = c_items + c_bounds;
            // ---------------- primary ray: s.group.intersect(&mut h, r)  render.rs:188-189 ----------------
            T best = inf<T>();
            unsigned best_item = 0;
            unsigned resume = inside ? 0u : kNever;
            unsigned i = 0;
            // the assembly loops count in bytes; a lane without a ray sleeps until the END node (offset nb), where every lane
            // is awake and the walk ends
            constexpr unsigned kStride = (unsigned)sizeof(Node<T>);
            const unsigned nb = ((VAR & 4) ? sc.n_fnodes : sc.n_nodes) * kStride;
            if constexpr ((VAR & 2) && !COUNT) {
                // lanes whose ray the loops walk: all of them, or what the cooperative walk of this quad hands back
                bool walk = inside;
                [[maybe_unused]] T cbest = inf<T>();
                [[maybe_unused]] unsigned citem = 0;
                if constexpr (COOP) {
                    if (coop_wave) coop_primary(cv, coop_lds[wave], coop_rays, dir.x, dir.y, dir.z, inside, cbest, citem, walk);
                }
                if (!coop_wave || __ballot(walk) != 0) {
                [[maybe_unused]] float fdx = (float)dir.x;
                if constexpr (COOP) { if (!walk) fdx = __builtin_nanf(""); }      // (a ray the cooperative walk has settled)
                if constexpr ((VAR & 16) != 0 && sizeof(T) == 8) {
                    // f64: the walk reads the scene's FNode stream (f32 filter terms; positions in ITS units) and fetches a node's own
                    // Node<double> record only when the filter lets some live lane through
                    constexpr unsigned kFStride = (unsigned)sizeof(FNode);
                    const unsigned nbf = ((VAR & 4) ? sc.n_fnodes : sc.n_nodes) * kFStride;
                    if constexpr ((VAR & 4) != 0) {
                        skip_primary_rot_filt_fused(sc.xfprim, nbf, dir.x, dir.y, dir.z, walk ? 0u : nbf, best, best_item, fdx, (float)dir.y, (float)dir.z, sc.fprim);
                        if (best_item != 0u && !(best_item & kNodeItem)) best_item = sc.xown[best_item / kFStride - 1u];
                    } else skip_primary_rot_filt(sc.xprim, nbf, dir.x, dir.y, dir.z, walk ? 0u : nbf, best, best_item, fdx, (float)dir.y, (float)dir.z, sc.prim);
                } else if constexpr ((VAR & 16) != 0 && sizeof(T) == 4) {
                    if constexpr ((VAR & 4) != 0) {
                        skip_primary_rot_filt_fused(sc.xfprim, nb, fdx, dir.y, dir.z, walk ? 0u : nb, best, best_item);
                        // a group's own sphere won: the walk recorded the offset behind its BOUND node
                        if (best_item != 0u && !(best_item & kNodeItem)) best_item = sc.xown[best_item / kStride - 1u];
                    } else skip_primary_rot_filt(sc.xprim, nb, fdx, dir.y, dir.z, walk ? 0u : nb, best, best_item);
                } else if constexpr ((VAR & 4) != 0) skip_primary_rot_fused(sc.fprim, nb, dir.x, dir.y, dir.z, walk ? 0u : nb, best, best_item);
                else skip_primary_rot(sc.prim, nb, dir.x, dir.y, dir.z, walk ? 0u : nb, best, best_item);
                best_item &= kNodeIndexMask;
                }
                if constexpr (COOP) {
                    if (coop_wave && !walk) { best = cbest; best_item = citem; }
                }
            } else {
            Node<T> nd = sc.prim[0];                                    // wave-uniform record -> SGPRs
            for (;;) {
                const bool active = i >= resume;
                // Sphere::distance_from_ray with the ray-independent parts pre-formed (primitive.rs:55-72)
                const T b = (nd.a0 * dir.x + nd.a1 * dir.y) + nd.a2 * dir.z;
                const T disc = (b * b - nd.a3) + nd.a4;
                unsigned ni;
                {
                    const bool pos = !(disc < T(0.0));
                    T d = inf<T>();
                    if (pos) {
                        const T s = (VAR & 1) ? sqrt_rn_lean(disc) : rsqrt_exact(disc);
                        const T t2 = b + s;
                        if (!(t2 < T(0.0))) {
                            const T t1 = b - s;
                            d = t1 > T(0.0) ? t1 : t2;
                        }
                    }
                    if constexpr (COUNT && sizeof(T) == 8) {
                        if (sc.xprim && active) {              // the f64 primary walk's f32 filter: never in the way of a finite distance
                            const FNode fn = sc.xprim[i];
                            const bool pass = primary_filter_pass(fn, (float)dir.x, (float)dir.y, (float)dir.z);
                            c_fpass += pass ? 1u : 0u;
                            c_fviol += (!pass && d < inf<T>()) ? 1u : 0u;
                            if (nd.is_bound()) {      // ... the step that needs no exact record (F64F.sure_enter_path): finite for sure, enters for sure
                                const float bf = __builtin_fmaf(fn.a2, (float)dir.z, __builtin_fmaf(fn.a1, (float)dir.y, fn.a0 * (float)dir.x));
                                if (bf >= fn.a3) {
                                    c_fviol += !(d < inf<T>() && d <= b && d > 0.0) ? 1u : 0u;
                                    if ((double)(bf * 0x1.002p+0f) < best) c_fviol += !(d < best) ? 1u : 0u;
                                }
                            }
                            if (nd.is_bound() && pos && fn.f5 > -inf<float>()) {      // ... and the BOUND step's root-free decision, in f64
                                int v = 1;
                                if (!(0.0 < b)) v = 0;
                                else if (b - best < 0.0) v = 2;
                                else if (disc * 0x1.00001p+0 <= (b - best) * (b - best)) v = 0;
                                c_fviol += ((v == 2 && !(d < best)) || (v == 0 && d < best)) ? 1u : 0u;
                            }
                        }
                    }
                    if constexpr (COUNT && sizeof(T) == 4) {
                        if (sc.xprim && active) {
                            const FNode fn = sc.xprim[i];
                            const bool pass = primary_filter_pass(fn, dir.x, dir.y, dir.z);
                            c_fpass += pass ? 1u : 0u;
                            c_fviol += (!pass && d < inf<T>()) ? 1u : 0u;
                            // the BOUND step's root-free decision (rt_skip_rot.hpp, bound_shortcut), held against the reference's own `d >= hit.distance`
                            if (nd.is_bound() && pos && fn.f5 > -inf<float>()) {
                                const int v = bound_shortcut_verdict((float)b, (float)disc, (float)best);
                                c_fviol += ((v == 2 && !(d < best)) || (v == 0 && d < best)) ? 1u : 0u;
                            }
                        }
                    }
                    if (nd.is_bound()) {                                // BOUND  group.rs:73
                        const bool cull = active && (d >= best);
                        if (cull) resume = nd.skip();
                        if (COUNT) c_bounds += active ? 1u : 0u;
                        ni = (__ballot(active && !cull) == 0) ? nd.skip() : i + 1;
                    } else {                                            // ITEM   primitive.rs:78-83
                        if (active && !(d >= best)) { best = d; best_item = nd.index(); }
                        if (COUNT) { c_items += active ? 1u : 0u; ++c_isteps; }
                        ni = i + 1;
                    }
                }
                if (COUNT) ++c_steps;
                if (ni >= n) break;
                nd = sc.prim[ni];
                i = ni;
            }
            }

            if (COUNT) c_ptotal += c_items + c_bounds - t_before;      // tests of this sample's primary ray
            // ---------------- shade  render.rs:190-199 ----------------
            bool need_shadow = false;
            T gdot = T(0.0);
            V3<T> sp = { T(0.0), T(0.0), T(0.0) };
            uint8_t state = kMiss;
            if (inside) {
                if (best == inf<T>()) {
                    if (!SPLIT) g = add(g, BACKGROUND);
                } else {
                    ++c_hits;
                    const Item<T> it = sc.items[best_item];
                    const V3<T> c = { it.cx, it.cy, it.cz };
                    const V3<T> nrm = normalized(add(eye, sub(mulf(dir, best), c)));       // primitive.rs:83
                    gdot = dot(nrm, light);
                    if (gdot >= T(0.0)) {
                        state = kAmbient;
                        if (!SPLIT) g = add(g, AMBIENT);
                    } else {
                        need_shadow = true;
                        ++c_shadow;
                        const V3<T> ns = mulf(nrm, best * rsqrt_exact(eps<T>()));
                        sp = add(add(eye, mulf(dir, best)), ns);
                    }
                }
            }

            // ---------------- shadow ray: any hit  render.rs:202-208 ----------------
            // hit.distance stays INF until the first hit, so a bound culls iff the ray misses it; the lane
            // retires at its first item hit (only has_missed() is asked afterwards).
            bool occluded = false;
            resume = need_shadow ? 0u : kNever;
            i = 0;
            if constexpr ((VAR & 2) && !COUNT) {
                bool walk_s = need_shadow;
                if constexpr (COOP) {
                    if (coop_wave && __ballot(need_shadow) != 0) coop_shadow(cv, coop_lds[wave], coop_rays, sp.x, sp.y, sp.z, sdir, need_shadow, occluded, walk_s);
                }
                if (__ballot(walk_s) != 0) {
                    resume = walk_s ? 0u : nb;                      // lanes without a shadow ray sleep until END
                    [[maybe_unused]] float q1 = 0.0f, q2 = 0.0f, fol = 0.0f, fa0 = 0.0f, fk1 = 0.0f, fkc = 0.0f;
                    if constexpr ((VAR & 16) != 0 && sizeof(T) == 4) {
                        const FilterConsts fc = *sc.fc;
                        shadow_filter_origin(fc, sp.x, sp.y, sp.z, q1, q2, fol);
                        if (!walk_s) q1 = inf<float>();          // no shadow ray: an in-plane origin at infinity is beyond every outer bound but END's
                        fa0 = fc.a0; fk1 = fc.k1; fkc = fc.kc;
                    }
                    if constexpr ((VAR & 16) != 0 && sizeof(T) == 8) {
                        // f64: the walk reads the FNodeS stream (positions in ITS units) and fetches a node's Node<double> record only when some live
                        // lane is inside the node's outer bound
                        constexpr unsigned kFStride = (unsigned)sizeof(FNodeS);
                        const unsigned nbf = ((VAR & 4) ? sc.n_fnodes : sc.n_nodes) * kFStride;
                        const FilterConsts fc = *sc.fc;
                        float fq1, fq2, fql;
                        shadow_filter_origin64(fc, sp.x, sp.y, sp.z, fq1, fq2, fql);
                        resume = walk_s ? 0u : nbf;
                        while (i < nbf) {
                            unsigned fin;
                            if constexpr ((VAR & 4) != 0) i = skip_shadow_rot_filt_fused(sc.xfshad, nbf, i, sp.x, sp.y, sp.z, sdir.x, sdir.y, sdir.z, resume, fin, fq1, fq2, fql, fc.a0, fc.k1, fc.kc, sc.fshad);
                            else i = skip_shadow_rot_filt(sc.xshad, nbf, i, sp.x, sp.y, sp.z, sdir.x, sdir.y, sdir.z, resume, fin, fq1, fq2, fql, fc.a0, fc.k1, fc.kc, sc.shad);
                            if (i >= nbf) break;
                            if (fin) { occluded = true; resume = nbf; }
                            i = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_min_u32(resume >= nbf ? nbf : (resume > i ? resume : i + kFStride)));
                        }
                        i = nb;                          // (the plain loop below has nothing left to do)
                    }
                    while (i < nb) {
                        unsigned fin;
                        if constexpr ((VAR & 16) != 0 && sizeof(T) == 4) {
                            if constexpr ((VAR & 4) != 0)
                                i = skip_shadow_rot_filt_fused(sc.xfshad, nb, i, sp.x, sp.y, sp.z, sdir.x, sdir.y, sdir.z, resume, fin, q1, q2, fol, fa0, fk1, fkc, sc.fshad);
                            else i = skip_shadow_rot_filt(sc.xshad, nb, i, sp.x, sp.y, sp.z, sdir.x, sdir.y, sdir.z, resume, fin, q1, q2, fol, fa0, fk1, fkc, sc.shad);
                        } else if constexpr ((VAR & 4) != 0) i = skip_shadow_rot_fused(sc.fshad, nb, i, sp.x, sp.y, sp.z, sdir.x, sdir.y, sdir.z, resume, fin);
                        else i = skip_shadow_rot(sc.shad, nb, i, sp.x, sp.y, sp.z, sdir.x, sdir.y, sdir.z, resume, fin);
                        if (i >= nb) break;
                        if (fin) { occluded = true; resume = nb; q1 = inf<float>(); }      // (a retired lane passes no bound any more)
                        // some lane retired at the node at i: go straight to the next node any lane still wants (nb: nobody is left)
                        i = (unsigned)__builtin_amdgcn_readfirstlane(
                            (int)wave_min_u32(resume >= nb ? nb : (resume > i ? resume : i + kStride)));
                    }
                }
            } else if (__ballot(need_shadow) != 0) {
                [[maybe_unused]] float fq1 = 0.0f, fq2 = 0.0f, fql = 0.0f;
                [[maybe_unused]] FilterConsts cfc{};
                if constexpr (COUNT && sizeof(T) == 4) { if (sc.xshad) { cfc = *sc.fc; shadow_filter_origin(cfc, sp.x, sp.y, sp.z, fq1, fq2, fql); } }
                if constexpr (COUNT && sizeof(T) == 8) { if (sc.xshad) { cfc = *sc.fc; shadow_filter_origin64(cfc, sp.x, sp.y, sp.z, fq1, fq2, fql); } }
                Node<T> nd = sc.shad[0];
                for (;;) {
                    const bool active = i >= resume;
                    const V3<T> v = { nd.a0 - sp.x, nd.a1 - sp.y, nd.a2 - sp.z };
                    const T b = dot(v, sdir);
                    const T disc = (b * b - dot(v, v)) + nd.a3;
                    bool hit = false;
                    {
                        const bool pos = !(disc < T(0.0));
                        if (pos) hit = !((b + ((VAR & 1) ? sqrt_rn_lean(disc) : rsqrt_exact(disc))) < T(0.0));
                    }
                    if constexpr (COUNT && sizeof(T) == 8) {
                        if (sc.xshad && active) {              // the f64 shadow walk's f32 bounds: MISS and HIT must be what the f64 test says
                            const int verdict = shadow_filter_verdict<true>(sc.xshad[i], cfc, fq1, fq2, fql);
                            c_fpass += verdict == 1 ? 1u : 0u;
                            c_fviol += ((verdict == 0 && hit) || (verdict == 2 && !hit)) ? 1u : 0u;
                        }
                    }
                    if constexpr (COUNT && sizeof(T) == 4) {
                        if (sc.xshad && active) {
                            const int verdict = shadow_filter_verdict<true>(sc.xshad[i], cfc, fq1, fq2, fql);
                            const int verdict2 = shadow_filter_verdict<false>(sc.xshad[i], cfc, fq1, fq2, fql);    // the two-ray loops' rounding
                            c_fpass += verdict == 1 ? 1u : 0u;              // tests the bounds leave to the reference's arithmetic
                            c_fviol += ((verdict == 0 && hit) || (verdict == 2 && !hit)) ? 1u : 0u;
                            c_fviol += ((verdict2 == 0 && hit) || (verdict2 == 2 && !hit)) ? 1u : 0u;
                        }
                    }
                    unsigned ni;
                    if (nd.is_bound()) {
                        const bool cull = active && !hit;
                        if (cull) resume = nd.skip();
                        if (COUNT) c_bounds += active ? 1u : 0u;
                        ni = (__ballot(active && hit) == 0) ? nd.skip() : i + 1;
                    } else {
                        const bool fin = active && hit;
                        if (COUNT) { c_items += active ? 1u : 0u; ++c_isteps; }
                        if (fin) { occluded = true; resume = kNever; }
                        if (__ballot(fin) != 0) {
                            // some lane retired: go straight to the next node any lane still wants
                            ni = (unsigned)__builtin_amdgcn_readfirstlane(
                                (int)wave_min_u32(resume == kNever ? kNever : (resume > i ? resume : i + 1)));
                        } else {
                            ni = i + 1;
                        }
                    }
                    if (COUNT) ++c_steps;
                    if (ni >= n) break;                                 // also ni == kNever: every lane retired
                    nd = sc.shad[ni];
                    i = ni;
                }
            }

            if (need_shadow) {
                if (!occluded) {
                    state = kLit;
                    if (!SPLIT) {
                        g = add(add(g, mulf(OBJECT, -gdot)), AMBIENT);      // render.rs:209
                        alpha += T(1.0);
                    }
                } else {
                    ++c_occ;
                    state = kShadowed;
                    if (!SPLIT) g = add(add(g, BACKGROUND), mulf(AMBIENT, -gdot));      // render.rs:212
                }
            }
            if (SPLIT && inside) {
                const size_t px_i = (size_t)(base + y * pitch + x);
                const size_t p = PACKED ? px_i * (spp * spp) + sample : (size_t)sample * sb.n_px + px_i;
                if constexpr (PACKED && sizeof(T) == 4) {
                    reinterpret_cast<uint32_t *>(sb.gdot)[p] = sample_word(state, gdot);      // k_resolve_words
                } else {
                    sb.gdot[p] = gdot;
                    sb.state[p] = state;
                }
            }
        }
    }

    if (!SPLIT && inside) {
        if (!ONE) {
            g = mulf(g, total_recip);
            alpha *= total_recip;
        }
        const size_t px = frame_w ? (size_t)y * frame_w + x : (size_t)(base + y * pitch + x);
        const unsigned rgba = scale_u8(g.x) | (scale_u8(g.y) << 8) | (scale_u8(g.z) << 16) | (scale_u8(alpha) << 24);
        reinterpret_cast<unsigned *>(out)[px] = rgba;
        if (COUNT) { if (lane_cost) lane_cost[px] = c_items + c_bounds; }   // the scene's cost map is rendered through this
    }

    if (trace && lane_cost && lane == 0) {
        const unsigned n_desc = (order && wg_first) ? wg_first[gridDim.x] : gridDim.x;
        // eight words per wave, in descriptor order:
        // start, end, HW_ID | XCC_ID << 16, descriptor | coop << 31, entry (the wave's first instruction), ack (the pixel store has been
        // acknowledged: s_endpgm waits for that too, so the slot is held until then), 0, 0
        uint32_t *rec = lane_cost + (((size_t)blockIdx.y * n_desc + di) * 4 + wave) * 8;
        rec[0] = (uint32_t)r_start;
        rec[1] = (uint32_t)__builtin_amdgcn_s_memrealtime();
        rec[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | (__builtin_amdgcn_s_getreg((31 << 11) | 20) << 16);  // HW_ID | XCC_ID << 16
        rec[3] = gblock | (coop_wave ? 0x80000000u : 0u);
        rec[4] = (uint32_t)r_entry;
        __builtin_amdgcn_s_waitcnt(0);          // vmcnt(0) expcnt(0) lgkmcnt(0): the pixel store (and the four words above) have landed
        rec[5] = (uint32_t)__builtin_amdgcn_s_memrealtime();
    }
    if (COUNT) {
        Counters *const stripe = counters + (gblock + blockIdx.y) % kCounterStripes;
        const unsigned long long prim = wave_sum(inside ? (SPLIT ? 1u : spp * spp) : 0u);
        const unsigned long long hits = wave_sum(c_hits), sh = wave_sum(c_shadow), oc = wave_sum(c_occ);
        const unsigned long long its = wave_sum(c_items), bds = wave_sum(c_bounds);
        const unsigned long long fpass = wave_sum(c_fpass), fviol = wave_sum(c_fviol), ptot = wave_sum(c_ptotal);
        if (lane == 0) {
            atomicAdd(&stripe->primary, prim);
            atomicAdd(&stripe->hits, hits);
            atomicAdd(&stripe->shadow, sh);
            atomicAdd(&stripe->occluded, oc);
            atomicAdd(&stripe->sphere_tests, its);
            atomicAdd(&stripe->bound_tests, bds);
            atomicAdd(&stripe->wave_steps, (unsigned long long)c_steps);
            atomicAdd(&stripe->wave_item_steps, (unsigned long long)c_isteps);
            atomicAdd(&stripe->primary_tests, ptot);
            if (fpass) atomicAdd(&stripe->filter_pass, fpass);
            if (fviol) atomicAdd(&stripe->filter_violations, fviol);
            atomicMax(&stripe->max_wave_steps, (unsigned long long)c_steps);
            atomicMax(&stripe->max_wave_cycles, (unsigned long long)(__builtin_amdgcn_s_memtime() - t_start));
            atomicMax(&stripe->max_wave_ref100mhz, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - r_start));
        }
    }
    }       // descriptors of this workgroup
}

// (The counting launches and the UNFILTERED f64 loops -- f64 scenes too large for the filter streams' 32-bit offsets.  Held to six waves
// per SIMD: the f64 walk waits for its node records like the f32 one, its 81 vector registers were one too many for the sixth wave -- 80
// without a spill, 1080p f64 69.8 -> 65.0 us; these loops own s[36:97], so a seventh wave is not to be had here: k_render_skip_f64 below.)
#ifndef RT_F64_SGPRS
#define RT_F64_SGPRS 96
#define RT_F64_WAVES 7
#endif
template <typename T, bool COUNT, int VAR, int MODE, bool COOP = false>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(6))) void k_render_skip(SkipArgs<T> args)
{
    unsigned long long r_entry = 0;
    if constexpr (!COUNT && (VAR & 8) != 0) r_entry = __builtin_amdgcn_s_memrealtime();
    const SkipArgs<T> &a = args;
    render_skip_body<T, COUNT, VAR, MODE, COOP>(a.order, a.wg_first, a.width, a.height, a.frame_w, a.out, a.tiles, a.n_tiles, a.spp_arg, a.counters, a.lane_cost, a.holes,
                     a.n_holes, a.sb, a.sc, a.cv, r_entry);
}

// The same kernel for f32 launches that do not count, held to 74 scalar registers -- 80 with the hardware's six, and 80 is what a CU admits
// EIGHT 256-thread workgroups at (MI355X_MICROARCH.md, "Residency": .sgpr_count <= 80 -> 8, 82 - 96 -> 7 whatever the compiler's occupancy
// remark says, 98+ -> 6).  The loops themselves own s[36:73]; everything the kernel keeps across them beyond s[0:35] is parked in
// vector-register lanes (v_writelane / v_readlane outside the loops; the kernel has 47 of its 64 vector registers to spare).  History: the
// unconstrained build has 106 (6 workgroups per CU); round 4 first capped it at 94 (-> 92: 7 per CU, 1080p 46.9 -> 44.3 us, and was
// taken for 8), then at 74 (-> 80: 8 per CU, 42.7 -> 41.8 us); round 5 asks for 82, which is the same 80 (the loops' s[36:73] + the
// hardware's six) without LLVM calling the loops' top eight registers reserved.  The walk waits for its node records half of its time: a
// wave more per SIMD is a fetch more in flight.  The f64 loops own s[36:97] and cannot live under such a limit, hence kernels of their
// own rather than an attribute on the template.
template <bool COUNT, int VAR, int MODE>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_num_sgpr(82))) void k_render_skip_f32(SkipArgs<float> args)
{
    unsigned long long r_entry = 0;
    if constexpr (!COUNT && (VAR & 8) != 0) r_entry = __builtin_amdgcn_s_memrealtime();
    const SkipArgs<float> &a = args;
    render_skip_body<float, COUNT, VAR, MODE, false>(a.order, a.wg_first, a.width, a.height, a.frame_w, a.out, a.tiles, a.n_tiles, a.spp_arg, a.counters, a.lane_cost, a.holes,
                     a.n_holes, a.sb, a.sc, a.cv, r_entry);
}
// ... and its flavour with the lane-cooperative walk (rt_coop.hpp), which keeps more state across the loops: at 74 it parks 41 values and
// the small passes it serves lose 2 % (800x600 28.7 -> 29.4 us); at 94 (7 workgroups per CU) they do not miss the eighth.
template <bool COUNT, int VAR, int MODE>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_num_sgpr(94))) void k_render_skip_f32_coop(SkipArgs<float> args)
{
    unsigned long long r_entry = 0;
    if constexpr (!COUNT && (VAR & 8) != 0) r_entry = __builtin_amdgcn_s_memrealtime();
    const SkipArgs<float> &a = args;
    render_skip_body<float, COUNT, VAR, MODE, true>(a.order, a.wg_first, a.width, a.height, a.frame_w, a.out, a.tiles, a.n_tiles, a.spp_arg, a.counters, a.lane_cost, a.holes,
                     a.n_holes, a.sb, a.sc, a.cv, r_entry);
}

// ... and the f64 launches of the FILTERED loops (VAR & 16), whose window ends at s89 (tools/gen_skip_asm.py F64F): 96 scalar registers with the
// hardware's six and 72 vector registers are what a SIMD admits seven waves at (round 5: 1080p f64 61.7 -> 59.9 us, 1024x768 spp 4 403.5 ->
// 389.2, interleaved; the spp-1 flavour parks two doubles in scratch across the primary walk -- 20 bytes, stored and loaded once per ray).
// LLVM calls s88 and s89 reserved at this limit, as it did s72 and s73 of the f32 kernel at 80: they are below the six the hardware adds.
template <int VAR, int MODE>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_num_sgpr(RT_F64_SGPRS), amdgpu_waves_per_eu(RT_F64_WAVES))) void k_render_skip_f64(SkipArgs<double> args)
{
    static_assert((VAR & 16) != 0 && (VAR & 2) != 0, "the filtered assembly loops");
    unsigned long long r_entry = 0;
    if constexpr ((VAR & 8) != 0) r_entry = __builtin_amdgcn_s_memrealtime();
    const SkipArgs<double> &a = args;
    render_skip_body<double, false, VAR, MODE, false>(a.order, a.wg_first, a.width, a.height, a.frame_w, a.out, a.tiles, a.n_tiles, a.spp_arg, a.counters, a.lane_cost, a.holes,
                     a.n_holes, a.sb, a.sc, a.cv, r_entry);
}

// ... with the lane-cooperative walk in f64 (round 6; rt_coop.hpp CNode64): the filtered loops need s[88:89] RESERVED as in k_render_skip_f64
// (tools/check_reserved_registers.py) -- amdgpu_num_sgpr(96) does that by itself --; the walk's state takes the kernel beyond the 72 vector
// registers of seven waves per SIMD, and the passes that take this flavour are the ones that wait for chains, not for slots.
#ifndef RT_F64_COOP_WAVES
#define RT_F64_COOP_WAVES 5
#endif
template <int VAR, int MODE>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_num_sgpr(RT_F64_SGPRS), amdgpu_waves_per_eu(RT_F64_COOP_WAVES))) void k_render_skip_f64_coop(SkipArgs<double> args)
{
    static_assert((VAR & 16) != 0 && (VAR & 2) != 0 && MODE == kSkipOne, "the filtered assembly loops, one sample per pixel");
    unsigned long long r_entry = 0;
    if constexpr ((VAR & 8) != 0) r_entry = __builtin_amdgcn_s_memrealtime();
    const SkipArgs<double> &a = args;
    render_skip_body<double, false, VAR, MODE, true>(a.order, a.wg_first, a.width, a.height, a.frame_w, a.out, a.tiles, a.n_tiles, a.spp_arg, a.counters, a.lane_cost, a.holes,
                     a.n_holes, a.sb, a.sc, a.cv, r_entry);
}

// Second pass of the SPLIT path: render.rs:233-252 for one pixel -- its samples' contributions accumulated strictly
// in the reference's order (ssx outer, ssy inner; each term added on its own, never pre-summed), then quantised.
template <typename T>
__global__ __launch_bounds__(kBlockThreads) void k_resolve_samples(SampleBuf<T> sb, unsigned spp, const TileDev *__restrict__ tiles,
                                                                  unsigned n_tiles, uint8_t *__restrict__ out, unsigned frame_w,
                                                                  bool packed)      // samples stored [pixel][sample], not [sample][pixel]
{
    unsigned lo = 0, hi = n_tiles - 1;
    while (lo < hi) {
        unsigned mid = (lo + hi + 1) >> 1;
        if (tiles[mid].blk_first <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const TileDev tile = tiles[lo];
    const unsigned lb = blockIdx.x - tile.blk_first;
    const unsigned x = tile.l + (lb % tile.blks_x) * kBlockW + (threadIdx.x & 15);
    const unsigned y = tile.b + (lb / tile.blks_x) * kBlockH + (threadIdx.x >> 4);
    if (!(x < tile.r && y < tile.t)) return;
    const size_t p = out_index(tile, x, y, 0);                   // the samples are always stored tile-major
    const V3<T> OBJECT = { T(0xae) / T(255.0), T(0x31) / T(255.0), T(0x31) / T(255.0) };
    const V3<T> BACKGROUND = { T(0x22) / T(255.0), T(0x0a) / T(255.0), T(0x0a) / T(255.0) };
    const V3<T> AMBIENT = { BACKGROUND.x * T(0.8), BACKGROUND.y * T(0.8), BACKGROUND.z * T(0.8) };
    const T ssf = T(spp);
    const T total_recip = T(1.0) / (ssf * ssf);
    V3<T> g = { T(0.0), T(0.0), T(0.0) };
    T alpha = T(0.0);
    const unsigned ns = spp * spp;
    for (unsigned k = 0; k < ns; ++k) {
        const size_t q = packed ? p * ns + k : (size_t)k * sb.n_px + p;
        const uint8_t st = sb.state[q];
        const T gdot = sb.gdot[q];
        if (st == kMiss) g = add(g, BACKGROUND);                                        // render.rs:191
        else if (st == kAmbient) g = add(g, AMBIENT);                                   // render.rs:196
        else if (st == kLit) { g = add(add(g, mulf(OBJECT, -gdot)), AMBIENT); alpha += T(1.0); }      // render.rs:209-210
        else g = add(add(g, BACKGROUND), mulf(AMBIENT, -gdot));                         // render.rs:212
    }
    g = mulf(g, total_recip);
    alpha *= total_recip;
    reinterpret_cast<unsigned *>(out)[out_index(tile, x, y, frame_w)] =
        scale_u8(g.x) | (scale_u8(g.y) << 8) | (scale_u8(g.z) << 16) | (scale_u8(alpha) << 24);
}

// The same for the f32 sample-packed passes (kSkipPacked, k_render_skip2): one word per sample (sample_word, rt_kernels.hpp), a pixel's
// NS samples contiguous -- a thread fetches them 16 bytes at a time and adds them in the reference's order.  Each sample adds
// (g + A) + B: {BACKGROUND, 0}, {AMBIENT, 0}, {OBJECT * -n.light, AMBIENT} (render.rs:209), {BACKGROUND, AMBIENT * -n.light} (render.rs:212);
// adding +0 to a non-negative sum changes no bit, so the single-term exits need no branch.
template <unsigned NS>
__global__ __launch_bounds__(kBlockThreads) void k_resolve_words(const uint4 *__restrict__ words, const TileDev *__restrict__ tiles, unsigned n_tiles,
                                                                uint8_t *__restrict__ out, unsigned frame_w)
{
    typedef float T;
    unsigned lo = 0, hi = n_tiles - 1;
    while (lo < hi) {
        unsigned mid = (lo + hi + 1) >> 1;
        if (tiles[mid].blk_first <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const TileDev tile = tiles[lo];
    const unsigned lb = blockIdx.x - tile.blk_first;
    const unsigned x = tile.l + (lb % tile.blks_x) * kBlockW + (threadIdx.x & 15);
    const unsigned y = tile.b + (lb / tile.blks_x) * kBlockH + (threadIdx.x >> 4);
    if (!(x < tile.r && y < tile.t)) return;
    const size_t p = out_index(tile, x, y, 0);                   // the samples are always stored tile-major
    const V3<T> OBJECT = { T(0xae) / T(255.0), T(0x31) / T(255.0), T(0x31) / T(255.0) };
    const V3<T> BACKGROUND = { T(0x22) / T(255.0), T(0x0a) / T(255.0), T(0x0a) / T(255.0) };
    const V3<T> AMBIENT = { BACKGROUND.x * T(0.8), BACKGROUND.y * T(0.8), BACKGROUND.z * T(0.8) };
    constexpr unsigned kSpp = NS == 4 ? 2u : NS == 16 ? 4u : 8u;
    const T ssf = T(kSpp);
    const T total_recip = T(1.0) / (ssf * ssf);
    V3<T> g = { T(0.0), T(0.0), T(0.0) };
    T alpha = T(0.0);
    const uint4 *src = words + p * (NS / 4u);
#pragma unroll 4
    for (unsigned j = 0; j < NS / 4u; ++j) {
        const uint4 q = src[j];
        const uint32_t w4[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
        for (unsigned k = 0; k < 4; ++k) {
            const uint32_t w = w4[k];
            const bool miss = w == kSampleMiss, amb = w == kSampleAmbient, lit = !miss && !amb && (w & 0x80000000u) != 0u, sh = !miss && !amb && !lit;
            const T m = __uint_as_float(w & 0x7fffffffu);           // -n.light of a lit / shadowed sample
            const V3<T> A = { lit ? OBJECT.x * m : amb ? AMBIENT.x : BACKGROUND.x, lit ? OBJECT.y * m : amb ? AMBIENT.y : BACKGROUND.y,
                              lit ? OBJECT.z * m : amb ? AMBIENT.z : BACKGROUND.z };
            const V3<T> B = { lit ? AMBIENT.x : sh ? AMBIENT.x * m : T(0.0), lit ? AMBIENT.y : sh ? AMBIENT.y * m : T(0.0),
                              lit ? AMBIENT.z : sh ? AMBIENT.z * m : T(0.0) };
            g = add(add(g, A), B);
            alpha += lit ? T(1.0) : T(0.0);
        }
    }
    g = mulf(g, total_recip);
    alpha *= total_recip;
    reinterpret_cast<unsigned *>(out)[out_index(tile, x, y, frame_w)] =
        scale_u8(g.x) | (scale_u8(g.y) << 8) | (scale_u8(g.z) << 16) | (scale_u8(alpha) << 24);
}

}  // namespace rt
