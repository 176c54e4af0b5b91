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

namespace rt {

// One stream node.  PRIMARY stream (all primary rays share Scene::eye):  a = {vx, vy, vz, vv, rr} with
// v = centre - eye, vv = dot(v, v), rr = radius * radius -- the ray-independent sub-expressions of
// primitive.rs:56-58, computed on the device with the same individually rounded operations.
// SHADOW stream (per-lane origin): a = {cx, cy, cz, rr, -}.
// skip != 0: BOUND node, skip = index of the first node after the group's subtree.  skip == 0: ITEM node.
template <typename T> struct alignas(sizeof(T) * 8) Node {
    T a0, a1, a2, a3, a4;
    uint32_t skip, item;
};
static_assert(sizeof(Node<float>) == 32 && sizeof(Node<double>) == 64, "node records are one aligned scalar-load unit");

// Host-built raw stream entry (before the device derives the two streams above).
template <typename T> struct RawNode {
    T cx, cy, cz, r;
    uint32_t skip, item;
};

template <typename T> struct SkipView {
    const Node<T> *prim;    // primary-ray stream
    const Node<T> *shad;    // shadow-ray stream
    const Item<T> *items;   // DFS items (centre of the winning item for the normal)
    uint32_t n_nodes;
    V3<T> light, eye;
};

// Derives both streams from the raw one: exact IEEE ops, no contraction (same products the CPU path forms).
template <typename T>
__global__ void k_build_streams(const RawNode<T> *__restrict__ raw, unsigned n, V3<T> eye, Node<T> *__restrict__ prim,
                                Node<T> *__restrict__ shad)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const RawNode<T> r = raw[i];
    const V3<T> v = { r.cx - eye.x, r.cy - eye.y, r.cz - eye.z };      // primitive.rs:56
    const T rr = r.r * r.r;                                           // primitive.rs:58
    Node<T> p; p.a0 = v.x; p.a1 = v.y; p.a2 = v.z; p.a3 = dot(v, v); p.a4 = rr; p.skip = r.skip; p.item = r.item;
    Node<T> s; s.a0 = r.cx; s.a1 = r.cy; s.a2 = r.cz; s.a3 = rr; s.a4 = T(0); s.skip = r.skip; s.item = r.item;
    prim[i] = p;
    shad[i] = s;
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (unsigned)__shfl_xor((int)v, o, 64));
    return v;
}

constexpr unsigned kNever = 0xFFFFFFFFu;

template <typename T, bool COUNT>
__global__ __launch_bounds__(kBlockThreads) void k_render_skip(SkipView<T> sc, unsigned width, unsigned height, unsigned spp,
                                                              const TileDev *__restrict__ tiles, unsigned n_tiles,
                                                              uint8_t *__restrict__ out, Counters *__restrict__ counters)
{
    unsigned lo = 0, hi = n_tiles - 1;
    while (lo < hi) {
        unsigned mid = (lo + hi + 1) >> 1;
        if (tiles[mid].blk_first <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const TileDev tile = tiles[lo];
    const unsigned lb = blockIdx.x - tile.blk_first;
    const unsigned bx = lb % tile.blks_x, by = lb / tile.blks_x;
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned x = tile.l + bx * kBlockW + (wave & 1) * 8 + (lane & 7);
    const unsigned y = tile.b + by * kBlockH + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = x < tile.r && y < tile.t;
    if (__ballot(inside) == 0) return;          // waves are independent here: no LDS, no barrier

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
    unsigned c_hits = 0, c_shadow = 0, c_occ = 0, c_items = 0, c_bounds = 0;

    for (unsigned ssx = 0; ssx < spp; ++ssx) {
        for (unsigned ssy = 0; ssy < spp; ++ssy) {
            const T xres = T(x) + T(ssx) / ssf;
            const T yres = T(y) + T(ssy) / ssf;
            V3<T> dir = { xres - half_w, (fh - yres) - half_h, fw };
            dir = normalized(dir);

            // ---------------- primary ray: s.group.intersect(&mut h, r)  render.rs:188-189 ----------------
            T best = inf<T>();
            unsigned best_item = 0;
            unsigned resume = inside ? 0u : kNever;
            unsigned i = 0;
            while (i < n) {
                const Node<T> nd = sc.prim[i];                          // wave-uniform record -> SGPRs
                const bool active = i >= resume;
                // Sphere::distance_from_ray with the ray-independent parts pre-formed (primitive.rs:55-72)
                const T b = (nd.a0 * dir.x + nd.a1 * dir.y) + nd.a2 * dir.z;
                const T disc = (b * b - nd.a3) + nd.a4;
                T d = inf<T>();
                if (!(disc < T(0.0))) {
                    const T s = rsqrt_exact(disc);
                    const T t2 = b + s;
                    if (!(t2 < T(0.0))) {
                        const T t1 = b - s;
                        d = t1 > T(0.0) ? t1 : t2;
                    }
                }
                if (nd.skip != 0u) {                                    // BOUND  group.rs:73
                    const bool cull = active && (d >= best);
                    if (cull) resume = nd.skip;
                    if (COUNT) c_bounds += active ? 1u : 0u;
                    i = (__ballot(active && !cull) == 0) ? nd.skip : i + 1;
                } else {                                                // ITEM   primitive.rs:78-83
                    if (active && !(d >= best)) { best = d; best_item = nd.item; }
                    if (COUNT) c_items += active ? 1u : 0u;
                    i = i + 1;
                }
            }

            // ---------------- shade  render.rs:190-199 ----------------
            bool need_shadow = false;
            T gdot = T(0.0);
            V3<T> sp = { T(0.0), T(0.0), T(0.0) };
            if (inside) {
                if (best == inf<T>()) {
                    g = add(g, BACKGROUND);
                } else {
                    ++c_hits;
                    const Item<T> it = sc.items[best_item];
                    const V3<T> c = { it.cx, it.cy, it.cz };
                    const V3<T> nrm = normalized(add(eye, sub(mulf(dir, best), c)));       // primitive.rs:83
                    gdot = dot(nrm, light);
                    if (gdot >= T(0.0)) {
                        g = add(g, AMBIENT);
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
            if (__ballot(need_shadow) != 0) {
                while (i < n) {
                    const Node<T> nd = sc.shad[i];
                    const bool active = i >= resume;
                    const V3<T> v = { nd.a0 - sp.x, nd.a1 - sp.y, nd.a2 - sp.z };
                    const T b = dot(v, sdir);
                    const T disc = (b * b - dot(v, v)) + nd.a3;
                    bool hit = false;
                    if (!(disc < T(0.0))) {
                        const T t2 = b + rsqrt_exact(disc);
                        hit = !(t2 < T(0.0));
                    }
                    if (nd.skip != 0u) {
                        const bool cull = active && !hit;
                        if (cull) resume = nd.skip;
                        if (COUNT) c_bounds += active ? 1u : 0u;
                        i = (__ballot(active && hit) == 0) ? nd.skip : i + 1;
                    } else {
                        const bool fin = active && hit;
                        if (COUNT) c_items += active ? 1u : 0u;
                        if (fin) { occluded = true; resume = kNever; }
                        if (__ballot(fin) != 0) {
                            // some lane retired: go straight to the next node any lane still wants
                            const unsigned nxt = (unsigned)__builtin_amdgcn_readfirstlane(
                                (int)wave_min_u32(resume == kNever ? kNever : (resume > i ? resume : i + 1)));
                            if (nxt == kNever) break;
                            i = nxt;
                        } else {
                            i = i + 1;
                        }
                    }
                }
            }

            if (need_shadow) {
                if (!occluded) {
                    g = add(add(g, mulf(OBJECT, -gdot)), AMBIENT);          // render.rs:209
                    alpha += T(1.0);
                } else {
                    ++c_occ;
                    g = add(add(g, BACKGROUND), mulf(AMBIENT, -gdot));      // render.rs:212
                }
            }
        }
    }

    if (inside) {
        g = mulf(g, total_recip);
        alpha *= total_recip;
        const unsigned tw = tile.r - tile.l;
        const size_t px = (size_t)tile.out_px + (size_t)(y - tile.b) * tw + (x - tile.l);
        const unsigned rgba = scale_u8(g.x) | (scale_u8(g.y) << 8) | (scale_u8(g.z) << 16) | (scale_u8(alpha) << 24);
        reinterpret_cast<unsigned *>(out)[px] = rgba;
    }

    if (COUNT) {
        const unsigned long long prim = wave_sum(inside ? spp * spp : 0u);
        const unsigned long long hits = wave_sum(c_hits), sh = wave_sum(c_shadow), oc = wave_sum(c_occ);
        const unsigned long long its = wave_sum(c_items), bds = wave_sum(c_bounds);
        if (lane == 0) {
            atomicAdd(&counters->primary, prim);
            atomicAdd(&counters->hits, hits);
            atomicAdd(&counters->shadow, sh);
            atomicAdd(&counters->occluded, oc);
            atomicAdd(&counters->sphere_tests, its);
            atomicAdd(&counters->bound_tests, bds);
        }
    }
}

}  // namespace rt
