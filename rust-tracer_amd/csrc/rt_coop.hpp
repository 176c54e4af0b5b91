// rt_coop.hpp -- the lane-cooperative walk of the heaviest pixels of a pass (f32; f64 since round 6: CNode64).
//
// The skip-pointer walk (rt_skip.hpp) gives every lane a ray and moves the whole wave through the node stream one node at a time:
// a pixel whose ray meets 400-500 nodes is a chain of that many DEPENDENT scalar loads, each an L2 round trip (a node is touched once
// per wave), and a frame waits for the few waves that hold such pixels while the rest of the chip is idle.  Here the lanes of a wave
// carry NODES instead: a wave owns a 4x4-pixel quad (16 rays) and keeps a work list of (ray, group) pairs in LDS; every round the
// lanes take the children of the pairs on top of the list -- one (ray, child) test per lane, the node records fetched by independent
// vector loads --, run the reference's own test on them (primitive.rs:55-72, operation for operation) and put every group whose bound
// returns a finite distance back on the list: a gather of the ray's CLOSURE, the nodes all of whose ancestors have a finite bound
// distance.  About eight dependent rounds of loads on the reference's pyramid instead of several hundred.
//
// Why the result is the reference's (TypedGroup::intersect, group.rs:72-83: `if bound.distance_from_ray(ray) >= hit.distance return`,
// Sphere::intersect, primitive.rs:77-84: strict `<`, so the first item in traversal order keeps a tie):
//   * The reference only descends into a group whose bound distance d is < hit.distance <= INF, so every node it visits is in the
//     closure; it visits a subset of it, in DFS order.
//   * Shadow rays (render.rs:202-208 only asks has_missed()): hit.distance stays INF until the first item hit, so the reference
//     visits exactly the closure until then -- the ray is occluded iff some item of the closure returns a finite distance.  No check.
//   * Primary rays: let F be the smallest distance over the closure's items and t the first item in DFS order (= smallest DFS item
//     index) that attains it.  If every ancestor bound a of t has d_a <= F, the reference's result is (F, t): when the DFS reaches a,
//     hit.distance is the minimum over items visited before a -- all of them closure items before t, hence > F >= d_a -- so a is
//     entered; t is reached with hit.distance > F and takes it; no later item has a smaller distance, and equal ones do not replace
//     it.  Each work-list pair carries the largest bound distance on its path (`anc`), the winner's is kept beside the minimum, and a
//     ray whose winner fails `anc <= F` (a ray that starts INSIDE a bound: the reference then compares the far intersection, SURVEY
//     H2) is handed back to the skip-pointer walk, as is every ray of a wave whose work list overflows.  Exact on every scene.
//   * Every distance compared is the reference's individually rounded one; distances are >= +0 (t2 is never -0: disc is never -0 and
//     b + root of opposite signs sums to +0), so their bit patterns order like the values and (distance bits << 32 | DFS item index)
//     is the key of one LDS ds_min_u64 per item hit.
//
// The node records live in a second copy of the hierarchy in SIBLING-CONTIGUOUS (breadth-first) order, so the children of a group
// are `count` consecutive records from `first` -- a pair on the work list is one 8-byte word pair {first | count << 24 | ray << 28, anc}.
#pragma once
#include "rt_kernels.hpp"

namespace rt {

// One node of the cooperative copy.  Primary array: a = {vx, vy, vz, vv}, a4 = rr (v = centre - eye: the same pre-formed terms as
// Node<T>, rt_skip.hpp, copied from that stream bit for bit); shadow array: a = {cx, cy, cz, rr}, a4 unused.
// count == 0: an ITEM, `first` is its DFS item index.  count > 0: a BOUND whose children are records [first, first + count).
typedef float rt_f32x3 __attribute__((ext_vector_type(3), aligned(4)));
struct alignas(32) CNode {
    float a0, a1, a2, a3;
    float a4;
    uint32_t first, count, pad;
};
static_assert(sizeof(CNode) == 32, "two 16-byte vector loads");
// ... of an f64 scene (round 6): the same terms as Node<double> carries them; bytes [32, 48) = {a4, first, count} are one 16-byte load
struct alignas(64) CNode64 {
    double a0, a1, a2, a3;
    double a4;
    uint32_t first, count;
    uint32_t pad[4];
};
static_assert(sizeof(CNode64) == 64, "three 16-byte vector loads of a 64-byte record");
template <typename T> struct CNodeOf { typedef CNode type; };
template <> struct CNodeOf<double> { typedef CNode64 type; };

constexpr unsigned kCoopRays = 16;            // a wave's quad: at most 4x4 pixels (level-1 descriptors; level 2: 2x2, level 3: one)
constexpr unsigned kCoopStack = 448;          // work-list capacity per wave (pairs)
constexpr unsigned kCoopMaxFanout = 15;       // `count` has four bits in a work-list word
constexpr uint32_t kCoopMaxNodes = 1u << 24;  // `first` has 24

struct CoopView {
    const void *prim = nullptr, *shad = nullptr;      // CNodeOf<T>::type arrays of the scene's precision
    uint32_t n_roots = 0;       // the top-level nodes are records [0, n_roots)
    uint32_t fanout = 0;        // the largest child count of the scene (and n_roots): lanes are dealt to (pair, child) by it; 0: no cooperative copy
};

struct CoopLds {
    float ray[kCoopRays][4];                    // primary: direction; shadow: origin
    unsigned long long best[kCoopRays];         // (distance bits << 32) | item, ~0 = no hit
    float best_anc[kCoopRays];                  // largest ancestor bound distance of the item that holds `best`
    uint32_t occluded;                          // shadow: bit per ray
    uint32_t pad;
    uint2 stack[kCoopStack];
};
static_assert(sizeof(CoopLds) <= 4096, "four waves and eight workgroups per CU: 160 KB of LDS");
// f64: a distance is 64 bits, so the minimum by (distance, DFS index) is kept in two words (coop_primary says how); `anc` travels as an f32
// ROUNDED UP -- the check `anc <= F` may then fail for a ray it would have passed (that ray is walked by the loops: never wrong), never
// pass one it should fail -- which keeps a work-list pair at eight bytes.
struct CoopLds64 {
    double ray[kCoopRays][4];
    unsigned long long best[kCoopRays];         // distance bits, ~0 = no hit
    uint32_t best_item[kCoopRays];              // the smallest DFS index among the items at that distance
    float best_anc[kCoopRays];
    uint32_t occluded;
    uint32_t pad;
    uint2 stack[kCoopStack];
};
static_assert(sizeof(CoopLds64) <= 4608, "four waves per workgroup: 18 KB");
template <typename T> struct CoopLdsOf { typedef CoopLds type; };
template <> struct CoopLdsOf<double> { typedef CoopLds64 type; };

__device__ __forceinline__ unsigned coop_lane_rank(unsigned long long mask)      // set bits of mask below this lane
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// sqrt_rn_lean (rt_math.hpp) for an operand that is known to be finite and >= 0 here (disc of a scene rt_scene_create accepted: no
// intermediate overflows): one range test instead of two.
__device__ __forceinline__ float coop_sqrt(float x)
{
    if (__builtin_expect(!(x >= 0x1p-96f), 0)) return __builtin_sqrtf(x);
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float r = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(r, h, g);
}

__device__ __forceinline__ double coop_sqrt(double x) { return __builtin_sqrt(x); }      // IEEE (rt_math.hpp: what the f64 loops use)

// Sphere::distance_from_ray with the ray-independent terms pre-formed (primitive.rs:55-72), as the C++ loop of k_render_skip forms it.
template <typename T>
__device__ __forceinline__ T coop_primary_distance(T vx, T vy, T vz, T vv, T rr, T dx, T dy, T dz)
{
    const T b = (vx * dx + vy * dy) + vz * dz;
    const T disc = (b * b - vv) + rr;
    T d = inf<T>();
    if (!(disc < T(0.0))) {
        const T s = coop_sqrt(disc);
        const T t2 = b + s;
        if (!(t2 < T(0.0))) {
            const T t1 = b - s;
            d = t1 > T(0.0) ? t1 : t2;
        }
    }
    return d;
}

// ... for a ray with its own origin: does the shadow ray hit the sphere (finite distance)?  primitive.rs:55-68
template <typename T>
__device__ __forceinline__ bool coop_shadow_hit(T cx, T cy, T cz, T rr, T ox, T oy, T oz, V3<T> sdir)
{
    const V3<T> v = { cx - ox, cy - oy, cz - oz };
    const T b = dot(v, sdir);
    const T disc = (b * b - dot(v, v)) + rr;
    bool hit = false;
    if (!(disc < T(0.0))) hit = !((b + coop_sqrt(disc)) < T(0.0));
    return hit;
}

// A bound distance (>= +0, finite) as the f32 the work list carries: the value itself in f32; in f64 the nearest f32 ABOVE it
__device__ __forceinline__ uint32_t coop_anc_bits(float d) { return __float_as_uint(d); }
__device__ __forceinline__ uint32_t coop_anc_bits(double d)
{
    const float f = (float)d;
    return __float_as_uint(f) + ((double)f < d ? 1u : 0u);
}

// The pairs a round takes from the top of the work list, dealt to the lanes: lane (e, k) gets child k of the e-th pair from the top.
struct CoopSlot { unsigned e, k, per; };
__device__ __forceinline__ CoopSlot coop_slot(unsigned fan)
{
    const unsigned lane = threadIdx.x & 63u;
    CoopSlot s;
    s.per = 64u / fan;
    s.e = lane / fan;
    s.k = lane - s.e * fan;
    return s;
}
__device__ __forceinline__ unsigned coop_uniform(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }

// The lanes of a wave hand each other rays, work-list pairs and minima through LDS.  A wave's LDS operations are issued and served in order,
// but that is the hardware's promise, not the language's: between a phase that writes and a phase in which OTHER lanes read, the writes are
// released and the compiler is told not to move LDS accesses across (no instruction beyond the s_waitcnt it would emit anyway).
__device__ __forceinline__ void coop_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The primary rays of a quad: lanes [0, n_rays) hold them (dir; `want`: the lane has a ray; n_rays = 16, 4 or 1).  Returns, in those lanes,
// hit.distance / the DFS index of the nearest item, and `failed` = the ray has to be walked by the skip-pointer loops instead (its winner is
// nearer than one of its ancestor bounds, or the wave's work list overflowed).  Wave-uniform control flow; `lds` is this wave's.
// PIPE (round 6, measured and NOT used): the records of the NEXT batch of pairs are requested before the current batch is evaluated -- a
// round is one trip to L2 plus ~75 instructions, a heavy quad makes about thirty of them, and whenever the list holds more than one batch
// the next one's pairs are known already; the order in which pairs are taken is free (a gather of the closure: minima by key), so the batch
// in flight simply counts as taken when the current one pushes.  Round 4 had dropped this for its registers (76 in the generic cooperative
// kernel); k_render_skip_fast_coop has them (40 -> 53 of 64), the waits land where they should (s_waitcnt vmcnt(2) at the head of a
// round: the previous batch's two loads are back, this one's two are in flight), every parity test passes -- and a cooperative wave takes
// 16.1 us instead of 13.7 (median, 800x600; the frame 25.2 against 25.4).  For the primary gather alone (an occluded shadow ray wants its
// items soon, which the deeper order of one batch at a time finds sooner): 14.06 against 14.26 us per wave, the frame the same.  Kept as an option.
// (the pair's ray travels with the batch: it is read from LDS while the records are on their way, not after they have arrived -- round 6)
template <typename T> struct CoopBatchP { uint2 e; float4 g, h; float rx, ry, rz; unsigned n; bool have; };      // g = {vx, vy, vz, vv}, h = {rr, first, count, -}
template <> struct CoopBatchP<double> { uint2 e; double2 g01, g23; double a4, rx, ry, rz; uint2 link; unsigned n; bool have; };
template <bool PIPE = false, typename T = float>
__device__ __forceinline__ void coop_primary(const CoopView &cv, typename CoopLdsOf<T>::type &lds, unsigned n_rays, T dx, T dy, T dz, bool want, T &best_out,
                                             unsigned &item_out, bool &failed_out)
{
    constexpr bool F64 = sizeof(T) == 8;
    typedef typename CNodeOf<T>::type Rec;
    const unsigned lane = threadIdx.x & 63u;
    const CoopSlot sl = coop_slot(cv.fanout);
    const bool mine = want && lane < n_rays;
    if (lane < n_rays) {
        lds.ray[lane][0] = dx; lds.ray[lane][1] = dy; lds.ray[lane][2] = dz;
        lds.best[lane] = ~0ull;
        lds.best_anc[lane] = 0.0f;
        if constexpr (F64) lds.best_item[lane] = 0xFFFFFFFFu;
    }
    // one pair per ray: (ray, the top level)
    const unsigned long long wm = __ballot(mine);
    if (mine) lds.stack[coop_lane_rank(wm)] = make_uint2(0u | (cv.n_roots << 24) | (lane << 28), 0u);
    unsigned top = coop_uniform((unsigned)__popcll(wm));
    bool overflow = false;
    coop_lds_sync();                             // rays, minima and the first pairs are written: every lane may read them
    // the pairs on top of the list, dealt to the lanes, and their records requested
    auto fetch = [&]() {
        CoopBatchP<T> b;
        b.n = min(sl.per, top);
        b.have = sl.e < b.n;
        b.e = lds.stack[b.have ? top - 1u - sl.e : 0u];
        top -= b.n;
        const unsigned cnt = (b.e.x >> 24) & 15u;
        const bool valid = b.have && sl.k < cnt;
        const unsigned node = valid ? (b.e.x & 0xFFFFFFu) + sl.k : 0u;
        const char *rec = static_cast<const char *>(cv.prim) + node * (unsigned)sizeof(Rec);        // (a 32-bit offset: fewer than 2^24 nodes)
        if constexpr (F64) {
            b.g01 = *reinterpret_cast<const double2 *>(rec);
            b.g23 = *reinterpret_cast<const double2 *>(rec + 16);
            const uint4 t = *reinterpret_cast<const uint4 *>(rec + 32);       // {a4 (rr), first, count}
            b.a4 = __hiloint2double((int)t.y, (int)t.x);
            b.link = make_uint2(t.z, t.w);
        } else {
            b.g = *reinterpret_cast<const float4 *>(rec);
            // ({rr, first, count}: three words -- a fourth, dead one would be handed to the next instruction as a register, and that instruction
            // would wait for the load)
            const rt_f32x3 h3 = *reinterpret_cast<const rt_f32x3 *>(rec + 16);
            b.h = make_float4(h3[0], h3[1], h3[2], 0.0f);
        }
        const unsigned ray = b.e.x >> 28;
        b.rx = lds.ray[ray][0]; b.ry = lds.ray[ray][1]; b.rz = lds.ray[ray][2];
        __builtin_amdgcn_sched_barrier(0);       // (requests first, arithmetic behind them: the scheduler would sink the LDS reads below the wait for the records)
        return b;
    };
    // one round: the reference's test on every (ray, child) of the batch, minima, and the groups that go back on the list
    auto eval = [&](const CoopBatchP<T> &b) {
        const unsigned cnt = (b.e.x >> 24) & 15u, ray = b.e.x >> 28;
        const bool valid = b.have && sl.k < cnt;
        const T rx = b.rx, ry = b.ry, rz = b.rz;
        T d;
        unsigned link_first, link_count;
        if constexpr (F64) {
            d = coop_primary_distance<double>(b.g01.x, b.g01.y, b.g23.x, b.g23.y, b.a4, rx, ry, rz);
            link_first = b.link.x; link_count = b.link.y;
        } else {
            d = coop_primary_distance<float>(b.g.x, b.g.y, b.g.z, b.g.w, b.h.x, rx, ry, rz);
            link_first = __float_as_uint(b.h.y); link_count = __float_as_uint(b.h.z);
        }
        const bool finite = valid && d < inf<T>();
        const float anc = __uint_as_float(b.e.y);
        // ITEM: keep the nearest by (distance, DFS index), and beside it the largest bound distance on its path
        const bool item_hit = finite && link_count == 0u;
        if (__ballot(item_hit) != 0ull) {
            if constexpr (F64) {
                // the distance's 64 bits are the key of the atomic minimum; the DFS index of the items AT the minimum is a second one.  A round
                // that lowers the minimum throws the index of the old one away first.
                const unsigned long long key = (unsigned long long)__double_as_longlong(d);
                const unsigned long long before = lds.best[ray];
                coop_lds_sync();                 // (every lane has read the minimum of the rounds before)
                if (item_hit) atomicMin(&lds.best[ray], key);
                coop_lds_sync();
                const bool at_min = item_hit && lds.best[ray] == key;
                if (at_min && key < before) lds.best_item[ray] = 0xFFFFFFFFu;
                coop_lds_sync();
                if (at_min) atomicMin(&lds.best_item[ray], link_first);
                coop_lds_sync();
                if (at_min && lds.best_item[ray] == link_first) lds.best_anc[ray] = anc;
            } else {
                const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | link_first;
                if (item_hit) atomicMin(&lds.best[ray], key);
                coop_lds_sync();                 // (every lane's minimum is in before any lane looks whether it holds it)
                if (item_hit && lds.best[ray] == key) lds.best_anc[ray] = anc;
            }
        }
        // BOUND with a finite distance: its children are wanted
        const bool push = finite && link_count != 0u;
        const unsigned long long pm = __ballot(push);
        const unsigned n_push = (unsigned)__popcll(pm);
        if (top + n_push > kCoopStack) { overflow = true; return; }
        if (push) lds.stack[top + coop_lane_rank(pm)] = make_uint2(link_first | (link_count << 24) | (ray << 28), max(b.e.y, coop_anc_bits(d)));      // (both >= +0: the bit patterns order like the values)
        top = coop_uniform(top + n_push);
        coop_lds_sync();                         // the pushed pairs are the next round's reads
    };
    if constexpr (!PIPE) {
        while (top > 0u && !overflow) { const CoopBatchP<T> b = fetch(); eval(b); }
    } else {
        // two batches in two sets of registers, by turns (a copy between rounds makes the compiler wait for the fetch it is meant to hide)
        CoopBatchP<T> a = fetch();
        while (a.n != 0u) {
            CoopBatchP<T> b = fetch();           // (empty when the list holds no more than `a`: then it is fetched behind a's pushes)
            eval(a);
            if (overflow) break;
            if (b.n == 0u) { b = fetch(); if (b.n == 0u) break; }
            a = fetch();
            eval(b);
            if (overflow) break;
            if (a.n == 0u) a = fetch();
        }
    }
    coop_lds_sync();
    T best = inf<T>();
    unsigned item = 0u;
    bool failed = false;
    if (mine) {
        const unsigned long long k = lds.best[lane];
        if (k != ~0ull) {
            if constexpr (F64) { best = __longlong_as_double((long long)k); item = lds.best_item[lane]; }
            else { best = __uint_as_float((unsigned)(k >> 32)); item = (unsigned)k; }
            failed = (T)lds.best_anc[lane] > best;         // an ancestor bound is farther than the winner: the reference may have culled it
        }
        failed = failed || overflow;
    }
    best_out = best; item_out = item; failed_out = failed;
}

// The shadow rays of a quad (any hit, render.rs:202-208): lanes [0, n_rays), origin (ox, oy, oz), `want`: the lane casts one.  Returns
// `occluded` in those lanes and `failed` (work list overflow: the skip-pointer loops decide).
template <typename T> struct CoopBatchS { uint2 e; float4 g; float rx, ry, rz; uint2 link; unsigned n; bool valid; };
template <> struct CoopBatchS<double> { uint2 e; double2 g01, g23; double rx, ry, rz; uint2 link; unsigned n; bool valid; };
template <bool PIPE = false, typename T = float>
__device__ __forceinline__ void coop_shadow(const CoopView &cv, typename CoopLdsOf<T>::type &lds, unsigned n_rays, T ox, T oy, T oz, V3<T> sdir, bool want,
                                            bool &occluded_out, bool &failed_out)
{
    constexpr bool F64 = sizeof(T) == 8;
    typedef typename CNodeOf<T>::type Rec;
    const unsigned lane = threadIdx.x & 63u;
    const CoopSlot sl = coop_slot(cv.fanout);
    const bool mine = want && lane < n_rays;
    if (lane < n_rays) { lds.ray[lane][0] = ox; lds.ray[lane][1] = oy; lds.ray[lane][2] = oz; }
    if (lane == 0u) lds.occluded = 0u;
    const unsigned long long wm = __ballot(mine);
    if (mine) lds.stack[coop_lane_rank(wm)] = make_uint2(0u | (cv.n_roots << 24) | (lane << 28), 0u);
    unsigned top = coop_uniform((unsigned)__popcll(wm));
    bool overflow = false;
    coop_lds_sync();
    auto fetch = [&]() {
        CoopBatchS<T> b;
        b.n = min(sl.per, top);
        const bool have = sl.e < b.n;
        b.e = lds.stack[have ? top - 1u - sl.e : 0u];
        const unsigned occ = lds.occluded;
        top -= b.n;
        const unsigned cnt = (b.e.x >> 24) & 15u, ray = b.e.x >> 28;
        b.valid = have && sl.k < cnt && ((occ >> ray) & 1u) == 0u;       // a ray that is occluded wants nothing more (PIPE: as of one round ago -- a test too many, never one too few)
        const unsigned node = b.valid ? (b.e.x & 0xFFFFFFu) + sl.k : 0u;
        const char *rec = static_cast<const char *>(cv.shad) + node * (unsigned)sizeof(Rec);
        if constexpr (F64) {
            b.g01 = *reinterpret_cast<const double2 *>(rec);
            b.g23 = *reinterpret_cast<const double2 *>(rec + 16);
            b.link = *reinterpret_cast<const uint2 *>(rec + 40);
        } else {
            b.g = *reinterpret_cast<const float4 *>(rec);
            b.link = *reinterpret_cast<const uint2 *>(rec + 20);
        }
        b.rx = lds.ray[ray][0]; b.ry = lds.ray[ray][1]; b.rz = lds.ray[ray][2];
        __builtin_amdgcn_sched_barrier(0);
        return b;
    };
    auto eval = [&](const CoopBatchS<T> &b) {
        const unsigned ray = b.e.x >> 28;
        const T rx = b.rx, ry = b.ry, rz = b.rz;
        bool hit;
        if constexpr (F64) hit = b.valid && coop_shadow_hit<double>(b.g01.x, b.g01.y, b.g23.x, b.g23.y, rx, ry, rz, sdir);
        else hit = b.valid && coop_shadow_hit<float>(b.g.x, b.g.y, b.g.z, b.g.w, rx, ry, rz, sdir);
        const bool item_hit = hit && b.link.y == 0u;
        if (__ballot(item_hit) != 0ull) {
            if (item_hit) atomicOr(&lds.occluded, 1u << ray);
        }
        const bool push = hit && b.link.y != 0u;
        const unsigned long long pm = __ballot(push);
        const unsigned n_push = (unsigned)__popcll(pm);
        if (top + n_push > kCoopStack) { overflow = true; return; }
        if (push) lds.stack[top + coop_lane_rank(pm)] = make_uint2(b.link.x | (b.link.y << 24) | (ray << 28), 0u);
        top = coop_uniform(top + n_push);
        coop_lds_sync();                         // pushed pairs and occluded bits are the next round's reads
    };
    if constexpr (!PIPE) {
        while (top > 0u && !overflow) { const CoopBatchS<T> b = fetch(); eval(b); }
    } else {
        CoopBatchS<T> a = fetch();
        while (a.n != 0u) {
            CoopBatchS<T> b = fetch();
            eval(a);
            if (overflow) break;
            if (b.n == 0u) { b = fetch(); if (b.n == 0u) break; }
            a = fetch();
            eval(b);
            if (overflow) break;
            if (a.n == 0u) a = fetch();
        }
    }
    coop_lds_sync();
    const unsigned occ = lds.occluded;
    occluded_out = mine && ((occ >> (lane & 15u)) & 1u) != 0u;
    // an occluded ray is settled whatever happened to the list afterwards: some item of its closure is hit
    failed_out = mine && overflow && !occluded_out;
}

// Device half of the cooperative copy: record j of the breadth-first arrays is node perm[j] of the (plain) skip streams, whose terms are
// copied bit for bit; link[j] = {first, count}.
template <typename T>
__global__ void k_build_coop(const void *prim_stream, const void *shad_stream, unsigned node_stride, const uint32_t *__restrict__ perm,
                             const uint2 *__restrict__ link, unsigned n, typename CNodeOf<T>::type *__restrict__ cprim, typename CNodeOf<T>::type *__restrict__ cshad)
{
    const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const T *p = reinterpret_cast<const T *>(static_cast<const char *>(prim_stream) + (size_t)perm[j] * node_stride);
    const T *s = reinterpret_cast<const T *>(static_cast<const char *>(shad_stream) + (size_t)perm[j] * node_stride);
    typename CNodeOf<T>::type a = {}, b = {};
    a.a0 = p[0]; a.a1 = p[1]; a.a2 = p[2]; a.a3 = p[3]; a.a4 = p[4];
    b.a0 = s[0]; b.a1 = s[1]; b.a2 = s[2]; b.a3 = s[3]; b.a4 = T(0.0);
    a.first = b.first = link[j].x; a.count = b.count = link[j].y;
    cprim[j] = a;
    cshad[j] = b;
}

}  // namespace rt
