// rt_skip_fast.hpp -- k_render_skip_fast / k_render_skip_fast_coop: the f32 hierarchy walk of a single-pass, ordered launch (spp 1, a dispatch list,
// no counters -- every steady-state frame of a scheduler: BASELINE configs 2 - 4) with a wave's FIXED costs cut down (round 6); the _coop flavour
// also walks the list's cooperative quads (rt_coop.hpp) and is what such lists run (rt_capi_launch.hpp).
//
// What tools/wave_timeline.py showed on the generic kernel (k_render_skip_f32, rt_skip.hpp): between two waves a wave slot stands 2.3 us --
// 0.16 store acknowledgement, 0.7 - 1.2 the hardware's relaunch, 1.4 the new wave's prologue -- and a wave that walks almost nothing still
// takes 4 us: more than half of a 1080p launch's slot time is spent outside the traversal loops.  The generic body serves every mode
// (raster and dealt lists, sample-parallel passes, counters, holes, two output layouts), keeps the values of all of them alive across the
// loops -- 18 - 40 of them parked in vector-register lanes, since the loops own s[36:73] of a kernel held to 80 scalar registers -- and
// meets its arguments in five dependent scalar-memory round trips before the first ray exists.
//
// This kernel is the same arithmetic (the same inline functions, the same generated loops) for ONE mode:
//   * arguments laid out by when they are needed, three scalar round trips in all: [order .. light] at entry (one s_load_dwordx16), the
//     descriptor, and -- after the primary walk, together with the winner's centre, a vector load the wave waits for anyway -- the shadow
//     walk's pointers and the filter's constants BY VALUE (no dependent fetch through a pointer);
//   * nothing parked: what shading needs of the entry batch (eye, light) waits in six vector registers, the pixel's address in two; the
//     late batch is requested where it is used;
//   * no raster search, no dealt loop, no sample loop (holes and cooperative descriptors: the _coop flavour, four words more in the entry batch).
// Launches that are not of this kind run the generic kernels (rt_capi.hip launch_skip_one).
#pragma once
#include "rt_skip.hpp"

#ifndef RT_FAST_COOP_SGPRS
#define RT_FAST_COOP_SGPRS 82
#endif

namespace rt {

struct FastArgs {
    // entry batch: dwords [0, 16)
    const BlockDesc *order;         // the dispatch list: one descriptor per workgroup
    const FNode *walk_prim;         // the primary walk's filtered stream (fused flavour: the compacted one)
    unsigned width, height;
    unsigned nb;                    // that stream's length in bytes
    unsigned frame_w;               // 0: tile-major output
    uint8_t *out;
    float eye[3], light[3];
    // late batch: dwords [16, 40)
    const Item<float> *items;
    const uint32_t *own;            // the compacted stream's own_item table
    const FNodeS *walk_shad;        // the shadow walk's filtered stream
    const Node<float> *exact_shad;  // ... and the exact records behind it
    float fc[16];                   // FilterConsts: m0, e1, e2, l, a0, k1, kc, ro2
    // (wave trace, hooks build)
    uint32_t *trace;
    // the COOP flavour (some quads of the pass are walked lane-cooperatively, rt_coop.hpp): dwords [42, 46) ride with the entry batch
    const uint64_t *holes;          // descriptors [0, n_holes): the 2x2-pixel quads of that block which cooperative descriptors render
    unsigned n_holes, pad_;
    CoopView cv;
};
static_assert(offsetof(FastArgs, items) == 64 && offsetof(FastArgs, fc) == 96 && offsetof(FastArgs, trace) == 160 && offsetof(FastArgs, holes) == 168,
              "the batches of k_render_skip_fast");
static_assert(offsetof(FilterConsts, a0) == 48 && offsetof(FilterConsts, ro2) == 60, "fc[16] is the head of FilterConsts");

// VAR: 19 (plain filtered streams) or 23 (fused); TRACE: the wave timeline's records (tools/wave_timeline.py; hooks build); COOP: the list
// carries cooperative descriptors and holes (rt_skip.hpp render_skip_body says how a pass is cut up for them)
template <int VAR, bool TRACE, bool COOP>
__device__ __forceinline__ void render_skip_fast_body(const FastArgs &args)
{
    typedef float T;
    constexpr bool FUSED = (VAR & 4) != 0;
    [[maybe_unused]] __shared__ CoopLds coop_lds[COOP ? kBlockThreads / 64 : 1];
    [[maybe_unused]] unsigned long long r_entry = 0, r_start = 0;
    if constexpr (TRACE) r_entry = __builtin_amdgcn_s_memrealtime();
    // ---- entry batch, then the descriptor ----
    const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
    rt_u32x16 q;
    [[maybe_unused]] rt_u32x4 qc = { 0u, 0u, 0u, 0u };
    if constexpr (COOP) asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0xa8\n\ts_waitcnt lgkmcnt(0)" : "=&s"(q), "=&s"(qc) : "s"(kp));
    else asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(q) : "s"(kp));
    typedef const rt_u32x4 __attribute__((address_space(4))) *desc_ptr;        // a BlockDesc as its four words
    const desc_ptr order = (desc_ptr)(((unsigned long long)q[1] << 32) | q[0]);
    const void *walk_prim = (const void *)(((unsigned long long)q[3] << 32) | q[2]);
    const unsigned width = q[4], height = q[5], nb = q[6], frame_w = q[7];
    const unsigned long long out_bits = ((unsigned long long)q[9] << 32) | q[8];
    // (eye and light are uniform, and needed after the primary walk: six vector registers instead of six scalar ones the loops would evict)
    float ex, ey, ez, lx, ly, lz;
    asm volatile("v_mov_b32_e32 %0, %6\n\tv_mov_b32_e32 %1, %7\n\tv_mov_b32_e32 %2, %8\n\tv_mov_b32_e32 %3, %9\n\tv_mov_b32_e32 %4, %10\n\tv_mov_b32_e32 %5, %11"
                 : "=v"(ex), "=v"(ey), "=v"(ez), "=v"(lx), "=v"(ly), "=v"(lz) : "s"(q[10]), "s"(q[11]), "s"(q[12]), "s"(q[13]), "s"(q[14]), "s"(q[15]));
    const V3<T> eye = { ex, ey, ez }, light = { lx, ly, lz };

    const rt_u32x4 raw = order[blockIdx.x];      // one s_load_dwordx4
    const unsigned bx0 = raw[0] & 0xFFFFu, by0 = raw[0] >> 16, tile_r = raw[1] & 0xFFFFu, tile_t = raw[1] >> 16, pitch = raw[2] & 0xFFFFu, base = raw[3];
    const unsigned level = (raw[2] >> kBlockNarrowShift) & 3u;      // 0: 8x8 pixels per wave; 1: 4x4 (16 live lanes); 2: 2x2
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned pw = 8u >> level, pbits = 3u - level;
    const unsigned x = bx0 + (wave & 1) * pw + (lane & (pw - 1));
    const unsigned y = by0 + (wave >> 1) * pw + ((lane >> pbits) & (pw - 1));
    bool inside = x < tile_r && y < tile_t && lane < pw * pw;
    [[maybe_unused]] bool coop_wave = false;
    [[maybe_unused]] const unsigned coop_rays = pw * pw;
    if constexpr (COOP) {
        // a cooperative descriptor's other waves have nothing to do; an ordinary block leaves its holes to the cooperative descriptors
        const unsigned coop_mask = (raw[2] >> kBlockCoopShift) & 15u;
        coop_wave = __builtin_amdgcn_readfirstlane((int)((coop_mask >> wave) & 1u)) != 0;
        if (coop_mask != 0u) inside = inside && coop_wave;
        else if (blockIdx.x < qc[2]) {
            typedef const unsigned long long __attribute__((address_space(4))) *hole_ptr;
            const unsigned long long hole = ((hole_ptr)(((unsigned long long)qc[1] << 32) | qc[0]))[blockIdx.x];
            inside = inside && ((hole >> (((y - by0) >> 1) * (8u >> level) + ((x - bx0) >> 1))) & 1ull) == 0ull;
        }
    }
    if (__ballot(inside) == 0) return;
    if constexpr (TRACE) r_start = __builtin_amdgcn_s_memrealtime();
    // where the pixel goes: known now, needed last -- as an address in two vector registers
    typedef unsigned __attribute__((address_space(1))) *pixel_ptr;
    pixel_ptr px_ptr;
    {
        const unsigned long long a = out_bits + 4ull * (frame_w ? (size_t)y * frame_w + x : (size_t)(base + y * pitch + x));
        unsigned lo = (unsigned)a, hi = (unsigned)(a >> 32);
        asm volatile("" : "+v"(lo), "+v"(hi));
        px_ptr = (pixel_ptr)(((unsigned long long)hi << 32) | lo);
    }

    // ---- render.rs:238-243, one sample ----
    const T fw = T(width), fh = T(height);
    const T half_w = fw / T(2.0), half_h = fh / T(2.0);
    V3<T> dir = { T(x) - half_w, (fh - T(y)) - half_h, fw };
    dir = normalized(dir);

    // ---- primary ray: s.group.intersect(&mut h, r)  render.rs:188-189 ----
    constexpr unsigned kStride = (unsigned)sizeof(Node<T>);
    T best = inf<T>();
    unsigned best_item = 0;
    // lanes whose ray the loops walk: all of them, or what the cooperative walk of this quad hands back
    bool walk = inside;
    [[maybe_unused]] T cbest = inf<T>();
    [[maybe_unused]] unsigned citem = 0;
    if constexpr (COOP) {
        if (coop_wave) coop_primary(args.cv, coop_lds[wave], coop_rays, dir.x, dir.y, dir.z, inside, cbest, citem, walk);
    }
    const bool loops_run = !COOP || !coop_wave || __ballot(walk) != 0;
    if (loops_run) {
        const float fdx = walk ? dir.x : __builtin_nanf("");        // (a lane without a ray carries a direction no bound lets through)
        if constexpr (FUSED) skip_primary_rot_filt_fused(walk_prim, nb, fdx, dir.y, dir.z, walk ? 0u : nb, best, best_item);
        else skip_primary_rot_filt(walk_prim, nb, fdx, dir.y, dir.z, walk ? 0u : nb, best, best_item);
    }

    // ---- late batch: requested here, where the wave is about to wait for the winner's centre anyway ----
    rt_u32x8 p;
    rt_u32x16 f;
    asm volatile("s_load_dwordx8 %0, %2, 0x40\n\ts_load_dwordx16 %1, %2, 0x60\n\ts_waitcnt lgkmcnt(0)" : "=&s"(p), "=&s"(f) : "s"(kp));
    typedef const Item<T> __attribute__((address_space(4))) *item_ptr;
    typedef const uint32_t __attribute__((address_space(4))) *u32_ptr;
    const item_ptr items = (item_ptr)(((unsigned long long)p[1] << 32) | p[0]);
    const u32_ptr own = (u32_ptr)(((unsigned long long)p[3] << 32) | p[2]);
    const void *walk_shad = (const void *)(((unsigned long long)p[5] << 32) | p[4]);
    const void *exact_shad = (const void *)(((unsigned long long)p[7] << 32) | p[6]);
    if constexpr (FUSED) {
        // a group's own sphere won: the walk recorded the offset behind its BOUND node
        if (best_item != 0u && !(best_item & kNodeItem)) best_item = own[best_item / kStride - 1u];
    }
    best_item &= kNodeIndexMask;
    if constexpr (COOP) {
        if (coop_wave && !walk) { best = cbest; best_item = citem; }
    }

    // ---- shade  render.rs:190-199 ----
    const V3<T> OBJECT = { T(0xae) / T(255.0), T(0x31) / T(255.0), T(0x31) / T(255.0) };
    const V3<T> BACKGROUND = { T(0x22) / T(255.0), T(0x0a) / T(255.0), T(0x0a) / T(255.0) };
    const V3<T> AMBIENT = { BACKGROUND.x * T(0.8), BACKGROUND.y * T(0.8), BACKGROUND.z * T(0.8) };
    const V3<T> sdir = mulf(light, T(-1.0));                        // render.rs:206
    V3<T> g = { T(0.0), T(0.0), T(0.0) };
    T alpha = T(0.0);
    bool need_shadow = false;
    T gdot = T(0.0);
    V3<T> sp = { T(0.0), T(0.0), T(0.0) };
    if (inside) {
        if (best == inf<T>()) g = add(g, BACKGROUND);
        else {
            const V3<T> c = { items[best_item].cx, items[best_item].cy, items[best_item].cz };
            const V3<T> nrm = normalized(add(eye, sub(mulf(dir, best), c)));       // primitive.rs:83
            gdot = dot(nrm, light);
            if (gdot >= T(0.0)) g = add(g, AMBIENT);
            else {
                need_shadow = true;
                const V3<T> ns = mulf(nrm, best * rsqrt_exact(eps<T>()));
                sp = add(add(eye, mulf(dir, best)), ns);
            }
        }
    }

    // ---- shadow ray: any hit  render.rs:202-208 ----
    bool occluded = false;
    bool walk_s = need_shadow;
    if constexpr (COOP) {
        if (coop_wave && __ballot(need_shadow) != 0) coop_shadow(args.cv, coop_lds[wave], coop_rays, sp.x, sp.y, sp.z, sdir, need_shadow, occluded, walk_s);
    }
    if (__ballot(walk_s) != 0) {
        FilterConsts fc;
        fc.m0[0] = __uint_as_float(f[0]); fc.m0[1] = __uint_as_float(f[1]); fc.m0[2] = __uint_as_float(f[2]);
        fc.e1[0] = __uint_as_float(f[3]); fc.e1[1] = __uint_as_float(f[4]); fc.e1[2] = __uint_as_float(f[5]);
        fc.e2[0] = __uint_as_float(f[6]); fc.e2[1] = __uint_as_float(f[7]); fc.e2[2] = __uint_as_float(f[8]);
        fc.l[0] = __uint_as_float(f[9]); fc.l[1] = __uint_as_float(f[10]); fc.l[2] = __uint_as_float(f[11]);
        fc.a0 = __uint_as_float(f[12]); fc.k1 = __uint_as_float(f[13]); fc.kc = __uint_as_float(f[14]); fc.ro2 = __uint_as_float(f[15]);
        float q1, q2, fol;
        shadow_filter_origin(fc, sp.x, sp.y, sp.z, q1, q2, fol);
        if (!walk_s) q1 = inf<float>();                 // no shadow ray: an in-plane origin at infinity is beyond every outer bound but END's
        // (the direction as scalars again: the loops take it as such)
        auto uniform = [](float v) { return __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); };      // (the bits, not the value)
        const float slx = uniform(sdir.x), sly = uniform(sdir.y), slz = uniform(sdir.z);
        unsigned resume = walk_s ? 0u : nb;             // lanes without a shadow ray sleep until END
        unsigned i = 0;
        while (i < nb) {
            unsigned fin;
            if constexpr (FUSED) i = skip_shadow_rot_filt_fused(walk_shad, nb, i, sp.x, sp.y, sp.z, slx, sly, slz, resume, fin, q1, q2, fol, fc.a0, fc.k1, fc.kc, exact_shad);
            else i = skip_shadow_rot_filt(walk_shad, nb, i, sp.x, sp.y, sp.z, slx, sly, slz, resume, fin, q1, q2, fol, fc.a0, fc.k1, fc.kc, exact_shad);
            if (i >= nb) break;
            if (fin) { occluded = true; resume = nb; q1 = inf<float>(); }      // (a retired lane passes no bound any more)
            // some lane retired at the node at i: go straight to the next node any lane still wants (nb: nobody is left)
            i = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_min_u32(resume >= nb ? nb : (resume > i ? resume : i + kStride)));
        }
    }
    if (need_shadow) {
        if (!occluded) {
            g = add(add(g, mulf(OBJECT, -gdot)), AMBIENT);      // render.rs:209
            alpha += T(1.0);
        } else g = add(add(g, BACKGROUND), mulf(AMBIENT, -gdot));      // render.rs:212
    }
    if (inside) *px_ptr = scale_u8(g.x) | (scale_u8(g.y) << 8) | (scale_u8(g.z) << 16) | (scale_u8(alpha) << 24);

    if constexpr (TRACE) {
        if (lane == 0) {
            uint32_t *rec = args.trace + ((size_t)blockIdx.x * 4 + wave) * 8;        // rt_skip.hpp: the same eight words
            rec[0] = (uint32_t)r_start;
            rec[1] = (uint32_t)__builtin_amdgcn_s_memrealtime();
            rec[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | (__builtin_amdgcn_s_getreg((31 << 11) | 20) << 16);
            rec[3] = blockIdx.x | (coop_wave ? 0x80000000u : 0u);
            rec[4] = (uint32_t)r_entry;
            __builtin_amdgcn_s_waitcnt(0);
            rec[5] = (uint32_t)__builtin_amdgcn_s_memrealtime();
        }
    }
}

template <int VAR, bool TRACE>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_num_sgpr(82))) void k_render_skip_fast(FastArgs args)
{
    render_skip_fast_body<VAR, TRACE, false>(args);
}
// ... with cooperative quads: the walk's state in LDS (16 KB per workgroup), more registers (what a CU admits is decided by the counts
// tests/test_kernel_resources.py pins)
template <int VAR, bool TRACE>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_num_sgpr(RT_FAST_COOP_SGPRS))) void k_render_skip_fast_coop(FastArgs args)
{
    render_skip_fast_body<VAR, TRACE, true>(args);
}

}  // namespace rt
