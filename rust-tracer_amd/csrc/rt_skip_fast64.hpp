// rt_skip_fast64.hpp -- k_render_skip_fast64_coop: rt_skip_fast.hpp's lean kernel for f64 scenes (round 6): the filtered f64 hierarchy walk of a
// single-pass, ordered launch (spp 1, a dispatch list, no counters), with or without cooperative quads, at EIGHT waves per SIMD.
//
// Why it exists.  The lane-cooperative walk (rt_coop.hpp, CNode64) took the small f64 frames from 57 - 63 us to 31, but inside the generic body
// (k_render_skip_f64_coop) it needs 96 vector registers -- five waves per SIMD where the plain f64 kernel runs seven --, so the library's trial
// left every pass that is bound by throughput (1280x720 and up; BASELINE config 3) on the plain kernel.  The generic body keeps the values of
// every mode alive across the loops; this one serves ONE mode with its arguments fetched where they are needed, as the f32 lean kernel does.
// It keeps so little across its loops that s[0:19] are enough for it, so the loops are the LOW-WINDOW copies of the filtered f64 loops
// (tools/gen_skip_asm.py F64F_LO: s[20:73] instead of s[36:89], constants in vector registers): .sgpr_count 80, 61 vector registers, no scratch.
// What made the eighth wave pay was keeping register TUPLES short: LLVM keeps a tuple whole, so a sixteen-word batch of which one word is an
// operand of a loop is parked -- all of it -- in vector-register lanes across that loop.  Hence: the walk's operands are copied into registers
// of their own, `items` / `own` are loaded again behind the primary walk, the shadow walk's arguments where it starts.  Same inline functions,
// same arithmetic, same bytes (the tests hold every f64 spp-1 frame of this kernel and of the generic ones against each other).
#pragma once
#include "rt_skip_fast.hpp"

namespace rt {

struct FastArgs64 {
    // entry batch: dwords [0, 16)
    const BlockDesc *order;
    const FNode *walk_prim;           // the primary walk's f32 filter stream (fused flavour: the compacted one)
    unsigned width, height;
    unsigned nbf;                     // that stream's length in bytes
    unsigned frame_w;                 // 0: tile-major output
    uint8_t *out;
    const Node<double> *exact_prim;   // the exact records behind it
    const Item<double> *items;
    const uint32_t *own;              // the compacted stream's own_item table
    // late batch: dwords [16, 48)
    double eye[3], light[3];
    const FNodeS *walk_shad;
    const Node<double> *exact_shad;
    float fc[16];                     // FilterConsts: m0, e1, e2, l, a0, k1, kc, ro2
    // (wave trace, hooks build)
    uint32_t *trace;
    // cooperative quads: dwords [50, 54) ride with the entry batch
    const uint64_t *holes;
    unsigned n_holes, pad_;
    CoopView cv;
};
static_assert(offsetof(FastArgs64, eye) == 64 && offsetof(FastArgs64, walk_shad) == 112 && offsetof(FastArgs64, fc) == 128 && offsetof(FastArgs64, trace) == 192 &&
              offsetof(FastArgs64, holes) == 200, "the batches of k_render_skip_fast64");

template <int VAR, bool TRACE, bool COOP>
__device__ __forceinline__ void render_skip_fast64_body(const FastArgs64 &args)
{
    typedef double T;
    constexpr bool FUSED = (VAR & 4) != 0;
    [[maybe_unused]] __shared__ CoopLds64 coop_lds[COOP ? kBlockThreads / 64 : 1];
    [[maybe_unused]] unsigned long long r_entry = 0, r_start = 0;
    if constexpr (TRACE) r_entry = __builtin_amdgcn_s_memrealtime();
    // ---- entry batch, then the descriptor ----
    const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
    rt_u32x16 q;
    [[maybe_unused]] rt_u32x4 qc = { 0u, 0u, 0u, 0u };
    if constexpr (COOP) asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0xc8\n\ts_waitcnt lgkmcnt(0)" : "=&s"(q), "=&s"(qc) : "s"(kp));
    else asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(q) : "s"(kp));
    typedef const rt_u32x4 __attribute__((address_space(4))) *desc_ptr;
    const desc_ptr order = (desc_ptr)(((unsigned long long)q[1] << 32) | q[0]);
    const void *walk_prim = (const void *)(((unsigned long long)q[3] << 32) | q[2]);
    const unsigned width = q[4], height = q[5], nbf = q[6], frame_w = q[7];
    const unsigned long long out_bits = ((unsigned long long)q[9] << 32) | q[8];
    const void *exact_prim = (const void *)(((unsigned long long)q[11] << 32) | q[10]);
    typedef const Item<T> __attribute__((address_space(1))) *item_ptr;
    typedef const uint32_t __attribute__((address_space(4))) *u32_ptr;

    const rt_u32x4 raw = order[blockIdx.x];
    const unsigned bx0 = raw[0] & 0xFFFFu, by0 = raw[0] >> 16, tile_r = raw[1] & 0xFFFFu, tile_t = raw[1] >> 16, pitch = raw[2] & 0xFFFFu, base = raw[3];
    const unsigned level = (raw[2] >> kBlockNarrowShift) & 3u;
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned pw = 8u >> level, pbits = 3u - level;
    const unsigned x = bx0 + (wave & 1) * pw + (lane & (pw - 1));
    const unsigned y = by0 + (wave >> 1) * pw + ((lane >> pbits) & (pw - 1));
    bool inside = x < tile_r && y < tile_t && lane < pw * pw;
    [[maybe_unused]] bool coop_wave = false;
    [[maybe_unused]] const unsigned coop_rays = pw * pw;
    if constexpr (COOP) {
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
    constexpr unsigned kFStride = (unsigned)sizeof(FNode);
    T best = inf<T>();
    unsigned best_item = 0;
    bool walk = inside;
    [[maybe_unused]] T cbest = inf<T>();
    [[maybe_unused]] unsigned citem = 0;
    if constexpr (COOP) {
        if (coop_wave) coop_primary<false, double>(args.cv, coop_lds[wave], coop_rays, dir.x, dir.y, dir.z, inside, cbest, citem, walk);
    }
    const bool loops_run = !COOP || !coop_wave || __ballot(walk) != 0;
    // (the walk's three scalar operands as registers of their own: the entry batch is ONE sixteen-register tuple, which stays allocated -- parked
    // in vector-register lanes across the loops, which leave the kernel s[0:19] -- as long as a single word of it is wanted)
    unsigned long long wp_bits, ep_bits;
    unsigned nbf_own;
    asm volatile("s_mov_b64 %0, %3\n\ts_mov_b64 %1, %4\n\ts_mov_b32 %2, %5" : "=&s"(wp_bits), "=&s"(ep_bits), "=&s"(nbf_own)
                 : "s"((unsigned long long)(uintptr_t)walk_prim), "s"((unsigned long long)(uintptr_t)exact_prim), "s"(nbf));
    const void *walk_prim_own = (const void *)(uintptr_t)wp_bits, *exact_prim_own = (const void *)(uintptr_t)ep_bits;
    if (loops_run) {
        // (f64: the loops ask who is awake at the top of a step -- a lane without a ray sleeps until END; its filter sees a NaN as well)
        const float fdx = walk ? (float)dir.x : __builtin_nanf("");
        if constexpr (FUSED) skip_primary_rot_filt_lo_fused(walk_prim_own, nbf_own, dir.x, dir.y, dir.z, walk ? 0u : nbf_own, best, best_item, fdx, (float)dir.y, (float)dir.z, exact_prim_own);
        else skip_primary_rot_filt_lo(walk_prim_own, nbf_own, dir.x, dir.y, dir.z, walk ? 0u : nbf_own, best, best_item, fdx, (float)dir.y, (float)dir.z, exact_prim_own);
    }
    // ---- late batch: eye, light, the shadow walk's pointers; the filter's constants (and what of the entry batch is wanted again) ----
    // (in pieces: a register tuple stays allocated as long as ONE of its words is wanted, and the shadow loops leave the kernel s[0:19] --
    // the two pointers they take are a tuple of their own, everything else is dead or in vector registers by then)
    rt_u32x8 pe;
    rt_u32x4 pl, pi;
    asm volatile("s_load_dwordx8 %0, %3, 0x40\n\ts_load_dwordx4 %1, %3, 0x60\n\ts_load_dwordx4 %2, %3, 0x30\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(pe), "=&s"(pl), "=&s"(pi) : "s"(kp));
    const unsigned long long items_bits = ((unsigned long long)pi[1] << 32) | pi[0];
    const u32_ptr own = (u32_ptr)(((unsigned long long)pi[3] << 32) | pi[2]);
    if (loops_run) {
        if constexpr (FUSED) {
            if (best_item != 0u && !(best_item & kNodeItem)) best_item = own[best_item / kFStride - 1u];
        }
        best_item &= kNodeIndexMask;
    }
    if constexpr (COOP) {
        if (coop_wave && !walk) { best = cbest; best_item = citem; }
    }
    const V3<T> eye = { __hiloint2double((int)pe[1], (int)pe[0]), __hiloint2double((int)pe[3], (int)pe[2]), __hiloint2double((int)pe[5], (int)pe[4]) };
    const V3<T> light = { __hiloint2double((int)pe[7], (int)pe[6]), __hiloint2double((int)pl[1], (int)pl[0]), __hiloint2double((int)pl[3], (int)pl[2]) };

    // ---- shade  render.rs:190-199 ----
    const V3<T> OBJECT = { T(0xae) / T(255.0), T(0x31) / T(255.0), T(0x31) / T(255.0) };
    const V3<T> BACKGROUND = { T(0x22) / T(255.0), T(0x0a) / T(255.0), T(0x0a) / T(255.0) };
    const V3<T> AMBIENT = { BACKGROUND.x * T(0.8), BACKGROUND.y * T(0.8), BACKGROUND.z * T(0.8) };
    V3<T> sdir = mulf(light, T(-1.0));                              // render.rs:206
    V3<T> g = { T(0.0), T(0.0), T(0.0) };
    T alpha = T(0.0);
    bool need_shadow = false;
    T gdot = T(0.0);
    V3<T> sp = { T(0.0), T(0.0), T(0.0) };
    if (inside) {
        if (best == inf<T>()) g = add(g, BACKGROUND);
        else {
            const item_ptr it = (item_ptr)(items_bits) + best_item;
            const V3<T> c = { it->cx, it->cy, it->cz };
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
        if (coop_wave && __ballot(need_shadow) != 0) coop_shadow<false, double>(args.cv, coop_lds[wave], coop_rays, sp.x, sp.y, sp.z, sdir, need_shadow, occluded, walk_s);
    }
    if (__ballot(walk_s) != 0) {
        // (the shadow walk's own arguments, where it starts: the filter's sixteen constants would otherwise wait in vector-register lanes)
        rt_u32x4 ps;
        rt_u32x16 f;
        unsigned nbf2;
        asm volatile("s_load_dwordx4 %0, %3, 0x70\n\ts_load_dwordx16 %1, %3, 0x80\n\ts_load_dword %2, %3, 0x18\n\ts_waitcnt lgkmcnt(0)" : "=&s"(ps), "=&s"(f), "=&s"(nbf2) : "s"(kp));
        const void *walk_shad = (const void *)(((unsigned long long)ps[1] << 32) | ps[0]);
        const void *exact_shad = (const void *)(((unsigned long long)ps[3] << 32) | ps[2]);
        FilterConsts fc;
        fc.m0[0] = __uint_as_float(f[0]); fc.m0[1] = __uint_as_float(f[1]); fc.m0[2] = __uint_as_float(f[2]);
        fc.e1[0] = __uint_as_float(f[3]); fc.e1[1] = __uint_as_float(f[4]); fc.e1[2] = __uint_as_float(f[5]);
        fc.e2[0] = __uint_as_float(f[6]); fc.e2[1] = __uint_as_float(f[7]); fc.e2[2] = __uint_as_float(f[8]);
        fc.l[0] = __uint_as_float(f[9]); fc.l[1] = __uint_as_float(f[10]); fc.l[2] = __uint_as_float(f[11]);
        fc.a0 = __uint_as_float(f[12]); fc.k1 = __uint_as_float(f[13]); fc.kc = __uint_as_float(f[14]); fc.ro2 = __uint_as_float(f[15]);
        float fq1, fq2, fql;
        shadow_filter_origin64(fc, sp.x, sp.y, sp.z, fq1, fq2, fql);
        // (the low-window loops take the direction and the constants from vector registers: copies made HERE, so that no word of the scalar
        // tuples they came from is wanted inside the loop)
        float fa0 = fc.a0, fk1 = fc.k1, fkc = fc.kc;
        asm volatile("" : "+v"(fa0), "+v"(fk1), "+v"(fkc));
        asm volatile("" : "+s"(sdir.x), "+s"(sdir.y), "+s"(sdir.z));
        constexpr unsigned kSStride = (unsigned)sizeof(FNodeS);
        const unsigned nbf = nbf2;
        unsigned resume = walk_s ? 0u : nbf;            // lanes without a shadow ray sleep until END
        unsigned i = 0;
        while (i < nbf) {
            unsigned fin;
            if constexpr (FUSED) i = skip_shadow_rot_filt_lo_fused(walk_shad, nbf, i, sp.x, sp.y, sp.z, sdir.x, sdir.y, sdir.z, resume, fin, fq1, fq2, fql, fa0, fk1, fkc, exact_shad);
            else i = skip_shadow_rot_filt_lo(walk_shad, nbf, i, sp.x, sp.y, sp.z, sdir.x, sdir.y, sdir.z, resume, fin, fq1, fq2, fql, fa0, fk1, fkc, exact_shad);
            if (i >= nbf) break;
            if (fin) { occluded = true; resume = nbf; }
            i = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_min_u32(resume >= nbf ? nbf : (resume > i ? resume : i + kSStride)));
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
            uint32_t *rec = args.trace + ((size_t)blockIdx.x * 4 + wave) * 8;
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

// The loops' low-window copies (tools/gen_skip_asm.py F64F_LO: s[20:73]): .sgpr_count 80 -- EIGHT waves per SIMD, where the generic f64 kernels
// (loops in s[36:89]) run seven.  Every register of the window is declared (nothing is reserved under amdgpu_num_sgpr(82)).
template <int VAR, bool TRACE>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_num_sgpr(82))) void k_render_skip_fast64_coop(FastArgs64 args)
{
    render_skip_fast64_body<VAR, TRACE, true>(args);
}

}  // namespace rt
