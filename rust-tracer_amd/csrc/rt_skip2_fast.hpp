// rt_skip2_fast.hpp -- k_render_skip2_fast: k_render_skip2 (two rays per lane, rt_skip2.hpp) with its arguments fetched where they are needed
// (round 6).
//
// k_render_skip2's loops own s[24:73] of a kernel held to 80 scalar registers, so the kernel itself has s[0:23] -- and its eleven arguments
// (a SkipView of thirteen pointers and two vectors among them) arrive in register TUPLES the compiler keeps whole: what is wanted behind a
// loop is parked in vector-register lanes in front of it and fetched back behind it, 155 v_writelane / v_readlane per wave of a launch whose
// waves issue 860 vector instructions each (config 5, tools/profile_sq_config5.sh).  Here the arguments are ONE struct, read in four short
// batches -- entry; behind the primary walk; in front of the shadow walk; in front of the stores --, the loops' scalar operands copied into
// registers of their own: nothing is parked.  Same inline functions, same generated loops, same bytes (the tests hold every frame of the
// two kernels against each other: RT_DEBUG_FAST_KERNEL 0 / 2).  Launches with a dispatch list (an order); the raster search of the first
// launches of a list stays with k_render_skip2.
#pragma once
#include "rt_skip2.hpp"
#include "rt_skip_fast.hpp"

namespace rt {

struct Fast2Args {
    // entry batch: dwords [0, 12)
    const BlockDesc *order;
    const uint32_t *wg_first;         // NULL: one descriptor per workgroup
    unsigned width, height, spp;
    unsigned nb;                      // the walk streams' length in bytes
    const void *walk_prim;            // xfprim (fused) / xprim
    unsigned frame_w, pad0;
    // behind the primary walk: dwords [12, 24)
    const Item<float> *items;
    const uint32_t *own;
    float eye[3], light[3];
    unsigned pad1[2];
    // in front of the shadow walk: dwords [24, 30)
    const void *walk_shad;            // xfshad / xshad
    const void *exact_shad;           // fshad / shad
    const FilterConsts *fc;
    // in front of the stores: dwords [30, 32)
    void *dst;                        // packed: the sample words (SampleBuf::gdot); one sample per pixel: the frame / the tiles
};
static_assert(offsetof(Fast2Args, items) == 48 && offsetof(Fast2Args, eye) == 64 && offsetof(Fast2Args, walk_shad) == 96 && offsetof(Fast2Args, dst) == 120 &&
              sizeof(Fast2Args) == 128, "the batches of k_render_skip2_fast");

template <int MODE, bool FUSED>
__global__ __launch_bounds__(kSkip2Threads) __attribute__((amdgpu_num_sgpr(82), amdgpu_waves_per_eu(8))) void k_render_skip2_fast(Fast2Args args)
{
    typedef float T;
    static_assert(MODE == kSkipOne || MODE == kSkipPacked, "two rays per lane: spp 1 or the sample-packed modes");
    constexpr bool PACKED = MODE == kSkipPacked, ONE = MODE == kSkipOne;
    constexpr unsigned R = kSkip2Rays;
    (void)args;
    const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
    typedef const rt_u32x4 __attribute__((address_space(4))) *desc_ptr;
    typedef const uint32_t __attribute__((address_space(4))) *u32_ptr;
    unsigned d_first = blockIdx.x, d_last = blockIdx.x + 1;
    {
        rt_u32x2 wf;
        asm volatile("s_load_dwordx2 %0, %1, 0x8\n\ts_waitcnt lgkmcnt(0)" : "=s"(wf) : "s"(kp));
        const unsigned long long wbits = ((unsigned long long)wf[1] << 32) | wf[0];
        if (wbits != 0ull) { const u32_ptr wg = (u32_ptr)wbits; d_first = wg[blockIdx.x]; d_last = wg[blockIdx.x + 1]; }
    }
    for (unsigned di = d_first; di < d_last; ++di) {
        // ---- entry batch, then the descriptor ----
        rt_u32x8 q;
        rt_u32x4 q2;
        asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x20\n\ts_waitcnt lgkmcnt(0)" : "=&s"(q), "=&s"(q2) : "s"(kp));
        const desc_ptr order = (desc_ptr)(((unsigned long long)q[1] << 32) | q[0]);
        const unsigned width = q[4], height = q[5], spp = ONE ? 1u : q[6];
        const rt_u32x4 raw = order[di];
        const unsigned bx0 = raw[0] & 0xFFFFu, by0 = raw[0] >> 16, tile_r = raw[1] & 0xFFFFu, tile_t = raw[1] >> 16, pitch = raw[2] & 0xFFFFu, base = raw[3];
        const unsigned level = (raw[2] >> kBlockNarrowShift) & 3u;
        const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        unsigned x[R], y, sample = 0;
        bool inside[R];
        if (PACKED) {
            const unsigned ns = spp * spp, lg = 31u - (unsigned)__builtin_clz(ns);
            const unsigned ppw = 8u >> (lg >> 1), pb = 3u - (lg >> 1);          // pixel group of a ray half: ppw x ppw pixels
            const unsigned pi = lane >> lg;
            sample = lane & (ns - 1u);
            y = by0 + (blockIdx.y / spp) * 2u * ppw + wave * ppw + (pi >> pb);
            for (unsigned h = 0; h < R; ++h) {
                x[h] = bx0 + (blockIdx.y % spp) * 2u * ppw + h * ppw + (pi & (ppw - 1u));
                inside[h] = x[h] < tile_r && y < tile_t;
            }
        } else {
            const unsigned pw = 8u >> level, pbits = 3u - level;
            y = by0 + wave * pw + ((lane >> pbits) & (pw - 1));
            for (unsigned h = 0; h < R; ++h) {
                x[h] = bx0 + h * pw + (lane & (pw - 1));
                inside[h] = x[h] < tile_r && y < tile_t && lane < pw * pw;
            }
            sample = blockIdx.y;
        }
        if (__ballot(inside[0] || inside[1]) == 0) continue;        // waves are independent here: no LDS, no barrier
        (void)pitch; (void)base;      // (wanted by the stores: the descriptor is read again there)

        const T ssf = T(spp);
        const T fw = T(width), fh = T(height);
        const T half_w = fw / T(2.0), half_h = fh / T(2.0);
        const unsigned ssx = PACKED ? sample / spp : 0u, ssy = PACKED ? sample % spp : 0u;
        const T yres = ONE ? T(y) : T(y) + T(ssy) / ssf;             // render.rs:238-243
        V3<T> dir[R];
        T dx[R], dy[R], dz[R];
        // (the walk's scalar operands as registers of their own: a tuple stays allocated as long as one word of it is wanted)
        unsigned long long wp_bits;
        unsigned nb;
        asm volatile("s_mov_b64 %0, %2\n\ts_mov_b32 %1, %3" : "=&s"(wp_bits), "=&s"(nb) : "s"(((unsigned long long)q2[1] << 32) | q2[0]), "s"(q[7]));
        unsigned resume[R];
        for (unsigned h = 0; h < R; ++h) {
            const T xres = ONE ? T(x[h]) : T(x[h]) + T(ssx) / ssf;
            dir[h] = normalized(V3<T>{ xres - half_w, (fh - yres) - half_h, fw });
            dx[h] = dir[h].x; dy[h] = dir[h].y; dz[h] = dir[h].z;
            resume[h] = inside[h] ? 0u : nb;                        // a lane half without a ray sleeps until the END node
        }

        // ---------------- primary rays: s.group.intersect(&mut h, r)  render.rs:188-189 ----------------
        T best[R];
        unsigned best_item[R];
        if constexpr (FUSED) skip2_primary_rot_fused((const void *)(uintptr_t)wp_bits, dx, dy, dz, resume, best, best_item);
        else skip2_primary_rot((const void *)(uintptr_t)wp_bits, dx, dy, dz, resume, best, best_item);

        // ---- behind the primary walk: items, own, eye, light ----
        rt_u32x4 pi4;
        rt_u32x8 pe;
        asm volatile("s_load_dwordx4 %0, %2, 0x30\n\ts_load_dwordx8 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(pi4), "=&s"(pe) : "s"(kp));
        typedef const Item<T> __attribute__((address_space(1))) *item_ptr;
        const unsigned long long items_bits = ((unsigned long long)pi4[1] << 32) | pi4[0];
        if constexpr (FUSED) {
            const u32_ptr own = (u32_ptr)(((unsigned long long)pi4[3] << 32) | pi4[2]);
            for (unsigned h = 0; h < R; ++h)  // a group's own sphere won: the walk recorded the offset behind its BOUND node
                if (best_item[h] != 0u && !(best_item[h] & kNodeItem)) best_item[h] = own[best_item[h] / (unsigned)sizeof(FNode) - 1u];
        }
        const V3<T> eye = { __uint_as_float(pe[0]), __uint_as_float(pe[1]), __uint_as_float(pe[2]) };
        const V3<T> light = { __uint_as_float(pe[3]), __uint_as_float(pe[4]), __uint_as_float(pe[5]) };
        const V3<T> OBJECT = { T(0xae) / T(255.0), T(0x31) / T(255.0), T(0x31) / T(255.0) };
        const V3<T> BACKGROUND = { T(0x22) / T(255.0), T(0x0a) / T(255.0), T(0x0a) / T(255.0) };
        const V3<T> AMBIENT = { BACKGROUND.x * T(0.8), BACKGROUND.y * T(0.8), BACKGROUND.z * T(0.8) };

        // ---------------- shade  render.rs:190-199 ----------------
        bool need_shadow[R];
        T gdot[R], ox[R], oy[R], oz[R];
        uint8_t state[R];
        for (unsigned h = 0; h < R; ++h) {
            need_shadow[h] = false;
            gdot[h] = T(0.0);
            ox[h] = oy[h] = oz[h] = T(0.0);
            state[h] = kMiss;
            if (inside[h] && !(best[h] == inf<T>())) {
                const item_ptr it = (item_ptr)items_bits + (best_item[h] & kNodeIndexMask);
                const V3<T> c = { it->cx, it->cy, it->cz };
                const V3<T> nrm = normalized(add(eye, sub(mulf(dir[h], best[h]), c)));       // primitive.rs:83
                gdot[h] = dot(nrm, light);
                if (gdot[h] >= T(0.0)) {
                    state[h] = kAmbient;
                } else {
                    need_shadow[h] = true;
                    const V3<T> ns = mulf(nrm, best[h] * rsqrt_exact(eps<T>()));
                    const V3<T> sp = add(add(eye, mulf(dir[h], best[h])), ns);
                    ox[h] = sp.x; oy[h] = sp.y; oz[h] = sp.z;
                }
            }
        }

        // ---------------- shadow rays: any hit  render.rs:202-208 ----------------
        bool occluded[R] = { false, false };
        if (__ballot(need_shadow[0] || need_shadow[1]) != 0) {
            rt_u32x4 ps;
            rt_u32x2 pf;
            unsigned nb2;
            asm volatile("s_load_dwordx4 %0, %3, 0x60\n\ts_load_dwordx2 %1, %3, 0x70\n\ts_load_dword %2, %3, 0x1c\n\ts_waitcnt lgkmcnt(0)" : "=&s"(ps), "=&s"(pf), "=&s"(nb2) : "s"(kp));
            const void *walk_shad = (const void *)(((unsigned long long)ps[1] << 32) | ps[0]);
            const void *exact_shad = (const void *)(((unsigned long long)ps[3] << 32) | ps[2]);
            const FilterConsts *fc = (const FilterConsts *)(((unsigned long long)pf[1] << 32) | pf[0]);
            for (unsigned h = 0; h < R; ++h) resume[h] = need_shadow[h] ? 0u : nb2;      // rays without a shadow ray sleep until END
            // one invocation: rays retire inside the loop (resume = nb + 1) and the walk goes on at the next wanted node
            if constexpr (FUSED) skip2_shadow_rot_filt_fused(walk_shad, nb2, ox, oy, oz, resume, fc, exact_shad);
            else skip2_shadow_rot_filt(walk_shad, nb2, ox, oy, oz, resume, fc, exact_shad);
            for (unsigned h = 0; h < R; ++h) occluded[h] = resume[h] == nb2 + 1u;
        }

        // ---- in front of the stores: the destination, and what of the entry batch and the descriptor is wanted again ----
        rt_u32x2 pd, po, pw2;
        unsigned fwv;
        asm volatile("s_load_dwordx2 %0, %4, 0x78\n\ts_load_dwordx2 %1, %4, 0x0\n\ts_load_dwordx2 %2, %4, 0x18\n\ts_load_dword %3, %4, 0x28\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(pd), "=&s"(po), "=&s"(pw2), "=&s"(fwv) : "s"(kp));      // dst; order; spp, nb; frame_w
        const rt_u32x4 raw2 = ((desc_ptr)(((unsigned long long)po[1] << 32) | po[0]))[di];
        const unsigned pitch2 = raw2[2] & 0xFFFFu, base2 = raw2[3];
        const unsigned long long dst_bits = ((unsigned long long)pd[1] << 32) | pd[0];
        typedef unsigned __attribute__((address_space(1))) *word_ptr;
        for (unsigned h = 0; h < R; ++h) {
            if (need_shadow[h]) state[h] = occluded[h] ? kShadowed : kLit;
            if (!inside[h]) continue;
            if (PACKED) {
                const unsigned spp2 = pw2[0];
                const size_t px_i = (size_t)(base2 + y * pitch2 + x[h]);
                const size_t p = px_i * (spp2 * spp2) + sample;
                ((word_ptr)dst_bits)[p] = sample_word(state[h], gdot[h]);          // k_resolve_words
            } else {
                // render.rs:233-252 for one sample: 0 + term, the mean over one sample and alpha * 1 are the identity bit for bit
                V3<T> g = { T(0.0), T(0.0), T(0.0) };
                T alpha = T(0.0);
                if (state[h] == kMiss) g = add(g, BACKGROUND);
                else if (state[h] == kAmbient) g = add(g, AMBIENT);
                else if (state[h] == kLit) { g = add(add(g, mulf(OBJECT, -gdot[h])), AMBIENT); alpha += T(1.0); }
                else g = add(add(g, BACKGROUND), mulf(AMBIENT, -gdot[h]));
                const size_t px = fwv ? (size_t)y * fwv + x[h] : (size_t)(base2 + y * pitch2 + x[h]);
                ((word_ptr)dst_bits)[px] = scale_u8(g.x) | (scale_u8(g.y) << 8) | (scale_u8(g.z) << 16) | (scale_u8(alpha) << 24);
            }
        }
    }       // descriptors of this workgroup
}

}  // namespace rt
