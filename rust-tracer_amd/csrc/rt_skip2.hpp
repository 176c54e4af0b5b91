// rt_skip2.hpp -- k_render_skip2: RT_TRAVERSAL_SKIP for f32 scenes with TWO rays per lane (rt_skip2_rot.hpp).
//
// The walk, the node streams and every arithmetic operation are k_render_skip's (rt_skip.hpp); what changes is who carries the
// rays.  A lane holds two of them in VGPR pairs, so that one packed instruction does one operation of the sphere test for both
// and the scalar bookkeeping of a step serves 128 rays (tools/gen_skip2_asm.py has the measured instruction costs).  A workgroup
// is 128 threads = two waves and renders what four waves of k_render_skip render:
//   kSkipOne    (spp 1): the 16x16 block of a descriptor; a wave takes a 16x8 patch, ray h of a lane is the pixel 8h to the right
//                        of the lane's own (narrowed descriptors: 2pw x pw patches in the first pw*pw lanes, pw = 8 >> level)
//   kSkipPacked (spp 2 / 4 / 8): a wave's lanes enumerate the samples of 4x4 / 2x2 / 1 pixels as in k_render_skip, and ray h
//                        belongs to the same sample of the pixel group ppw pixels to the right; blockIdx.y picks the sub-block
// Launches that count tests, f64 and other spp stay with k_render_skip (there is no C++ flavour of
// these loops; tests/test_gpu_parity.py compares the two kernels' frames byte for byte).
#pragma once
#include "rt_skip.hpp"
#include "rt_skip2_rot.hpp"

namespace rt {

constexpr unsigned kSkip2Rays = 2;
constexpr unsigned kSkip2Threads = kBlockThreads / kSkip2Rays;

// FILT: the shadow walk reads the two-sided bounds of the filtered streams (skip2_shadow_rot_filt_fused) instead of forming the
// reference's sixteen operations at every node; the primary walk is filtered in both flavours.
// Round 4: the loops' operands are bound to the loops' own registers (tools/gen_skip2_asm.py: eight copies and eight registers fewer at
// the statement) -- 64 vector registers, no scratch (round 3 had forced 64 on a 72-register kernel and paid 12 bytes of scratch for it) --
// and the loops' fifty scalar registers start at s24, the kernel held to 74: .sgpr_count 80, which is what a CU admits EIGHT workgroups'
// worth of waves per SIMD at (MI355X_MICROARCH.md "Residency": 82 - 96 admit seven whatever the occupancy remark says); what the kernel
// keeps across the loops beyond s[0:23] is parked in vector-register lanes.  The walk waits for node records like the one-ray walk does:
// config 5 2.93 -> 2.86 (64 vector registers, seven per SIMD) -> 2.83 ms (eight), the 100,000-sphere frame 19.9 -> 18.7 ms.
// FUSED = false: a scene whose bounds have no sphere of their own (the automatic hierarchy of an arbitrary sphere list): the plain filtered
// streams, the plain-stream loops (FILT only).
template <int MODE, bool FILT, bool FUSED = true>
__global__ __launch_bounds__(kSkip2Threads) __attribute__((amdgpu_num_sgpr(82), amdgpu_waves_per_eu(8))) void k_render_skip2(SkipView<float> sc, unsigned width, unsigned height, unsigned spp_arg,
                                                               const TileDev *__restrict__ tiles, unsigned n_tiles, uint8_t *__restrict__ out,
                                                               SampleBuf<float> sb, unsigned frame_w, const BlockDesc *__restrict__ order,
                                                               const uint32_t *__restrict__ wg_first)
{
    typedef float T;
    static_assert(MODE == kSkipOne || MODE == kSkipPacked, "two rays per lane: spp 1 or the sample-packed modes");
    static_assert(FUSED || FILT, "the plain-stream loops exist in the filtered flavour only");
    constexpr bool PACKED = MODE == kSkipPacked, ONE = MODE == kSkipOne;
    constexpr unsigned R = kSkip2Rays;
    const unsigned spp = ONE ? 1u : spp_arg;
    unsigned d_first = blockIdx.x, d_last = blockIdx.x + 1;
    if (order && wg_first) { d_first = wg_first[blockIdx.x]; d_last = wg_first[blockIdx.x + 1]; }
    for (unsigned di = d_first; di < d_last; ++di) {
        unsigned bx0, by0, tile_r, tile_t, pitch, base;
        unsigned level = 0;
        if (order) {
            const BlockDesc bd = load_block_desc(order, di);
            bx0 = bd.x0; by0 = bd.y0; tile_r = bd.r; tile_t = bd.t; pitch = bd.pitch & 0xFFFFu; base = bd.base;
            level = (bd.pitch >> kBlockNarrowShift) & 3u;         // (a cooperative mask in the bits above is ignored: those quads are walked like any other)
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

        const T ssf = T(spp);
        const T fw = T(width), fh = T(height);
        const T half_w = fw / T(2.0), half_h = fh / T(2.0);
        const V3<T> eye = sc.eye, light = sc.light;
        const V3<T> OBJECT = { T(0xae) / T(255.0), T(0x31) / T(255.0), T(0x31) / T(255.0) };
        const V3<T> BACKGROUND = { T(0x22) / T(255.0), T(0x0a) / T(255.0), T(0x0a) / T(255.0) };
        const V3<T> AMBIENT = { BACKGROUND.x * T(0.8), BACKGROUND.y * T(0.8), BACKGROUND.z * T(0.8) };
        const V3<T> sdir = mulf(light, T(-1.0));                        // render.rs:206
        constexpr unsigned kStride = (unsigned)sizeof(Node<T>);
        const unsigned nb = (FUSED ? sc.n_fnodes : sc.n_nodes) * kStride;

        const unsigned ssx = PACKED ? sample / spp : 0u, ssy = PACKED ? sample % spp : 0u;
        const T yres = ONE ? T(y) : T(y) + T(ssy) / ssf;             // render.rs:238-243
        V3<T> dir[R];
        T dx[R], dy[R], dz[R];
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
        if constexpr (FUSED) {
            skip2_primary_rot_fused(sc.xfprim, dx, dy, dz, resume, best, best_item);
            for (unsigned h = 0; h < R; ++h)  // a group's own sphere won: the walk recorded the offset behind its BOUND node
                if (best_item[h] != 0u && !(best_item[h] & kNodeItem)) best_item[h] = sc.xown[best_item[h] / (unsigned)sizeof(FNode) - 1u];
        } else skip2_primary_rot(sc.xprim, dx, dy, dz, resume, best, best_item);

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
                const Item<T> it = sc.items[best_item[h] & kNodeIndexMask];
                const V3<T> c = { it.cx, it.cy, it.cz };
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
            for (unsigned h = 0; h < R; ++h) resume[h] = need_shadow[h] ? 0u : nb;      // rays without a shadow ray sleep until END
            if constexpr (FILT) {
                // one invocation: rays retire inside the loop (resume = nb + 1) and the walk goes on at the next wanted node
                if constexpr (FUSED) skip2_shadow_rot_filt_fused(sc.xfshad, nb, ox, oy, oz, resume, sc.fc, sc.fshad);
                else skip2_shadow_rot_filt(sc.xshad, nb, ox, oy, oz, resume, sc.fc, sc.shad);
                for (unsigned h = 0; h < R; ++h) occluded[h] = resume[h] == nb + 1u;
            } else if constexpr (FUSED) {
                unsigned i = 0;
                while (i < nb) {
                    unsigned fin[R];
                    i = (unsigned)__builtin_amdgcn_readfirstlane((int)skip2_shadow_rot_fused(sc.fshad, nb, i, ox, oy, oz, sdir.x, sdir.y, sdir.z, resume, fin));
                    if (i >= nb) break;
                    unsigned want = nb;
                    for (unsigned h = 0; h < R; ++h) {
                        if (fin[h]) { occluded[h] = true; resume[h] = nb; }
                        const unsigned w = resume[h] >= nb ? nb : (resume[h] > i ? resume[h] : i + kStride);
                        want = w < want ? w : want;
                    }
                    // some ray retired at the node at i: go straight to the next node any ray still wants (nb: nobody is left)
                    i = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_min_u32(want));
                }
            }
        }

        for (unsigned h = 0; h < R; ++h) {
            if (need_shadow[h]) state[h] = occluded[h] ? kShadowed : kLit;
            if (!inside[h]) continue;
            if (PACKED) {
                const size_t px_i = (size_t)(base + y * pitch + x[h]);
                const size_t p = px_i * (spp * spp) + sample;
                reinterpret_cast<uint32_t *>(sb.gdot)[p] = sample_word(state[h], gdot[h]);          // k_resolve_words
            } else {
                // render.rs:233-252 for one sample: 0 + term, the mean over one sample and alpha * 1 are the identity bit for bit
                V3<T> g = { T(0.0), T(0.0), T(0.0) };
                T alpha = T(0.0);
                if (state[h] == kMiss) g = add(g, BACKGROUND);
                else if (state[h] == kAmbient) g = add(g, AMBIENT);
                else if (state[h] == kLit) { g = add(add(g, mulf(OBJECT, -gdot[h])), AMBIENT); alpha += T(1.0); }
                else g = add(add(g, BACKGROUND), mulf(AMBIENT, -gdot[h]));
                const size_t px = frame_w ? (size_t)y * frame_w + x[h] : (size_t)(base + y * pitch + x[h]);
                reinterpret_cast<unsigned *>(out)[px] = scale_u8(g.x) | (scale_u8(g.y) << 8) | (scale_u8(g.z) << 16) | (scale_u8(alpha) << 24);
            }
        }
    }       // descriptors of this workgroup
}

}  // namespace rt
