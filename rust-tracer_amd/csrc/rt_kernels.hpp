// rt_kernels.hpp -- gfx950 kernels of the per-pixel ray-sphere path.
//
// k_render_fused: one thread per pixel does everything Renderer::render_region does for that pixel
// (render.rs:218-255): supersample loop (ssx outer, ssy inner), primary-ray generation, nearest-hit scan over
// the item array staged through LDS in chunks, shade, shadow any-hit scan, sequential f32 accumulation,
// f32 -> u8 quantisation.  Written for wave64: a wave owns an 8x8 pixel patch so its rays stay coherent.
#pragma once
#include "rt_math.hpp"

namespace rt {

constexpr int kBlockThreads = 256;     // 4 waves: a 16x16 pixel block, one 8x8 patch per wave
constexpr int kBlockW = 16, kBlockH = 16;

// One entry per requested ImageRegion (render.rs:42-48) plus where its pixels go and which blocks cover it.
struct TileDev {
    uint16_t l, t, r, b;
    uint32_t out_px;       // first pixel of this tile in the tile-major output
    uint32_t blk_first;    // first 16x16 block index of this tile in the launch grid
    uint32_t blks_x;       // blocks per tile row
};

template <typename T> struct SceneView {
    const Item<T> *items;  // DFS order
    uint32_t n_items;
    V3<T> light, eye;      // Scene::directional_light (unit), Scene::eye  render.rs:138-142
};

// Counter slots are striped (kCounterStripes copies, picked by block index, summed on the host): tens of thousands of
// waves adding to ONE address serialise at ~88 atomics/us, which made a stats pass 15x slower than the render itself.
constexpr unsigned kCounterStripes = 256;

struct Counters {          // same meaning as the reference-side ray statistics
    unsigned long long primary, hits, shadow, occluded;
    unsigned long long sphere_tests, bound_tests;   // per-ray tests executed (SKIP traversal counts them exactly)
    unsigned long long wave_steps, max_wave_steps;  // SKIP: node visits summed over waves / of the busiest wave (diagnostic)
    unsigned long long max_wave_cycles, max_wave_ref100mhz;   // diagnostic: s_memtime / s_memrealtime span of the longest wave
};

__device__ __forceinline__ unsigned long long wave_sum(unsigned v)
{
    unsigned long long s = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    return s;
}

template <typename T, int CHUNK>
__global__ __launch_bounds__(kBlockThreads) void k_render_fused(SceneView<T> sc, unsigned width, unsigned height,
                                                               unsigned spp, const TileDev *__restrict__ tiles,
                                                               unsigned n_tiles, uint8_t *__restrict__ out,
                                                               Counters *__restrict__ counters)
{
    __shared__ Item<T> s_items[CHUNK];

    // ---- which tile, which 16x16 block of it (wave-uniform) ----
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

    // ---- render.rs:219-229 ----
    const T ssf = T(spp);
    const T total_recip = T(1.0) / (ssf * ssf);
    const T fw = T(width), fh = T(height);
    const T half_w = fw / T(2.0), half_h = fh / T(2.0);
    const V3<T> eye = sc.eye, light = sc.light;
    const unsigned n = sc.n_items;

    // Renderer::raytrace constants render.rs:172-186
    const V3<T> OBJECT = { T(0xae) / T(255.0), T(0x31) / T(255.0), T(0x31) / T(255.0) };
    const V3<T> BACKGROUND = { T(0x22) / T(255.0), T(0x0a) / T(255.0), T(0x0a) / T(255.0) };
    const V3<T> AMBIENT = { BACKGROUND.x * T(0.8), BACKGROUND.y * T(0.8), BACKGROUND.z * T(0.8) };

    V3<T> g = { T(0.0), T(0.0), T(0.0) };
    T alpha = T(0.0);
    unsigned c_hits = 0, c_shadow = 0, c_occ = 0;

    for (unsigned ssx = 0; ssx < spp; ++ssx) {
        for (unsigned ssy = 0; ssy < spp; ++ssy) {
            // render.rs:238-243
            const T xres = T(x) + T(ssx) / ssf;
            const T yres = T(y) + T(ssy) / ssf;
            V3<T> dir = { xres - half_w, (fh - yres) - half_h, fw };
            dir = normalized(dir);

            // ---- primary ray: nearest hit, strict `<`, first item in DFS order wins ties (primitive.rs:79) ----
            T best = inf<T>();
            unsigned best_i = 0;
            for (unsigned base = 0; base < n; base += CHUNK) {
                const unsigned cnt = min((unsigned)CHUNK, n - base);
                __syncthreads();
                for (unsigned j = threadIdx.x; j < cnt; j += kBlockThreads) s_items[j] = sc.items[base + j];
                __syncthreads();
                if (inside) {
                    for (unsigned j = 0; j < cnt; ++j) {
                        const Item<T> it = s_items[j];
                        const T d = distance_from_ray(it.cx, it.cy, it.cz, it.r, eye, dir);
                        if (!(d >= best)) { best = d; best_i = base + j; }
                    }
                }
            }

            // ---- shade (render.rs:190-199); the normal is a pure function of (ray, distance, centre), so it is
            // computed once for the final nearest item instead of on every improvement (primitive.rs:83) ----
            bool need_shadow = false;
            T gdot = T(0.0);
            V3<T> sp = { T(0.0), T(0.0), T(0.0) };
            if (inside) {
                if (best == inf<T>()) {
                    g = add(g, BACKGROUND);
                } else {
                    ++c_hits;
                    const Item<T> it = sc.items[best_i];
                    const V3<T> c = { it.cx, it.cy, it.cz };
                    const V3<T> nrm = normalized(add(eye, sub(mulf(dir, best), c)));
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

            // ---- shadow ray: any hit (render.rs:202-208).  The block stops scanning once no lane is pending. ----
            const V3<T> sdir = mulf(light, T(-1.0));
            bool pending = need_shadow, occluded = false;
            for (unsigned base = 0; base < n; base += CHUNK) {
                if (!__syncthreads_or(pending ? 1 : 0)) break;
                const unsigned cnt = min((unsigned)CHUNK, n - base);
                for (unsigned j = threadIdx.x; j < cnt; j += kBlockThreads) s_items[j] = sc.items[base + j];
                __syncthreads();
                if (pending) {
                    for (unsigned j = 0; j < cnt; ++j) {
                        const Item<T> it = s_items[j];
                        const T d = distance_from_ray(it.cx, it.cy, it.cz, it.r, sp, sdir);
                        if (!(d >= inf<T>())) { occluded = true; pending = false; break; }
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
        g = mulf(g, total_recip);                                           // render.rs:249-250
        alpha *= total_recip;
        const unsigned tw = tile.r - tile.l;
        const size_t px = (size_t)tile.out_px + (size_t)(y - tile.b) * tw + (x - tile.l);
        const unsigned rgba = scale_u8(g.x) | (scale_u8(g.y) << 8) | (scale_u8(g.z) << 16) | (scale_u8(alpha) << 24);
        reinterpret_cast<unsigned *>(out)[px] = rgba;
    }

    if (counters) {
        counters += blockIdx.x % kCounterStripes;
        const unsigned long long prim = wave_sum(inside ? spp * spp : 0u);
        const unsigned long long hits = wave_sum(c_hits), sh = wave_sum(c_shadow), oc = wave_sum(c_occ);
        if (lane == 0) {
            atomicAdd(&counters->primary, prim);
            atomicAdd(&counters->hits, hits);
            atomicAdd(&counters->shadow, sh);
            atomicAdd(&counters->occluded, oc);
        }
    }
}

// Exhaustive self-test of sqrt_rn_lean against the compiler's IEEE sqrt: every f32 bit pattern in [first, first+count).
__global__ void k_selftest_sqrt(unsigned first, unsigned long long count, unsigned long long *mismatches, unsigned *first_bad)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += stride) {
        const unsigned bits = first + (unsigned)k;
        const float x = __uint_as_float(bits);
        const float a = sqrt_rn_lean(x), b = rsqrt_exact(x);
        const bool same = __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
        if (!same) { ++bad; atomicMin(first_bad, bits); }
    }
    if (bad) atomicAdd(mismatches, bad);
}

// RGBABuffer::set_pixels_from_buffer (render.rs:112-126): tile-major tiles -> row-major frame, 4 B per lane.
__global__ __launch_bounds__(kBlockThreads) void k_blit_tiles(unsigned width, const TileDev *__restrict__ tiles, unsigned n_tiles,
                                                             const unsigned *__restrict__ src, unsigned *__restrict__ frame)
{
    unsigned lo = 0, hi = n_tiles - 1;
    while (lo < hi) {
        unsigned mid = (lo + hi + 1) >> 1;
        if (tiles[mid].blk_first <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const TileDev tile = tiles[lo];
    const unsigned lb = blockIdx.x - tile.blk_first;
    const unsigned bx = lb % tile.blks_x, by = lb / tile.blks_x;
    const unsigned x = tile.l + bx * kBlockW + (threadIdx.x & 15);
    const unsigned y = tile.b + by * kBlockH + (threadIdx.x >> 4);
    if (x < tile.r && y < tile.t)
        frame[(size_t)y * width + x] = src[(size_t)tile.out_px + (size_t)(y - tile.b) * (tile.r - tile.l) + (x - tile.l)];
}

}  // namespace rt
