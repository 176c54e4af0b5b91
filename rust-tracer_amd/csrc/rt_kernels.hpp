// rt_kernels.hpp -- definitions shared by the gfx950 kernels of the per-pixel ray-sphere path (tile table, ray
// counters, wave reductions) and the small kernels around them: the device blit (set_pixels_from_buffer) and the
// sqrt self-test.  The render kernels live in rt_skip.hpp (hierarchy walk, default) and rt_flat.hpp (flat LDS scan).
// All of them are written for wave64: a wave owns an 8x8 pixel patch so its rays stay coherent.
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

// One 16x16 block of a pass, ready to use: where it is, where its tile ends, and where its pixels go in a tile-major
// output (pixel (x, y) -> base + y * pitch + x, modulo 2^32; a row-major frame uses y * frame_w + x instead).  A pass hands
// the render kernel one descriptor per workgroup IN DISPATCH ORDER (most expensive block first), so a workgroup finds its
// work with one 16-byte scalar load instead of a binary search over the tile table (9 dependent loads for 510 buckets,
// 13 for 8,160 one-block tiles) followed by an index lookup.
struct BlockDesc {
    uint16_t x0, y0, r, t;     // block origin; the tile's right / top edge (exclusive) for clipping
    uint32_t pitch, base;
};
static_assert(sizeof(BlockDesc) == 16, "one s_load_dwordx4");
// Descriptor i of a dispatch list (i wave-uniform) as ONE scalar load.  Left to itself the compiler fetches the two 16-bit clip edges with
// per-lane global_load_ushort -- they are compared with per-lane coordinates -- and every wave then starts with a vector-memory round trip
// on top of the scalar ones: part of the 1.4 us a wave spent before its first ray (round 6, tools/wave_timeline.py "prologue").
__device__ __forceinline__ BlockDesc load_block_desc(const BlockDesc *__restrict__ list, unsigned i)
{
    const uint4 raw = *reinterpret_cast<const uint4 *>(list + i);
    BlockDesc d;
    d.x0 = (uint16_t)(raw.x & 0xFFFFu); d.y0 = (uint16_t)(raw.x >> 16); d.r = (uint16_t)(raw.y & 0xFFFFu); d.t = (uint16_t)(raw.y >> 16);
    d.pitch = raw.z; d.base = raw.w;
    return d;
}
// Narrow blocks (pitch >> 16 = level; a tile is at most 65,535 wide): at level 1 the descriptor covers an 8x8 quadrant of
// a block and each of the workgroup's four waves traces a 4x4 patch with 16 live lanes.  A wave walks the union of its rays'
// nodes and the pass is as long as its longest wave, so the few most expensive blocks are dealt out as four such workgroups
// each: their chains get ~17 % shorter (1080p: 74 -> 66 us with the ~30 heaviest blocks narrowed; narrowing hundreds costs
// throughput).  Level 2 (2x2 patches, sixteen workgroups per block) is used in passes too small to fill the chip.
constexpr uint32_t kBlockNarrowShift = 16;       // (pitch >> 16) & 3: 0 full, 1: 4x4 pixels per wave, 2: 2x2
// A narrow descriptor with a mask in bits 20..23 is COOPERATIVE (rt_coop.hpp): wave w of its workgroup traces its quad cooperatively when
// bit 20 + w is set and has nothing to do otherwise (those pixels belong to the block's ordinary descriptor).
constexpr uint32_t kBlockCoopShift = 20;
// wg_first (optional): n_wg + 1 offsets into d -- workgroup w renders the descriptors [wg_first[w], wg_first[w + 1]).
// holes (optional): descriptors [0, n_holes) are 16x16 blocks some of whose 2x2-pixel quads (one bit each, row-major 8x8) are rendered by
// cooperative descriptors further down the list (the launch then needs the COOP flavour of k_render_skip and its LDS).
// ev0 / ev1 (optional): the library is timing this order against others (rt_capi.hip pick_order): record them around the launch.
struct BlockList { const BlockDesc *d = nullptr; uint32_t n = 0; const uint32_t *wg_first = nullptr; uint32_t n_wg = 0; const uint64_t *holes = nullptr; uint32_t n_holes = 0;
                   hipEvent_t ev0 = nullptr, ev1 = nullptr;
                   // the same list without cooperative quads (for launches that cannot walk them)
                   const BlockDesc *plain_d = nullptr; uint32_t plain_n = 0; const uint32_t *plain_wg_first = nullptr; uint32_t plain_n_wg = 0; };

// Outcome of one sample, stored by the sample-parallel paths and consumed by k_resolve_samples in the reference's
// accumulation order: the four exits of Renderer::raytrace (render.rs:190-213) and n.light where it is needed.
enum SampleState : uint8_t { kMiss = 0, kAmbient = 1, kLit = 2, kShadowed = 3 };

// f32 sample-packed passes of the hierarchy walk ([pixel][sample] order) store ONE word per sample in the gdot array and leave the state
// array alone: a sample needs n.light only when a shadow ray was cast, i.e. when n.light < 0 (render.rs:196-199) -- its sign bit is
// known, so the bit says whether the shadow ray was occluded; the two other exits are NaN payloads (no n.light is a NaN: DESIGN.md 2).
constexpr uint32_t kSampleMiss = 0x7fc00001u, kSampleAmbient = 0x7fc00002u;
__device__ __forceinline__ uint32_t sample_word(uint8_t state, float gdot)
{
    if (state == 0) return kSampleMiss;                 // kMiss
    if (state == 1) return kSampleAmbient;              // kAmbient
    const uint32_t b = __float_as_uint(gdot);           // sign bit set
    return state == 2 ? b : (b & 0x7fffffffu);          // kLit : kShadowed
}

template <typename T> struct SampleBuf {
    T *gdot;             // [spp*spp][n_px]  n.light of the sample (meaningful for kLit / kShadowed)
    uint8_t *state;      // [spp*spp][n_px]
    unsigned n_px;
};

// Counter slots are striped (kCounterStripes copies, picked by block index, summed on the host): tens of thousands of
// waves adding to ONE address serialise at ~88 atomics/us, which made a stats pass 15x slower than the render itself.
constexpr unsigned kCounterStripes = 256;

// Where pixel (x, y) of `tile` goes: tile-major (each bucket its own row-major RGBABuffer, render.rs:69-71) when
// frame_w == 0, or straight into a row-major frame of that width (the result of set_pixels_from_buffer, render.rs:112-126).
__device__ __forceinline__ size_t out_index(const TileDev &tile, unsigned x, unsigned y, unsigned frame_w)
{
    return frame_w ? (size_t)y * frame_w + x : (size_t)tile.out_px + (size_t)(y - tile.b) * (tile.r - tile.l) + (x - tile.l);
}

struct Counters {          // same meaning as the reference-side ray statistics
    unsigned long long primary, hits, shadow, occluded;
    unsigned long long sphere_tests, bound_tests;   // per-ray tests executed (SKIP traversal counts them exactly)
    unsigned long long wave_steps, max_wave_steps;  // SKIP: node visits summed over waves / of the busiest wave (diagnostic)
    unsigned long long max_wave_cycles, max_wave_ref100mhz;   // diagnostic: s_memtime / s_memrealtime span of the longest wave
    unsigned long long wave_item_steps;                       // diagnostic: node visits that were ITEM nodes
    // SKIP, f32, counting launches: per-ray tests the filtered loops' bound lets through / tests with a finite distance that
    // the bound would have ruled out (must be 0: rt_debug_count(RT_DEBUG_COUNT_FILTER_VIOLATIONS))
    unsigned long long filter_pass, filter_violations;
    unsigned long long primary_tests;                         // SKIP: the part of sphere_tests + bound_tests made for primary rays
};

__device__ __forceinline__ unsigned long long wave_sum(unsigned v)
{
    unsigned long long s = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    return s;
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

// The same for rcp_rn_lean against the compiler's IEEE division 1.0f / x.
__global__ void k_selftest_rcp(unsigned first, unsigned long long count, unsigned long long *mismatches, unsigned *first_bad)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += stride) {
        const unsigned bits = first + (unsigned)k;
        const float x = __uint_as_float(bits);
        const float a = rcp_rn_lean(x), b = 1.0f / x;
        bool same = __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
        // normalized<float> inlines both lean sequences behind one range test: the same comparison for the value it multiplies by
        const V3<float> n = normalized(V3<float>{ x, 0.0f, 0.0f });
        const float ref = x * (1.0f / rsqrt_exact(x * x + 0.0f * 0.0f + 0.0f * 0.0f));
        same = same && (__float_as_uint(n.x) == __float_as_uint(ref) || (n.x != n.x && ref != ref));
        if (!same) { ++bad; atomicMin(first_bad, bits); }
    }
    if (bad) atomicAdd(mismatches, bad);
}

// Host data into device memory WITHOUT a copy engine: `src` is the device alias of pinned host memory, read over PCIe by the lanes.  The
// first hipMemcpy* of a process costs 7.5 ms (asynchronous: the copy queue's set-up) and its first blocking one as much again
// (tools/init_probe.hip, round 6); a kernel launch costs 0.4.  What `make image` waits for in rt_scene_create is a few MB, once.
__global__ __launch_bounds__(256) void k_upload_words(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, size_t n_words)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void k_zero_words(uint32_t *__restrict__ dst, size_t n_words)        // (hipMemsetAsync loads the runtime's own kernels first)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) dst[i] = 0u;
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

// samples_per_pixel == 0 (render.rs:219-250: no sample is taken, the sum 0 is multiplied by (0 * 0).recip() = inf, and `NaN as u8` is 0): every
// listed pixel is {0, 0, 0, 0}.  One workgroup per tile, into a row-major frame.
__global__ __launch_bounds__(kBlockThreads) void k_zero_tiles(unsigned width, const TileDev *__restrict__ tiles, unsigned *__restrict__ frame)
{
    const TileDev tile = tiles[blockIdx.x];
    const unsigned tw = tile.r - tile.l, n = tw * (tile.t - tile.b);
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) frame[(size_t)(tile.b + i / tw) * width + (tile.l + i % tw)] = 0u;
}

// The writer's conversion on the device (render.rs:373-401): tile-major RGBA tiles -> their place in a row-major frame in the FILE's pixel
// format -- BPP 3: R, G, B (P6, render.rs:392-396), BPP 1: ((r + g + b) as f32 / 3.0) as u8 (P5, render.rs:399), BPP 4: RGBA (the blit).
// One workgroup per tile, a wave per row (rows w, w + 4, ...), a lane per aligned 32-bit word of the row's bytes: the destination may be
// pinned HOST memory, where 256 contiguous bytes per wave store travel as full PCIe writes; the unaligned head / tail bytes of a row
// segment (odd widths) go out as byte stores.  Bytes outside the listed tiles are not touched.
template <int BPP>
__device__ __forceinline__ unsigned encoded_byte(const unsigned *__restrict__ row_src, unsigned rel)      // byte `rel` of the row segment's encoding
{
    const unsigned px = rel / BPP, rgba = row_src[px];
    if constexpr (BPP == 1) return (unsigned)(unsigned char)(((float)(rgba & 255u) + (float)((rgba >> 8) & 255u) + (float)((rgba >> 16) & 255u)) / 3.0f);
    else return (rgba >> (8u * (rel % BPP))) & 255u;
}

template <int BPP>
__global__ __launch_bounds__(kBlockThreads) void k_encode_tiles(unsigned width, const TileDev *__restrict__ tiles, unsigned n_tiles,
                                                               const unsigned *__restrict__ src, unsigned char *__restrict__ frame)
{
    if (blockIdx.x >= n_tiles) return;
    const TileDev tile = tiles[blockIdx.x];
    const unsigned tw = (unsigned)tile.r - tile.l, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (unsigned y = tile.b + wave; y < tile.t; y += kBlockThreads / 64) {
        const unsigned *row_src = src + (size_t)tile.out_px + (size_t)(y - tile.b) * tw;
        const size_t a = ((size_t)y * width + tile.l) * BPP, e = a + (size_t)tw * BPP;      // this row segment's bytes [a, e) of the frame
        const size_t a4 = (a + 3) & ~(size_t)3, e4 = e & ~(size_t)3;
        if (a4 >= e4) {                                                                    // shorter than an aligned word: bytes
            for (size_t j = a + lane; j < e; j += 64) frame[j] = (unsigned char)encoded_byte<BPP>(row_src, (unsigned)(j - a));
            continue;
        }
        for (size_t j = a4 + 4 * (size_t)lane; j < e4; j += 256) {
            const unsigned rel = (unsigned)(j - a);
            unsigned w;
            if constexpr (BPP == 4) w = row_src[rel >> 2];
            else w = encoded_byte<BPP>(row_src, rel) | encoded_byte<BPP>(row_src, rel + 1) << 8 | encoded_byte<BPP>(row_src, rel + 2) << 16 |
                     encoded_byte<BPP>(row_src, rel + 3) << 24;
            *reinterpret_cast<unsigned *>(frame + j) = w;
        }
        if (lane < a4 - a) frame[a + lane] = (unsigned char)encoded_byte<BPP>(row_src, lane);
        if (lane < e - e4) frame[e4 + lane] = (unsigned char)encoded_byte<BPP>(row_src, (unsigned)(e4 - a) + lane);
    }
}

}  // namespace rt
