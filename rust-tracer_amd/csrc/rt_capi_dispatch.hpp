// rt_capi_dispatch.hpp -- part of rt_capi.hip: how a pass is cut into workgroups -- the cost map, exact per-pixel counts of a list's heaviest
// blocks, dispatch orders (narrow tiers, cooperative quads), their trial at run time, the cache of device tile tables.
// (included by rt_capi.hip where its text used to stand: nothing here is a header of its own)
// spp > 1 runs sample-parallel (one thread per sample + a resolve pass) unless spp*spp exceeds grid.y's limit.
bool use_split(unsigned spp) { return spp > 1 && (unsigned long long)spp * spp <= 65535ull; }
// spp 2 / 4 / 8: the samples of a pixel fill 4 / 16 / 64 lanes of a wave (rt_skip.hpp, kSkipPacked)
bool packed_samples(unsigned spp)
{
    return (spp == 2 || spp == 4 || spp == 8) && knob(RT_DEBUG_PACKED_SAMPLES) != 0;
}

// Two rays per lane (rt_skip2.hpp) unless csrc/rt_debug.h RT_DEBUG_SKIP_RAYS says otherwise.  A wave of 128 rays walks the union of more
// paths and a frame is as long as its heaviest waves, so the second ray pays once there is enough work to be throughput-bound.
// Measured with both kernels' walks behind their conservative bounds (round 3: tools/skip2_sweep.sh, profiles/r03d_skip2_sweep.log;
// one / two rays per lane, us per launch).  21,845 spheres, spp 1: 1920x1080 45.5 / 57.5, 2304x1296 65.6 / 61.7, 2560x1440 77.8 / 71.1,
// 3840x2160 153.9 / 131.6; sample-packed: 1024x768 spp 2 108.5 / 121.2, 640x480 spp 4 147.8 / 152.6, 800x600 spp 4 202.0 / 192.5,
// 1024x768 spp 4 (`make image`) 263.6 / 244.0, 2048x2048 spp 4 952 / 828.  87,381 spheres, spp 1: 2560x1440 92.4 / 115.8, 3200x1800
// 138.5 / 131.7, 3840x2160 181.5 / 153.8; sample-packed: 1280x720 spp 2 163.3 / 175.1, 640x480 spp 4 187.9 / 194.6, 800x600 spp 4
// 247.7 / 242.5, 1920x1080 spp 4 724 / 625, 4096x4096 spp 4 4206 / 3420.  (Round 2, before the bounds: spp 1 from 3.5 M / 6 M pixels,
// sample-packed modes only on the large scene.)
// End of round 4, both kernels at eight waves per SIMD (the one-ray kernel gained more from its eighth than the two-ray kernel: it was the
// one waiting more).  21,845 spheres, spp 1: 1920x1080 41.4 / 58.4, 2560x1440 69.3 / 70.2, 3200x1800 99.7 / 96.9, 3840x2160 133.4 / 127.6;
// sample-packed: 1920x1080 spp 2 158.6 / 165.8, 640x480 spp 4 125.5 / 140.8, 800x600 spp 4 170.9 / 177.9, 1024x768 spp 4 206.7 / 210.9,
// 1920x1080 spp 4 475 / 465, 2048x2048 spp 4 732 / 700.  87,381 spheres, spp 1: 2560x1440 95.1 / 117.7, 3840x2160 163.3 / 146.4;
// sample-packed: 1024x768 spp 4 271.5 / 257.9, 1920x1080 spp 4 591 / 519, 4096x4096 spp 4 3437 / 2841.
bool skip2_by_default(uint64_t total_px, unsigned spp, uint32_t n_nodes)
{
    const bool large_scene = n_nodes >= 65536u;
    if (spp == 1) return total_px >= (large_scene ? 5000000ull : 4000000ull);
    return total_px * spp * spp >= (large_scene ? 6000000ull : 20000000ull);
}

constexpr size_t kMaxCachedTables = 32;

// Device copy of `tab`: from the scene's cache when seen before (or cacheable now), else through the context.
rt_status device_table(rt_scene *s, Context *c, const std::vector<rt::TileDev> &tab, hipStream_t stream, const rt::TileDev **out,
                       int slot = 0, const rt_options *o = nullptr, rt::BlockList *order_out = nullptr, bool cacheable = true, bool will_be_timed = false);

// Copies the tile table through the context's pinned buffer; truly asynchronous on `stream`.  A table of one or two tiles
// is read by the kernel straight from the pinned copy instead (one PCIe read per workgroup beats a copy operation on the stream).
constexpr size_t kZeroCopyTableTiles = 2;

rt_status upload_tiles(Context *c, const std::vector<rt::TileDev> &tab, hipStream_t stream, int slot, const rt::TileDev **out)
{
    const size_t tab_bytes = tab.size() * sizeof(rt::TileDev);
    // A second table through the same slot within one lease (the batches of rt_render_tiles_stream / rt_render_frame_stream once the scene's
    // table cache is full): the copy queued for the previous batch may not have read the pinned staging yet, and that batch's kernels --
    // on either of the context's streams -- may still be reading the device copy.  Rare and slow on purpose: wait for all of it.
    if (c->tiles_live[slot]) {
        HIP_TRY(hipStreamSynchronize(stream));
        if (c->stream && c->stream != stream) HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->stream2 && c->stream2 != stream) HIP_TRY(hipStreamSynchronize(c->stream2));
    }
    c->tiles_live[slot] = true;
    if (c->tiles_cap[slot] < tab.size()) {
        if (c->d_tiles[slot]) HIP_TRY(hipFree(c->d_tiles[slot]));
        if (c->h_tiles[slot]) HIP_TRY(hipHostFree(c->h_tiles[slot]));
        c->d_tiles[slot] = nullptr; c->h_tiles[slot] = nullptr; c->tiles_cap[slot] = 0;
        HIP_TRY(hipMalloc(&c->d_tiles[slot], tab_bytes));
        HIP_TRY(hipHostMalloc(&c->h_tiles[slot], tab_bytes, hipHostMallocDefault));
        c->tiles_cap[slot] = tab.size();
    }
    memcpy(c->h_tiles[slot], tab.data(), tab_bytes);
    if (tab.size() <= kZeroCopyTableTiles) {
        void *alias = nullptr;
        if (hipHostGetDevicePointer(&alias, c->h_tiles[slot], 0) == hipSuccess) { *out = static_cast<const rt::TileDev *>(alias); return RT_OK; }
        (void)hipGetLastError();
    }
    HIP_TRY(hipMemcpyAsync(c->d_tiles[slot], c->h_tiles[slot], tab_bytes, hipMemcpyHostToDevice, stream));
    *out = c->d_tiles[slot];
    return RT_OK;
}

constexpr unsigned kCostRes = 256;
// the cost arena's tile-table area: one tile for the map itself; up to kExactBlocks 16x16 blocks when a tile list's heaviest blocks are
// counted again at the frame's own resolution (exact_block_costs)
constexpr unsigned kExactBlocks = 1024;             // 262,144 pixels: an eighth of a 1080p frame, half of an 800x600 one
constexpr size_t kCostArenaPx = (size_t)kExactBlocks * rt::kBlockW * rt::kBlockH > (size_t)kCostRes * kCostRes ? (size_t)kExactBlocks * rt::kBlockW * rt::kBlockH : (size_t)kCostRes * kCostRes;     // pixels each of the arena's areas holds
constexpr size_t kCostTileBytes = (kExactBlocks * sizeof(rt::TileDev) + 255) & ~(size_t)255;
constexpr size_t kTableStageBytes = 256 * 1024;       // pinned staging for the tile tables of new lists (a 1080p list of 64x64 buckets: 10 KB)

// The scene's cost map: one counting render of a kCostRes^2 image (same camera: x spans the same field of view at every
// width), each lane storing the number of tests its pixel took.
// The pinned host side of the cost map: [map: kCostRes^2 words | tile-table area | staging for the tile tables of new lists].  Made by
// rt_scene_create (0.15 ms); the device side and the counting render wait until a tile list wants dispatch orders (start_cost_map): a
// process that renders ONE frame (`make image`) never pays for them.
rt_status alloc_cost_host(rt_scene *s)
{
    constexpr size_t kPx = kCostArenaPx * 4;
    HIP_TRY(hipHostMalloc(&s->h_cost, kPx + kCostTileBytes + kTableStageBytes, hipHostMallocDefault));
    s->h_tab_stage = static_cast<char *>(s->h_cost) + kPx + kCostTileBytes;
    return RT_OK;
}

// Enqueues the counting render of the cost map on the scene's own stream (by whoever first asks for the map: cost_map_of, normally the
// scene's worker thread).  No copy engine: the one tile is read from pinned memory, the lanes store their counts into the pinned map.
template <typename T>
rt_status start_cost_map(rt_scene *s)
{
    constexpr unsigned R = kCostRes;
    const rt::TileDev tile{ 0, (uint16_t)R, (uint16_t)R, 0, 0u, 0u, R / rt::kBlockW };
    if (!s->h_cost) return RT_ERR_OUT_OF_MEMORY;
    // ONE device allocation, kept until the scene goes (hipMalloc / hipFree wait for a busy device): tile | frame | costs | counters
    constexpr size_t kTileBytes = kCostTileBytes, kPx = kCostArenaPx * 4, kCnt = sizeof(rt::Counters) * rt::kCounterStripes;
    HIP_TRY(hipMalloc(&s->d_cost_arena, kTileBytes + 2 * kPx + kCnt));
    hipStream_t stream = s->cost_stream;
    char *base = static_cast<char *>(s->d_cost_arena);
    uint8_t *d_out = reinterpret_cast<uint8_t *>(base + kTileBytes);
    rt::Counters *d_cnt = reinterpret_cast<rt::Counters *>(base + kTileBytes + 2 * kPx);
    memcpy(static_cast<char *>(s->h_cost) + kPx, &tile, sizeof tile);
    void *h_alias = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&h_alias, s->h_cost, 0));
    const rt::TileDev *tile_alias = reinterpret_cast<const rt::TileDev *>(static_cast<char *>(h_alias) + kPx);
    hipLaunchKernelGGL(rt::k_zero_words, dim3(64), dim3(256), 0, stream, reinterpret_cast<uint32_t *>(d_cnt), kCnt / 4);
    rt::SampleBuf<T> sb{ nullptr, nullptr, R * R };
    hipLaunchKernelGGL((rt::k_render_skip<T, true, 1, rt::kSkipLoop>), dim3((R / rt::kBlockW) * (R / rt::kBlockH)), dim3(rt::kBlockThreads), 0, stream,
                       skip_args<T>(s, nullptr, nullptr, R, R, 0u, d_out, tile_alias, 1u, 1u, d_cnt, static_cast<uint32_t *>(h_alias), sb));
    HIP_TRY(hipGetLastError());
    s->cost_started = true;
    return RT_OK;
}

// NULL when the scene has no hierarchy (or the map could not be made: ordering is an optimisation, never an error).
const std::vector<uint32_t> *cost_map_of(rt_scene *s)
{
    std::call_once(s->cost_once, [s] {
        if (s->n_nodes == 0) return;
        if ((s->precision == RT_F32 ? start_cost_map<float>(s) : start_cost_map<double>(s)) != RT_OK) { s->cost_started = false; (void)hipGetLastError(); return; }
        if (hipStreamSynchronize(s->cost_stream) != hipSuccess) { (void)hipGetLastError(); return; }
        const uint32_t *h = static_cast<const uint32_t *>(s->h_cost);
        s->cost_map.assign(h, h + (size_t)kCostRes * kCostRes);
    });
    return s->cost_map.empty() ? nullptr : &s->cost_map;
}

// Tests per pixel at the FRAME's resolution for a few blocks of a tile list (round 6).  The scene's cost map has one cell per 7.5 pixels of
// a 1080p frame, and the rays that meet several hundred nodes follow silhouettes thinner than that: a threshold on the map picks some of a
// heavy pixel's neighbours and misses the pixel, and the wave that keeps it is as long as ever (tools/wave_timeline.py, the cooperative walk
// at 1080p: the quads walked in 14 us, the frame's longest wave still 41).  So the blocks the map ranks highest are counted again, exactly:
// one counting launch over those blocks alone (<= 1,024 blocks = 262,144 pixels, ~0.1 ms of device time, once per tile list, on the scene's own
// stream), each lane storing the number of tests its pixel took.  px[i * 256 + (y - y0) * 16 + (x - x0)] for block i of `blocks`.
struct ExactCosts { std::vector<uint32_t> block; std::vector<uint32_t> px; uint32_t top = 0; };      // block: raster index of the counted blocks
template <typename T>
rt_status exact_block_costs(rt_scene *s, const std::vector<rt::BlockDesc> &raster, const std::vector<uint32_t> &blocks, unsigned w, unsigned h, ExactCosts &out)
{
    out = ExactCosts{};
    if (blocks.empty() || blocks.size() > kExactBlocks || !s->d_cost_arena || !s->h_cost) return RT_OK;
    std::lock_guard<std::mutex> lk(s->exact_mu);                 // one counting launch at a time through the scene's arena
    constexpr size_t kPx = kCostArenaPx * 4, kCnt = sizeof(rt::Counters) * rt::kCounterStripes;
    char *base = static_cast<char *>(s->d_cost_arena);
    uint8_t *d_out = reinterpret_cast<uint8_t *>(base + kCostTileBytes);
    rt::Counters *d_cnt = reinterpret_cast<rt::Counters *>(base + kCostTileBytes + 2 * kPx);
    // (like the map itself: the tile table is read from the pinned arena, the counts are stored into it -- the map was copied out of it
    // when it was collected)
    void *h_alias = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&h_alias, s->h_cost, 0));
    rt::TileDev *h_tiles = reinterpret_cast<rt::TileDev *>(static_cast<char *>(s->h_cost) + kPx);
    const rt::TileDev *tile_alias = reinterpret_cast<const rt::TileDev *>(static_cast<char *>(h_alias) + kPx);
    uint32_t *h_counts = static_cast<uint32_t *>(s->h_cost);
    std::vector<rt::TileDev> tiles(blocks.size());
    for (size_t i = 0; i < blocks.size(); ++i) {
        const rt::BlockDesc &d = raster[blocks[i]];
        // a 16 x 16 tile of its own, clipped like the block; 256 pixels of the tile-major output each
        tiles[i] = rt::TileDev{ d.x0, (uint16_t)std::min<unsigned>(d.y0 + rt::kBlockH, d.t), (uint16_t)std::min<unsigned>(d.x0 + rt::kBlockW, d.r), d.y0,
                                (uint32_t)(i * rt::kBlockW * rt::kBlockH), (uint32_t)i, 1u };
    }
    hipStream_t stream = s->cost_stream;
    memcpy(h_tiles, tiles.data(), tiles.size() * sizeof(rt::TileDev));
    memset(h_counts, 0, blocks.size() * rt::kBlockW * rt::kBlockH * 4);           // (pixels outside a clipped block are not stored)
    hipLaunchKernelGGL(rt::k_zero_words, dim3(64), dim3(256), 0, stream, reinterpret_cast<uint32_t *>(d_cnt), kCnt / 4);
    rt::SampleBuf<T> sb{ nullptr, nullptr, (unsigned)(blocks.size() * rt::kBlockW * rt::kBlockH) };
    hipLaunchKernelGGL((rt::k_render_skip<T, true, 1, rt::kSkipLoop>), dim3((unsigned)blocks.size()), dim3(rt::kBlockThreads), 0, stream,
                       skip_args<T>(s, nullptr, nullptr, w, h, 0u, d_out, tile_alias, (unsigned)tiles.size(), 1u, d_cnt, static_cast<uint32_t *>(h_alias), sb));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    const std::vector<uint32_t> raw(h_counts, h_counts + blocks.size() * rt::kBlockW * rt::kBlockH);
    // the kernel stores tile-major with the tile's own pitch (its clipped width): re-pitch to 16
    out.block = blocks;
    out.px.assign(raw.size(), 0u);
    for (size_t i = 0; i < blocks.size(); ++i) {
        const unsigned tw = (unsigned)tiles[i].r - tiles[i].l, th = (unsigned)tiles[i].t - tiles[i].b;
        for (unsigned y = 0; y < th; ++y)
            for (unsigned x = 0; x < tw; ++x) {
                const uint32_t v = raw[i * 256 + (size_t)y * tw + x];
                out.px[i * 256 + y * 16 + x] = v;
                out.top = std::max(out.top, v);
            }
    }
    return RT_OK;
}

// Dispatch order of a pass's 16x16 blocks: descending estimated cost (the largest cost-map value under the block),
// ties in grid order.  The frame is as long as its last wave's chain of dependent node steps and the chains differ by
// more than 10x across the image, so the long ones have to start first (measured at 1080p: 141 -> 115 us for the
// same one-block tiles in raster vs. descending order).
constexpr uint64_t kFixedBlockCost = 8;      // what a block costs besides its tests (ray set-up, store), in units of the cost map
constexpr size_t kNarrowMax = 64;
constexpr uint64_t kNarrowPercent = 60;
constexpr size_t kNarrowPassBlocks = 16384;
constexpr size_t kNarrowLevel2Blocks = 4096;
// the cooperative walk (rt_coop.hpp): blocks whose estimate reaches kCoopPercent of the pass's largest (and kCoopMinCost tests), at most
// 1 / kCoopMaxShare of the pass
constexpr uint64_t kCoopPercent = 40, kCoopMinCost = 96;
constexpr size_t kCoopMaxShare = 8, kCoopPassBlocks = 4096;
constexpr unsigned kCoopLevel = 2;
constexpr unsigned kCoopRestLevel = 0;      // what is left of a block with holes: 0 = one descriptor (8x8 pixels per wave), 1 = four (4x4 per wave)

// `passes`: how many times the render kernel walks the list in one launch (one per sample in the sample-parallel path).
// Workgroups a launch keeps resident at once: 8 waves per SIMD, 4 waves per workgroup, 256 CUs.
constexpr size_t kResidentWorkgroups = 2048;

// coop (optional): the scene's cooperative copy; holes (with coop): one 64-bit word per descriptor [0, holes->size()) of the list -- the
// 2x2-pixel quads of that 16x16 block (bit (y >> 1) * 8 + (x >> 1)) which cooperative descriptors further down the list render instead.
void block_order(const std::vector<uint32_t> *map, const std::vector<rt::TileDev> &tab, unsigned w, unsigned h, unsigned passes,
                 std::vector<rt::BlockDesc> &descs, std::vector<uint32_t> &wg_first, const rt::CoopView *coop = nullptr, std::vector<uint64_t> *holes = nullptr,
                 int coop_percent = -1,            // 0: no cooperative quads; > 0: from that share of the pass's largest estimate; -1: as rt_debug.h says (default share)
                 const ExactCosts *exact = nullptr, uint32_t exact_thr = 0,      // cooperative quads by EXACT tests per pixel (exact_block_costs) from exact_thr on, instead
                 std::vector<uint32_t> *heaviest = nullptr)                      // out: the raster indices of the heaviest blocks (what exact_block_costs is asked for); descs is not made
{
    wg_first.clear();
    if (holes) holes->clear();
    constexpr int R = (int)kCostRes;
    std::vector<uint32_t> cost;
    std::vector<rt::BlockDesc> raster;
    auto map_col = [&](unsigned x) { return std::clamp((int)((uint64_t)x * R / w), 0, R - 1); };
    auto map_row = [&](unsigned y) { return std::clamp((int)std::floor(((double)y - h / 2.0) * R / w + R / 2.0), 0, R - 1); };
    // the largest map value under the pixels [x0, x1] x [y0, y1], grown by `grow` cells on every side
    auto map_max = [&](unsigned x0, unsigned y0, unsigned x1, unsigned y1, int grow) {
        uint32_t m = 0;
        for (int Y = std::max(0, map_row(y0) - grow); Y <= std::min(R - 1, map_row(y1) + grow); ++Y)
            for (int X = std::max(0, map_col(x0) - grow); X <= std::min(R - 1, map_col(x1) + grow); ++X) m = std::max(m, (*map)[(size_t)Y * R + X]);
        return m;
    };
    for (const rt::TileDev &t : tab) {
        const unsigned bys = ((unsigned)(t.t - t.b) + rt::kBlockH - 1) / rt::kBlockH;
        const uint32_t pitch = (uint32_t)t.r - t.l;
        for (unsigned by = 0; by < bys; ++by)
            for (unsigned bx = 0; bx < t.blks_x; ++bx) {
                const unsigned x0 = t.l + bx * rt::kBlockW, y0 = t.b + by * rt::kBlockH;
                raster.push_back(rt::BlockDesc{ (uint16_t)x0, (uint16_t)y0, t.r, t.t, pitch, t.out_px - t.b * pitch - t.l });
                uint32_t m = 0;
                if (map) m = map_max(x0, y0, std::min<unsigned>(x0 + rt::kBlockW, t.r) - 1, std::min<unsigned>(y0 + rt::kBlockH, t.t) - 1, 0);
                cost.push_back(m);
            }
    }
    std::vector<uint32_t> order(cost.size());
    for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&cost](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
    if (heaviest) {
        heaviest->assign(order.begin(), order.begin() + std::min<size_t>(order.size(), kExactBlocks));
        while (!heaviest->empty() && cost[heaviest->back()] == 0) heaviest->pop_back();        // (nothing to count where the map sees nothing)
        return;
    }
    // The most expensive blocks go out as four narrow workgroups each (rt_kernels.hpp, kBlockNarrow): those whose cost
    // estimate is at least kNarrowPercent of the pass's maximum, at most kNarrowMax and 1/128 of the pass (rt_debug.h can
    // override the cap for A/B runs -- the table is built once per tile list, when it is first seen).
    size_t n_narrow = 0;
    if (knob(RT_DEBUG_PRINT_COSTS) > 0 && !cost.empty()) {
        fprintf(stderr, "[rtrace_hip] block costs, descending:");
        for (size_t i = 0; i < order.size(); i = i < 64 ? i + 4 : i * 2) fprintf(stderr, " #%zu=%u", i, cost[order[i]]);
        fprintf(stderr, "\n");
    }
    // The quads whose rays meet the most nodes are walked COOPERATIVELY (rt_coop.hpp; single-pass f32 launches of scenes that have the
    // cooperative copy, blocks not dealt to workgroups on the host).  A block that holds such quads -- 2x2 pixels whose estimate (the map
    // cells under them) reaches the threshold -- goes out as its
    // ordinary descriptor plus a 64-bit word of HOLES, the quads its own waves leave out, and one cooperative descriptor per 4x4-pixel
    // region that has holes (level 2: a 2x2 quad = 4 rays per wave; rt_debug.h can ask for 16 rays or one): the waves of those trace
    // the holes and nothing else.  RT_DEBUG_COOP: 0 never, 2 every quad of every block (tests).
    size_t n_coop = 0;
    uint64_t coop_thr = 0;
    const long long coop_knob = knob(RT_DEBUG_COOP);
    const bool coop_all = coop_knob == 2;
    if (coop && holes && coop->fanout != 0u && passes == 1 && knob(RT_DEBUG_WG_POLICY) <= 0 && coop_knob != 0 && coop_percent != 0 && !cost.empty() && (map || coop_all)) {
        if (coop_all) n_coop = order.size();
        else {
            const uint64_t top = cost[order[0]];
            const long long t = knob(RT_DEBUG_COOP_THR), m = knob(RT_DEBUG_COOP_MAX);
            coop_thr = t >= 0 ? (uint64_t)t : std::max<uint64_t>(kCoopMinCost, top * (uint64_t)(coop_percent > 0 ? coop_percent : (int)kCoopPercent) / 100);
            // (a pass of more blocks is throughput-bound from its first to its last wave -- 1080p: DESIGN.md 4.1 -- and a cooperative test costs
            // four to five times the vector instructions of a test of the skip-pointer walk: it only pays where waves wait for a few chains)
            const size_t cap = m >= 0 ? (size_t)m : order.size() > kCoopPassBlocks ? 0 : order.size() / kCoopMaxShare;
            while (n_coop < order.size() && n_coop < cap && cost[order[n_coop]] >= coop_thr) ++n_coop;
        }
    }
    // ... or by EXACT tests per pixel (exact_block_costs: the counted blocks, this list's heaviest by the map): a 2x2-pixel quad is walked
    // cooperatively when one of its pixels took exact_thr tests or more.  The blocks that hold such quads move to the front of the order
    // (the holes of a pass are indexed by descriptor position); what is left of them goes out as 4x4-pixel quarters, like the narrow tier.
    std::vector<uint64_t> exact_hole;                    // of order[0 .. n_coop)
    const bool use_exact = exact && exact_thr > 0 && !exact->block.empty() && coop && holes && coop->fanout != 0u && passes == 1 && knob(RT_DEBUG_WG_POLICY) <= 0 &&
                           coop_knob != 0 && !coop_all && !cost.empty();
    const uint32_t top_estimate = cost.empty() ? 0u : cost[order[0]];
    if (use_exact) {
        std::vector<uint64_t> hole_of(cost.size(), 0ull);
        for (size_t k = 0; k < exact->block.size(); ++k) {
            const uint32_t *px = &exact->px[k * 256];
            uint64_t hole = 0;
            for (unsigned qy = 0; qy < 8; ++qy)
                for (unsigned qx = 0; qx < 8; ++qx) {
                    const uint32_t m = std::max(std::max(px[(2 * qy) * 16 + 2 * qx], px[(2 * qy) * 16 + 2 * qx + 1]), std::max(px[(2 * qy + 1) * 16 + 2 * qx], px[(2 * qy + 1) * 16 + 2 * qx + 1]));
                    if (m >= exact_thr) hole |= 1ull << (qy * 8u + qx);
                }
            hole_of[exact->block[k]] = hole;
        }
        std::stable_partition(order.begin(), order.end(), [&hole_of](uint32_t b) { return hole_of[b] != 0ull; });
        n_coop = 0;
        while (n_coop < order.size() && hole_of[order[n_coop]] != 0ull) exact_hole.push_back(hole_of[order[n_coop++]]);
        coop_thr = exact_thr;
    }
    if (map && !cost.empty()) {
        const long long e = knob(RT_DEBUG_NARROW_MAX);
        // a pass of more blocks than kNarrowPassBlocks is throughput-bound: narrowing only adds work there (3840x2160 + 2 %)
        // (and only in single-pass launches: the packed sample-parallel mapping has its own, finer ray packets)
        // (behind a cooperative tier the next blocks are narrowed more generously: tools/coop_sweep.py, 800x600 28.4 -> 26.4 us)
        size_t cap = passes > 1 ? 0 : e >= 0 ? (size_t)e : order.size() > kNarrowPassBlocks ? 0 : std::min<size_t>(kNarrowMax, order.size() / (n_coop && !coop_all && !use_exact ? 32 : 128));
        // (exact holes: those blocks are narrow already.  A pass that fills the chip keeps the tier as large as without them; a small one -- over
        // when its first waves are -- narrows generously behind them: what it waits for next are 8x8 waves whose sixty-four moderate rays walk a
        // long UNION of paths, 800x600: 24 us for pixels of fewer than 150 tests each)
        const bool small_exact = use_exact && order.size() <= kCoopPassBlocks;
        if (use_exact && e < 0) cap = small_exact ? std::min<size_t>(4 * kNarrowMax, order.size() / 8) : cap > n_coop ? cap - n_coop : 0;
        // (with the heaviest blocks walked cooperatively, "expensive" is measured against the cooperative threshold)
        const uint64_t top = use_exact && !small_exact ? top_estimate : n_coop && !coop_all ? coop_thr : cost[order[0]];
        while (n_coop + n_narrow < order.size() && n_narrow < cap && cost[order[n_coop + n_narrow]] > 0 &&
               (uint64_t)cost[order[n_coop + n_narrow]] * 100 >= top * kNarrowPercent)
            ++n_narrow;
    }
    descs.clear();
    descs.reserve(order.size() + 15 * n_narrow + 63 * n_coop);
    std::vector<uint32_t> dcost;                          // cost estimate of every descriptor, descending
    dcost.reserve(order.size() + 15 * n_narrow + 63 * n_coop);
    if (n_coop) {
        // level 1: workgroups of 8x8 pixels (a 4x4 quad = 16 rays per wave); 2: 4x4 (2x2 = 4 rays per wave); 3: 2x2 (one ray per wave)
        const long long lk = knob(RT_DEBUG_COOP_LEVEL);
        const unsigned level = lk >= 1 && lk <= 3 ? (unsigned)lk : kCoopLevel, step = 16u >> level, cnt = 1u << level, quad = step / 2u;
        const unsigned grain = std::max(2u, quad);           // a hole is decided for `grain` x `grain` pixels at once: whole cooperative quads
        // the chains follow silhouettes thinner than a map cell: where cells are small (a few pixels) their neighbours count too
        const int grow = (w + R - 1) / R <= 4 ? 1 : 0;
        const unsigned rest_level = knob(RT_DEBUG_COOP_REST) >= 0 ? (unsigned)std::min(1ll, knob(RT_DEBUG_COOP_REST)) : use_exact ? 1u : kCoopRestLevel;
        std::vector<rt::BlockDesc> cdescs;
        std::vector<uint32_t> ccost;
        for (size_t i = 0; i < n_coop; ++i) {
            const rt::BlockDesc &d = raster[order[i]];
            uint64_t hole = use_exact ? exact_hole[i] : 0ull;
            for (unsigned gy = 0; !use_exact && gy < 16u; gy += grain)
                for (unsigned gx = 0; gx < 16u; gx += grain) {
                    const unsigned px0 = d.x0 + gx, py0 = d.y0 + gy;
                    if (!(px0 < d.r && py0 < d.t)) continue;                    // outside a clipped edge tile
                    if (!(coop_all || map_max(px0, py0, std::min<unsigned>(px0 + grain, d.r) - 1, std::min<unsigned>(py0 + grain, d.t) - 1, grow) >= coop_thr)) continue;
                    for (unsigned sy = 0; sy < grain; sy += 2)
                        for (unsigned sx = 0; sx < grain; sx += 2) hole |= 1ull << (((gy + sy) >> 1) * 8u + ((gx + sx) >> 1));
                }
            if (rest_level == 0u || !hole) { descs.push_back(d); dcost.push_back(cost[order[i]]); holes->push_back(hole); }
            else
                for (unsigned qy = 0; qy < 2; ++qy)             // what is left of the block as four quarters (a 4x4 patch per wave), each with its 4x4 holes
                    for (unsigned qx = 0; qx < 2; ++qx) {
                        rt::BlockDesc n = d;
                        n.x0 = (uint16_t)(d.x0 + qx * 8u); n.y0 = (uint16_t)(d.y0 + qy * 8u);
                        if (!(n.x0 < d.r && n.y0 < d.t)) continue;
                        uint64_t sub = 0;
                        for (unsigned sy = 0; sy < 4; ++sy)
                            for (unsigned sx = 0; sx < 4; ++sx)
                                if ((hole >> ((qy * 4u + sy) * 8u + qx * 4u + sx)) & 1ull) sub |= 1ull << (sy * 4u + sx);
                        if (sub == 0xFFFFull) continue;          // nothing left of this quarter
                        n.pitch |= 1u << rt::kBlockNarrowShift;
                        descs.push_back(n); dcost.push_back(cost[order[i]]); holes->push_back(sub);
                    }
            if (!hole) continue;
            for (unsigned qy = 0; qy < cnt; ++qy)
                for (unsigned qx = 0; qx < cnt; ++qx) {
                    rt::BlockDesc n = d;
                    n.x0 = (uint16_t)(d.x0 + qx * step); n.y0 = (uint16_t)(d.y0 + qy * step);
                    uint32_t mask = 0;
                    for (unsigned wv = 0; wv < 4; ++wv) {
                        const unsigned lx = qx * step + (wv & 1u) * quad, ly = qy * step + (wv >> 1) * quad;      // the wave's quad, inside the block
                        if ((hole >> ((ly >> 1) * 8u + (lx >> 1))) & 1ull) mask |= 1u << wv;
                    }
                    if (!mask) continue;
                    n.pitch |= (level << rt::kBlockNarrowShift) | (mask << rt::kBlockCoopShift);
                    cdescs.push_back(n); ccost.push_back(cost[order[i]]);
                }
        }
        descs.insert(descs.end(), cdescs.begin(), cdescs.end());
        dcost.insert(dcost.end(), ccost.begin(), ccost.end());
    }
    for (size_t i = n_coop; i < order.size(); ++i) {
        const rt::BlockDesc &d = raster[order[i]];
        if (i >= n_coop + n_narrow) { descs.push_back(d); dcost.push_back(cost[order[i]]); continue; }
        // 4x4 pixels per wave; 2x2 in a pass so small that its waves all start at once anyway (800x600: 63 -> 52 us; at 1080p the
        // sixteen-fold wave count of 2x2 costs more throughput than the shorter chains buy)
        const long long l2 = knob(RT_DEBUG_NARROW_L2);
        const unsigned level = (l2 >= 0 ? (long long)(i - n_coop) < l2 : (order.size() <= kNarrowLevel2Blocks && n_coop == 0)) ? 2u : 1u, step = 16u >> level, cnt = 1u << level;
        for (unsigned qy = 0; qy < cnt; ++qy)
            for (unsigned qx = 0; qx < cnt; ++qx) {
                rt::BlockDesc n = d;
                n.x0 = (uint16_t)(d.x0 + qx * step); n.y0 = (uint16_t)(d.y0 + qy * step);
                n.pitch |= level << rt::kBlockNarrowShift;
                if (n.x0 < d.r && n.y0 < d.t) { descs.push_back(n); dcost.push_back(cost[order[i]]); }      // parts outside a clipped edge tile have no pixels
            }
    }
    // A sample-parallel pass has one workgroup per block AND sample: 49,152 for `make image`, 1,048,576 for BASELINE config 5,
    // each living a few microseconds.  Past 32,768 workgroups the blocks are dealt out here instead, about eight to a
    // workgroup (8,192 .. 65,536 workgroups per launch): descriptors in descending cost, each to the workgroup with the least
    // estimated work so far (longest-processing-time-first); a workgroup renders its descriptors in that order, so the long
    // chains still start first.  Measured (tools/knob_sweep.py wg_policy): make image 303 -> 278 us, config 5 4.67 -> 4.26 ms;
    // passes the dispatcher can deal one block at a time (1080p, 4K at spp 1) lose by it -- its dynamic balancing beats
    // a static deal by estimated cost -- and are left alone.
    const long long policy = knob(RT_DEBUG_WG_POLICY);
    const size_t total_wg = (size_t)descs.size() * std::max(1u, passes);
    size_t n_wg = 0;
    if (policy > 0) n_wg = kResidentWorkgroups * (size_t)policy / std::max(1u, passes);
    else if (policy < 0 && total_wg > 32768) n_wg = std::clamp<size_t>(total_wg / 8, 8192, 65536) / std::max(1u, passes);
    if (map && n_wg >= 64 && descs.size() > n_wg && n_coop == 0) {        // (the holes of a cooperative pass are indexed by descriptor position)
        std::vector<std::vector<uint32_t>> lists(n_wg);
        std::vector<std::pair<uint64_t, uint32_t>> heap;                      // (load, workgroup), min-heap
        heap.reserve(n_wg);
        for (uint32_t g = 0; g < n_wg; ++g) heap.emplace_back(0ull, g);
        auto cmp = [](const std::pair<uint64_t, uint32_t> &a, const std::pair<uint64_t, uint32_t> &b) { return a > b; };
        std::make_heap(heap.begin(), heap.end(), cmp);
        for (uint32_t i = 0; i < descs.size(); ++i) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            auto &top = heap.back();
            lists[top.second].push_back(i);
            top.first += (uint64_t)dcost[i] + kFixedBlockCost;
            std::push_heap(heap.begin(), heap.end(), cmp);
        }
        std::vector<rt::BlockDesc> dealt;
        dealt.reserve(descs.size());
        wg_first.reserve(n_wg + 1);
        for (const auto &l : lists) {
            wg_first.push_back((uint32_t)dealt.size());
            for (uint32_t i : l) dealt.push_back(descs[i]);
        }
        wg_first.push_back((uint32_t)dealt.size());
        descs.swap(dealt);
    }
}

bool block_order_enabled() { return knob(RT_DEBUG_BLOCK_ORDER) != 0; }     // read per call: A/B timing interleaves both


void release_order(rt_scene::Order &od)           // (its arrays live in the table's arena)
{
    for (hipEvent_t e : od.e0) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : od.e1) if (e) (void)hipEventDestroy(e);
    od = rt_scene::Order{};
}

// The dispatch order a launch of this table uses (called under the scene's lock).  While a table with several candidates is undecided, its
// launches take turns: every candidate is handed out kOrderTrialSamples times with a pair of events (the caller records them around the
// launch; all of them may be in flight at once -- a caller that enqueues far ahead of the device is not waited for); results are collected
// here as they complete, and once every sample is in, the candidate with the smallest one stays.
constexpr int kOrderTrialSamples = 3;
rt::BlockList pick_order(rt_scene::CachedTable &t, bool will_be_timed)      // will_be_timed: a non-counting hierarchy-walk launch (launch_render records the pair)
{
    if (t.orders.empty()) return rt::BlockList{};
    // (orders[0] never has holes: what a launch that cannot walk cooperatively -- counters, f64, two rays per lane -- falls back to)
    auto list_of = [&t](const rt_scene::Order &od) {
        rt::BlockList l{ od.dev_order, od.n_order, od.dev_wg, od.n_wg, od.dev_holes, od.n_holes };
        l.plain_d = t.orders[0].dev_order; l.plain_n = t.orders[0].n_order; l.plain_wg_first = t.orders[0].dev_wg; l.plain_n_wg = t.orders[0].n_wg;
        return l;
    };
    if (t.chosen >= 0) return list_of(t.orders[(size_t)t.chosen]);
    bool all_done = true;
    for (auto &od : t.orders) {
        while (od.harvested < od.issued && hipEventQuery(od.e1[od.harvested % kOrderTrialSamples]) == hipSuccess) {
            float ms = 0.f;
            const int slot = od.harvested % kOrderTrialSamples;
            if (hipEventElapsedTime(&ms, od.e0[slot], od.e1[slot]) == hipSuccess && ms > 0.f) { od.best_ms = std::min(od.best_ms, ms); ++od.good; }
            ++od.harvested;
        }
        (void)hipGetLastError();                         // hipErrorNotReady is not an error here
        all_done = all_done && od.good >= kOrderTrialSamples;
    }
    auto best_known = [&t] {
        size_t best = 0;
        for (size_t i = 1; i < t.orders.size(); ++i) if (t.orders[i].best_ms < t.orders[best].best_ms) best = i;
        return best;
    };
    if (all_done) {
        const size_t best = best_known();
        t.chosen = (int)best;
        if (knob(RT_DEBUG_PRINT_STEPS) > 0) {
            fprintf(stderr, "[rtrace_hip] dispatch orders of a %zu-tile list, ms:", t.host.size());
            for (const auto &od : t.orders) fprintf(stderr, " %.4f%s", od.best_ms, od.dev_holes ? "c" : "");
            fprintf(stderr, " -> #%zu\n", best);
        }
        return list_of(t.orders[best]);
    }
    for (size_t k = 0; will_be_timed && k < t.orders.size(); ++k) {
        rt_scene::Order &od = t.orders[(t.turn + k) % t.orders.size()];
        const int in_flight = od.issued - od.harvested;
        if (od.good + in_flight >= kOrderTrialSamples) continue;     // (in_flight < kOrderTrialSamples follows: the slot is free)
        t.turn = (unsigned)((t.turn + k + 1) % t.orders.size());
        rt::BlockList l = list_of(od);
        l.ev0 = od.e0[od.issued % kOrderTrialSamples]; l.ev1 = od.e1[od.issued % kOrderTrialSamples];
        ++od.issued;
        return l;
    }
    return list_of(t.orders[best_known()]);              // every sample is in flight: the best known so far, untimed
}

// Something about the dispatch was asked for explicitly (rt_debug.h: tests, A/B tools): then a tile list's orders are made at once, by the
// caller, so that its very first launch already runs what was asked for.
bool order_knobs_set()
{
    for (int k : { RT_DEBUG_COOP, RT_DEBUG_COOP_THR, RT_DEBUG_COOP_MAX, RT_DEBUG_COOP_LEVEL, RT_DEBUG_COOP_REST, RT_DEBUG_NARROW_MAX, RT_DEBUG_NARROW_L2, RT_DEBUG_WG_POLICY,
                   RT_DEBUG_PRINT_COSTS, RT_DEBUG_EXACT_COSTS })
        if (knob(k) >= 0) return true;
    return knob(RT_DEBUG_ASYNC_ORDERS) == 0;
}

// The candidate dispatch orders of one tile list: the plain one first; where the cooperative walk could serve the pass and nothing was
// asked for explicitly, a few thresholds in percent of the pass's largest estimate (pick_order tries them: chosen = -1).
rt_status build_orders(rt_scene *s, const std::vector<uint32_t> *map, const std::vector<rt::TileDev> &tab, unsigned w, unsigned h, unsigned passes,
                       std::vector<rt_scene::Order> &orders, int &chosen, void **arena_out)
{
    const bool coop_pass = s->coop.fanout != 0u && passes == 1;       // (f32; f64 scenes whose filtered streams exist)
    uint64_t total_px = 0, total_blocks = 0;
    for (const rt::TileDev &td : tab) {
        total_px += (uint64_t)(td.r - td.l) * (td.t - td.b);
        total_blocks += (uint64_t)td.blks_x * (((unsigned)(td.t - td.b) + rt::kBlockH - 1) / rt::kBlockH);
    }
    const long long rays = knob(RT_DEBUG_SKIP_RAYS);
    const bool two_rays = s->precision == RT_F32 && (rays < 0 ? skip2_by_default(total_px, 1, s->fused ? s->n_fnodes : s->n_nodes) : rays == 2);      // k_render_skip2 (f32) knows no cooperative quads
    // Candidate 0 is ALWAYS the plain order of a pass that could walk cooperatively (coop_percent 0: no holes, no cooperative descriptors):
    // pick_order and launch_skip_one hand it to every launch that cannot take holes (counters, f64, two rays per lane).
    // Candidates: the plain order first; then cooperative thresholds.  Where the scene's stream is there to count the list's heaviest blocks
    // again at the frame's own resolution (exact_block_costs), the thresholds are shares of the largest EXACT count of tests per pixel and the
    // quads are picked pixel by pixel -- any pass size; without it (or when a control of rt_debug.h asks for the old way) shares of the map's
    // largest estimate, small passes only.
    struct Want { int pc; uint32_t exact_thr; };
    std::vector<Want> wants;
    ExactCosts exact;
    const bool knobs = knob(RT_DEBUG_COOP) >= 0 || knob(RT_DEBUG_COOP_THR) >= 0 || knob(RT_DEBUG_COOP_MAX) >= 0;
    if (coop_pass && map && !two_rays && !knobs && knob(RT_DEBUG_EXACT_COSTS) != 0) {
        std::vector<rt::BlockDesc> none; std::vector<uint32_t> none_wg, heaviest;
        block_order(map, tab, w, h, passes, none, none_wg, nullptr, nullptr, 0, nullptr, 0, &heaviest);
        std::vector<rt::BlockDesc> raster;
        for (const rt::TileDev &t : tab) {          // (block_order's raster enumeration: the indices `heaviest` holds)
            const unsigned bys = ((unsigned)(t.t - t.b) + rt::kBlockH - 1) / rt::kBlockH;
            const uint32_t pitch = (uint32_t)t.r - t.l;
            for (unsigned by = 0; by < bys; ++by)
                for (unsigned bx = 0; bx < t.blks_x; ++bx)
                    raster.push_back(rt::BlockDesc{ (uint16_t)(t.l + bx * rt::kBlockW), (uint16_t)(t.b + by * rt::kBlockH), t.r, t.t, pitch, 0u });
        }
        if ((s->precision == RT_F32 ? exact_block_costs<float>(s, raster, heaviest, w, h, exact) : exact_block_costs<double>(s, raster, heaviest, w, h, exact)) != RT_OK) {
            (void)hipGetLastError(); exact = ExactCosts{};
        }
    }
    bool asked = false;
    if (!exact.block.empty() && exact.top >= kCoopMinCost) {
        wants.push_back({ 0, 0u });
        // (a pass that fills the chip several times over pays for every cooperative quad in throughput: high thresholds; one that is over when its
        // first waves are -- 800x600: every wave dispatched by 10 us of 26 -- waits for chains only: the trial's times fell all the way to the
        // lowest threshold there, so its candidates reach lower)
        const bool small = total_blocks <= kCoopPassBlocks;
        // (round 6, the trial's own times: the best candidate was the HIGHEST threshold of the five at 1024x768, 1280x720, 1080p and in f64
        // everywhere -- profiles/r06_order_trials.log --, so the range now reaches further up, at both sizes)
        for (unsigned pc : { small ? 85u : 95u, small ? 70u : 88u, small ? 58u : 80u, small ? 48u : 70u, small ? 40u : 58u, small ? 33u : 48u })
            wants.push_back({ 0, std::max<uint32_t>((uint32_t)kCoopMinCost, exact.top * pc / 100u) });
    } else if (coop_pass && map && !two_rays && total_blocks <= kCoopPassBlocks && !knobs)
        wants = { { 0, 0u }, { 28, 0u }, { 34, 0u }, { 40, 0u }, { 48, 0u }, { 58, 0u } };
    else if (coop_pass && !two_rays && knobs && (knob(RT_DEBUG_COOP) > 0 || knob(RT_DEBUG_COOP_THR) >= 0 || knob(RT_DEBUG_COOP_MAX) >= 0)) { wants = { { 0, 0u }, { -1, 0u } }; asked = true; }       // as asked, behind the plain one
    else if (coop_pass) wants = { { 0, 0u } };
    else wants = { { -1, 0u } };
    // every candidate on the host first, then ONE device allocation for all their arrays: hipMalloc / hipFree wait for a busy device,
    // and this may run in the background of a caller who keeps it busy
    struct Host { std::vector<rt::BlockDesc> order; std::vector<uint32_t> wg_first; std::vector<uint64_t> holes; bool any_hole = false; };
    std::vector<Host> cand;
    for (const Want &wt : wants) {
        Host c;
        block_order(map, tab, w, h, passes, c.order, c.wg_first, coop_pass ? &s->coop : nullptr, &c.holes, wt.pc, wt.exact_thr ? &exact : nullptr, wt.exact_thr);
        c.any_hole = std::any_of(c.holes.begin(), c.holes.end(), [](uint64_t v) { return v != 0; });
        if (&wt != &wants[0] && !c.any_hole) continue;     // the same dispatch as the plain one
        if (!cand.empty() && wt.exact_thr && cand.back().any_hole && cand.back().holes == c.holes && cand.back().order.size() == c.order.size()) continue;   // (two thresholds, the same quads)
        cand.push_back(std::move(c));
    }
    auto up = [](size_t n) { return (n + 255) & ~(size_t)255; };
    size_t bytes = 0;
    for (const Host &c : cand)
        bytes += up(c.order.size() * sizeof(rt::BlockDesc)) + (c.any_hole ? up(c.holes.size() * sizeof(uint64_t)) : 0) + (c.wg_first.empty() ? 0 : up(c.wg_first.size() * sizeof(uint32_t)));
    char *arena = nullptr;
    hipError_t e = hipMalloc(&arena, std::max<size_t>(bytes, 256));
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(dispatch orders)", __LINE__);
    auto fail = [&](hipError_t err) { for (auto &od : orders) release_order(od); orders.clear(); (void)hipFree(arena); return hip_fail(err, "dispatch orders", __LINE__); };
    size_t off = 0;
    for (const Host &c : cand) {
        rt_scene::Order od;
        od.dev_order = reinterpret_cast<rt::BlockDesc *>(arena + off); od.n_order = (uint32_t)c.order.size();
        if ((e = hipMemcpyAsync(od.dev_order, c.order.data(), c.order.size() * sizeof(rt::BlockDesc), hipMemcpyHostToDevice, s->cost_stream)) != hipSuccess) return fail(e);
        off += up(c.order.size() * sizeof(rt::BlockDesc));
        if (c.any_hole) {
            od.dev_holes = reinterpret_cast<uint64_t *>(arena + off); od.n_holes = (uint32_t)c.holes.size();
            if ((e = hipMemcpyAsync(od.dev_holes, c.holes.data(), c.holes.size() * sizeof(uint64_t), hipMemcpyHostToDevice, s->cost_stream)) != hipSuccess) return fail(e);
            off += up(c.holes.size() * sizeof(uint64_t));
        }
        if (!c.wg_first.empty()) {
            od.dev_wg = reinterpret_cast<uint32_t *>(arena + off); od.n_wg = (uint32_t)c.wg_first.size() - 1;
            if ((e = hipMemcpyAsync(od.dev_wg, c.wg_first.data(), c.wg_first.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s->cost_stream)) != hipSuccess) return fail(e);
            off += up(c.wg_first.size() * sizeof(uint32_t));
        }
        orders.push_back(od);
    }
    if ((e = hipStreamSynchronize(s->cost_stream)) != hipSuccess) return fail(e);       // (the candidates' host arrays go out of scope; the orders are in device memory from here on)
    if (!orders.empty() && orders[0].dev_holes) {       // (cannot happen: candidate 0 is made with coop_percent 0 wherever holes are possible)
        for (auto &od : orders) release_order(od);
        orders.clear(); (void)hipFree(arena);
        snprintf(g_err, sizeof g_err, "internal: the plain dispatch order has holes");
        return RT_ERR_INVALID_ARGUMENT;
    }
    chosen = 0;
    if (orders.size() > 1 && asked) chosen = 1;      // asked for explicitly
    else if (orders.size() > 1) {
        chosen = -1;                                                // to be decided by measurement
        for (auto &od : orders)
            for (int k = 0; k < kOrderTrialSamples; ++k) {
                e = hipEventCreate(&od.e0[k]);
                if (e == hipSuccess) e = hipEventCreate(&od.e1[k]);
                if (e != hipSuccess) return fail(e);
            }
    }
    *arena_out = arena;
    return RT_OK;
}

// Builder threads must not outlive the HIP runtime: a process that exits without destroying its scenes (a Python interpreter does not run
// every finalizer) still has them joined, by an exit handler registered when the first one is started -- later than the runtime's own
// teardown was registered, hence run before it.
void stop_worker(rt_scene *s);
std::mutex g_live_mu;
std::vector<rt_scene *> g_live_scenes;           // scenes that ever started a builder and are not destroyed yet

void join_builders_at_exit()
{
    std::vector<rt_scene *> live;
    { std::lock_guard<std::mutex> lk(g_live_mu); live.swap(g_live_scenes); }
    for (rt_scene *sc : live) {
        for (std::thread &b : sc->builders) if (b.joinable()) b.join();
        stop_worker(sc);
    }
}

void note_builder(rt_scene *s)
{
    static std::once_flag once;
    std::call_once(once, [] { std::atexit(join_builders_at_exit); });
    std::lock_guard<std::mutex> lk(g_live_mu);
    if (std::find(g_live_scenes.begin(), g_live_scenes.end(), s) == g_live_scenes.end()) g_live_scenes.push_back(s);
}

void forget_scene(rt_scene *s)
{
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live_scenes.erase(std::remove(g_live_scenes.begin(), g_live_scenes.end(), s), g_live_scenes.end());
}

// The scene's worker: runs the jobs handed to it one after the other; on stop, the ones still queued as well (they are finite and somebody
// may be waiting for `building` to clear).
void worker_main(rt_scene *s)
{
    knobs_at_default();                           // it only ever serves lists that were first seen with no dispatch control set
    (void)hipSetDevice(s->device);
    for (;;) {
        std::function<void()> job;
        bool stopping;
        {
            std::unique_lock<std::mutex> lk(s->wmu);
            s->wcv.wait(lk, [s] { return s->wstop || !s->wjobs.empty(); });
            if (s->wjobs.empty()) return;
            job = std::move(s->wjobs.front());
            s->wjobs.pop_front();
            stopping = s->wstop;
        }
        // let the caller's first launch (and whoever waits for it) have the runtime to itself: the orders' allocations and blocking copies
        // took 20-50 us out of a one-shot caller's first frame when they started at once, and nobody misses them for another 0.3 ms
        if (!stopping) std::this_thread::sleep_for(std::chrono::microseconds(300));
        job();
    }
}

void stop_worker(rt_scene *s)
{
    if (!s->worker.joinable()) return;
    { std::lock_guard<std::mutex> lk(s->wmu); s->wstop = true; }
    s->wcv.notify_all();
    s->worker.join();
}

// The same from a thread of its own (see device_table): cost map, orders, uploads -- then the finished orders are handed to table `index`
// under the scene's lock.  Whatever fails here only costs the ordering: the table keeps rendering through the tile table.
void build_orders_async(rt_scene *s, size_t index, std::vector<rt::TileDev> tab, unsigned w, unsigned h, unsigned passes)
{
    knobs_at_default();                           // this thread only exists because no dispatch control was set when the list was first seen
    std::vector<rt_scene::Order> orders;
    int chosen = 0;
    void *arena = nullptr;
    bool ok = hipSetDevice(s->device) == hipSuccess;
    if (ok) {
        const std::vector<uint32_t> *map = cost_map_of(s);
        // (the uploads are blocking copies: the data is in device memory when they return.  No device-wide synchronise here -- the caller's
        // own launches keep the device busy and it would wait for all of them)
        ok = build_orders(s, map, tab, w, h, passes, orders, chosen, &arena) == RT_OK;
    }
    (void)hipGetLastError();
    std::lock_guard<std::mutex> lk(s->mu);
    rt_scene::CachedTable &t = s->tables[index];
    if (ok) { t.orders = std::move(orders); t.chosen = chosen; t.order_arena = arena; }
    t.building = false;
}

rt_status device_table(rt_scene *s, Context *c, const std::vector<rt::TileDev> &tab, hipStream_t stream, const rt::TileDev **out, int slot,
                       const rt_options *o, rt::BlockList *order_out, bool cacheable, bool will_be_timed)
{
    const size_t bytes = tab.size() * sizeof(rt::TileDev);
    const unsigned w = o ? o->width : 0u, h = o ? o->height : 0u;
    const unsigned passes = (o && use_split(o->samples_per_pixel)) ? (unsigned)o->samples_per_pixel * o->samples_per_pixel : 1u;
    if (order_out) *order_out = rt::BlockList{};
    StageClock clk;
    // the cooperative walk's controls (rt_debug.h) are part of a dispatch table's identity: tests render one tile list with and without
    long long coop_key = 0;
    if (o && order_out)
        for (int k : { RT_DEBUG_COOP, RT_DEBUG_COOP_THR, RT_DEBUG_COOP_MAX, RT_DEBUG_COOP_LEVEL, RT_DEBUG_COOP_REST, RT_DEBUG_NARROW_MAX, RT_DEBUG_NARROW_L2, RT_DEBUG_SKIP_RAYS, RT_DEBUG_EXACT_COSTS })
            coop_key = coop_key * 1000003ll + (knob(k) + 2);
    if (cacheable) {
        std::lock_guard<std::mutex> lk(s->mu);
        for (auto &t : s->tables)
            if ((!o || (t.w == w && t.h == h && t.passes == passes && (!order_out || t.coop_key == coop_key))) && t.host.size() == tab.size() &&
                memcmp(t.host.data(), tab.data(), bytes) == 0) {
                *out = t.dev;
                if (t.landed) {                      // uploaded on its first caller's stream: has it arrived?
                    if (hipEventQuery(t.landed) == hipSuccess) { (void)hipEventDestroy(t.landed); t.landed = nullptr; }
                    else {
                        (void)hipGetLastError();
                        if (stream != t.landed_on) HIP_TRY(hipStreamWaitEvent(stream, t.landed, 0));
                    }
                }
                if (order_out && block_order_enabled()) *order_out = pick_order(t, will_be_timed);
                return RT_OK;
            }
        if (s->tables.size() < kMaxCachedTables) {
            rt_scene::CachedTable t;
            // The dispatch orders (and the scene's cost map they are made from) cost the host a few milliseconds: unless something was asked
            // for explicitly (rt_debug.h), they are made by the scene's worker thread while this and the next launches find their blocks through
            // the tile table -- a one-shot caller (`make image`) never waits for them, a scheduler gets them a few frames in.
            const bool want_orders = o && order_out;
            const bool in_background = want_orders && !order_knobs_set();
            HIP_TRY(hipMalloc(&t.dev, bytes));
            auto drop = [&t] { (void)hipFree(t.dev); for (auto &od : t.orders) release_order(od); if (t.order_arena) (void)hipFree(t.order_arena);
                               if (t.landed) (void)hipEventDestroy(t.landed); };
            // The table itself: through the scene's pinned staging on the caller's own stream when the list is new to a caller in a hurry
            // (the launch that follows is behind it on that stream; launches on other streams wait for `landed`) -- a blocking copy and the
            // device-wide synchronise it needs cost the first frame 50 us.
            const size_t staged = (bytes + 255) & ~(size_t)255;
            bool async_copy = false;
            if (in_background && s->h_tab_stage && s->tab_stage_used + staged <= kTableStageBytes &&
                hipEventCreateWithFlags(&t.landed, hipEventDisableTiming) == hipSuccess) {
                char *h = s->h_tab_stage + s->tab_stage_used;
                memcpy(h, tab.data(), bytes);
                if (upload_words(t.dev, h, bytes, stream) == RT_OK && hipEventRecord(t.landed, stream) == hipSuccess) {      // (a kernel, not the copy engine: rt_kernels.hpp k_upload_words)
                    s->tab_stage_used += staged;
                    t.landed_on = stream;
                    async_copy = true;
                } else { (void)hipGetLastError(); (void)hipEventDestroy(t.landed); t.landed = nullptr; }
            } else (void)hipGetLastError();
            hipError_t e = async_copy ? hipSuccess : hipMemcpy(t.dev, tab.data(), bytes, hipMemcpyHostToDevice);    // blocking, once per table
            clk.lap("tile table upload");
            if (e != hipSuccess) { drop(); return hip_fail(e, "hipMemcpy(tile table)", __LINE__); }
            if (want_orders && !in_background) {
                const std::vector<uint32_t> *map = cost_map_of(s);
                clk.lap("cost map (cached after 1st)");
                rt_status bst = build_orders(s, map, tab, w, h, passes, t.orders, t.chosen, &t.order_arena);
                clk.lap("dispatch orders");
                if (bst != RT_OK) { drop(); return bst; }
            }
            // The copies above are blocking for the host, but the render kernel runs on another (non-blocking) stream: make sure
            // the tables have landed in device memory before anything can be launched against them (once per tile list).
            if (!async_copy) {
                e = hipDeviceSynchronize();
                if (e != hipSuccess) { drop(); return hip_fail(e, "hipDeviceSynchronize(tile tables)", __LINE__); }
            }
            clk.lap("device sync");
            t.building = in_background;
            t.host = tab; t.w = w; t.h = h; t.passes = passes; t.coop_key = coop_key;
            *out = t.dev;
            s->tables.push_back(std::move(t));
            if (in_background) {
                const size_t index = s->tables.size() - 1;
                note_builder(s);
                if (s->worker.joinable()) {
                    { std::lock_guard<std::mutex> wl(s->wmu); s->wjobs.emplace_back([s, index, tab, w, h, passes] { build_orders_async(s, index, tab, w, h, passes); }); }
                    s->wcv.notify_one();
                } else s->builders.emplace_back([s, index, tab, w, h, passes] { knobs_at_default(); build_orders_async(s, index, tab, w, h, passes); });
            }
            if (order_out && block_order_enabled()) *order_out = pick_order(s->tables.back(), will_be_timed);
            return RT_OK;
        }
    }
    if (!c) { *out = nullptr; return RT_OK; }                       // cache full and no context to upload through
    return upload_tiles(c, tab, stream, slot, out);
}

