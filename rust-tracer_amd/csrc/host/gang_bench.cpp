// gang_bench.cpp -- the several-GPU path of ONE process, timed: the 1920x1080 frame of the default scene through rt_gang_* on devices
// 0..N-1 (include/rtrace_hip.h: buckets dealt round-robin in the scheduler's order render.rs:273-298, one ncclGather of the u8 shards to
// the first GPU, blit there -- what replaces the channel of render.rs:271,293,301 across GPUs).  No torch.distributed call on the path:
// bench.py runs this as a child process after its one-process-per-GPU measurement (`native_gang`, never `value`).  Prints one JSON line.
//
//   gang_bench [--devices N] [--frames K] [--width W --height H --spp S --level L]
//   (tests/c/gang_bench_test, built with -DRT_TEST_HOOKS: + --rccl-stand-in <library>, all N ranks on device 0 through the stand-in for
//   librccl.so -- how the N = 8 deal is executed on a one-GPU box)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "render.hpp"
#ifdef RT_TEST_HOOKS
#include "../rt_debug.h"
#endif

using namespace rtrace;
using Clock = std::chrono::steady_clock;

namespace {

double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

void die(rt_status st, const char *what)
{
    if (st == RT_OK) return;
    fprintf(stderr, "gang_bench: %s: %s -- %s\n", what, rt_strerror(st), rt_last_error_message());
    exit(3);
}

uint32_t crc32_of(const uint8_t *p, size_t n)
{
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
        init = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

}  // namespace

int main(int argc, char **argv)
{
    int n_devices = 1, frames = 200, level = 8;
    unsigned width = 1920, height = 1080, spp = 1;
    std::string stand_in;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string a = argv[i];
        if (a == "--devices") n_devices = atoi(argv[i + 1]);
        else if (a == "--frames") frames = atoi(argv[i + 1]);
        else if (a == "--width") width = (unsigned)atoi(argv[i + 1]);
        else if (a == "--height") height = (unsigned)atoi(argv[i + 1]);
        else if (a == "--spp") spp = (unsigned)atoi(argv[i + 1]);
        else if (a == "--level") level = atoi(argv[i + 1]);
#ifdef RT_TEST_HOOKS
        else if (a == "--rccl-stand-in") stand_in = argv[i + 1];
#endif
        else { fprintf(stderr, "gang_bench: unknown option %s\n", a.c_str()); return 2; }
    }
    n_devices = std::clamp(n_devices, 1, 64); frames = std::clamp(frames, 2, 100000);
#ifdef RT_TEST_HOOKS
    if (!stand_in.empty()) die(rt_debug_rccl_library(stand_in.c_str()), "rt_debug_rccl_library");
#endif
    const Scene scene = Scene::with_level((uint32_t)level);
    std::vector<int> ids;
    for (int d = 0; d < n_devices; ++d) ids.push_back(stand_in.empty() ? d : 0);
    const Clock::time_point t_create = Clock::now();
    rt_status gst = RT_OK;
    const std::shared_ptr<DeviceGang> gang_ptr = DeviceGang::try_create(scene, ids, &gst);      // (RCCL or a device missing: the library's message)
    die(gst, "rt_gang_create");
    const DeviceGang &gang = *gang_ptr;
    const double create_ms = ms_since(t_create);
    const RenderOptions o{ (uint16_t)width, (uint16_t)height, (uint16_t)spp };
    const rt_options opts{ o.width, o.height, o.samples_per_pixel };
    const std::vector<ImageRegion> bl = Renderer::buckets(o);
    const rt_region *regs = reinterpret_cast<const rt_region *>(bl.data());
    const uint32_t n = (uint32_t)bl.size();
    const size_t frame_bytes = (size_t)width * height * 4;

    // two pinned frames, written by the root GPU's blit kernel itself (rt_host_alloc memory is recognised by address)
    void *pinned[2] = { nullptr, nullptr };
    for (void *&p : pinned) die(rt_host_alloc(frame_bytes, &p), "rt_host_alloc");
    std::vector<uint8_t *> dst((size_t)frames);
    for (int f = 0; f < frames; ++f) dst[(size_t)f] = static_cast<uint8_t *>(pinned[f & 1]);

    // one frame at a time: render -> gather -> blit -> done (what a caller who wants THIS frame waits for)
    for (int f = 0; f < 5; ++f) die(rt_gang_render_frame(gang.handle(), &opts, RT_TRAVERSAL_SKIP, regs, n, dst[0], nullptr), "rt_gang_render_frame");
    std::vector<double> lat;
    for (int f = 0; f < 30; ++f) {
        const Clock::time_point t0 = Clock::now();
        die(rt_gang_render_frame(gang.handle(), &opts, RT_TRAVERSAL_SKIP, regs, n, dst[0], nullptr), "rt_gang_render_frame");
        lat.push_back(ms_since(t0));
    }
    std::sort(lat.begin(), lat.end());
    // frames pipelined inside one call: render(f + 1) over gather(f) + blit(f)
    die(rt_gang_render_frames(gang.handle(), &opts, RT_TRAVERSAL_SKIP, regs, n, dst.data(), (uint32_t)std::min(frames, 20), nullptr), "rt_gang_render_frames");
    memset(pinned[0], 0, frame_bytes); memset(pinned[1], 0, frame_bytes);      // what is checked below was produced by the timed call
    std::vector<double> per;
    for (int rep = 0; rep < 3; ++rep) {
        const Clock::time_point t0 = Clock::now();
        die(rt_gang_render_frames(gang.handle(), &opts, RT_TRAVERSAL_SKIP, regs, n, dst.data(), (uint32_t)frames, nullptr), "rt_gang_render_frames");
        per.push_back(ms_since(t0) / frames);
    }
    std::sort(per.begin(), per.end());
    const uint32_t crc0 = crc32_of(static_cast<uint8_t *>(pinned[0]), frame_bytes), crc1 = crc32_of(static_cast<uint8_t *>(pinned[1]), frame_bytes);
    rt_stats st{};
    die(rt_gang_render_frame(gang.handle(), &opts, RT_TRAVERSAL_SKIP, regs, n, dst[0], &st), "rt_gang_render_frame(stats)");
    const double rays = (double)(st.primary + st.shadow);
    printf("{\"devices\": %d, \"stand_in\": %s, \"workload\": \"%ux%u spp %u L%d, %u buckets\", \"gang_create_ms\": %.2f, \"ms_per_frame\": %.4f, \"ms_per_frame_min_max\": [%.4f, %.4f], "
           "\"frames_per_call\": %d, \"value\": %.1f, \"unit\": \"Mrays/s\", \"frame_latency_ms\": %.4f, \"frame_latency_min_ms\": %.4f, \"frame_crc32\": %u, \"both_buffers_equal\": %s, "
           "\"primary\": %llu, \"shadow\": %llu}\n",
           n_devices, stand_in.empty() ? "false" : "true", width, height, spp, level, n, create_ms, per[per.size() / 2], per.front(), per.back(), frames,
           rays / per[per.size() / 2] / 1e3, lat[lat.size() / 2], lat.front(), crc0, crc0 == crc1 ? "true" : "false",
           (unsigned long long)st.primary, (unsigned long long)st.shadow);
    for (void *p : pinned) rt_host_free(p);
    return crc0 == crc1 ? 0 : 4;
}
