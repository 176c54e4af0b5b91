// seam_bench.cpp -- timing of the boundary the reference binds, from native threads (what the Rust pool threads would do):
//
//   host_tiles     rt_render_tiles, the whole frame's buckets in ONE call, bytes delivered to host memory
//                  (pageable Vec<u8>-like memory / rt_host_alloc'd memory)
//   host_region    the literal closure of render.rs:283-294: one rt_render_region call per 64x64 bucket, from
//                  RTRACEMAXPROCS = 1 and = T pool threads (concurrent callers are merged into shared device passes)
//   end_to_end     frame:     render + D2H + PPM encode + file write of one frame (upload excluded)
//                  scheduler: Renderer::render (bucket scheduler, channel of 4, PPMStdoutRGBABufferWriter) with 1 and T pool threads
//
// Workload = bench.py's: 1920x1080, default scene (pyramid level 8), spp 1, the hierarchy traversal.  Prints one JSON object.
// bench.py runs this as a child process and embeds the object in its line.
//
//   seam_bench [--frames N] [--threads T] [--width W --height H --spp S --level L] [--dir /dev/shm]
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>

#include "render.hpp"
#ifdef RT_TEST_HOOKS
#include "../rt_debug.h"      // diagnostic legs only when built against the -DRT_TEST_HOOKS library; bench.py runs the product
#endif

using namespace rtrace;
using Clock = std::chrono::steady_clock;

namespace {

double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

struct Series { std::vector<double> v; };
void stats_json(const char *name, Series &s, double rays, bool last = false)
{
    std::sort(s.v.begin(), s.v.end());
    const double med = s.v[s.v.size() / 2];
    printf("  \"%s\": {\"ms_per_frame\": %.4f, \"min\": %.4f, \"max\": %.4f, \"frames\": %zu, \"Mrays_per_s\": %.1f}%s\n", name, med, s.v.front(),
           s.v.back(), s.v.size(), rays / med / 1e3, last ? "" : ",");
}

void die(rt_status st, const char *what)
{
    if (st == RT_OK) return;
    fprintf(stderr, "seam_bench: %s: %s -- %s\n", what, rt_strerror(st), rt_last_error_message());
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
    int frames = 30, threads = (int)std::max(1u, std::thread::hardware_concurrency()), level = 8;
    unsigned width = 1920, height = 1080, spp = 1;
    std::string dir = "/dev/shm";
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string a = argv[i];
        if (a == "--frames") frames = atoi(argv[i + 1]);
        else if (a == "--threads") threads = atoi(argv[i + 1]);
        else if (a == "--width") width = (unsigned)atoi(argv[i + 1]);
        else if (a == "--height") height = (unsigned)atoi(argv[i + 1]);
        else if (a == "--spp") spp = (unsigned)atoi(argv[i + 1]);
        else if (a == "--level") level = atoi(argv[i + 1]);
        else if (a == "--dir") dir = argv[i + 1];
#ifdef RT_TEST_HOOKS
        else if (a == "--leaders") rt_debug_set(RT_DEBUG_COALESCE, atoi(argv[i + 1]));       // diagnostic: merged passes in flight (0: no merging)
        else if (a == "--frame-ahead") rt_debug_set(RT_DEBUG_FRAME_AHEAD, atoi(argv[i + 1])); // diagnostic: 0 / 1 / 2 (rt_debug.h)
#endif
        else { fprintf(stderr, "seam_bench: unknown option %s\n", a.c_str()); return 2; }
    }
    frames = std::max(frames, 3); threads = std::max(threads, 1);

    const Scene scene = Scene::with_level((uint32_t)level);
    Backend be;
    be.devices.push_back(std::make_shared<DeviceScene>(scene, 0));
    rt_scene *h = be.devices[0]->handle();
    const RenderOptions o{ (uint16_t)width, (uint16_t)height, (uint16_t)spp };
    const rt_options opts{ o.width, o.height, o.samples_per_pixel };
    const std::vector<ImageRegion> bl = Renderer::buckets(o);
    const rt_region *regs = reinterpret_cast<const rt_region *>(bl.data());
    const uint32_t n = (uint32_t)bl.size();
    const size_t bytes = (size_t)rt_tiles_rgba_bytes(regs, n);
    std::vector<size_t> off(n + 1, 0);
    for (uint32_t i = 0; i < n; ++i) off[i + 1] = off[i] + bl[i].area() * 4;

    // reference bytes + ray counts of the frame (one counted call)
    std::vector<uint8_t> ref(bytes);
    rt_stats st{};
    die(rt_render_tiles(h, &opts, RT_TRAVERSAL_SKIP, regs, n, ref.data(), &st), "rt_render_tiles");
    const double rays = (double)(st.primary + st.shadow);
    const uint32_t ref_crc = crc32_of(ref.data(), bytes);

    printf("{\n  \"workload\": \"%ux%u, pyramid level %d, spp %u, hierarchy traversal, %u buckets\", \"rays_per_frame\": %.0f, \"bytes_per_frame\": %zu,\n"
           "  \"threads\": %d, \"tile_major_crc32\": %u,\n", width, height, level, spp, n, rays, bytes, threads, ref_crc);

    auto check = [&](const uint8_t *p, const char *what) {
        if (memcmp(p, ref.data(), bytes) != 0) { fprintf(stderr, "seam_bench: %s delivered different bytes\n", what); exit(4); }
    };

    // ---------------- host_tiles ----------------
    {
        std::vector<uint8_t> pageable(bytes);
        Series s;
        for (int f = 0; f < frames + 3; ++f) {
            const auto t0 = Clock::now();
            die(rt_render_tiles(h, &opts, RT_TRAVERSAL_SKIP, regs, n, pageable.data(), nullptr), "rt_render_tiles");
            if (f >= 3) s.v.push_back(ms_since(t0));
        }
        check(pageable.data(), "host_tiles (pageable)");
        stats_json("host_tiles_pageable", s, rays);
        void *pinned = nullptr;
        die(rt_host_alloc(bytes, &pinned), "rt_host_alloc");
        Series p;
        for (int f = 0; f < frames + 3; ++f) {
            const auto t0 = Clock::now();
            die(rt_render_tiles(h, &opts, RT_TRAVERSAL_SKIP, regs, n, static_cast<uint8_t *>(pinned), nullptr), "rt_render_tiles");
            if (f >= 3) p.v.push_back(ms_since(t0));
        }
        check(static_cast<uint8_t *>(pinned), "host_tiles (pinned)");
        stats_json("host_tiles_pinned", p, rays);
        die(rt_host_free(pinned), "rt_host_free");
        // the same frame delivered in the FILE's pixel format (P6 payload, 3 B/px), converted and placed by the device (rt_render_frame_stream)
        void *rgb = nullptr;
        die(rt_host_alloc((size_t)width * height * 3, &rgb), "rt_host_alloc");
        Series q;
        for (int f = 0; f < frames + 3; ++f) {
            const auto t0 = Clock::now();
            die(rt_render_frame_stream(h, &opts, RT_TRAVERSAL_SKIP, regs, n, RT_FRAME_RGB, static_cast<uint8_t *>(rgb), nullptr, nullptr), "rt_render_frame_stream");
            if (f >= 3) q.v.push_back(ms_since(t0));
        }
        stats_json("host_frame_rgb_pinned", q, rays);
        die(rt_host_free(rgb), "rt_host_free");
    }

    // ---------------- host_region ----------------
    // the pool is persistent, like the reference's ThreadPool (main.rs:83): thread start-up is not part of a frame
    for (int nt : { 1, threads }) {
        std::vector<uint8_t> frame(bytes);
        Series s;
        const int reps = std::max(3, nt == 1 ? frames / 6 : frames / 2);
        ThreadPool pool((size_t)nt);
#ifdef RT_TEST_HOOKS
        const long long calls0 = rt_debug_count(RT_DEBUG_COUNT_REGION_CALLS), passes0 = rt_debug_count(RT_DEBUG_COUNT_REGION_PASSES);
#endif
        for (int f = 0; f < reps + 1; ++f) {
            std::fill(frame.begin(), frame.end(), 0);
            const auto t0 = Clock::now();
            SyncChannel<int> done(n);
            for (uint32_t i = 0; i < n; ++i)
                pool.execute([&, i] {
                    die(rt_render_region(h, &opts, RT_TRAVERSAL_SKIP, &regs[i], frame.data() + off[i], nullptr), "rt_render_region");
                    done.send(1);
                });
            for (uint32_t i = 0; i < n; ++i) (void)done.recv();
            if (f >= 1) s.v.push_back(ms_since(t0));
        }
        check(frame.data(), "host_region");
        char name[64];
        snprintf(name, sizeof name, "host_region_%s", nt == 1 ? "1_thread" : "T_threads");
        stats_json(name, s, rays);
#ifdef RT_TEST_HOOKS
        printf("  \"%s_calls_per_device_pass\": %.2f,\n", name, (double)(rt_debug_count(RT_DEBUG_COUNT_REGION_CALLS) - calls0) /
                                                                  (double)std::max(1ll, rt_debug_count(RT_DEBUG_COUNT_REGION_PASSES) - passes0));
#endif
        if (threads == 1) break;
    }

    // ---------------- end_to_end: one frame call + PPM ----------------
    const std::string path = dir + "/seam_bench_" + std::to_string((long)getpid()) + ".tga";
    {
        void *pinned = nullptr;
        die(rt_host_alloc((size_t)width * height * 4, &pinned), "rt_host_alloc");
        const rt_region whole{ 0, (uint16_t)height, (uint16_t)width, 0 };
        Series s;
        for (int f = 0; f < frames + 2; ++f) {
            const auto t0 = Clock::now();
            die(rt_render_tiles(h, &opts, RT_TRAVERSAL_SKIP, &whole, 1, static_cast<uint8_t *>(pinned), nullptr), "rt_render_tiles");
            // the library's own writer, fed the whole frame as ONE RGBABuffer: one conversion (on this thread), one write of the file
            FileOrAnyWriter sink;
            sink.f = fopen(path.c_str(), "wb");
            sink.is_file = true;
            if (!sink.f) { fprintf(stderr, "seam_bench: cannot write %s\n", path.c_str()); return 5; }
            {
                PPMStdoutRGBABufferWriter writer(true, sink);
                writer.begin((uint16_t)width, (uint16_t)height);
                writer.write_rgba_buffer(RGBABuffer(ImageRegion{ 0, (uint16_t)height, (uint16_t)width, 0 }, static_cast<const uint8_t *>(pinned), RGBABuffer::View{}));
            }
            fclose(sink.f);
            if (f >= 2) s.v.push_back(ms_since(t0));
        }
        stats_json("end_to_end_frame_cpu_encode", s, rays);
        // the same frame with the P6 encoding on the device (rt_render_frame_stream): the writer's image is memory the GPU writes, the
        // frame arrives converted (6.2 MB over PCIe instead of 8.3), the file is one positioned write of header + pixels
        Series d;
        uint32_t crc_file = 0;
        for (int f = 0; f < frames + 2; ++f) {
            const auto t0 = Clock::now();
            FileOrAnyWriter sink;
            sink.f = fopen(path.c_str(), "wb");
            sink.is_file = true;
            if (!sink.f) { fprintf(stderr, "seam_bench: cannot write %s\n", path.c_str()); return 5; }
            {
                PPMStdoutRGBABufferWriter writer(true, sink);
                writer.begin((uint16_t)width, (uint16_t)height);
                die(rt_render_frame_stream(h, &opts, RT_TRAVERSAL_SKIP, regs, n, writer.frame_format(), writer.pixels(), nullptr, nullptr), "rt_render_frame_stream");
                writer.buckets_arrived(bl.data(), bl.size());
            }
            fclose(sink.f);
            if (f >= 2) d.v.push_back(ms_since(t0));
        }
        stats_json("end_to_end_frame", d, rays);
        {
            // the file the device-encoded path wrote must be the file the CPU-encoded path writes: P6 header + R, G, B of the reference frame
            std::vector<uint8_t> want;
            char hdr[64];
            const int hl = snprintf(hdr, sizeof hdr, "P6\n%u %u\n255\n", width, height);
            want.insert(want.end(), hdr, hdr + hl);
            std::vector<uint8_t> frame((size_t)width * height * 4);
            for (uint32_t i = 0; i < n; ++i)
                for (uint16_t y = bl[i].b; y < bl[i].t; ++y)
                    memcpy(frame.data() + ((size_t)y * width + bl[i].l) * 4, ref.data() + off[i] + (size_t)(y - bl[i].b) * bl[i].width() * 4, (size_t)bl[i].width() * 4);
            for (size_t px = 0; px < (size_t)width * height; ++px) { want.push_back(frame[4 * px]); want.push_back(frame[4 * px + 1]); want.push_back(frame[4 * px + 2]); }
            std::vector<uint8_t> got(want.size() + 1);
            FILE *rf = fopen(path.c_str(), "rb");
            const size_t nread = rf ? fread(got.data(), 1, got.size(), rf) : 0;
            if (rf) fclose(rf);
            if (nread != want.size() || memcmp(got.data(), want.data(), want.size()) != 0) { fprintf(stderr, "seam_bench: the device-encoded file differs from the reference bytes\n"); return 4; }
            crc_file = crc32_of(want.data(), want.size());
        }
        printf("  \"end_to_end_frame_file_crc32\": %u,\n", crc_file);
        die(rt_host_free(pinned), "rt_host_free");
    }

    // ---------------- end_to_end: the reference's scheduler + writer ----------------
    for (int nt : { 1, threads }) {
        ThreadPool pool((size_t)nt);
        Series s;
        const int reps = std::max(3, frames / 3);
        double part[5] = { 0, 0, 0, 0, 0 };                           // fopen | Renderer::render | the writer's Drop (second write of the file) | fclose | of render: inside the writer
        struct TimedWriter : PPMStdoutRGBABufferWriter {              // how much of Renderer::render is the consumer's own work (conversion + the first write of the file)
            using PPMStdoutRGBABufferWriter::PPMStdoutRGBABufferWriter;
            double in_writer = 0;
            void write_rgba_buffer(const RGBABuffer &b) override
            {
                const auto t = Clock::now();
                PPMStdoutRGBABufferWriter::write_rgba_buffer(b);
                in_writer += ms_since(t);
            }
            bool accepts_device_frames() const override { return true; }
            void buckets_arrived(const ImageRegion *regions, size_t count) override
            {
                const auto t = Clock::now();
                PPMStdoutRGBABufferWriter::buckets_arrived(regions, count);
                in_writer += ms_since(t);
            }
        };
        for (int f = 0; f < reps + 1; ++f) {
            const auto t0 = Clock::now();
            FileOrAnyWriter sink;
            sink.f = fopen(path.c_str(), "wb");
            sink.is_file = true;
            if (!sink.f) { fprintf(stderr, "seam_bench: cannot write %s\n", path.c_str()); return 5; }
            const double t_open = ms_since(t0);
            double t_render = 0;
            {
                TimedWriter writer(true, sink);
                Renderer::render(o, be, writer, pool);
                t_render = ms_since(t0);
                if (f >= 1) part[4] += writer.in_writer;
            }
            const double t_drop = ms_since(t0);
            fclose(sink.f);
            const double t_all = ms_since(t0);
            if (f >= 1) {
                s.v.push_back(t_all);
                part[0] += t_open; part[1] += t_render - t_open; part[2] += t_drop - t_render; part[3] += t_all - t_drop;
            }
        }
        char name[64];
        snprintf(name, sizeof name, "end_to_end_scheduler_%s", nt == 1 ? "1_thread" : "T_threads");
        stats_json(name, s, rays);
        printf("  \"%s_parts_ms\": {\"fopen_truncate\": %.3f, \"render_and_first_write\": %.3f, \"drop_second_write\": %.3f, \"fclose\": %.3f, \"of_render_inside_the_writer\": %.3f}%s\n", name,
               part[0] / reps, part[1] / reps, part[2] / reps, part[3] / reps, part[4] / reps, nt == threads ? "" : ",");
        if (threads == 1) break;
    }
    unlink(path.c_str());
    printf("}\n");
    return 0;
}
