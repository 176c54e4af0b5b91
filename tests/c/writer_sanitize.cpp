// Drives PPMStdoutRGBABufferWriter (csrc/host/render.cpp) with clipped buckets in a scrambled order, P5 and P6: built with
// -fsanitize=address,undefined by tests/test_host_sanitizers.py (CPU only).
#include "render.hpp"
#include <cstdio>
#include <vector>
using namespace rtrace;
int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "/tmp/writer_sanitize.ppm";
    for (int rgb = 0; rgb < 2; ++rgb) {
        FileOrAnyWriter sink; sink.f = fopen(path, "wb"); sink.is_file = true;
        {
            PPMStdoutRGBABufferWriter w(rgb != 0, sink);
            w.begin(200, 130);
            RenderOptions o{200, 130, 1};
            auto bs = Renderer::buckets(o);
            for (size_t k = 0; k < bs.size(); ++k) {
                const ImageRegion &r = bs[(k * 5) % bs.size()];
                std::vector<uint8_t> px(r.area() * 4, (uint8_t)(k + 1));
                w.write_rgba_buffer(RGBABuffer(r, px.data(), RGBABuffer::View{}));
            }
        }
        fclose(sink.f);
        FILE *f = fopen(path, "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fclose(f);
        printf("%s file bytes %ld (expected %ld + header)\n", rgb ? "P6" : "P5", n, (long)200 * 130 * (rgb ? 3 : 1));
    }
    // The device path's bookkeeping (buckets_arrived; rt_render_frame_stream fills pixels() on a GPU box -- here a loop stands in for it):
    // batches of whole bucket rows; a rewrite after the first batch shows the complete rows only and leaves the rest of the file a hole;
    // a second writer reuses the first one's image (the spare) without zeroing it.
    for (int pass = 0; pass < 2; ++pass) {
        FileOrAnyWriter sink; sink.f = fopen(path, "wb"); sink.is_file = true;
        const unsigned W = 200, H = 130;
        long partial = -1;
        {
            PPMStdoutRGBABufferWriter w(true, sink);
            w.begin((uint16_t)W, (uint16_t)H);
            RenderOptions o{(uint16_t)W, (uint16_t)H, 1};
            auto bs = Renderer::buckets(o);                  // 4 x 3 buckets
            const size_t per_row = 4;
            for (size_t first = 0; first < bs.size(); first += per_row) {
                for (size_t k = first; k < first + per_row; ++k)
                    for (unsigned y = bs[k].b; y < bs[k].t; ++y)
                        for (unsigned x = bs[k].l; x < bs[k].r; ++x)
                            for (unsigned c = 0; c < 3; ++c) w.pixels()[((size_t)y * W + x) * 3 + c] = (uint8_t)(x + y + c + pass);
                w.buckets_arrived(bs.data() + first, per_row);      // the first call writes the file (render.rs:427-432)
                if (first == 0) {
                    FILE *f = fopen(path, "rb"); std::vector<uint8_t> got(W * H * 3 + 64); const size_t n = fread(got.data(), 1, got.size(), f); fclose(f);
                    partial = (long)n;
                    const size_t hl = n - (size_t)W * H * 3;
                    for (unsigned y = 0; y < H; ++y)
                        for (unsigned x = 0; x < W * 3; ++x) {
                            const uint8_t want = y < 64 ? (uint8_t)(x / 3 + y + x % 3 + pass) : 0;      // rows below the first bucket row: a hole
                            if (got[hl + (size_t)y * W * 3 + x] != want) { printf("partial rewrite wrong at %u %u\n", x, y); return 1; }
                        }
                }
            }
        }
        fclose(sink.f);
        FILE *f = fopen(path, "rb"); std::vector<uint8_t> got(W * H * 3 + 64); const size_t n = fread(got.data(), 1, got.size(), f); fclose(f);
        const size_t hl = n - (size_t)W * H * 3;
        for (unsigned y = 0; y < H; ++y)
            for (unsigned x = 0; x < W * 3; ++x)
                if (got[hl + (size_t)y * W * 3 + x] != (uint8_t)(x / 3 + y + x % 3 + pass)) { printf("final image wrong at %u %u\n", x, y); return 1; }
        printf("device path pass %d: partial file %ld bytes, final %zu bytes, header %zu\n", pass, partial, n, hl);
    }
    return 0;
}
