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
    return 0;
}
