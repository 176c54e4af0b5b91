// render.hpp -- host-side mirror of render.rs around the device seam.
//
// Everything here is what STAYS on the host when the backend is dropped in behind the crate: RenderOptions,
// ImageRegion, RGBABuffer, the RGBABufferWriter trait, the PPM writer and the 64x64 bucket scheduler.  The one call
// that moved is the closure body of render.rs:283-294 (RGBABuffer::new + Renderer::render_region): it is now
// rt_render_tiles() on a batch of buckets (include/rtrace_hip.h).  No pixel arithmetic happens on the host.
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <optional>
#include <stdexcept>
#include <typeinfo>
#include <string>
#include <vector>

#include "../../../include/rtrace_hip.h"
#include "scene.hpp"
#include "threadpool.hpp"

namespace rtrace {

struct RenderOptions {                                            // render.rs:33-38
    uint16_t width, height, samples_per_pixel;
};

struct ImageRegion {                                              // render.rs:42-72
    uint16_t l, t, r, b;
    uint16_t width() const { return r - l; }
    uint16_t height() const { return t - b; }
    size_t area() const { return (size_t)width() * (size_t)height(); }
    bool contains(const ImageRegion &o) const { return o.l >= l && o.b >= b && o.t <= t && o.r <= r; }
    size_t buffer_offset(uint16_t x, uint16_t y) const { return (size_t)(y - b) * width() + (size_t)(x - l); }
    bool operator==(const ImageRegion &o) const { return l == o.l && t == o.t && r == o.r && b == o.b; }
};

class RGBABuffer {                                                // render.rs:74-135
public:
    struct View {};                                               // tag: the buffer does not own its bytes
    explicit RGBABuffer(const ImageRegion &r) : buf_(r.area() * components(), 0), reg_(r) {}
    RGBABuffer(const ImageRegion &r, const uint8_t *src) : buf_(src, src + r.area() * components()), reg_(r) {}
    // A bucket as the device delivered it (pinned staging of rt_render_tiles_stream): valid while the writer is being called with
    // it, which is all the reference's consumer loop needs (render.rs:301-307 drops each buffer after write_rgba_buffer).
    RGBABuffer(const ImageRegion &r, const uint8_t *src, View) : view_(src), reg_(r) {}
    static size_t components() { return 4; }
    const uint8_t *data() const { return view_ ? view_ : buf_.data(); }
    size_t size() const { return reg_.area() * components(); }

    void set_pixels_from_buffer(const RGBABuffer &b)              // render.rs:112-126
    {
        if (!reg_.contains(b.reg_)) throw std::runtime_error("assertion failed: self.reg.contains(&b.reg)");
        if (view_) throw std::runtime_error("set_pixels_from_buffer on a view");
        const size_t w = (size_t)b.reg_.width() * components();
        const uint8_t *src = b.data();
        if (reg_ == b.reg_) { buf_.assign(src, src + b.size()); return; }
        for (uint16_t y = b.reg_.b; y < b.reg_.t; ++y) {
            const size_t bl = reg_.buffer_offset(b.reg_.l, y) * components();
            const size_t their = b.reg_.buffer_offset(b.reg_.l, y) * components();
            std::copy(src + their, src + their + w, buf_.begin() + bl);
        }
    }
    const std::vector<uint8_t> &buffer() const { return buf_; }   // owned buffers only
    const ImageRegion &region() const { return reg_; }

private:
    std::vector<uint8_t> buf_;
    const uint8_t *view_ = nullptr;
    ImageRegion reg_;
};

struct RGBABufferWriter {                                         // trait, render.rs:20-30
    virtual ~RGBABufferWriter() = default;
    virtual void begin(uint16_t x, uint16_t y) = 0;
    virtual void write_rgba_buffer(const RGBABuffer &buffer) = 0;
};

struct FileOrAnyWriter {                                          // render.rs:313-316
    FILE *f = nullptr;
    bool is_file = false;
};

class PPMStdoutRGBABufferWriter : public RGBABufferWriter {       // render.rs:319-434
public:
    PPMStdoutRGBABufferWriter(bool write_rgb, FileOrAnyWriter &out) : out_(out), rgb_(write_rgb) {}
    ~PPMStdoutRGBABufferWriter() override                         // Drop, render.rs:331-335
    {
        try { write_buffer_with_header(); } catch (...) {}
        release_image();
    }
    PPMStdoutRGBABufferWriter(const PPMStdoutRGBABufferWriter &) = delete;
    PPMStdoutRGBABufferWriter &operator=(const PPMStdoutRGBABufferWriter &) = delete;

    void begin(uint16_t x, uint16_t y) override                   // render.rs:411-420
    {
        width_ = x; height_ = y;
        // The reference keeps the frame as an RGBABuffer and turns it into P6 / P5 bytes on every write of the file
        // (render.rs:373-401).  Here the frame is kept in the file's own pixel format: a bucket is converted once, when it
        // arrives (while the device is still rendering the next ones), and a write of the file is the header plus one write.
        allocate_image((size_t)x * y * (rgb_ ? 3 : 1));
        rows_with_data_ = 0;
        zero_pending_ = true;                                     // (the image is zeroed when the first bucket is blitted on this thread; a frame the
        row_cover_.assign(y, 0); complete_rows_ = 0;              // device delivers is never zeroed: its writes cover complete rows only)
        emptied_ = false; full_length_ = false;                   // a second frame through the same writer: its first write empties the file again,
                                                                  // so rows it has not reached read as zeros, never as the previous frame's (render.rs:366)
    }
    void write_rgba_buffer(const RGBABuffer &buffer) override     // render.rs:422-433
    {
        blit_encoded(buffer);                                     // set_pixels_from_buffer (render.rs:112-126) + the conversion of render.rs:392-399
        buffer_dirty_ = true;
        const auto now = std::chrono::steady_clock::now();
        if (out_.is_file && (!last_written_at_ || *last_written_at_ + std::chrono::seconds(1) <= now)) {
            last_written_at_ = now;
            write_buffer_with_header();
        }
    }
    void write_buffer_with_header();                              // render.rs:359-407

    // The device side of the writer (not in the reference): the image is kept in the file's own pixel format in memory the GPU can write
    // (rt_host_alloc), so Renderer::render lets the device convert and place the buckets (rt_render_frame_stream) and tells the writer
    // which ones have arrived -- the same bookkeeping as write_rgba_buffer (dirty flag, the once-per-second rewrite), without its copy.
    uint8_t *pixels() { return pixels_; }                         // width * height * (rgb ? 3 : 1) bytes, row-major, 4-byte aligned
    rt_frame_format frame_format() const { return rgb_ ? RT_FRAME_RGB : RT_FRAME_GREY; }
    virtual void buckets_arrived(const ImageRegion *regions, size_t n);
    // Renderer::render takes the device path only for THIS class: a subclass that overrides write_rgba_buffer to see every bucket keeps
    // seeing them (it may opt in by overriding this together with buckets_arrived)
    virtual bool accepts_device_frames() const { return typeid(*this) == typeid(PPMStdoutRGBABufferWriter); }

private:
    void blit_encoded(const RGBABuffer &b);
    static bool positioned_write(int fd, const uint8_t *p, size_t n);
    void allocate_image(size_t pixel_bytes);
    void release_image();
    FileOrAnyWriter &out_;
    std::optional<uint16_t> width_, height_;
    // the file as it is written: [pad][header][pixels], `pixels_` 4-byte aligned, header right in front of it -- ONE write per rewrite
    uint8_t *image_ = nullptr, *pixels_ = nullptr;
    size_t image_cap_ = 0, pixel_bytes_ = 0, header_len_ = 0;
    bool image_pinned_ = false;
    bool rgb_;
    std::optional<std::chrono::steady_clock::time_point> last_written_at_;
    bool buffer_dirty_ = false;
    bool emptied_ = false;                                        // this writer has emptied the file once (render.rs:366 does it on every write)
    bool full_length_ = false;                                    // ... and given it its final length
    uint32_t rows_with_data_ = 0;                                 // rows [0, n) of the image may hold pixels; the rest is still zero
    bool zero_pending_ = false;                                   // begin() has not zeroed the image yet
    std::vector<uint32_t> row_cover_;                             // device path: pixels of row y that have arrived
    uint32_t complete_rows_ = 0;                                  // ... rows [0, n) have arrived completely
};

// Device copies of a Scene on one GPU (replaces handing Arc<Scene> to the pool threads, render.rs:279).
class DeviceScene {
public:
    DeviceScene(const Scene &scene, int device);
    ~DeviceScene() { rt_scene_destroy(h_); }
    DeviceScene(const DeviceScene &) = delete;
    DeviceScene &operator=(const DeviceScene &) = delete;
    rt_scene *handle() const { return h_; }

private:
    rt_scene *h_ = nullptr;
};

// The Scene on several GPUs of this node (rt_gang: one process, buckets dealt round-robin, one RCCL gather to the root GPU).
class DeviceGang {
public:
    DeviceGang(const Scene &scene, const std::vector<int> &devices);
    // nullptr + *status when the gang cannot be created (RT_ERR_UNSUPPORTED: no RCCL on this machine)
    static std::shared_ptr<DeviceGang> try_create(const Scene &scene, const std::vector<int> &devices, rt_status *status);
    explicit DeviceGang(rt_gang *h) : h_(h) {}
    ~DeviceGang() { rt_gang_destroy(h_); }
    DeviceGang(const DeviceGang &) = delete;
    DeviceGang &operator=(const DeviceGang &) = delete;
    rt_gang *handle() const { return h_; }

private:
    rt_gang *h_ = nullptr;
};

struct RenderStats {
    uint64_t primary = 0, hits = 0, shadow = 0, occluded = 0, sphere_tests = 0, bound_tests = 0;
    double device_ms = 0;
};

struct Backend {                                                  // additions that do not exist in the reference
    std::vector<std::shared_ptr<DeviceScene>> devices;            // buckets are dealt round-robin over these (host buffers per device)
    std::shared_ptr<DeviceGang> gang;                             // set: the frame goes through rt_gang_render_frame instead (RCCL gather)
    rt_traversal traversal = RT_TRAVERSAL_SKIP;
    size_t buckets_per_call = 0;                                  // 0 = split each device's buckets evenly over the pool (<= 64 per call)
    bool want_stats = false;                                      // collect RenderStats (ray / test counters, device time)
    bool strict_64 = false;                                       // reproduce assert!(w % 64 == 0 && h % 64 == 0), render.rs:265-266
};

struct Renderer {
    // render.rs:260-310.  `pool` keeps the reference's meaning (RTRACEMAXPROCS / --num-cores host scheduler threads).
    static RenderStats render(const RenderOptions &o, const Backend &backend, RGBABufferWriter &writer, ThreadPool &pool);
    static std::vector<ImageRegion> buckets(const RenderOptions &o);
};

}  // namespace rtrace
