#include "render.hpp"

#include <unistd.h>

#include <cstring>
#include <exception>
#include <mutex>

namespace rtrace {

static constexpr uint16_t CHUNK_SIZE = 64;                        // render.rs:264
static constexpr size_t kMaxBucketsPerCall = 64, kMinBucketsPerCall = 16;
static constexpr size_t kDeviceEncodeFromPixels = (size_t)6 << 20;     // Renderer::render: frames from here on arrive in the file's pixel format, converted by the device      // per device call (a pass of < 16 buckets leaves most CUs idle)

static void check(rt_status st, const char *what)
{
    if (st != RT_OK)
        throw std::runtime_error(std::string(what) + ": " + rt_strerror(st) + " -- " + rt_last_error_message());
}

DeviceScene::DeviceScene(const Scene &scene, int device)
{
    const FlatScene f = scene.flatten();
    const RFloat light[3] = { scene.directional_light.x, scene.directional_light.y, scene.directional_light.z };
    const RFloat eye[3] = { scene.eye.x, scene.eye.y, scene.eye.z };
    const rt_precision prec = sizeof(RFloat) == 4 ? RT_F32 : RT_F64;
    static_assert(sizeof(rt_range) == 2 * sizeof(int32_t), "ranges are {first, count} pairs");
    check(rt_scene_create(device, prec, f.items.data(), (uint32_t)(f.items.size() / 4), light, eye, f.bounds.data(),
                          reinterpret_cast<const rt_range *>(f.ranges.data()), (uint32_t)(f.ranges.size() / 2), &h_),
          "rt_scene_create");
}

static rt_status gang_create(const Scene &scene, const std::vector<int> &devices, rt_gang **h)
{
    const FlatScene f = scene.flatten();
    const RFloat light[3] = { scene.directional_light.x, scene.directional_light.y, scene.directional_light.z };
    const RFloat eye[3] = { scene.eye.x, scene.eye.y, scene.eye.z };
    const rt_precision prec = sizeof(RFloat) == 4 ? RT_F32 : RT_F64;
    return rt_gang_create(devices.data(), (int)devices.size(), prec, f.items.data(), (uint32_t)(f.items.size() / 4), light, eye, f.bounds.data(),
                          reinterpret_cast<const rt_range *>(f.ranges.data()), (uint32_t)(f.ranges.size() / 2), h);
}

DeviceGang::DeviceGang(const Scene &scene, const std::vector<int> &devices) { check(gang_create(scene, devices, &h_), "rt_gang_create"); }

std::shared_ptr<DeviceGang> DeviceGang::try_create(const Scene &scene, const std::vector<int> &devices, rt_status *status)
{
    rt_gang *h = nullptr;
    *status = gang_create(scene, devices, &h);
    return *status == RT_OK ? std::make_shared<DeviceGang>(h) : nullptr;
}

// RGBA -> RGB for n pixels (alpha dropped, render.rs:392-396).  The SSSE3 body moves 4 pixels per shuffle.
__attribute__((target("ssse3"))) static void rgba_to_rgb_ssse3(const uint8_t *b, uint8_t *w, size_t n)
{
    typedef char v16 __attribute__((vector_size(16)));
    const v16 mask = { 0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14, -1, -1, -1, -1 };
    size_t i = 0;
    for (; i + 8 <= n; i += 4, b += 16, w += 12) {            // the 16-byte store writes 4 bytes past the 12 it means: keep a margin
        v16 v;
        memcpy(&v, b, 16);
        v = __builtin_ia32_pshufb128(v, mask);
        memcpy(w, &v, 16);
    }
    for (; i < n; ++i, b += 4, w += 3) { w[0] = b[0]; w[1] = b[1]; w[2] = b[2]; }
}

static void rgba_to_rgb(const uint8_t *b, uint8_t *w, size_t n)
{
    static const bool ssse3 = __builtin_cpu_supports("ssse3");
    if (ssse3) { rgba_to_rgb_ssse3(b, w, n); return; }
    for (size_t i = 0; i < n; ++i, b += 4, w += 3) { w[0] = b[0]; w[1] = b[1]; w[2] = b[2]; }
}

// Pinning memory costs about a millisecond per 6 MB and so does giving it back: a process that writes frame after frame (a new writer per
// frame, like seam_bench or a viewer) keeps ONE released image around for the next writer.
namespace {
struct SpareImage {
    std::mutex mu;
    uint8_t *p = nullptr; size_t cap = 0; bool pinned = false;
    ~SpareImage() { if (p) { if (pinned) rt_host_free(p); else free(p); } }
} g_spare;
}  // namespace

void PPMStdoutRGBABufferWriter::allocate_image(size_t pixel_bytes)
{
    char header[64];
    const int hl = snprintf(header, sizeof header, "%s\n%u %u\n255\n", rgb_ ? "P6" : "P5", (unsigned)*width_, (unsigned)*height_);
    const size_t front = ((size_t)hl + 3) & ~(size_t)3, need = front + pixel_bytes + 16;      // (+ 16: the SSSE3 conversion stores whole vectors)
    if (image_cap_ < need) {
        release_image();
        std::lock_guard<std::mutex> lk(g_spare.mu);
        if (g_spare.p && g_spare.cap >= need) { image_ = g_spare.p; image_cap_ = g_spare.cap; image_pinned_ = g_spare.pinned; g_spare.p = nullptr; g_spare.cap = 0; }
    }
    if (image_cap_ < need) {
        void *p = nullptr;
        if (rt_host_alloc(need, &p) == RT_OK) image_pinned_ = true;          // memory the device writes directly; without a device: plain memory
        else { p = malloc(need); image_pinned_ = false; }
        if (!p) throw std::bad_alloc();
        image_ = static_cast<uint8_t *>(p);
        image_cap_ = need;
    }
    pixels_ = image_ + front;
    pixel_bytes_ = pixel_bytes;
    header_len_ = (size_t)hl;
    memcpy(pixels_ - hl, header, (size_t)hl);
}

void PPMStdoutRGBABufferWriter::release_image()
{
    if (!image_) return;
    {
        std::lock_guard<std::mutex> lk(g_spare.mu);
        if (image_cap_ > g_spare.cap) { std::swap(image_, g_spare.p); std::swap(image_cap_, g_spare.cap); std::swap(image_pinned_, g_spare.pinned); }
    }
    if (image_) { if (image_pinned_) rt_host_free(image_); else free(image_); }
    image_ = pixels_ = nullptr; image_cap_ = 0;
}

void PPMStdoutRGBABufferWriter::buckets_arrived(const ImageRegion *regions, size_t n)
{
    if (!width_ || !height_) throw std::runtime_error("begin() called");
    for (size_t i = 0; i < n; ++i) {
        if (regions[i].r > *width_ || regions[i].t > *height_) throw std::runtime_error("assertion failed: self.reg.contains(&b.reg)");
        rows_with_data_ = std::max<uint32_t>(rows_with_data_, regions[i].t);
        for (uint16_t y = regions[i].b; y < regions[i].t; ++y) row_cover_[y] += regions[i].width();
    }
    // The device wrote these buckets into an image nobody zeroed (6 MB of memset per frame for nothing when every pixel arrives): a
    // rewrite of the file shows the rows that have arrived COMPLETELY -- Renderer::render's batches are whole bucket rows, so that is
    // every bucket that has arrived -- and leaves the rest of the file a hole (zeros), like rows no bucket has reached.
    while (complete_rows_ < *height_ && row_cover_[complete_rows_] >= *width_) ++complete_rows_;
    buffer_dirty_ = true;
    const auto now = std::chrono::steady_clock::now();            // render.rs:427-432
    if (out_.is_file && (!last_written_at_ || *last_written_at_ + std::chrono::seconds(1) <= now)) {
        last_written_at_ = now;
        write_buffer_with_header();
    }
}

void PPMStdoutRGBABufferWriter::blit_encoded(const RGBABuffer &b)
{
    const ImageRegion &r = b.region();
    if (!width_ || !height_ || r.r > *width_ || r.t > *height_) throw std::runtime_error("assertion failed: self.reg.contains(&b.reg)");
    const size_t bpp = rgb_ ? 3 : 1, pitch = (size_t)*width_ * bpp;
    if (zero_pending_) { memset(pixels_, 0, pixel_bytes_); zero_pending_ = false; }      // pixels no bucket has reached read as zeros
    const uint8_t *src = b.data();
    rows_with_data_ = std::max<uint32_t>(rows_with_data_, r.t);
    for (uint16_t y = r.b; y < r.t; ++y, src += (size_t)r.width() * 4) {
        uint8_t *dst = pixels_ + (size_t)y * pitch + (size_t)r.l * bpp;
        if (rgb_) rgba_to_rgb(src, dst, r.width());
        else for (size_t i = 0; i < r.width(); ++i) dst[i] = (uint8_t)(((float)src[4 * i] + (float)src[4 * i + 1] + (float)src[4 * i + 2]) / 3.0f);      // render.rs:399
    }
}

// header + pixels in ONE positioned write.  (Writing a large image from several threads side by side was tried: on tmpfs it doubles the
// time of the write AND of the truncating fopen before it -- the page cache serialises them; profiles/r05_write_helpers_ab.log.)
bool PPMStdoutRGBABufferWriter::positioned_write(int fd, const uint8_t *p, size_t n)
{
    for (size_t done = 0; done < n;) {
        const ssize_t w = pwrite(fd, p + done, n - done, (off_t)done);
        if (w <= 0) return false;
        done += (size_t)w;
    }
    return true;
}

void PPMStdoutRGBABufferWriter::write_buffer_with_header()
{
    if (!buffer_dirty_) return;
    FILE *out = out_.f;
    if (!width_ || !height_) throw std::runtime_error("begin() called");
    const uint8_t *file_image = pixels_ - header_len_;            // header + pixels, contiguous
    if (!out_.is_file) {
        // a device-delivered frame is never zeroed (buckets_arrived): if it stopped short -- a failed batch, then this Drop write -- the rows
        // that did not arrive completely are memory nobody wrote (a reused image: the previous frame's rows); they read as zeros, like rows
        // no bucket has reached on the other path
        if (zero_pending_ && complete_rows_ < *height_) {
            const size_t pitch = (size_t)*width_ * (rgb_ ? 3 : 1), have = std::min(pixel_bytes_, (size_t)complete_rows_ * pitch);
            memset(pixels_ + have, 0, pixel_bytes_ - have);
        }
        if (fwrite(file_image, 1, header_len_ + pixel_bytes_, out) != header_len_ + pixel_bytes_) throw std::runtime_error("called `Result::unwrap()` on an `Err` value: write");
        fflush(out);
        buffer_dirty_ = false;
        return;
    }
    // set_len(0) + seek(Start(0)) + header + pixels (render.rs:366-407): the same bytes in the file afterwards, with less work for the
    // page cache (two thirds of a 1080p frame's 1.9 ms through the scheduler were these writes: seam_bench's *_parts_ms).  The file is
    // emptied on this writer's FIRST write only -- later writes overwrite in place: same length, no pages to give back and take again --
    // and rows no bucket has reached yet are left as a hole behind the last row that has data: a hole reads as the zeros those rows
    // hold in the image.  1080p, the write that the first batch of buckets triggers: 1.1 of 6.2 MB.  Header and pixels lie back to
    // back in the image, so a rewrite is ONE positioned write.
    fflush(out);
    const int fd = fileno(out);
    if (!emptied_) {
        if (ftruncate(fd, 0) != 0) throw std::runtime_error("called `Result::unwrap()` on an `Err` value: ftruncate");
        emptied_ = true;
    }
    const size_t pitch = (size_t)*width_ * (rgb_ ? 3 : 1);
    const size_t live = header_len_ + std::min(pixel_bytes_, (size_t)(zero_pending_ ? complete_rows_ : rows_with_data_) * pitch);
    if (!positioned_write(fd, file_image, live)) throw std::runtime_error("called `Result::unwrap()` on an `Err` value: write");
    if (!full_length_) {                                          // (rows behind the last one that has data: a hole; later rewrites leave the length alone)
        if (ftruncate(fd, (off_t)(header_len_ + pixel_bytes_)) != 0) throw std::runtime_error("called `Result::unwrap()` on an `Err` value: ftruncate");
        full_length_ = true;
    }
    fseek(out, 0, SEEK_END);
    buffer_dirty_ = false;
}

std::vector<ImageRegion> Renderer::buckets(const RenderOptions &o)
{
    // render.rs:273-298: row-major, y outer.  The reference asserts w % 64 == 0 && h % 64 == 0 (render.rs:265-266);
    // edge buckets are clipped here instead so that 800x600 and 1920x1080 render (a pixel does not depend on its bucket).
    std::vector<ImageRegion> out;
    for (uint32_t y = 0; y < o.height; y += CHUNK_SIZE)
        for (uint32_t x = 0; x < o.width; x += CHUNK_SIZE)
            out.push_back(ImageRegion{ (uint16_t)x, (uint16_t)std::min<uint32_t>(y + CHUNK_SIZE, o.height),
                                       (uint16_t)std::min<uint32_t>(x + CHUNK_SIZE, o.width), (uint16_t)y });
    return out;
}

RenderStats Renderer::render(const RenderOptions &o, const Backend &be, RGBABufferWriter &writer, ThreadPool &pool)
{
    if (be.devices.empty() && !be.gang) throw std::runtime_error("no device scene: the HIP backend has no CPU fallback");
    if (be.strict_64 && (o.width % CHUNK_SIZE != 0 || o.height % CHUNK_SIZE != 0))                 // render.rs:265-266
        throw std::runtime_error("TODO: handle chunk sizes");
    writer.begin(o.width, o.height);

    const std::vector<ImageRegion> all = buckets(o);
    size_t count = all.size();
    // width or height 0: the scheduler's loops (render.rs:273-298) produce no bucket, the writer never becomes dirty and its Drop writes
    // nothing (render.rs:361-363): an empty file, exit code 0.  (The C ABI rejects an empty tile list.)
    if (all.empty()) return RenderStats{};
    static_assert(sizeof(ImageRegion) == sizeof(rt_region), "ImageRegion is layout-compatible with rt_region");
    if (be.gang) {
        // Several GPUs: gang calls of at most 64 buckets per GPU (bucket i -> GPU i % N, one RCCL gather of the u8 shards to the
        // root GPU, blit there -- straight into this pinned frame); after every call its buckets go to the writer, like the
        // channel's consumer loop: a long render still delivers tiles while it runs (render.rs:301-307, 427-432).
        const rt_options gopts{ o.width, o.height, o.samples_per_pixel };
        const size_t frame_bytes = (size_t)o.width * o.height * 4;
        void *pinned = nullptr;
        check(rt_host_alloc(frame_bytes, &pinned), "rt_host_alloc");
        struct Free { void *p; ~Free() { rt_host_free(p); } } free_frame{ pinned };
        uint8_t *frame = static_cast<uint8_t *>(pinned);
        int n_gpus = 1;
        check(rt_gang_size(be.gang->handle(), &n_gpus), "rt_gang_size");
        const size_t per_call = be.buckets_per_call ? be.buckets_per_call : kMaxBucketsPerCall * (size_t)n_gpus;
        RenderStats total;
        std::vector<uint8_t> tile;
        for (size_t first = 0; first < all.size(); first += per_call) {
            const size_t cnt = std::min(per_call, all.size() - first);
            rt_stats st{};
            check(rt_gang_render_frame(be.gang->handle(), &gopts, be.traversal, reinterpret_cast<const rt_region *>(all.data() + first), (uint32_t)cnt,
                                       frame, be.want_stats ? &st : nullptr),
                  "rt_gang_render_frame");
            total.primary += st.primary; total.hits += st.hits; total.shadow += st.shadow; total.occluded += st.occluded;
            total.sphere_tests += st.sphere_tests; total.bound_tests += st.bound_tests; total.device_ms += st.device_ms;
            for (size_t i = first; i < first + cnt; ++i) {
                const ImageRegion &r = all[i];
                tile.resize(r.area() * 4);
                for (uint16_t y = r.b; y < r.t; ++y)
                    memcpy(tile.data() + (size_t)(y - r.b) * r.width() * 4, frame + ((size_t)y * o.width + r.l) * 4, (size_t)r.width() * 4);
                writer.write_rgba_buffer(RGBABuffer(r, tile.data(), RGBABuffer::View{}));
                count -= 1;
            }
        }
        if (count != 0) throw std::runtime_error("We really should have processed all chunks here");
        return total;
    }
    if (be.devices.size() == 1 && !be.want_stats && be.buckets_per_call == 0) {
        // One GPU, no counters: the whole bucket list in ONE streaming call.  The device renders batch after batch into pinned
        // staging and the callback -- this thread, the channel's consumer (render.rs:301-307) -- receives each bucket as a view of
        // that staging as soon as its batch is complete, while later batches are still rendering: no per-batch host call, no
        // intermediate copies.  (The pool keeps its meaning for the paths below; here the producers are the GPU's workgroups.)
        const rt_options opts1{ o.width, o.height, o.samples_per_pixel };
        // (from kDeviceEncodeFromPixels on: measured end to end, frame + file, same box -- 4096 x 4096: 14.9 ms against 19.6 ms with the
        // conversion on this thread; 2560 x 1440: 3.0 against 2.6, 1920 x 1080: 1.6 against 1.2 -- the file write reads what the GPU wrote
        // from DRAM, what this thread converted from its cache, and below a few million pixels that outweighs the conversion;
        // profiles/r05_end_to_end_encoders.log)
        if (auto *ppm = dynamic_cast<PPMStdoutRGBABufferWriter *>(&writer);
            ppm && ppm->accepts_device_frames() && ppm->pixels() && (size_t)o.width * o.height >= kDeviceEncodeFromPixels) {
            // The library's own writer keeps its image in the file's pixel format in memory the GPU can write: the device converts and
            // places the buckets itself (rt_render_frame_stream: 6.2 MB of P6 payload over PCIe for a 1080p frame instead of 8.3 MB of RGBA,
            // no conversion on this thread), and the writer is told which buckets have arrived -- its first write of the file after the
            // first batch, the once-per-second rewrite, the final write on Drop stay what they are (render.rs:427-432, 331-335).
            struct Arrived { PPMStdoutRGBABufferWriter *writer; const ImageRegion *all; size_t delivered = 0; std::exception_ptr err; } ctx{ ppm, all.data(), 0, nullptr };
            check(rt_render_frame_stream(be.devices[0]->handle(), &opts1, be.traversal, reinterpret_cast<const rt_region *>(all.data()), (uint32_t)all.size(),
                                         ppm->frame_format(), ppm->pixels(),
                                         [](void *user, uint32_t first, uint32_t n) {
                                             Arrived *c = static_cast<Arrived *>(user);
                                             if (c->err) return;
                                             try { c->writer->buckets_arrived(c->all + first, n); c->delivered += n; } catch (...) { c->err = std::current_exception(); }
                                         }, &ctx),
                  "rt_render_frame_stream");
            if (ctx.err) std::rethrow_exception(ctx.err);
            if (ctx.delivered != all.size()) throw std::runtime_error("We really should have processed all chunks here");
            return RenderStats{};
        }
        struct Ctx { RGBABufferWriter *writer; size_t delivered = 0; std::exception_ptr err; } ctx{ &writer, 0, nullptr };
        check(rt_render_tiles_stream(be.devices[0]->handle(), &opts1, be.traversal, reinterpret_cast<const rt_region *>(all.data()), (uint32_t)all.size(),
                                     [](void *user, uint32_t, const rt_region *region, const uint8_t *rgba) {
                                         Ctx *c = static_cast<Ctx *>(user);
                                         if (c->err) return;                       // a failing writer: keep draining, report at the end
                                         try {
                                             c->writer->write_rgba_buffer(RGBABuffer(ImageRegion{ region->l, region->t, region->r, region->b }, rgba, RGBABuffer::View{}));
                                             c->delivered += 1;
                                         } catch (...) { c->err = std::current_exception(); }
                                     }, &ctx),
              "rt_render_tiles_stream");
        if (ctx.err) std::rethrow_exception(ctx.err);
        if (ctx.delivered != all.size()) throw std::runtime_error("We really should have processed all chunks here");
        return RenderStats{};
    }
    // Deal buckets round-robin over the devices, then cut each device's list into batches: one rt_render_tiles call
    // per batch (a launch per 64x64 bucket would leave 255 of 256 CUs idle).
    const size_t ndev = be.devices.size();
    struct Batch { size_t dev; std::vector<ImageRegion> regs; };
    std::vector<Batch> batches;
    for (size_t d = 0; d < ndev; ++d) {
        std::vector<ImageRegion> mine;
        for (size_t i = d; i < all.size(); i += ndev) mine.push_back(all[i]);
        const size_t calls = std::max<size_t>(1, (pool.size() + ndev - 1) / ndev);
        // at most kMaxBucketsPerCall per device call: enough to fill 256 CUs (64 buckets = 1,024 workgroups), few enough that
        // finished buckets keep reaching the writer during a long render (tiles in completion order, the once-per-second
        // rewrite of the file: render.rs:301-307, 427-432) even with RTRACEMAXPROCS = 1
        const size_t per = be.buckets_per_call ? be.buckets_per_call
                                               : std::clamp((mine.size() + calls - 1) / calls, kMinBucketsPerCall, kMaxBucketsPerCall);
        for (size_t i = 0; i < mine.size(); i += per)
            batches.push_back(Batch{ d, std::vector<ImageRegion>(mine.begin() + i, mine.begin() + std::min(mine.size(), i + per)) });
    }

    // One message per finished bucket plus one end-of-batch marker (which carries the batch's failure, if any).
    struct Msg { std::optional<RGBABuffer> buf; bool batch_end = false; std::exception_ptr err; };
    SyncChannel<Msg> chan(4);                                     // sync_channel::<RGBABuffer>(4)  render.rs:271
    const rt_options opts{ o.width, o.height, o.samples_per_pixel };
    RenderStats total;
    std::mutex stats_mu;

    for (const Batch &b : batches) {
        pool.execute([&, b] {
            std::exception_ptr err;
            try {
                size_t px = 0;
                for (const ImageRegion &r : b.regs) px += r.area();
                std::vector<uint8_t> rgba(px * 4);
                rt_stats st{};
                // counters are opt-in: a counted launch runs the counting flavour of the kernels (same bytes, ~3x slower)
                check(rt_render_tiles(be.devices[b.dev]->handle(), &opts, be.traversal,
                                      reinterpret_cast<const rt_region *>(b.regs.data()), (uint32_t)b.regs.size(), rgba.data(),
                                      be.want_stats ? &st : nullptr),
                      "rt_render_tiles");
                if (be.want_stats) {
                    std::lock_guard<std::mutex> lk(stats_mu);
                    total.primary += st.primary; total.hits += st.hits; total.shadow += st.shadow; total.occluded += st.occluded;
                    total.sphere_tests += st.sphere_tests; total.bound_tests += st.bound_tests; total.device_ms += st.device_ms;
                }
                size_t off = 0;
                for (const ImageRegion &r : b.regs) {
                    chan.send(Msg{ RGBABuffer(r, rgba.data() + off), false, nullptr });      // tx.send(b)  render.rs:293
                    off += r.area() * 4;
                }
            } catch (...) {
                err = std::current_exception();
            }
            chan.send(Msg{ std::nullopt, true, err });
        });
    }

    // Read the results and pass them to the writer  render.rs:300-307
    std::exception_ptr first_err;
    for (size_t open = batches.size(); open > 0;) {
        Msg m = chan.recv();
        if (m.batch_end) {
            if (m.err && !first_err) first_err = m.err;
            --open;
        } else if (!first_err) {
            // a failing writer (ftruncate / fwrite) must not unwind past the pool jobs: they hold references to this frame's
            // locals and may be blocked in chan.send -- remember the failure and keep draining until every batch has ended
            try {
                writer.write_rgba_buffer(*m.buf);
                count -= 1;
            } catch (...) {
                first_err = std::current_exception();
            }
        }
    }
    if (first_err) std::rethrow_exception(first_err);             // the reference panics here
    if (count != 0) throw std::runtime_error("We really should have processed all chunks here");   // render.rs:308-309
    return total;
}

}  // namespace rtrace
