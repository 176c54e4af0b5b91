/*
 * rtrace_hip.h -- C ABI of the MI355X (gfx950) backend for rust-tracer's per-tile hot path.
 *
 * Drop-in seam (there is no FFI in the reference today; this is the natural one):
 * the body of the closure the bucket scheduler hands to its thread pool,
 *
 *     /root/reference/src/rust/render.rs:283-294
 *         let mut b = RGBABuffer::new(&ImageRegion{l: x, r: x + 64, b: y, t: y + 64});
 *         Renderer::render_region(&opts, tscene.deref(), &mut b);          // render.rs:218-255
 *         tx.send(b)
 *
 * i.e. `pub fn render_region(o: &RenderOptions, scene: &Scene, buf: &mut RGBABuffer)` and everything it
 * calls (Renderer::raytrace render.rs:171-215, TypedGroup::intersect group.rs:72-83,
 * Sphere::intersect / distance_from_ray primitive.rs:55-84, RGBABuffer::set_pixel_from_vector render.rs:92-109).
 * A Rust maintainer binds these symbols from an `extern "C"` block (INTEGRATION.md shows the block);
 * plain pointers and sizes only, no C++ or torch types, never unwinds, returns integer status codes
 * where the reference panics.
 *
 * Results: the RGBA bytes are bit-identical to the reference CPU path for the same Scene / RenderOptions /
 * ImageRegion (f32; every + - * / sqrt individually rounded, no FMA contraction, reference operation order).
 *
 * Threading: every entry point may be called concurrently from several host threads on one rt_scene*
 * (the reference calls render_region from up to RTRACEMAXPROCS pool threads, render.rs:283); each call
 * works on its own HIP stream and workspace.
 */
#ifndef RTRACE_HIP_H
#define RTRACE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTRACE_HIP_ABI_VERSION 4

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = 1,  /* NULL pointer, n == 0, width or height 0, non-finite / non-positive-radius sphere ... (NOT
                                     samples_per_pixel == 0: that is the reference's black frame, see rt_options)           */
    RT_ERR_INVALID_REGION = 2,    /* region outside the image or t <= b / r <= l  (reference: assert!/index panic)  */
    RT_ERR_NO_DEVICE = 3,         /* no gfx950 device visible, or device index out of range                        */
    RT_ERR_HIP = 4,               /* a HIP runtime call or kernel launch failed; rt_last_error_message() has detail */
    RT_ERR_OUT_OF_MEMORY = 5,
    RT_ERR_UNSUPPORTED = 6        /* e.g. RT_TRAVERSAL_SKIP on a scene created without subtree bounds              */
} rt_status;

/* `pub type RFloat = f32` (vec.rs:6).  RT_F64 is the type-alias swap of BASELINE config 3 (every hard-coded
 * f32 constant promoted); the reference pins nothing for it. */
typedef enum rt_precision { RT_F32 = 0, RT_F64 = 1 } rt_precision;

/* How a ray visits the items.  Both give the reference's pixels on scenes whose bounds enclose their
 * subtrees and whose eye lies outside every bound (the default scene).
 *   FLAT : every item in DFS order, strict `<` nearest for primary rays, any-hit for shadow rays.
 *   SKIP : the same DFS array plus the reference's own group bounds as skip ranges, culled per ray with the
 *          reference's rule `bound.distance_from_ray(ray) >= hit.distance` (group.rs:73) -- reproduces the
 *          hierarchy exactly, including its inside-the-bound behaviour. */
typedef enum rt_traversal { RT_TRAVERSAL_FLAT = 0, RT_TRAVERSAL_SKIP = 1 } rt_traversal;

/* RenderOptions, render.rs:33-38 (u16 fields there too).  samples_per_pixel = k means k*k samples; k = 0 is the reference's black
 * frame (no sample taken, 0 * inf = NaN, `NaN as u8` = 0: every listed pixel {0, 0, 0, 0}).  width and height must be >= 1 at this
 * boundary: an empty image has no bucket to hand over (the host scheduler returns before calling, like render.rs:273-298). */
typedef struct rt_options {
    uint16_t width, height, samples_per_pixel;
} rt_options;

/* ImageRegion, render.rs:42-48: pixels x in [l, r), y in [b, t); y = 0 is the TOP image row and row 0 of
 * the tile buffer is y == b (buffer_offset render.rs:69-71). */
typedef struct rt_region {
    uint16_t l, t, r, b;
} rt_region;

/* A group's bound and the contiguous range of DFS item indices of its subtree (for RT_TRAVERSAL_SKIP). */
typedef struct rt_range {
    int32_t first, count;
} rt_range;

/* Ray counters with the reference's meaning (they must equal the CPU path's exactly) + device time. */
typedef struct rt_stats {
    uint64_t primary;       /* primary samples traced = sum(area) * spp^2                    */
    uint64_t hits;          /* primary rays that hit an item                                 */
    uint64_t shadow;        /* shadow rays cast (hit and n.light < 0), render.rs:199-207      */
    uint64_t occluded;      /* shadow rays that hit something                                */
    uint64_t sphere_tests;  /* ray x item tests (FLAT: rays * n_items; SKIP: Sphere::intersect calls the
                               reference's traversal makes, shadow rays stopping at their first hit)        */
    uint64_t bound_tests;   /* bound.distance_from_ray calls (group.rs:73); 0 for FLAT          */
    uint64_t tests_executed;/* ray x record tests the kernels actually ran: SKIP = sphere_tests + bound_tests;
                               FLAT = primary * n_items + (shadow rays) * first chunk + (survivors) * rest --
                               the any-hit passes stop early, so FLAT's figure is below sphere_tests         */
    uint64_t primary_tests; /* of sphere_tests + bound_tests (FLAT: of sphere_tests), the tests made for PRIMARY rays; the rest were made
                               for shadow rays (8 vs 16 arithmetic operations per test with the ray-independent terms pre-formed)  */
    double device_ms;       /* hipEvent time of all kernels of this call on its stream.  A call that asks for
                               stats runs the counting flavour of the kernels (same bytes, about 2.5x slower):
                               it is not the product's speed -- time calls without stats with your own events */
    uint64_t longest_wave_cycles, longest_wave_ref100mhz;   /* (ABI 4; SKIP) the counting launch's longest wave on the shader clock (s_memtime) and on the
                               constant 100 MHz reference (s_memrealtime): cycles / ref100mhz * 100 = the clock in MHz that launch ran at.
                               Diagnostic -- what bench.py prices its roofline's peak at next to the nominal 2.4 GHz; 0 for FLAT */
} rt_stats;

typedef struct rt_scene rt_scene;   /* opaque: device copies of a Scene (render.rs:138-142) */

/* (ABI 4; host-side, touches no device) The automatic bounding-sphere hierarchy for an ARBITRARY sphere list -- SURVEY.md 8f.4; the reference has
 * no such builder (group.rs:28-65 builds its pyramid only), so this is an input generator, not reference arithmetic: median splits along the
 * longest axis until at most leaf_size spheres remain, near-minimal enclosing spheres as bounds, the half nearer to `eye` first (eye = NULL:
 * the split's own order).  The one implementation both hosts use (csrc/host/hierarchy.hpp).
 *   spheres     double[4 * n]: cx, cy, cz, radius
 *   items_out   REAL[4 * n]  in the tree's DFS order;  order_out uint64[n] (optional): items_out[k] = spheres[order_out[k]]
 *   bounds_out  REAL[4 * 2n], ranges_out rt_range[2n]: room for the at most 2n - 1 groups; *n_groups_out of them are written (DFS pre-order)
 * ready for rt_scene_create(dfs_items = items_out, bounds = bounds_out, ranges = ranges_out, n_bounds = *n_groups_out). */
rt_status rt_build_hierarchy(const double *spheres, uint32_t n, uint32_t leaf_size, const double *eye, rt_precision precision,
                             void *items_out, void *bounds_out, rt_range *ranges_out, uint64_t *order_out, uint32_t *n_groups_out);

/* Number of usable devices (RT_ERR_NO_DEVICE and *n = 0 when there is none). */
rt_status rt_device_count(int *n);

/* Uploads a Scene.  Replaces Arc<Scene> construction for the device side (render.rs:144-166, main.rs:23).
 *   dfs_items   REAL[4*n_items] = {cx,cy,cz,radius} in traversal (DFS, insertion) order -- the order
 *               TypedGroup::intersect visits Pair::Item children (group.rs:77-82); ties between items at
 *               exactly equal distance go to the first in this order (primitive.rs:79).
 *   light_unit  REAL[3] Scene::directional_light, already normalised by the host in REAL (render.rs:154-159).
 *   eye         REAL[3] Scene::eye.
 *   bounds/ranges/n_bounds  optional (NULL/NULL/0): group bounds REAL[4*n_bounds] with their item ranges in DFS
 *               pre-order (outer group before the groups nested in it); required for RT_TRAVERSAL_SKIP.
 * REAL is float for RT_F32 and double for RT_F64.  All values must be finite, |coordinate| <= 1e15 (items, bounds, eye),
 * radius > 0, light_unit of squared length within 2e-3 of 1 (RT_ERR_INVALID_ARGUMENT otherwise): within these bounds no
 * intermediate of the path overflows, so no NaN can arise and every comparison agrees with the reference's whichever way it is written.
 * The caller keeps ownership of every host buffer; nothing is retained but the returned handle. */
rt_status rt_scene_create(int device, rt_precision precision,
                          const void *dfs_items, uint32_t n_items,
                          const void *light_unit, const void *eye,
                          const void *bounds, const rt_range *ranges, uint32_t n_bounds,
                          rt_scene **out);

rt_status rt_scene_destroy(rt_scene *scene);

/* What rt_scene_create found out about the scene (diagnostic; selects nothing the caller has to know about).
 *   RT_SCENE_HAS_BOUNDS      created with subtree bounds: RT_TRAVERSAL_SKIP is available
 *   RT_SCENE_CONCENTRIC      every group bound is directly followed by an item with the same centre, bit for bit (the
 *                            reference's pyramid, group.rs:37-41): the f32 traversal loops test that item inside the
 *                            bound's step -- same tests, same order, same values, one node step fewer per entered group */
enum { RT_SCENE_HAS_BOUNDS = 1u, RT_SCENE_CONCENTRIC = 2u };
rt_status rt_scene_traits(const rt_scene *scene, uint32_t *traits);

/* What rt_scene_create took on the host (diagnostic: a one-shot caller -- `make image` -- pays it once per process): total_ms for the whole
 * call, stream_ms of it for creating the scene's stream.  In a process that has no stream yet that is where the RUNTIME makes its first
 * hardware queue (~19 ms on MI355X / ROCm 7.2, whoever creates the first stream or launches the first kernel: tools/init_probe.hip); the
 * library's own work -- allocations, uploads, deriving the streams; its code object is loaded meanwhile -- is the difference. */
rt_status rt_scene_setup_cost(const rt_scene *scene, double *total_ms, double *stream_ms);

/* rt_render_tiles with delivery in completion order (the reference's channel, render.rs:271,301-307): the buckets are rendered in
 * batches that are all enqueued at once, and `callback` is invoked -- on the calling thread -- for every bucket of a batch as soon
 * as that batch is complete, while later batches are still rendering.  tile_index: the bucket's position in `tiles`; rgba: its
 * RGBABuffer bytes (row 0 = region.b), valid only during the callback.  Returns when every bucket has been delivered. */
typedef void (*rt_tile_callback)(void *user, uint32_t tile_index, const rt_region *region, const uint8_t *rgba);
rt_status rt_render_tiles_stream(rt_scene *scene, const rt_options *options, rt_traversal traversal,
                                 const rt_region *tiles, uint32_t n_tiles, rt_tile_callback callback, void *user);

/* The same pass for a writer that keeps its image in the FILE's pixel format (PPMStdoutRGBABufferWriter, render.rs:373-401): every listed
 * bucket is rendered and put -- converted on the device -- into its place in a row-major frame of options->width x options->height pixels:
 *   RT_FRAME_RGBA  4 B/px  what set_pixels_from_buffer leaves in the writer's image (render.rs:112-126, 422-424)
 *   RT_FRAME_RGB   3 B/px  R, G, B -- the P6 payload (render.rs:392-396: alpha dropped)
 *   RT_FRAME_GREY  1 B/px  ((r + g + b) as f32 / 3.0) as u8 -- the P5 payload (render.rs:399)
 * frame_out: HOST memory, 4-byte aligned, width * height * {4, 3, 1} bytes; bytes outside the listed buckets are left alone.  Memory from
 * rt_host_alloc / rt_host_register is written by the device directly (a 1080p P6 image: 6.2 MB over PCIe instead of 8.3 MB and no
 * conversion on the CPU); pageable memory goes through pinned staging and a CPU copy.  Batches as for rt_render_tiles_stream: `callback`
 * (may be NULL) is invoked on the calling thread after each batch is in place -- tiles [first_tile, first_tile + n_tiles) of the list --
 * while later batches are still rendering (the writer's once-per-second rewrite, render.rs:427-432, hangs off it). */
typedef enum rt_frame_format { RT_FRAME_RGBA = 0, RT_FRAME_RGB = 1, RT_FRAME_GREY = 2 } rt_frame_format;
typedef void (*rt_batch_callback)(void *user, uint32_t first_tile, uint32_t n_tiles);
rt_status rt_render_frame_stream(rt_scene *scene, const rt_options *options, rt_traversal traversal,
                                 const rt_region *tiles, uint32_t n_tiles, rt_frame_format format, uint8_t *frame_out,
                                 rt_batch_callback callback, void *user);

/* Host memory for RGBABuffer storage (render.rs:74-90 allocates it with vec![0; area * 4]) that the device can reach
 * directly.  rt_render_tiles / rt_render_region recognise such memory by address (any pointer inside a range this library
 * pinned -- memory pinned by other means counts as pageable) and then
 * the render kernel stores its pixels straight into it over PCIe: no device-side copy of the frame, no staging, no CPU copy.
 * Pageable memory (a plain Vec<u8>) works everywhere too, but every byte then takes a bounce through pinned staging and a CPU
 * copy (about 3x slower for a 1080p frame).
 *   rt_host_alloc / rt_host_free        pinned, device-mapped allocation
 *   rt_host_register / rt_host_unregister   pin memory the caller already owns (e.g. the writer's frame, render.rs:323);
 *                                       the caller keeps it alive and unregisters before freeing it */
rt_status rt_host_alloc(size_t bytes, void **out);
rt_status rt_host_free(void *p);
rt_status rt_host_register(void *p, size_t bytes);
rt_status rt_host_unregister(void *p);

/* Renderer::render_region for a batch of regions in ONE device pass (a literal launch per 64x64 bucket would
 * starve 256 CUs).  rgba_out (HOST memory, see rt_host_alloc) receives the tiles back to back
 * ("tile-major"): tile i starts at 4 * sum_{j<i} area(j) and is its own row-major RGBABuffer (render.rs:74-109).
 * A single region {0, height, width, 0} therefore yields the row-major frame.  stats may be NULL. */
rt_status rt_render_tiles(rt_scene *scene, const rt_options *options, rt_traversal traversal,
                          const rt_region *tiles, uint32_t n_tiles,
                          uint8_t *rgba_out, rt_stats *stats);

/* Same, but rgba_out is DEVICE memory on the scene's device and the work is enqueued on `hip_stream`
 * (a hipStream_t passed as void*; NULL = the null stream) without waiting for it: the caller synchronises
 * (and may hand the buffer straight to an RCCL gather).  stats (may be NULL) receives counters
 * only when the call can read them back, i.e. it is filled after an internal stream sync if non-NULL. */
rt_status rt_render_tiles_device(rt_scene *scene, const rt_options *options, rt_traversal traversal,
                                 const rt_region *tiles, uint32_t n_tiles,
                                 void *rgba_out_device, void *hip_stream, rt_stats *stats);

/* rt_render_tiles_device + rt_blit_tiles_device in one pass: every listed bucket is rendered straight into its place
 * in a row-major RGBA frame of options->width x options->height in device memory -- what the writer's image holds
 * after write_rgba_buffer() for each bucket (render.rs:422-424).  Pixels outside the listed buckets are left alone. */
rt_status rt_render_frame_device(rt_scene *scene, const rt_options *options, rt_traversal traversal,
                                 const rt_region *tiles, uint32_t n_tiles,
                                 void *frame_rgba_device, void *hip_stream, rt_stats *stats);

/* Single bucket: the exact shape of the reference call (render.rs:283-294), made from up to RTRACEMAXPROCS pool threads
 * at once.  A request for a bucket of the scheduler's own grid (64 x 64, row-major from (0, 0), edge buckets clipped:
 * render.rs:273-298) is served from a pass that renders the WHOLE grid once per frame into pinned staging -- the call is then a
 * 16 KB copy.  A bucket is handed out once per pass (asking for it again starts the caller's next frame), and the pass for the next
 * frame is started while the current one is being handed out: a scene is immutable, so its bytes are those of a pass started later;
 * one pass too many is rendered when the caller stops.  Other requests: concurrent calls on one scene are merged into shared device
 * passes (whoever finds no pass running leads the next one and renders every request waiting at that moment); a lone caller gets
 * one pass per call.  With stats != NULL the call runs on its own. */
rt_status rt_render_region(rt_scene *scene, const rt_options *options, rt_traversal traversal,
                           const rt_region *region, uint8_t *rgba_out, rt_stats *stats);

/* RGBABuffer::set_pixels_from_buffer (render.rs:112-126) on the device: row-wise blit of tile-major tiles (the
 * layout rt_render_tiles_device writes, or several such shards after an RCCL gather) into a row-major RGBA frame
 * of options->width x options->height, i.e. what PPMStdoutRGBABufferWriter::write_rgba_buffer does per bucket
 * (render.rs:422-424).  src_px_offset[i] is tile i's first pixel in src (in pixels, 4 B each); NULL means the
 * tiles lie back to back in list order.  Enqueued on hip_stream without waiting. */
rt_status rt_blit_tiles_device(rt_scene *scene, const rt_options *options, const rt_region *tiles, uint32_t n_tiles,
                               const uint32_t *src_px_offset, const void *src_tile_major_device,
                               void *frame_rgba_device, void *hip_stream);

/* ---- several GPUs of one node, one process -------------------------------------------------------------
 * The reference joins its pool threads' buckets through one channel (render.rs:271, 293, 301).  A gang does the same across
 * GPUs: the Scene is replicated on every listed device, the buckets of a frame are dealt round-robin in list order
 * (bucket i -> devices[i % n], the scheduler's row-major order render.rs:273-298), every device renders its shard
 * tile-major, and ONE RCCL gather of the equal-length u8 shards (ncclGather over xGMI, root = devices[0]; communicators from
 * ncclCommInitAll) brings them to the root GPU, which blits them into the row-major frame (set_pixels_from_buffer,
 * render.rs:112-126) and hands that frame to the host.  The bytes are identical for every n.  librccl.so is loaded when the
 * first gang is created (RT_ERR_UNSUPPORTED if it cannot be); single-GPU rendering never needs it. */
typedef struct rt_gang rt_gang;
rt_status rt_gang_create(const int *devices, int n_devices, rt_precision precision,
                         const void *dfs_items, uint32_t n_items, const void *light_unit, const void *eye,
                         const void *bounds, const rt_range *ranges, uint32_t n_bounds, rt_gang **out);
rt_status rt_gang_destroy(rt_gang *gang);
rt_status rt_gang_size(const rt_gang *gang, int *n_devices);
/* frame_rgba_host: width * height * 4 bytes, row-major (what the writer's image holds after write_rgba_buffer() for every
 * listed bucket; pixels outside the listed buckets are unspecified).  stats (may be NULL): counters summed over the devices,
 * device_ms = the slowest device's render. */
rt_status rt_gang_render_frame(rt_gang *gang, const rt_options *options, rt_traversal traversal,
                               const rt_region *tiles, uint32_t n_tiles, uint8_t *frame_rgba_host, rt_stats *stats);
/* n_frames frames of the same tile list, frame f to frames_rgba_host[f] (each as for rt_gang_render_frame): the gather, the blit and
 * the copy of frame f run under the render of frame f + 1 (double-buffered shards; what dist.py's run_pipeline does across
 * processes).  A frame buffer from rt_host_alloc / rt_host_register is written by the root GPU's blit kernel directly.
 * stats (may be NULL): the counters of ONE frame. */
rt_status rt_gang_render_frames(rt_gang *gang, const rt_options *options, rt_traversal traversal,
                                const rt_region *tiles, uint32_t n_tiles, uint8_t *const *frames_rgba_host, uint32_t n_frames,
                                rt_stats *stats);

/* Bytes rt_render_tiles writes for this tile list (4 * total area), or 0 on an invalid list. */
uint64_t rt_tiles_rgba_bytes(const rt_region *tiles, uint32_t n_tiles);

/* Device self-test: compares the traversal loops' lean correctly-rounded f32 sqrt with the compiler's IEEE sqrt on
 * ALL 2^32 bit patterns; *mismatches must come back 0 (first_bad_bits = 0xFFFFFFFF).  ~10 ms on an MI355X. */
rt_status rt_selftest_sqrt(int device, uint64_t *mismatches, uint32_t *first_bad_bits);
/* The same for the lean reciprocal of Vector::normalized (vec.rs:87-95, `len.recip()`): against the compiler's IEEE division
 * 1.0f / x on all 2^32 bit patterns, and through normalized() itself. */
rt_status rt_selftest_rcp(int device, uint64_t *mismatches, uint32_t *first_bad_bits);

const char *rt_strerror(rt_status status);
/* Detail of the last RT_ERR_HIP on the calling thread (static thread-local storage; never NULL). */
const char *rt_last_error_message(void);
int rt_abi_version(void);

/* Which kernels the LAST render call of the calling thread launched (read-only diagnostic; the bytes never depend on it):
 *   RT_LAUNCH_TWO_RAYS         the hierarchy walk with two rays per lane (k_render_skip2: large frames / large scenes)
 *   RT_LAUNCH_COOPERATIVE      the launch carried lane-cooperative quads (the heaviest 2x2-pixel quads of a small pass)
 *   RT_LAUNCH_SAMPLE_PARALLEL  one thread per SAMPLE + an ordered resolve pass (samples_per_pixel > 1)
 *   RT_LAUNCH_ORDERED          blocks dispatched most-expensive-first from the scene's cost map (else: through the tile table)
 *   RT_LAUNCH_FLAT_PIPELINE    RT_TRAVERSAL_FLAT's wavefront pipeline
 *   RT_LAUNCH_COUNTING         the counting flavour of the kernels (the call asked for rt_stats)
 *   RT_LAUNCH_FAST_KERNEL      the single-pass kernel of steady-state frames (f32, one sample per pixel, ordered: k_render_skip_fast) */
enum { RT_LAUNCH_TWO_RAYS = 1u, RT_LAUNCH_COOPERATIVE = 2u, RT_LAUNCH_SAMPLE_PARALLEL = 4u, RT_LAUNCH_ORDERED = 8u, RT_LAUNCH_FLAT_PIPELINE = 16u,
       RT_LAUNCH_COUNTING = 32u, RT_LAUNCH_FAST_KERNEL = 64u };
uint32_t rt_last_launch_flags(void);

/* The toolchain this library was built with, e.g. "hipcc: HIP version: 7.2.x ... | clang ... | kernels <sha1 of the kernel sources>"
 * (static storage).  The generated traversal loops are gfx950 assembly whose register windows were validated against THIS compiler:
 * tests/test_kernel_resources.py pins the string together with every hot kernel's register counts. */
const char *rt_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* RTRACE_HIP_H */
