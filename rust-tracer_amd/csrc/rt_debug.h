/*
 * rt_debug.h -- diagnostic controls of librtrace_hip.so.  NOT part of the drop-in ABI (include/rtrace_hip.h): nothing a
 * renderer needs is here, and the library reads no environment variable.  The parity tests and tools/ use these to pick a
 * traversal-loop flavour or switch an optimisation off and prove that no byte and no counter changes.
 *
 * Every control is process-wide and atomic; set it between calls, not during one.
 */
#ifndef RTRACE_HIP_DEBUG_H
#define RTRACE_HIP_DEBUG_H

#include "../../include/rtrace_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rt_debug_key {
    RT_DEBUG_SKIP_VARIANT = 0,   /* k_render_skip VAR bits (rt_skip.hpp): 0/1 C++ loops, 3 generated assembly loops, 7 their fused
                                    flavour (dropped for scenes that are not concentric), + 16 the filtered loops (f32: a conservative
                                    bound in front of every test).  Default 23 */
    RT_DEBUG_BLOCK_ORDER = 1,    /* 0: dispatch a pass's blocks in raster order instead of most-expensive-first.  Default 1 */
    RT_DEBUG_NARROW_MAX = 2,     /* cap on the number of blocks dealt out as narrow workgroups (read when a tile list is first seen) */
    RT_DEBUG_PACKED_SAMPLES = 3, /* 0: spp 2/4/8 use the plain sample-parallel mapping.  Default 1 */
    RT_DEBUG_PRINT_STEPS = 4,    /* 1: a counted call prints its node-step statistics on stderr */
    RT_DEBUG_PRINT_COSTS = 5,    /* 1: print the block cost estimates when a dispatch order is built */
    RT_DEBUG_HOST_COPY = 6,      /* how rt_render_tiles returns bytes to host memory: 0 the library's choice, 1 one copy to the
                                    caller's pointer, 2 pinned staging + CPU copy, 3 the kernel stores into pinned host memory */
    RT_DEBUG_COALESCE = 7,       /* 0: concurrent rt_render_region calls are not merged into shared passes; n > 0: at most n merged passes in flight.  Default 2 */
    RT_DEBUG_LDS_BYTES = 8,      /* dynamic LDS reserved per render workgroup: caps the waves per SIMD (160 KiB / n workgroups per CU) for
                                    occupancy experiments; the kernels do not touch it */
    RT_DEBUG_WG_POLICY = 9,      /* 0: one workgroup per block and sample (the dispatcher deals); n > 0: n x 2,048 workgroups per launch, blocks
                                    dealt on the host; default: dealt only past 32,768 workgroups (read when a tile list is first seen) */
    RT_DEBUG_NARROW_L2 = 10,     /* of the narrow blocks, how many of the most expensive go out as sixteen 2x2-pixel-per-wave workgroups
                                    (the rest as four 4x4 ones); default: all in a pass of <= 4,096 blocks, none otherwise */
    RT_DEBUG_FLAT_KERNELS = 11,  /* 0: the f32 flat traversal runs round 1's LDS-staged packed-math kernels (rt_flat_wf.hpp, what f64 always
                                    runs) instead of the scalar-fed scan (rt_flat_sc.hpp) */
    RT_DEBUG_SKIP_RAYS = 12,     /* rays per lane of the f32 hierarchy walk (assembly loops): 1 = k_render_skip always; 2 = k_render_skip2 (two rays per lane on
                                    packed math, rt_skip2.hpp) wherever it exists (spp 1, 2, 4, 8; launches that do not count tests);
                                    default: the library's choice per workload (large spp-1 frames, large scenes) */
    RT_DEBUG_FRAME_AHEAD = 13,   /* 0: rt_render_region never serves a bucket from a whole-grid pass rendered ahead (every call its own device pass,
                                    or merged with concurrent ones: RT_DEBUG_COALESCE); 1: whole-grid passes, each rendered when its frame is first
                                    asked for; default (2): the pass for the next frame is started as soon as the current one is being handed out */
    RT_DEBUG_FILTER_RO_PERCENT = 14, /* read by rt_scene_create: the radius around the scene's centroid in which the filtered loops' bounds cover
                                    shadow-ray origins, in percent of the library's own value.  A small value leaves real origins
                                    uncovered -- they must then fall back to the reference's arithmetic at every node (tests).  Default 100 */
    RT_DEBUG_COOP = 15,          /* the lane-cooperative walk of the heaviest quads (rt_coop.hpp; read when a tile list is first seen): 0 never,
                                    1 asked for at the default threshold (40 % of the pass's largest estimate) without the trial, 2 every quad of
                                    every block (parity tests); default: the library TRIES thresholds of 28 / 34 / 40 / 48 / 58 % against the plain
                                    dispatch over a list's first launches and keeps the fastest (rt_capi.hip build_orders, pick_order) */
    RT_DEBUG_COOP_THR = 16,      /* ... the cost-map value (tests per pixel) from which a quad is walked cooperatively; default: see RT_DEBUG_COOP */
    RT_DEBUG_COOP_MAX = 17,      /* ... cap on the number of 16x16 blocks that are split for it; default: an eighth of the pass */
    RT_DEBUG_COOP_LEVEL = 18,    /* ... rays per cooperative wave: 1 = 16 (4x4 pixels), 2 = 4 (2x2), 3 = one; default 2 */
    RT_DEBUG_COOP_REST = 19,     /* ... what is left of a block with cooperative quads: 0 = one descriptor (8x8 pixels per wave), 1 = four (4x4 per wave) */
    RT_DEBUG_ASYNC_ORDERS = 20,  /* 0: the dispatch orders of a tile list (and the scene's cost map) are made by the first call that uses the list, as they are
                                    whenever another dispatch control here is set; default: by a background thread, the first launches finding their blocks
                                    through the tile table */
    RT_DEBUG_FAST_KERNEL = 21,   /* the lean kernels (rt_skip_fast.hpp f32, rt_skip_fast64.hpp f64, rt_skip2_fast.hpp two rays per lane: one mode each, arguments fetched
                                    where they are needed): 0 never -- the generic kernels --, 2 wherever one exists; default: f32 where the list has cooperative
                                    quads, f64 every ordered spp-1 launch, two rays per lane every launch with a dispatch list but a dealt list of a
                                    plain-stream scene.  All must render the same bytes */
    RT_DEBUG_EXACT_COSTS = 22,   /* 0: cooperative quads are picked by the scene's 256 x 256 cost map only (small passes), not by counting a tile list's heaviest
                                    blocks again at the frame's own resolution (rt_capi.hip exact_block_costs).  Default 1 (read when a tile list is first seen) */
    RT_DEBUG_KEYS = 23
} rt_debug_key;

/* value < 0 restores the default. */
rt_status rt_debug_set(int key, long long value);

/* Process-wide event counts since load (diagnostic): how the merged rt_render_region passes went. */
typedef enum rt_debug_counter {
    RT_DEBUG_COUNT_REGION_CALLS = 0,    /* rt_render_region calls that went through the merging path */
    RT_DEBUG_COUNT_REGION_PASSES = 1,   /* device passes they were rendered in */
    RT_DEBUG_COUNT_FILTER_PASS = 2,     /* f32 hierarchy walk, calls that return rt_stats: per-ray tests the filtered loops' bound lets through */
    RT_DEBUG_COUNT_FILTER_VIOLATIONS = 3,   /* ... and tests with a finite distance that the bound would have ruled out: must stay 0 */
    RT_DEBUG_COUNT_PRIMARY_TESTS = 4,   /* hierarchy walk: of the LAST call that returned rt_stats, the tests (items + bounds) made for primary rays
                                           (the rest of sphere_tests + bound_tests were made for shadow rays); not cumulative */
    RT_DEBUG_COUNT_FRAME_AHEAD_PASSES = 5,  /* whole-grid passes rendered for rt_render_region's frame-ahead */
    RT_DEBUG_COUNT_TWO_RAY_LAUNCHES = 6,    /* hierarchy-walk passes that ran k_render_skip2 (two rays per lane) instead of k_render_skip */
    RT_DEBUG_COUNT_COOP_LAUNCHES = 7,       /* hierarchy-walk passes whose launch carried cooperative quads (k_render_skip<..., COOP>) */
    RT_DEBUG_COUNTERS = 8
} rt_debug_counter;
long long rt_debug_count(int counter);

/* Per-wave timeline (tools/wave_timeline.py): while a path is set, every launch of the assembly loops records each wave's
 * start / end / placement and writes the records to the file (synchronous, overwritten per launch).  NULL switches it off. */
rt_status rt_debug_wave_trace(const char *path);

/* Test infrastructure for the flat scan's conservative filter (rt_flat_sc.hpp, flat_filter_constant): evaluates, for every primary
 * ray of a width x height x spp frame and every item of an f32 scene, the exact discriminant and the filter's bound.
 * counts = { pairs with disc >= 0, pairs with bound >= 0, pairs with disc >= 0 but bound < 0 } for the primary filter, then the same
 * three for the shadow filter on rays from a point of each primary ray towards the light: counts[2] and counts[5] must be 0. */
rt_status rt_debug_flat_filter_check(rt_scene *scene, uint32_t width, uint32_t height, uint32_t spp, unsigned long long counts[6]);

/* Test infrastructure for the multi-GPU paths.  rt_debug_gang_layout: the sharding arithmetic of rt_gang_render_frame(s) without a
 * device -- for bucket i its device (i % n_devices) and its first pixel inside that device's tile-major shard; per device the pixels
 * of its shard; the padded shard length every device sends.  rt_debug_shard_costs: what the scene's cost map predicts for the shards
 * (sum of tests per pixel under each device's buckets): cost[n_devices]. */
rt_status rt_debug_gang_layout(const rt_region *tiles, uint32_t n_tiles, uint32_t n_devices, uint32_t *device_of, uint32_t *px_offset,
                               uint64_t *shard_px, uint64_t *padded_px);
/* A stand-in for librccl.so (tests/c/fake_rccl.cpp), by path: gangs created while it is set bind ITS ncclCommInitAll / ncclGroupStart / ncclGather /
 * ... instead of librccl.so's and may list one device several times (several ranks on one GPU) -- how the N > 1 code of rt_gang_* is
 * executed on a one-GPU box.  NULL: librccl.so again.  A gang keeps the library it was created with. */
rt_status rt_debug_rccl_library(const char *path);
rt_status rt_debug_shard_costs(rt_scene *scene, const rt_options *options, const rt_region *tiles, uint32_t n_tiles, uint32_t n_devices, double *cost);

#ifdef __cplusplus
}
#endif
#endif /* RTRACE_HIP_DEBUG_H */
