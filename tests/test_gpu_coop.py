"""The lane-cooperative walk (csrc/rt_coop.hpp): a wave gathers the closure of its quad's rays breadth-first -- lanes carry nodes,
not rays -- and must return exactly what the reference's DFS with its `bound.distance >= hit.distance` cull returns
(/root/reference/src/rust/group.rs:72-83, primitive.rs:77-84).  csrc/rt_debug.h RT_DEBUG_COOP = 2 forces it on every quad of every
block; by default the library picks the quads whose rays meet the most nodes.  Every case here is held against the CPU oracle."""
import os
import zlib

import numpy as np
import pytest

import oracle
import rust_tracer_amd as rta
from tests import util
from tests.test_gpu_parity import _vector_cases, bucket_list

pytestmark = pytest.mark.gpu

SKIP = rta.RT_TRAVERSAL_SKIP
HIER_EXIT = oracle.MODE_HIERARCHY | oracle.MODE_ANYHIT_EXIT
COOP_LAUNCHES = rta.capi.DEBUG_COUNT_COOP_LAUNCHES


def coop_launches():
    return rta.capi.debug_count(COOP_LAUNCHES)


def render(scene, w, h, regs, mode):
    """mode: 0 never cooperative, 2 every quad, -1 the library's choice.  Returns the bytes and the number of cooperative launches -- None on the
    library that ships (tests/conftest.py RTRACE_PARITY_ON_PRODUCT: no control, no counter): there the frame is rendered as the library
    chooses, twice (the second launch finds the dispatch orders the first one ordered), and still has to be the oracle's."""
    if not rta.capi.HAVE_TEST_HOOKS:
        data, _ = scene.device().render_tiles((w, h, 1), regs, SKIP, want_stats=True)
        return data, None
    with rta.capi.debug(rta.capi.DEBUG_COOP, mode):
        before = coop_launches()
        data, _ = scene.device().render_tiles((w, h, 1), regs, SKIP, want_stats=False)
        return data, coop_launches() - before


def tile_crcs(data, regs):
    out, off = [], 0
    for (l, t, r, b) in regs:
        k = (r - l) * (t - b) * 4
        out.append(zlib.crc32(data[off:off + k].tobytes()) & 0xFFFFFFFF)
        off += k
    return out


@pytest.mark.parametrize("name", ["config2_800x600", "config3_1920x1080_f32"])
def test_baseline_frames_forced_and_by_default(name):
    # BASELINE configs 2 and 3 (f32): every quad cooperative, asked for with the default threshold, and none -- the committed oracle CRCs
    case = next(c for c in _vector_cases() if c["name"] == name)
    w, h = case["width"], case["height"]
    regs = bucket_list(w, h)
    # (asked for without a threshold, rt_debug.h RT_DEBUG_COOP = 1, the library still leaves passes of more than 4,096 blocks alone: 1080p)
    for mode, expect_coop in ((2, True), (1, name == "config2_800x600"), (0, False)):
        s = rta.Scene.default()
        data, n = render(s, w, h, regs, mode)
        assert n is None or (n > 0) == expect_coop, (mode, n)
        assert tile_crcs(data, regs) == case["tile_crc32"], (name, mode)


def test_the_library_tries_its_dispatch_orders_and_every_one_renders_the_frame():
    # Left alone (rt_debug.h untouched) the library builds several dispatch orders for a tile list whose pass could use the cooperative
    # walk -- none, and a few thresholds -- and times them against each other over its first launches (rt_capi.hip pick_order) before it
    # settles.  Whatever order a launch got: the same bytes, the oracle's.
    case = next(c for c in _vector_cases() if c["name"] == "config2_800x600")
    w, h = case["width"], case["height"]
    regs = bucket_list(w, h)
    s = rta.Scene.default()
    before = coop_launches()
    for _ in range(24):
        data, _ = s.device().render_tiles((w, h, 1), regs, SKIP, want_stats=False)
        assert tile_crcs(data, regs) == case["tile_crc32"]
    assert coop_launches() > before                     # some of the candidates are cooperative
    counted, st = s.device().render_tiles((w, h, 1), regs, SKIP, want_stats=True)          # a counting launch renders the plain order
    assert tile_crcs(counted, regs) == case["tile_crc32"]
    for k in ("primary", "hits", "shadow", "occluded", "sphere_tests", "bound_tests"):
        assert st[k] == case["stats"][k], k


def test_dispatch_orders_made_in_the_background_arrive_and_change_no_byte():
    # the default, outside the test suite: a new tile list's first launches find their blocks through the tile table while a thread of
    # the library renders the scene's cost map and builds the candidate orders; a few frames in the ordered (here: partly cooperative)
    # dispatch takes over.  Every frame on the way is the oracle's.
    import time
    case = next(c for c in _vector_cases() if c["name"] == "config2_800x600")
    w, h = case["width"], case["height"]
    regs = bucket_list(w, h)
    with rta.capi.debug(rta.capi.DEBUG_ASYNC_ORDERS, -1):
        s = rta.Scene.default()
        d = s.device()
        before, t0, frames = coop_launches(), time.time(), 0
        data, _ = d.render_tiles((w, h, 1), regs, SKIP, want_stats=False)
        assert tile_crcs(data, regs) == case["tile_crc32"]                      # the very first launch: no order yet
        while coop_launches() == before and time.time() - t0 < 20.0:
            data, _ = d.render_tiles((w, h, 1), regs, SKIP, want_stats=False)
            frames += 1
            assert tile_crcs(data, regs) == case["tile_crc32"]
        assert coop_launches() > before, "no cooperative candidate after %d frames" % frames
        for _ in range(20):
            data, _ = d.render_tiles((w, h, 1), regs, SKIP, want_stats=False)
            assert tile_crcs(data, regs) == case["tile_crc32"]
        d.close()                                                                # joins the builder


def test_a_new_tile_list_is_uploaded_behind_its_first_callers_kernels_and_other_streams_wait_for_it():
    # the default outside the test suite: a new list's tile table goes to the device through pinned staging on the FIRST caller's stream (no
    # blocking copy, no device-wide synchronise in a one-shot caller's first frame).  A caller on another stream may use the cached table
    # before that copy has run -- here it is queued behind 50 ms of somebody else's work -- and has to wait for it, not read an empty table.
    import torch
    case = next(c for c in _vector_cases() if c["name"] == "config2_800x600")
    w, h = case["width"], case["height"]
    regs = bucket_list(w, h)
    n = sum((r - l) * (t - b) for (l, t, r, b) in regs) * 4
    with rta.capi.debug(rta.capi.DEBUG_ASYNC_ORDERS, -1):
        s = rta.Scene.default()
        d = s.device()
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        outa, outb = torch.zeros(n, dtype=torch.uint8, device="cuda"), torch.zeros(n, dtype=torch.uint8, device="cuda")
        big = torch.randn(8192, 8192, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(sa):
            for _ in range(40):
                big = big @ big * 1e-4                                           # keeps stream A busy while the calls below return
        d.render_tiles_device((w, h, 1), regs, outa.data_ptr(), sa.cuda_stream, SKIP)       # first sight of the list: the table's copy is queued on A
        d.render_tiles_device((w, h, 1), regs, outb.data_ptr(), sb.cuda_stream, SKIP)       # B finds the list cached
        torch.cuda.synchronize()
        assert tile_crcs(outb.cpu().numpy(), regs) == case["tile_crc32"]
        assert tile_crcs(outa.cpu().numpy(), regs) == case["tile_crc32"]
        d.close()


def _native_threads():
    with open("/proc/self/status") as f:
        return int(next(line for line in f if line.startswith("Threads:")).split()[1])


def test_scenes_come_and_go_with_their_worker_threads():
    # every scene starts a worker thread (it makes the dispatch orders of new tile lists) and hands it work in its first frame: scenes
    # destroyed at once, after one frame (the job may not have started), after the orders arrived, several lists queued behind each other
    # -- no frame differs, nothing hangs, no thread is left behind
    case = next(c for c in _vector_cases() if c["name"] == "config2_800x600")
    w, h = case["width"], case["height"]
    regs = bucket_list(w, h)
    with rta.capi.debug(rta.capi.DEBUG_ASYNC_ORDERS, -1):
        warm = rta.Scene.default(5).device()
        warm.render_tiles((64, 64, 1), bucket_list(64, 64), SKIP, want_stats=False)
        warm.close()
        before = _native_threads()
        for k in range(40):
            d = rta.Scene.default(8 if k % 4 == 1 else 5 + k % 3).device()
            if k % 4 == 1:
                data, _ = d.render_tiles((w, h, 1), regs, SKIP, want_stats=False)
                assert tile_crcs(data, regs) == case["tile_crc32"]
            elif k % 4 == 2:
                for size in ((320, 200), (640, 360), (333, 77), (w, h)):                         # four new lists queued one behind the other
                    d.render_tiles((size[0], size[1], 1), bucket_list(*size), SKIP, want_stats=False)
            elif k % 4 == 3:
                for _ in range(12):
                    d.render_tiles((w, h, 1), regs, SKIP, want_stats=False)
            d.close()
        assert _native_threads() <= before


def test_level9_pyramid_and_ragged_tiles():
    # 87,381 spheres (BASELINE config 5's scene) on ragged 50x50 tiles whose last block row / column is clipped to 2 pixels
    s, o = rta.Scene.default(9), oracle.Scene.default(level=9)
    w, h = 1280, 720
    regs = [(x, y + 50, x + 50, y) for y in range(280, 480, 50) for x in range(560, 760, 50)]
    data, n = render(s, w, h, regs, 2)
    assert n is None or n > 0
    off = 0
    for (l, t, r, b) in regs:
        ref, _ = o.render_region(w, h, 1, l, t, r, b, HIER_EXIT)
        k = (r - l) * (t - b) * 4
        np.testing.assert_array_equal(data[off:off + k].reshape(t - b, r - l, 4), ref)
        off += k


@pytest.mark.parametrize("seed", range(3000, 3030))
def test_random_nested_scenes(seed):
    # bounds that do not enclose their subtrees, the eye inside bounds every other scene (then the winner's ancestors are farther than
    # the winner and the ray goes back to the skip-pointer loops), random light, ragged sizes
    rng = np.random.default_rng(seed)
    depth, fan, leaf = int(rng.integers(2, 6)), int(rng.integers(2, 5)), int(rng.integers(1, 4))
    items, bounds, ranges = util.random_nested_scene(seed, depth=depth, fan=fan, leaf_items=leaf, concentric=seed % 3 == 1)
    far = seed % 2 == 0
    eye = (float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-9.0, -7.0) if far else rng.uniform(-4.5, -1.0)))
    light = (float(rng.uniform(-2, 2)), float(rng.uniform(-3, -0.5)), float(rng.uniform(-2, 2)))
    s, o = util.scene_pair_ranges(items, bounds, ranges, rta.RT_F32, light=light, eye=eye)
    w, h = int(rng.integers(2, 7)) * 32 + int(rng.integers(0, 17)), int(rng.integers(2, 7)) * 24 + int(rng.integers(0, 13))
    regs = bucket_list(w, h)
    ref, _, _ = o.render(w, h, 1, os.cpu_count() or 1, HIER_EXIT)
    data, n = render(s, w, h, regs, 2)
    assert n is None or n > 0
    np.testing.assert_array_equal(util.stitch((w, h), regs, data), ref)


@pytest.mark.parametrize("scale", [1e-20, 1e-10, 1e6, 5e13])
def test_scaled_scenes(scale):
    # every square a denormal (1e-20: the lean root's general path) ... coordinates just under the validation bound
    for seed, concentric in ((31, False), (32, True)):
        items, bounds, ranges = util.random_nested_scene(seed, depth=3, fan=3, leaf_items=2, concentric=concentric)
        sc = lambda a: (np.asarray(a, dtype=np.float64) * scale).astype(np.float32).astype(np.float64)
        eye = tuple(float(v) for v in sc((0.07, -0.12, -3.1)))
        s, o = util.scene_pair_ranges(sc(items), sc(bounds), ranges, rta.RT_F32, eye=eye)
        w, h = 96, 72
        regs = bucket_list(w, h)
        ref, rst, _ = o.render(w, h, 1, os.cpu_count() or 1, HIER_EXIT)
        assert rst["hits"] > 500 and rst["shadow"] > 100
        data, n = render(s, w, h, regs, 2)
        assert n is None or n > 0
        np.testing.assert_array_equal(util.stitch((w, h), regs, data), ref)


def test_inside_bound_quirk_and_tie_break():
    # the eye inside a bound -- the reference culls the group that holds the nearer item.  The gather finds that item,
    # sees that its ancestor is farther than it, and hands the ray back: the bytes are the hierarchy's, not the flat scan's.
    s, o = util.scene_pair_ranges(util.INSIDE_ITEMS, util.INSIDE_BOUNDS, util.INSIDE_RANGES)
    regs = [(0, 64, 64, 0)]
    ref, _ = o.render_region(64, 64, 1, 0, 64, 64, 0, HIER_EXIT)
    flat, _ = o.render_region(64, 64, 1, 0, 64, 64, 0, oracle.MODE_FLAT)
    assert not np.array_equal(ref, flat)
    data, n = render(s, 64, 64, regs, 2)
    assert n is None or n > 0
    np.testing.assert_array_equal(data.reshape(64, 64, 4), ref)
    # two spheres hit at exactly the same f32 distance: the first in DFS order keeps the hit (primitive.rs:79, strict <)
    s, o = util.scene_pair_spheres(util.TIE_SPHERES, util.TIE_BOUND)
    ref, _ = o.render_region(64, 64, 1, 0, 64, 64, 0, HIER_EXIT)
    data, n = render(s, 64, 64, regs, 2)
    assert n is None or n > 0
    np.testing.assert_array_equal(data.reshape(64, 64, 4), ref)


def test_work_list_overflow_goes_back_to_the_loops():
    # 14 groups of 14 groups of one item, all bounds around the whole scene: every ray enters everything, 16 rays x 14 x 14 pairs do
    # not fit the wave's work list (448 pairs) -- the quad's rays are walked by the skip-pointer loops instead.  Same bytes.
    rng = np.random.default_rng(7)
    items, bounds, ranges = [], [], []
    bounds.append((0.0, 0.0, 0.0, 6.0)); ranges.append(None)
    for _ in range(14):
        bi = len(bounds)
        bounds.append((0.0, 0.0, 0.0, 5.5)); ranges.append(None)
        first = len(items)
        for _ in range(14):
            c = rng.uniform(-1.0, 1.0, 3)
            bounds.append((0.0, 0.0, 0.0, 5.0)); ranges.append((len(items), 1))
            items.append((c[0], c[1], c[2], float(rng.uniform(0.05, 0.2))))
        ranges[bi] = (first, len(items) - first)
    ranges[0] = (0, len(items))
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    s, o = util.scene_pair_ranges(f32(items), f32(bounds), np.asarray(ranges, dtype=np.int32), eye=(0.0, 0.0, -9.0))
    w, h = 96, 64
    regs = bucket_list(w, h)
    ref, rst, _ = o.render(w, h, 1, os.cpu_count() or 1, HIER_EXIT)
    assert rst["hits"] > 100
    data, n = render(s, w, h, regs, 2)
    assert n is None or n > 0
    np.testing.assert_array_equal(util.stitch((w, h), regs, data), ref)


# ---- f64 (round 6): the same walk on CNode64 records -- the distance's 64 bits and the DFS index are two minima, `anc` travels as an f32
# rounded up (rt_coop.hpp).  The oracle's f64 instantiation is the checker (parity unpinned: the reference holds no f64 vector).

def render64(scene, w, h, regs, mode):
    if not rta.capi.HAVE_TEST_HOOKS:
        data, _ = scene.device().render_tiles((w, h, 1), regs, SKIP, want_stats=True)
        return data, None
    with rta.capi.debug(rta.capi.DEBUG_COOP, mode):
        before = coop_launches()
        data, _ = scene.device().render_tiles((w, h, 1), regs, SKIP, want_stats=False)
        return data, coop_launches() - before


def test_f64_baseline_frame_forced_by_default_and_never():
    case = next(c for c in _vector_cases() if c["name"] == "config3_1920x1080_f64")
    w, h = case["width"], case["height"]
    regs = bucket_list(w, h)
    for mode, expect_coop in ((2, True), (0, False)):
        s = rta.Scene.default(8, rta.RT_F64)
        data, n = render64(s, w, h, regs, mode)
        assert n is None or (n > 0) == expect_coop, (mode, n)
        assert tile_crcs(data, regs) == case["tile_crc32"], mode
    # 800x600 in f64 (the frame that waits longest for one pixel's chain: 56 us before the walk existed in f64) against the oracle
    s, o = util.scene_pair_default(rta.RT_F64)
    w, h = 800, 600
    regs = bucket_list(w, h)
    ref, _, _ = o.render(w, h, 1, os.cpu_count() or 1, HIER_EXIT)
    for mode in (2, -1):
        data, n = render64(s, w, h, regs, mode)
        assert n is None or mode != 2 or n > 0
        np.testing.assert_array_equal(util.stitch((w, h), regs, data), ref)


@pytest.mark.parametrize("seed", range(3100, 3120))
def test_f64_random_nested_scenes(seed):
    rng = np.random.default_rng(seed)
    depth, fan, leaf = int(rng.integers(2, 6)), int(rng.integers(2, 5)), int(rng.integers(1, 4))
    items, bounds, ranges = util.random_nested_scene(seed, depth=depth, fan=fan, leaf_items=leaf, concentric=seed % 3 == 1)
    far = seed % 2 == 0
    eye = (float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-9.0, -7.0) if far else rng.uniform(-4.5, -1.0)))
    light = (float(rng.uniform(-2, 2)), float(rng.uniform(-3, -0.5)), float(rng.uniform(-2, 2)))
    s, o = util.scene_pair_ranges(items, bounds, ranges, rta.RT_F64, light=light, eye=eye)
    w, h = int(rng.integers(2, 7)) * 32 + int(rng.integers(0, 17)), int(rng.integers(2, 7)) * 24 + int(rng.integers(0, 13))
    regs = bucket_list(w, h)
    ref, _, _ = o.render(w, h, 1, os.cpu_count() or 1, HIER_EXIT)
    data, n = render64(s, w, h, regs, 2)
    assert n is None or n > 0
    np.testing.assert_array_equal(util.stitch((w, h), regs, data), ref)


def test_f64_inside_bound_quirk_tie_break_and_overflow():
    regs = [(0, 64, 64, 0)]
    s, o = util.scene_pair_ranges(util.INSIDE_ITEMS, util.INSIDE_BOUNDS, util.INSIDE_RANGES, rta.RT_F64)
    ref, _ = o.render_region(64, 64, 1, 0, 64, 64, 0, HIER_EXIT)
    flat, _ = o.render_region(64, 64, 1, 0, 64, 64, 0, oracle.MODE_FLAT)
    assert not np.array_equal(ref, flat)
    data, n = render64(s, 64, 64, regs, 2)
    assert n is None or n > 0
    np.testing.assert_array_equal(data.reshape(64, 64, 4), ref)
    # two spheres at exactly the same f64 distance on the middle column: the first in DFS order keeps the hit -- both orders
    for spheres in (util.TIE_SPHERES, util.TIE_SPHERES[::-1]):
        s, o = util.scene_pair_spheres(spheres, util.TIE_BOUND, rta.RT_F64)
        ref, _ = o.render_region(64, 64, 1, 0, 64, 64, 0, HIER_EXIT)
        data, n = render64(s, 64, 64, regs, 2)
        assert n is None or n > 0
        np.testing.assert_array_equal(data.reshape(64, 64, 4), ref)
    # the work list overflows (14 x 14 groups around the whole scene): the loops walk those rays
    rng = np.random.default_rng(7)
    items, bounds, ranges = [], [], []
    bounds.append((0.0, 0.0, 0.0, 6.0)); ranges.append(None)
    for _ in range(14):
        bi = len(bounds)
        bounds.append((0.0, 0.0, 0.0, 5.5)); ranges.append(None)
        first = len(items)
        for _ in range(14):
            c = rng.uniform(-1.0, 1.0, 3)
            bounds.append((0.0, 0.0, 0.0, 5.0)); ranges.append((len(items), 1))
            items.append((c[0], c[1], c[2], float(rng.uniform(0.05, 0.2))))
        ranges[bi] = (first, len(items) - first)
    ranges[0] = (0, len(items))
    s, o = util.scene_pair_ranges(np.asarray(items), np.asarray(bounds), np.asarray(ranges, dtype=np.int32), rta.RT_F64, eye=(0.0, 0.0, -9.0))
    w, h = 96, 64
    regs = bucket_list(w, h)
    ref, rst, _ = o.render(w, h, 1, os.cpu_count() or 1, HIER_EXIT)
    assert rst["hits"] > 100
    data, n = render64(s, w, h, regs, 2)
    assert n is None or n > 0
    np.testing.assert_array_equal(util.stitch((w, h), regs, data), ref)


def test_scenes_without_a_cooperative_copy_render_as_before():
    # a group with more children than a work-list word can count (15): no cooperative copy, the control changes nothing
    rng = np.random.default_rng(11)
    items = [(float(c[0]), float(c[1]), float(c[2]), 0.2) for c in rng.uniform(-1.5, 1.5, (40, 3))]
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    s, o = util.scene_pair_ranges(f32(items), f32([(0.0, 0.0, 0.0, 4.0)]), np.asarray([(0, 40)], dtype=np.int32))
    regs = bucket_list(96, 64)
    ref, _, _ = o.render(96, 64, 1, os.cpu_count() or 1, HIER_EXIT)
    data, n = render(s, 96, 64, regs, 2)
    assert n is None or n == 0
    np.testing.assert_array_equal(util.stitch((96, 64), regs, data), ref)


def test_100k_spheres_automatic_hierarchy():
    from tests.scenes import hundred_thousand_spheres
    s = rta.Scene.from_spheres_auto(hundred_thousand_spheres())
    o = oracle.Scene.from_ranges(s.items.astype(np.float64), s.bounds.astype(np.float64), s.ranges)
    w, h = 512, 384
    regs = bucket_list(w, h)
    ref, _, _ = o.render(w, h, 1, os.cpu_count() or 1, HIER_EXIT)
    data, n = render(s, w, h, regs, 2)
    assert n is None or n > 0
    np.testing.assert_array_equal(util.stitch((w, h), regs, data), ref)
