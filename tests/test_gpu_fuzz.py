"""Randomised parity: nested scenes of random depth / fan-out / bound tightness, random eye (often inside bounds), random
image sizes and sample counts -- the HIP path (both traversals, counted and plain launches) against the CPU oracle."""
import os

import numpy as np
import pytest

import oracle
import rust_tracer_amd as rta
from tests import util

pytestmark = pytest.mark.gpu

HIER_EXIT = oracle.MODE_HIERARCHY | oracle.MODE_ANYHIT_EXIT


@pytest.mark.parametrize("seed", range(2000, 2040))
def test_random_scene(seed):
    rng = np.random.default_rng(seed)
    depth, fan, leaf = int(rng.integers(2, 6)), int(rng.integers(2, 5)), int(rng.integers(1, 4))
    # every third scene is concentric (a group's first child sits at the centre of its bound): the fused traversal loops
    items, bounds, ranges = util.random_nested_scene(seed, depth=depth, fan=fan, leaf_items=leaf, concentric=seed % 3 == 1)
    eye = (float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-4.5, -1.0)))
    light = (float(rng.uniform(-2, 2)), float(rng.uniform(-3, -0.5)), float(rng.uniform(-2, 2)))
    precision = rta.RT_F64 if seed % 5 == 0 else rta.RT_F32
    s, o = util.scene_pair_ranges(items, bounds, ranges, precision, light=light, eye=eye)
    w, h, spp = int(rng.integers(2, 7)) * 32 + int(rng.integers(0, 17)), int(rng.integers(2, 7)) * 24 + int(rng.integers(0, 13)), int(rng.integers(1, 4))
    regs = [tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, spp))]
    skip, st = s.device().render_tiles((w, h, spp), regs, rta.RT_TRAVERSAL_SKIP)
    ref, rst, _ = o.render(w, h, spp, os.cpu_count() or 1, HIER_EXIT)
    np.testing.assert_array_equal(util.stitch((w, h), regs, skip), ref)
    assert util.all_stats(st) == util.all_stats(rst)
    flat, fst = s.device().render_tiles((w, h, spp), regs, rta.RT_TRAVERSAL_FLAT)
    fref, frst, _ = o.render(w, h, spp, os.cpu_count() or 1, oracle.MODE_FLAT)
    np.testing.assert_array_equal(util.stitch((w, h), regs, flat), fref)
    assert util.ray_stats(fst) == util.ray_stats(frst)
    if precision == rta.RT_F32:
        # the flat scan's conservative filter, pair by pair (every ray x every item of this frame): no candidate rejected.  (The hierarchy
        # walk's bounds were checked test by test by the counting launch above: conftest asserts rt_debug_count(FILTER_VIOLATIONS) == 0.)
        if rta.capi.HAVE_TEST_HOOKS:
            c = rta.capi.flat_filter_check(s.device()._h, w, h, spp)
            assert c[2] == 0 and c[5] == 0, c


@pytest.mark.parametrize("variant", [0, 1, 3, 7, 19, 23])
@pytest.mark.parametrize("precision", [rta.RT_F32, rta.RT_F64], ids=["f32", "f64"])
@pytest.mark.parametrize("scale", [1e-20, 1e-10, 1e6, 1e12, 5e13])
def test_scaled_scenes_every_loop_flavour(scale, precision, variant):
    # The whole scene (items, bounds, eye) scaled by 1e-20 ... 5e13: at 1e-20 the f32 squares (rr, vv, b*b ~ 1e-40) are
    # denormals, so every root goes through the traversal loops' scaled `tiny` branches (and the C++ lean sqrt's general path);
    # at 5e13 coordinates sit just under the 1e15 validation bound and squares reach 1e29.  Denormals are kept on both sides.
    # All four loop flavours (0/1 C++, 3 generated assembly, 7 fused; the library drops the fused bit for the nested scene)
    # against the oracle: pixels, alpha and every counter.
    for seed, concentric in ((31, False), (32, True)):
        items, bounds, ranges = util.random_nested_scene(seed, depth=3, fan=3, leaf_items=2, concentric=concentric)
        R = np.float32 if precision == rta.RT_F32 else np.float64
        sc = lambda a: (np.asarray(a, dtype=np.float64) * scale).astype(R).astype(np.float64)
        eye = tuple(float(v) for v in sc((0.07, -0.12, -3.1)))
        s, o = util.scene_pair_ranges(sc(items), sc(bounds), ranges, precision, eye=eye)
        w, h, spp = 96, 72, 2
        regs = [tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, spp))]
        ref, rst, _ = o.render(w, h, spp, os.cpu_count() or 1, HIER_EXIT)
        assert rst["hits"] > 500 and rst["shadow"] > 100           # the scaled scene is still in view
        with util.loop_flavour(variant):
            plain, _ = s.device().render_tiles((w, h, spp), regs, rta.RT_TRAVERSAL_SKIP, want_stats=False)
            counted, st = s.device().render_tiles((w, h, spp), regs, rta.RT_TRAVERSAL_SKIP, want_stats=True)
        np.testing.assert_array_equal(util.stitch((w, h), regs, plain), ref)
        np.testing.assert_array_equal(counted, plain)
        assert util.all_stats(st) == util.all_stats(rst)


@pytest.mark.parametrize("seed", [2101, 2104, 2107, 2110, 2113, 2116, 2119, 2122])
def test_two_rays_per_lane_walk_on_random_concentric_scenes(seed):
    # k_render_skip2 (rt_skip2.hpp: two rays per lane on packed math, generated rt_skip2_rot.hpp) serves fused f32 scenes at spp 1 and
    # the sample-packed spp 2 / 4 / 8; by default only large frames use it, csrc/rt_debug.h RT_DEBUG_SKIP_RAYS = 2 forces it.  Random
    # concentric scenes, random eye (often inside bounds), ragged image sizes: pixels and alpha against the oracle, bytes against
    # the one-ray kernel.
    rng = np.random.default_rng(seed)
    depth, fan, leaf = int(rng.integers(2, 6)), int(rng.integers(2, 5)), int(rng.integers(1, 4))
    items, bounds, ranges = util.random_nested_scene(seed, depth=depth, fan=fan, leaf_items=leaf, concentric=True)
    eye = (float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-4.5, -1.0)))
    light = (float(rng.uniform(-2, 2)), float(rng.uniform(-3, -0.5)), float(rng.uniform(-2, 2)))
    s, o = util.scene_pair_ranges(items, bounds, ranges, rta.RT_F32, light=light, eye=eye)
    for spp in (1, 2, 4, 8):
        w, h = int(rng.integers(2, 7)) * 32 + int(rng.integers(0, 17)), int(rng.integers(2, 5)) * 24 + int(rng.integers(0, 13))
        regs = [tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, spp))]
        ref, rst, _ = o.render(w, h, spp, os.cpu_count() or 1, HIER_EXIT)
        with util.control(rta.capi.DEBUG_SKIP_RAYS, 2):
            two, _ = s.device().render_tiles((w, h, spp), regs, rta.RT_TRAVERSAL_SKIP, want_stats=False)
        with util.control(rta.capi.DEBUG_SKIP_RAYS, 1):
            one, _ = s.device().render_tiles((w, h, spp), regs, rta.RT_TRAVERSAL_SKIP, want_stats=False)
        np.testing.assert_array_equal(util.stitch((w, h), regs, two), ref)
        np.testing.assert_array_equal(two, one)


@pytest.mark.parametrize("seed", [2201, 2204, 2207, 2210, 2213, 2216, 2219, 2222])
def test_two_rays_per_lane_walk_on_random_scenes_whose_bounds_have_no_sphere_of_their_own(seed):
    # round 4: the plain-stream flavour of k_render_skip2 (non-concentric hierarchies: what the library builds for an arbitrary sphere
    # list) -- bounds that need not enclose their subtrees, the eye often inside them, ragged sizes, spp 1 / 2 / 4 / 8
    rng = np.random.default_rng(seed)
    depth, fan, leaf = int(rng.integers(2, 6)), int(rng.integers(2, 5)), int(rng.integers(1, 4))
    items, bounds, ranges = util.random_nested_scene(seed, depth=depth, fan=fan, leaf_items=leaf, concentric=False)
    eye = (float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-4.5, -1.0)))
    light = (float(rng.uniform(-2, 2)), float(rng.uniform(-3, -0.5)), float(rng.uniform(-2, 2)))
    s, o = util.scene_pair_ranges(items, bounds, ranges, rta.RT_F32, light=light, eye=eye)
    before = rta.capi.debug_count(rta.capi.DEBUG_COUNT_TWO_RAY_LAUNCHES) if rta.capi.HAVE_TEST_HOOKS else None
    for spp in (1, 2, 4, 8):
        w, h = int(rng.integers(2, 7)) * 32 + int(rng.integers(0, 17)), int(rng.integers(2, 5)) * 24 + int(rng.integers(0, 13))
        regs = [tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, spp))]
        ref, rst, _ = o.render(w, h, spp, os.cpu_count() or 1, HIER_EXIT)
        with util.control(rta.capi.DEBUG_SKIP_RAYS, 2):
            two, _ = s.device().render_tiles((w, h, spp), regs, rta.RT_TRAVERSAL_SKIP, want_stats=False)
        with util.control(rta.capi.DEBUG_SKIP_RAYS, 1):
            one, _ = s.device().render_tiles((w, h, spp), regs, rta.RT_TRAVERSAL_SKIP, want_stats=False)
        np.testing.assert_array_equal(util.stitch((w, h), regs, two), ref)
        np.testing.assert_array_equal(two, one)
    assert before is None or rta.capi.debug_count(rta.capi.DEBUG_COUNT_TWO_RAY_LAUNCHES) >= before + 4


@pytest.mark.parametrize("scale", [1e-20, 1e-10, 1e6, 5e13])
def test_two_rays_per_lane_walk_on_scaled_scenes(scale):
    # the exact path of the two-ray loops per half, including the scaled `tiny` branches (1e-20: every square is a denormal)
    items, bounds, ranges = util.random_nested_scene(32, depth=3, fan=3, leaf_items=2, concentric=True)
    sc = lambda a: (np.asarray(a, dtype=np.float64) * scale).astype(np.float32).astype(np.float64)
    eye = tuple(float(v) for v in sc((0.07, -0.12, -3.1)))
    s, o = util.scene_pair_ranges(sc(items), sc(bounds), ranges, rta.RT_F32, eye=eye)
    for (w, h, spp) in ((96, 72, 2), (75, 50, 1), (40, 33, 4)):
        regs = [tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, spp))]
        ref, rst, _ = o.render(w, h, spp, os.cpu_count() or 1, HIER_EXIT)
        assert rst["hits"] > 200 and rst["shadow"] > 50
        with util.control(rta.capi.DEBUG_SKIP_RAYS, 2):
            two, _ = s.device().render_tiles((w, h, spp), regs, rta.RT_TRAVERSAL_SKIP, want_stats=False)
        np.testing.assert_array_equal(util.stitch((w, h), regs, two), ref)


def test_inputs_that_could_overflow_are_rejected():
    # the no-NaN argument rests on these bounds: coordinates and eye within 1e15, light a unit vector
    items = np.array([[0, 0, 0, 1.0]])
    for kw in (dict(eye=(0, 0, -2e15)), dict(light=(0.0, -3.0, 0.0)), dict(light=(0.0, -1.01, 0.0)), dict(light=(0.5, -0.5, 0.5)),
               dict(items=np.array([[2e15, 0, 0, 1.0]]))):
        it = kw.pop("items", items)
        light = np.asarray(kw.pop("light", rta.normalized((-1, -3, 2), rta.RT_F32)), dtype=np.float64)
        s = rta.Scene(it, light, kw.pop("eye", (0, 0, -4)), np.array([[0, 0, 0, 3.0]]), np.array([[0, 1]], dtype=np.int32))
        with pytest.raises(rta.RtError) as e:
            s.device()
        assert e.value.status == rta.capi.RT_ERR_INVALID_ARGUMENT
