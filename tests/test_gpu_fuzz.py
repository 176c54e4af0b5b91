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
