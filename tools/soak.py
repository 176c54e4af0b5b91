#!/usr/bin/env python3
"""Soak / fuzz run for the generated assembly loops (GPU box): bitwise stability over many launches, and random nested scenes on
which every hierarchy-walk flavour (C++, assembly, fused assembly, filtered, two rays per lane) must agree with the counted C++ flavour
(which the parity tests pin to the CPU path) and the filtered scalar-fed flat scan with round 1's LDS kernels.
usage: soak.py [seconds]"""
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RTRACE_HIP_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "c", "librtrace_hip_test.so"))      # rt_debug.h's controls: the hooks build
import numpy as np
import torch
import rust_tracer_amd as rta
from tests import scenes as util        # numpy-only scene generators (the oracle is not loaded by this tool)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
t_end = time.time() + budget
stream = torch.cuda.current_stream().cuda_stream

# 1. bitwise stability of the default path, many launches back to back
s = rta.Scene.default()
d = s.device()
t_phase = time.time()
for (w, h, spp, trav, share, batch) in ((1920, 1080, 1, rta.RT_TRAVERSAL_SKIP, 0.15, 200), (1024, 768, 4, rta.RT_TRAVERSAL_SKIP, 0.10, 200),
                                       (800, 600, 1, rta.RT_TRAVERSAL_SKIP, 0.10, 200),            # round 4: partly lane-cooperative once the library's orders are in
                                       (640, 480, 1, rta.RT_TRAVERSAL_SKIP, 0.05, 200),
                                       (3840, 2160, 1, rta.RT_TRAVERSAL_SKIP, 0.05, 50),           # k_render_skip2 by default
                                       (1920, 1080, 1, rta.RT_TRAVERSAL_FLAT, 0.10, 10)):
    regs = d._regions([tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, spp))])
    out = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    d.render_frame_device((w, h, spp), regs, out.data_ptr(), stream, trav)
    torch.cuda.synchronize()
    want = zlib.crc32(out.cpu().numpy().tobytes())
    n = 0
    t_phase += budget * share
    while time.time() < t_phase:
        for _ in range(batch):
            d.render_frame_device((w, h, spp), regs, out.data_ptr(), stream, trav)
        torch.cuda.synchronize()
        assert zlib.crc32(out.cpu().numpy().tobytes()) == want, "frame changed after %d launches" % n
        n += batch
    print("stable: %dx%d spp %d %s, %d launches" % (w, h, spp, "flat" if trav == rta.RT_TRAVERSAL_FLAT else "skip", n), flush=True)

# 2. random nested scenes (every other one concentric): flavours 0/3/7 (no counters) vs flavour 1 with counters
seed = 1000
checked = 0
filter_tests = 0
coop_scenes = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    depth, fan, leaf = int(rng.integers(2, 6)), int(rng.integers(2, 5)), int(rng.integers(1, 4))
    items, bounds, ranges = util.random_nested_scene(seed, depth=depth, fan=fan, leaf_items=leaf, concentric=seed % 2 == 1)
    # round 3: the filtered loops' constants depend on the light, the eye and the scene's extent -- vary all three.  One scene in
    # eight puts the eye inside the hierarchy, one in four scales every coordinate (margins are relative, subnormals are not).
    eye = (float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-4.5, -1.0)))
    if seed % 8 == 6:
        eye = tuple(float(v) for v in rng.uniform(-0.8, 0.8, 3))
    light = rng.normal(size=3)
    if seed % 3 == 0 or np.linalg.norm(light) < 0.2:
        light = np.array((-1.0, -3.0, 2.0))
    k = float(2.0 ** int(rng.integers(-30, 31))) if seed % 4 == 2 else 1.0
    items, bounds, eye = items * k, bounds * k, tuple(v * k for v in eye)
    prec = rta.RT_F64 if seed % 4 == 3 else rta.RT_F32
    sc = rta.Scene(items, rta.normalized(tuple(light), prec), eye, bounds, ranges, prec)
    dv = sc.device()
    w, h, spp = int(rng.integers(3, 9)) * 32, int(rng.integers(3, 9)) * 24, int(rng.integers(1, 3))
    regs = [tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, spp))]
    rta.capi.debug_set(rta.capi.DEBUG_SKIP_VARIANT, 1)
    ref, st = dv.render_tiles((w, h, spp), regs, rta.RT_TRAVERSAL_SKIP, want_stats=True)
    flavours = (0, 3, 7, 19, 23)             # (f64: 19 / 23 put the f32 filter in front of the primary walk)
    for v in flavours:
        rta.capi.debug_set(rta.capi.DEBUG_SKIP_VARIANT, v)
        got, _ = dv.render_tiles((w, h, spp), regs, rta.RT_TRAVERSAL_SKIP, want_stats=False)
        assert np.array_equal(got, ref), "seed %d: flavour %d differs" % (seed, v)
    if spp == 1:
        # round 4 (f64: round 6): every quad through the lane-cooperative gather (rt_coop.hpp) -- rays whose winner is nearer than an ancestor bound go back to the loops
        rta.capi.debug_set(rta.capi.DEBUG_SKIP_VARIANT, -1)
        with rta.capi.debug(rta.capi.DEBUG_COOP, 2):
            got, _ = dv.render_tiles((w, h, spp), regs, rta.RT_TRAVERSAL_SKIP, want_stats=False)
        assert np.array_equal(got, ref), "seed %d: the cooperative walk differs" % seed
        coop_scenes += 1
    if True:
        # the counting launch evaluated the filtered loops' bounds NEXT TO the reference's arithmetic for every test it made and
        # counts a violation whenever a bound rules out what the arithmetic finds (rt_skip.hpp, COUNT mode)
        assert rta.capi.debug_count(rta.capi.DEBUG_COUNT_FILTER_VIOLATIONS) == 0, "seed %d: a bound ruled out a hit" % seed
        filter_tests += st["sphere_tests"] + st["bound_tests"]
    rta.capi.debug_set(rta.capi.DEBUG_SKIP_VARIANT, -1)
    if prec == rta.RT_F32:
        if True:                                                 # the two-ray kernel: fused loops for concentric scenes, plain-stream loops for the others
            with rta.capi.debug(rta.capi.DEBUG_SKIP_RAYS, 2):
                got, _ = dv.render_tiles((w, h, spp), regs, rta.RT_TRAVERSAL_SKIP, want_stats=False)
            assert np.array_equal(got, ref), "seed %d: two rays per lane differ" % seed
    # both precisions: the filtered flat scan (f32: scalar-fed, two rays per lane; f64: rt_flat_f64.hpp) against round 1's unfiltered LDS kernels
    flat, _ = dv.render_tiles((w, h, spp), regs, rta.RT_TRAVERSAL_FLAT, want_stats=False)
    flat = flat.copy()
    with rta.capi.debug(rta.capi.DEBUG_FLAT_KERNELS, 0):
        lds, _ = dv.render_tiles((w, h, spp), regs, rta.RT_TRAVERSAL_FLAT, want_stats=False)
    assert np.array_equal(flat, lds), "seed %d: filtered flat scan differs from the LDS kernels" % seed
    dv.close()
    checked += 1
    seed += 1
    if checked % 2000 == 0:
        print("fuzz: %d scenes so far" % checked, flush=True)        # a long run must not look hung
print("fuzz: %d random scenes, all flavours identical (hierarchy: C++ / assembly / fused / filtered / two rays; flat: filtered scan / LDS kernels); "
      "the lane-cooperative walk on %d of them; 0 filter violations in %d counted tests" % (checked, coop_scenes, filter_tests))
