"""Shared helpers for the parity tests: the same scene built on both sides (product host mirror / oracle)."""
import numpy as np

import oracle
import rust_tracer_amd as rta

PREC = {rta.RT_F32: oracle.F32, rta.RT_F64: oracle.F64}


def product_env(**extra):
    """os.environ for a child process that must load the PRODUCT library (rust-tracer_amd/librtrace_hip.so), not the -DRT_TEST_HOOKS build
    tests/conftest.py points this process at."""
    import os
    env = {k: v for k, v in os.environ.items() if k != "RTRACE_HIP_LIBRARY"}
    env.update(extra)
    return env

THREE_SPHERES = [(0.0, -1.0, 0.0, 1.0), (-1.2, 0.2, 0.0, 0.5), (1.2, 0.2, 0.0, 0.5)]
THREE_BOUND = (0.0, -1.0, 0.0, 3.0)

# two overlapping spheres mirrored in x: every ray of the pixel column x == width/2 (dir.x == 0 exactly) hits both
# at exactly the same f32 distance, but their normals differ in x and light.x != 0 -> the colour tells who won.
TIE_SPHERES = [(-0.3, 0.0, 0.0, 1.0), (0.3, 0.0, 0.0, 1.0)]
TIE_BOUND = (0.0, 0.0, 0.0, 3.0)


def scene_pair_default(precision=rta.RT_F32, level=8):
    return rta.Scene.default(level, precision), oracle.Scene.default(PREC[precision], level)


def scene_pair_spheres(spheres, bound, precision=rta.RT_F32, light=(-1.0, -3.0, 2.0), eye=(0.0, 0.0, -4.0)):
    return (rta.Scene.from_spheres(spheres, bound, light, eye, precision),
            oracle.Scene.from_spheres(spheres, bound, light, eye, PREC[precision]))


def stitch(options, regions, data):
    """tile-major bytes -> uint8[h, w, 4] frame (host-side set_pixels_from_buffer)."""
    w, h = options[0], options[1]
    frame = np.zeros((h, w, 4), dtype=np.uint8)
    off = 0
    for (l, t, r, b) in regions:
        n = (r - l) * (t - b) * 4
        frame[b:t, l:r] = data[off:off + n].reshape(t - b, r - l, 4)
        off += n
    assert off == data.size
    return frame


def ray_stats(st):
    return tuple(int(st[k]) for k in ("primary", "hits", "shadow", "occluded"))


# the eye lies INSIDE group A's bound, so A's bound "distance" is its exit distance (primitive.rs:70-71)
# and A is culled (group.rs:73) although it holds the nearer item Y.  hierarchy -> X, flat scan -> Y.
INSIDE_ITEMS = [(0.0, 0.0, -4.0 + 2.3, 0.4), (0.0, 0.0, -4.0 + 2.0, 0.5)]
INSIDE_BOUNDS = [(0.0, 0.0, -2.0, 10.0), (0.0, 0.0, -4.0 + 0.3, 2.3)]
INSIDE_RANGES = [(0, 2), (1, 1)]


def scene_pair_ranges(items, bounds, ranges, precision=rta.RT_F32, light=(-1.0, -3.0, 2.0), eye=(0.0, 0.0, -4.0)):
    s = rta.Scene(np.asarray(items, dtype=np.float64), rta.normalized(light, precision), eye,
                  np.asarray(bounds, dtype=np.float64), np.asarray(ranges, dtype=np.int32), precision)
    return s, oracle.Scene.from_ranges(items, bounds, ranges, light, eye, PREC[precision])


from tests.scenes import random_nested_scene  # noqa: E402,F401  (pure numpy; tools/soak.py uses it without the oracle)


def all_stats(st):
    return tuple(int(st[k]) for k in ("primary", "hits", "shadow", "occluded", "sphere_tests", "bound_tests"))


def loop_flavour(variant):
    """with util.loop_flavour(v): the traversal-loop flavour v (csrc/rt_debug.h RT_DEBUG_SKIP_VARIANT) for the launches inside.  The library that
    ships has no such control (tests/conftest.py RTRACE_PARITY_ON_PRODUCT): there flavour 23 -- the filtered assembly loops, fused where the
    scene is concentric: what a scene gets by itself -- is simply the launch as it is, and every other flavour skips the test."""
    import contextlib
    import pytest
    if rta.capi.HAVE_TEST_HOOKS:
        return rta.capi.debug(rta.capi.DEBUG_SKIP_VARIANT, variant)
    if variant == 23:
        return contextlib.nullcontext()
    pytest.skip("loop flavour %d needs csrc/rt_debug.h: the library that ships runs its own choice (23)" % variant)


def control(key, value):
    """with util.control(rta.capi.DEBUG_X, v): a control of csrc/rt_debug.h for the launches inside -- where the library has them.  On the library
    that ships (no control) the block runs the library's own choice: the oracle comparison inside still holds, the forced flavour is what
    the ordinary run of the suite covers.  Use it only where nothing inside asserts WHICH flavour ran."""
    import contextlib
    if rta.capi.HAVE_TEST_HOOKS:
        return rta.capi.debug(key, value)
    return contextlib.nullcontext()
