import os
import sys

import pytest

# PyTorch-ROCm wheels bundle their own libamdhip64; librtrace_hip.so links the system one.  Both have the same SONAME,
# so whichever is loaded first serves the whole process -- and torch only works on its own.  Tests that hand torch
# tensors / streams to the C ABI therefore need torch imported BEFORE the backend library is first loaded.
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The parity tests need the diagnostic controls of csrc/rt_debug.h (loop flavours, forced cooperative walk, violation counters, the stand-in
# for librccl.so): they load tests/c/librtrace_hip_test.so -- the product's sources built with -DRT_TEST_HOOKS -- instead of the product
# library, which has none of them.  Child processes that must run the PRODUCT (rtrace, bench.py) get an environment without this.
# (tests/util.py product_env)
#
# RTRACE_PARITY_ON_PRODUCT=1 (tests/test_gpu_product.py starts such a child `pytest -m gpu`): this process loads the PRODUCT library instead and
# runs every test that needs no control -- the library that ships under the same oracle comparisons.  A test that reaches for a control
# (capi.debug / debug_set / debug_count / the stand-in for librccl.so ...) is SKIPPED at that point; the fixture below then only compares the
# counting launch with the one that does not count.
PRODUCT_RUN = os.environ.get("RTRACE_PARITY_ON_PRODUCT") == "1"
TEST_LIB = os.path.join(ROOT, "tests", "c", "librtrace_hip_test.so")
if PRODUCT_RUN:
    os.environ.pop("RTRACE_HIP_LIBRARY", None)
else:
    os.environ["RTRACE_HIP_LIBRARY"] = TEST_LIB


# The test modules import the package (and the oracle) at module scope, and both need their built libraries: on a fresh
# checkout (the .so files are git-ignored) build them here, before collection, instead of failing with an ImportError.
if not (os.path.exists(os.path.join(ROOT, "rust-tracer_amd", "librtrace_hip.so")) and os.path.exists(TEST_LIB) and
        os.path.exists(os.path.join(ROOT, "oracle", "librt_oracle.so")) and
        os.path.exists(os.path.join(ROOT, "rust-tracer_amd", "rtrace"))):
    import __graft_entry__
    __graft_entry__.build()

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if PRODUCT_RUN:
        import rust_tracer_amd as rta
        assert rta.capi.LIB_PATH == rta.capi.PRODUCT_LIB_PATH and not rta.capi.HAVE_TEST_HOOKS, "RTRACE_PARITY_ON_PRODUCT: the product library did not load"

        def needs_hooks(what):
            pytest.skip("needs a control of csrc/rt_debug.h (%s): not in the library that ships" % what)
        rta.capi._need_hooks = needs_hooks


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _check_both_launch_flavours(request, monkeypatch):
    """The kernels have two flavours: with ray/test counters (C++ traversal loops) and without (generated assembly
    loops, the flavour bench.py and the async entry points use).  Every -m gpu test that renders with counters is
    made to render once more without them, and the bytes must be identical -- so all oracle comparisons cover both."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import numpy as np
    import rust_tracer_amd as rta
    orig = rta.DeviceScene.render_tiles
    if PRODUCT_RUN:
        def both_product(self, options, regions, traversal=None, want_stats=True, out=None):
            import time
            data, st = orig(self, options, regions, traversal, want_stats, out)
            if want_stats:
                plain, _ = orig(self, options, regions, traversal, False)
                assert np.array_equal(plain, data), "the launch without counters renders different bytes"
                if options[2] == 1 and traversal in (None, rta.RT_TRAVERSAL_SKIP):
                    # the library makes a new tile list's dispatch orders in the background (a few ms): the launches above found their blocks
                    # through the tile table.  Once more when the orders are there -- f32: k_render_skip_fast, what a scheduler's frames run
                    for _ in range(100):
                        again, _ = orig(self, options, regions, traversal, False)
                        assert np.array_equal(again, data), "a later launch of the same list renders different bytes (%r)" % (rta.capi.last_launch(),)
                        if "ordered" in rta.capi.last_launch():
                            break
                        time.sleep(0.002)
            return data, st
        monkeypatch.setattr(rta.DeviceScene, "render_tiles", both_product)
        yield
        return
    # the dispatch orders of a tile list are normally made by a background thread while the first launches walk the tile table
    # (rt_capi.hip build_orders_async); the tests want the ordered / narrowed / cooperative dispatch in the very launch they look at
    rta.capi.debug_set(rta.capi.DEBUG_ASYNC_ORDERS, 0)

    def both(self, options, regions, traversal=None, want_stats=True, out=None):
        data, st = orig(self, options, regions, traversal, want_stats, out)
        if want_stats:
            plain, _ = orig(self, options, regions, traversal, False)
            assert np.array_equal(plain, data), "the launch without counters renders different bytes"
            if options[2] == 1 and traversal in (None, rta.RT_TRAVERSAL_SKIP):
                # ... and once more with EVERY quad walked by the lane-cooperative gather (csrc/rt_coop.hpp; f32 scenes that have the
                # cooperative copy -- for the others the control changes nothing)
                with rta.capi.debug(rta.capi.DEBUG_COOP, 2):
                    coop, _ = orig(self, options, regions, traversal, False)
                assert np.array_equal(coop, data), "the lane-cooperative walk renders different bytes"
                # ... and with the lean kernel (rt_skip_fast.hpp: f32, a dispatch list) where `plain` ran the generic one, and the other way round
                for fast in (0, 2):
                    with rta.capi.debug(rta.capi.DEBUG_FAST_KERNEL, fast):
                        other, _ = orig(self, options, regions, traversal, False)
                    assert np.array_equal(other, data), "k_render_skip_f32 and k_render_skip_fast render different bytes (%d)" % fast
                with rta.capi.debug(rta.capi.DEBUG_COOP, 2), rta.capi.debug(rta.capi.DEBUG_FAST_KERNEL, 0):
                    coop0, _ = orig(self, options, regions, traversal, False)
                assert np.array_equal(coop0, data), "the lane-cooperative walk in the generic kernel renders different bytes"
            elif traversal in (None, rta.RT_TRAVERSAL_SKIP):
                # several samples per pixel: where two rays per lane walk the pass (large passes; or asked for), k_render_skip2 and its lean twin
                # (rt_skip2_fast.hpp) -- `plain` ran one of them, this is the other
                for fast in (0, 2):
                    with rta.capi.debug(rta.capi.DEBUG_FAST_KERNEL, fast):
                        other, _ = orig(self, options, regions, traversal, False)
                    assert np.array_equal(other, data), "the generic and the lean kernels render different bytes (%d)" % fast
        return data, st

    monkeypatch.setattr(rta.DeviceScene, "render_tiles", both)
    yield
    # every counting launch of the f32 hierarchy walk also evaluated the filtered loops' bounds for each test it made
    # (rt_skip.hpp): none that returned a finite distance may have been ruled out
    assert rta.capi.debug_count(rta.capi.DEBUG_COUNT_FILTER_VIOLATIONS) == 0, "the filtered loops' bound ruled out a hit"
