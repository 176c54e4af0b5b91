"""The PRODUCT library on the GPU.  The pytest process itself runs on tests/c/librtrace_hip_test.so (the same sources with -DRT_TEST_HOOKS:
the parity tests need csrc/rt_debug.h's controls); what ships -- rust-tracer_amd/librtrace_hip.so, no rt_debug_* entry point, only the loop
flavours a scene gets by itself -- renders the committed vectors here in a child process of its own (one library per process), through
every entry point that needs no control: counted and uncounted launches, both traversals, host and device destinations."""
import json
import os
import subprocess
import sys

import pytest

from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import json, os, sys, zlib
import numpy as np
import torch                                       # before the backend library (tests/conftest.py says why)
sys.path.insert(0, sys.argv[1])
import rust_tracer_amd as rta
from rust_tracer_amd import capi
from tests import util                             # (imports the oracle's binding for its scene helpers; nothing below calls the oracle)
from tests.test_gpu_parity import _scene_for, bucket_list

assert capi.LIB_PATH == capi.PRODUCT_LIB_PATH and not capi.HAVE_TEST_HOOKS and "RT_TEST_HOOKS" not in capi.build_info()
for name in capi.DEBUG_SYMBOLS:
    assert not hasattr(capi.lib, name), name
cases = json.load(open(os.path.join(sys.argv[1], "tests", "golden", "oracle_vectors.json")))["cases"]
big = int(sys.argv[2])
done = []
for case in cases:
    w, h, spp = case["width"], case["height"], case["spp"]
    if w * h * spp * spp > big:
        continue
    s = _scene_for(case)
    d = s.device()
    regs = bucket_list(w, h, spp)
    travs = [rta.RT_TRAVERSAL_FLAT] if case.get("flat") else [rta.RT_TRAVERSAL_SKIP]
    if not case.get("flat") and case["scene"] not in ("inside", "hundred_thousand_spheres") and w * h * spp * spp <= 2100000:
        travs.append(rta.RT_TRAVERSAL_FLAT)
    for trav in travs:
        for want_stats in (True, False):           # the counting kernels (C++ loops) and the product's own (generated assembly loops)
            data, st = d.render_tiles((w, h, spp), regs, trav, want_stats)
            off = 0
            for i, (l, t, r, b) in enumerate(regs):
                n = (r - l) * (t - b) * 4
                assert zlib.crc32(data[off:off + n].tobytes()) & 0xFFFFFFFF == case["tile_crc32"][i], (case["name"], trav, want_stats, i)
                off += n
            if want_stats:
                for k in ("primary", "hits", "shadow", "occluded"):
                    assert st[k] == case["stats"][k], (case["name"], k)
                if trav == rta.RT_TRAVERSAL_SKIP:
                    assert (st["sphere_tests"], st["bound_tests"]) == (case["stats"]["sphere_tests"], case["stats"]["bound_tests"]), case["name"]
                    assert 0 < st["primary_tests"] < st["sphere_tests"] + st["bound_tests"]
                flags = capi.last_launch()
                assert "counting" in flags and ("flat_pipeline" in flags) == (trav == rta.RT_TRAVERSAL_FLAT), flags
            else:
                assert "counting" not in capi.last_launch()
    # the asynchronous entry point into device memory (what bench.py times), twice: through the tile table, then in dispatch order
    if not case.get("flat"):
        frame = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
        for _ in range(2):
            frame.zero_()
            d.render_frame_device((w, h, spp), d._regions(regs), frame.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert zlib.crc32(frame.cpu().numpy().tobytes()) & 0xFFFFFFFF == case["frame_crc32"], case["name"]
    done.append(case["name"])
print("OK " + json.dumps(done))
"""


def test_the_product_library_renders_the_committed_vectors(tmp_path):
    script = tmp_path / "product_vectors.py"
    script.write_text(_SCRIPT)
    r = subprocess.run([sys.executable, str(script), ROOT, str(1024 * 768 * 16)], capture_output=True, text=True, timeout=900, env=util.product_env())
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("OK ")]
    assert line, r.stdout[-2000:]
    done = json.loads(line[0][3:])
    for name in ("config1_three_spheres_64x64", "config2_800x600", "config3_1920x1080_f32", "config3_1920x1080_f64", "make_image_1024x768_spp4",
                 "level9_320x256_spp2", "tie_break_64x64", "inside_bound_hierarchy_64x64", "inside_bound_flat_64x64", "config5_100k_512x512_spp4"):
        assert name in done, done


def test_smoke_runs_on_the_product_library():
    # __graft_entry__.smoke() is what the driver runs before the bench: in a fresh process it must load the product library
    code = ("import sys; sys.path.insert(0, %r); import torch; import __graft_entry__ as g; g.smoke(); import rust_tracer_amd as rta; "
            "assert not rta.capi.HAVE_TEST_HOOKS; print('library', rta.capi.LIB_PATH)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=util.product_env())
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    assert "smoke ok" in r.stdout and "librtrace_hip.so" in r.stdout and "librtrace_hip_test" not in r.stdout


def test_the_parity_suite_on_the_library_that_ships():
    """the oracle comparisons of the whole `-m gpu` suite -- fuzz scenes, ragged regions, 2 .. 300 samples per pixel, the
    tie-break scenes in f32 and f64, zero samples, the largest frame, config 5 at full size -- run once more in a child pytest whose process
    loads rust-tracer_amd/librtrace_hip.so itself (tests/conftest.py, RTRACE_PARITY_ON_PRODUCT): what ships, not its -DRT_TEST_HOOKS twin.
    Tests that reach for a control of csrc/rt_debug.h skip themselves there; the rest must pass, and there must be many of them."""
    import re
    if os.environ.get("RTRACE_PARITY_ON_PRODUCT") == "1":
        pytest.skip("this IS the child run")
    log = os.path.join(ROOT, "gpurun_out", "product_parity.log")
    os.makedirs(os.path.dirname(log), exist_ok=True)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-v", "-m", "gpu", "-p", "no:cacheprovider", "-rs"],
                       env=util.product_env(RTRACE_PARITY_ON_PRODUCT="1"), cwd=ROOT, capture_output=True, text=True, timeout=2400)
    open(log, "w").write(r.stdout + r.stderr)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0, (tail, r.stdout[-3000:], r.stderr[-2000:])
    m = re.search(r"(\d+) passed", tail)
    assert m, tail
    assert "failed" not in tail and "error" not in tail, tail
    assert int(m.group(1)) >= MIN_PRODUCT_PASSES, "only %s tests ran on the product library: %s" % (m.group(1), tail)


MIN_PRODUCT_PASSES = 300
