#!/usr/bin/env python3
"""Generates tests/golden/oracle_vectors.json from the CPU oracle -- AFTER the oracle has been pinned on the
reference's own image (make_golden.py / tests/test_oracle_golden.py).  These vectors are therefore pinned only
transitively: they cover what the reference's PNG cannot (other resolutions, the alpha channel, clipped buckets,
pyramid level 9, the f64 type-alias swap, synthetic tie / inside-bound scenes).

Each case: per-bucket CRC32 of the RGBA tile bytes (64x64 buckets, row-major, clipped) + the ray counters."""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from tests import util  # noqa: E402


def tile_crcs(frame):
    h, w = frame.shape[:2]
    out = []
    for y in range(0, h, 64):
        for x in range(0, w, 64):
            out.append(zlib.crc32(np.ascontiguousarray(frame[y:y + 64, x:x + 64]).tobytes()) & 0xFFFFFFFF)
    return out


def case(name, scene, w, h, spp, mode=oracle.MODE_HIERARCHY | oracle.MODE_ANYHIT_EXIT):
    frame, st, n = scene.render(w, h, spp, nthreads=os.cpu_count() or 1, mode=mode)
    return {"name": name, "width": w, "height": h, "spp": spp, "buckets": n, "stats": st,
            "frame_crc32": zlib.crc32(frame.tobytes()) & 0xFFFFFFFF, "tile_crc32": tile_crcs(frame)}


def main():
    cases = []
    d32, d64 = oracle.Scene.default(oracle.F32), oracle.Scene.default(oracle.F64)
    three = oracle.Scene.from_spheres(util.THREE_SPHERES, util.THREE_BOUND)
    cases.append(dict(case("config1_three_spheres_64x64", three, 64, 64, 1), scene="three_spheres", precision="f32"))
    cases.append(dict(case("default_64x128_spp2", d32, 64, 128, 2), scene="default8", precision="f32"))
    cases.append(dict(case("config2_800x600", d32, 800, 600, 1), scene="default8", precision="f32"))
    cases.append(dict(case("config3_1920x1080_f32", d32, 1920, 1080, 1), scene="default8", precision="f32"))
    cases.append(dict(case("config3_1920x1080_f64", d64, 1920, 1080, 1), scene="default8", precision="f64"))
    cases.append(dict(case("make_image_1024x768_spp4", d32, 1024, 768, 4), scene="default8", precision="f32"))
    cases.append(dict(case("level9_320x256_spp2", oracle.Scene.default(oracle.F32, 9), 320, 256, 2), scene="default9", precision="f32"))
    # BASELINE config 5 at full size (4096^2, pyramid level 9 = 87,381 spheres, spp 4): bench.py checks the frame its timed
    # launches leave behind against this CRC; the -m gpu tests compare spot buckets bit for bit
    cases.append(dict(case("config5_4096x4096_spp4_L9", oracle.Scene.default(oracle.F32, 9), 4096, 4096, 4), scene="default9", precision="f32"))
    tie = oracle.Scene.from_spheres(util.TIE_SPHERES, util.TIE_BOUND)
    cases.append(dict(case("tie_break_64x64", tie, 64, 64, 1), scene="tie", precision="f32"))
    inside = oracle.Scene.from_ranges(util.INSIDE_ITEMS, util.INSIDE_BOUNDS, util.INSIDE_RANGES)
    cases.append(dict(case("inside_bound_hierarchy_64x64", inside, 64, 64, 1), scene="inside", precision="f32"))
    cases.append(dict(case("inside_bound_flat_64x64", inside, 64, 64, 1, oracle.MODE_FLAT), scene="inside", precision="f32", flat=True))
    with open(os.path.join(HERE, "oracle_vectors.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_oracle_vectors.py (CPU oracle, pinned on the reference image first)",
                   "cases": cases}, f, indent=1)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
