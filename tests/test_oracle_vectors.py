"""CPU: the committed oracle vectors are reproducible from the oracle (guards the fixture against drift) and
carry the ray statistics SURVEY.md 8(d) lists."""
import json
import os
import zlib

import numpy as np
import pytest

import oracle
from tests import util

CASES = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.json")))["cases"]


def by_name(n):
    return next(c for c in CASES if c["name"] == n)


def test_survey_ray_statistics_are_in_the_vectors():
    s = by_name("config2_800x600")["stats"]
    assert (s["primary"], s["hits"], s["shadow"], s["occluded"]) == (480000, 359528, 275032, 136797)
    s = by_name("config3_1920x1080_f32")["stats"]
    assert (s["primary"], s["hits"], s["shadow"], s["occluded"]) == (2073600, 1777280, 1337403, 730313)
    s = by_name("make_image_1024x768_spp4")["stats"]
    assert (s["primary"], s["hits"], s["shadow"], s["occluded"]) == (12582912, 9430527, 7211901, 3586443)
    assert by_name("config3_1920x1080_f32")["buckets"] == 510 and by_name("config2_800x600")["buckets"] == 130


@pytest.mark.parametrize("name", ["config1_three_spheres_64x64", "default_64x128_spp2", "config2_800x600", "tie_break_64x64",
                                  "inside_bound_hierarchy_64x64", "inside_bound_flat_64x64"])
def test_vectors_reproduce_from_the_oracle(name):
    c = by_name(name)
    scene = {"three_spheres": lambda: oracle.Scene.from_spheres(util.THREE_SPHERES, util.THREE_BOUND),
             "default8": lambda: oracle.Scene.default(),
             "tie": lambda: oracle.Scene.from_spheres(util.TIE_SPHERES, util.TIE_BOUND),
             "inside": lambda: oracle.Scene.from_ranges(util.INSIDE_ITEMS, util.INSIDE_BOUNDS, util.INSIDE_RANGES)}[c["scene"]]()
    mode = oracle.MODE_FLAT if c.get("flat") else oracle.MODE_HIERARCHY | oracle.MODE_ANYHIT_EXIT
    frame, st, n = scene.render(c["width"], c["height"], c["spp"], nthreads=os.cpu_count() or 1, mode=mode)
    assert zlib.crc32(frame.tobytes()) & 0xFFFFFFFF == c["frame_crc32"]
    assert st == c["stats"] and n == c["buckets"]


def test_f64_differs_from_f32_it_is_a_separate_golden():
    # the type-alias swap changes pixels; f64 has its own vector (parity unpinned by the reference)
    assert by_name("config3_1920x1080_f64")["frame_crc32"] != by_name("config3_1920x1080_f32")["frame_crc32"]
