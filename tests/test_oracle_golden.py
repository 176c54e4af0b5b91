"""Pins the f32 oracle on the only pixel-level vector the reference ships: its own output image
src/img/rtrace-output.png (`make image` settings), committed as decoded pixels under tests/golden/."""
import json
import os
import zlib

import numpy as np

import oracle


def test_make_image_golden_zero_differing_pixels(golden_dir):
    ref = np.load(os.path.join(golden_dir, "make_image_1024x768_spp4_rgb.npz"))["rgb"]
    meta = json.load(open(os.path.join(golden_dir, "make_image_1024x768_spp4_tiles.json")))
    s = oracle.Scene.default()
    img, st, n = s.render(1024, 768, 4, nthreads=os.cpu_count() or 1)
    assert n == 192
    assert int((img[:, :, :3] != ref).any(axis=2).sum()) == 0
    assert zlib.crc32(np.ascontiguousarray(img[:, :, :3]).tobytes()) & 0xFFFFFFFF == meta["frame_crc32"]
    # SURVEY.md 8(d) parity-anchor row
    assert (st["primary"], st["hits"], st["shadow"], st["occluded"]) == (12582912, 9430527, 7211901, 3586443)


def test_golden_spot_values(golden_dir):
    # SURVEY.md Appendix A
    ref = np.load(os.path.join(golden_dir, "make_image_1024x768_spp4_rgb.npz"))["rgb"]
    assert ref[0, 0].tolist() == [34, 10, 10]
    assert (ref[:57] == np.array([34, 10, 10], dtype=np.uint8)).all()
    assert ref.reshape(-1, 3).max(axis=0).tolist() == [201, 57, 57]
    assert ref[767, 512].tolist() == [102, 29, 29]
