#!/usr/bin/env python3
"""Generates the committed golden fixtures.  Runs only in the build container (needs /root/reference).

1. make_image_1024x768_spp4_rgb.npz -- the decoded pixels of the reference's own output image
   /root/reference/src/img/rtrace-output.png (`make image`: --samples-per-pixel=4 --width=1024 --height=768,
   Makefile:7).  This is DATA the reference ships (expected output), the only pixel-level pin it has.
2. make_image_1024x768_spp4_tiles.json -- CRC32 of each 64x64 bucket's RGB bytes (row-major bucket order of
   render.rs:273-298), derived from (1), for cheap per-tile localisation of a mismatch.

Everything else under tests/golden/ (other resolutions, alpha, L9, f64, synthetic scenes) is produced by
make_oracle_vectors.py from the oracle AFTER the oracle has been pinned on (1) -- transitively pinned only.
"""
import json
import os
import zlib

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/src/img/rtrace-output.png"


def main():
    rgb = np.array(Image.open(SRC).convert("RGB"), dtype=np.uint8)
    assert rgb.shape == (768, 1024, 3), rgb.shape
    np.savez_compressed(os.path.join(HERE, "make_image_1024x768_spp4_rgb.npz"), rgb=rgb)
    crcs = []
    for ty in range(0, 768, 64):
        for tx in range(0, 1024, 64):
            crcs.append(zlib.crc32(np.ascontiguousarray(rgb[ty:ty + 64, tx:tx + 64]).tobytes()) & 0xFFFFFFFF)
    meta = {
        "source": "src/img/rtrace-output.png of Byron/rust-tracer (decoded with PIL)",
        "width": 1024, "height": 768, "samples_per_pixel": 4, "pyramid_level": 8,
        "tile": 64, "tiles_x": 16, "tiles_y": 12,
        "frame_crc32": zlib.crc32(rgb.tobytes()) & 0xFFFFFFFF,
        "tile_crc32": crcs,
        "spot": {"(0,0)": rgb[0, 0].tolist(), "(x=512,y=767)": rgb[767, 512].tolist(),
                 "max": rgb.reshape(-1, 3).max(axis=0).tolist()},
    }
    with open(os.path.join(HERE, "make_image_1024x768_spp4_tiles.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", meta["frame_crc32"], len(crcs), "tiles")


if __name__ == "__main__":
    main()
