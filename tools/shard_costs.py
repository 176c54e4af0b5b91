#!/usr/bin/env python3
"""What the scene's cost map predicts for the shards of a frame dealt `tile_id % N` over N GPUs (rt_debug_shard_costs): max / mean
shard cost -- the expectation the first multi-GPU hardware run can be checked against.  usage: shard_costs.py  (needs one GPU: the
cost map is a counting render)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RTRACE_HIP_LIBRARY", os.path.join(ROOT, "tests", "c", "librtrace_hip_test.so"))      # the controls of csrc/rt_debug.h live in the hooks build
import numpy as np
import rust_tracer_amd as rta


def main():
    out = {}
    for name, (w, h, spp, level) in {"1080p": (1920, 1080, 1, 8), "config5": (4096, 4096, 4, 9)}.items():
        s = rta.Scene.default(level)
        regs = [tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, spp))]
        rows = {}
        for n in (1, 2, 4, 8):
            c = rta.capi.shard_costs(s.device()._h, (w, h, spp), regs, n)
            rows[str(n)] = {"max_over_mean": round(float(c.max() / c.mean()), 4), "shard_costs_relative": [round(float(v / c.sum()), 5) for v in c]}
        out[name] = rows
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
