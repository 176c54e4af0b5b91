#!/usr/bin/env python3
"""What a rank's shard of an N-GPU frame takes to render, measured on ONE GPU: the tile list rank 0 of an N-GPU `tiles` run gets (buckets
i % N == 0), the library left to its own choices (dispatch orders tried and settled during the warm-up) -> profiles/expected_shard_render.json,
which bench.py quotes beside the first hardware curve (`expected_shard_render_us`).  usage: shard_expect.py   (GPU box)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RTRACE_HIP_LIBRARY", os.path.join(ROOT, "tests", "c", "librtrace_hip_test.so"))      # the controls of csrc/rt_debug.h live in the hooks build
import numpy as np
import torch
import rust_tracer_amd as rta
from bench import git_head, kernel_src_sha


def shard_us(w, h, spp, level, n, launches, rounds=6):
    scene = rta.Scene.default(level)
    dev = scene.device(0)
    regs = [tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, spp))][::n]
    regs_c = dev._regions(regs)
    out = torch.zeros(sum((r - l) * (t - b) for (l, t, r, b) in regs) * 4, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    times = []
    for r in range(rounds + 12):                # (the library tries its dispatch orders over a list's first ~150 launches: measure after that)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            dev.render_tiles_device((w, h, spp), regs_c, out.data_ptr(), stream, rta.RT_TRAVERSAL_SKIP)
        e1.record()
        torch.cuda.synchronize()
        if r >= 12:
            times.append(e0.elapsed_time(e1) / launches * 1e3)
    return round(float(np.median(times)), 1)


def main():
    res = {"source": "tools/shard_expect.py: rank 0's tile list (buckets i % N == 0) rendered on one MI355X, median us per launch",
           "git_head": git_head(), "kernel_src_sha": kernel_src_sha(), "1080p": {}, "config5": {}}
    for n in (1, 2, 4, 8, 16):
        res["1080p"][str(n)] = shard_us(1920, 1080, 1, 8, n, 20)
        res["config5"][str(n)] = shard_us(4096, 4096, 4, 9, n, 3, rounds=3)
        print("N = %2d: 1080p %.1f us   config 5 %.1f us" % (n, res["1080p"][str(n)], res["config5"][str(n)]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for d in ("profiles", "gpurun_out"):           # (gpurun only carries gpurun_out/ back: copy it into profiles/ afterwards)
        json.dump(res, open(os.path.join(ROOT, d, "expected_shard_render.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
