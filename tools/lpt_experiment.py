#!/usr/bin/env python3
"""Experiment: how much of the frame time is dispatch order?  Renders the 1080p frame as 16x16 tiles (= one workgroup each)
in (a) raster order, (b) descending cost order (cost = the tile's own test count), (c) ascending, and prints us per frame.
usage: lpt_experiment.py [w h]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RTRACE_HIP_LIBRARY", os.path.join(ROOT, "tests", "c", "librtrace_hip_test.so"))      # the controls of csrc/rt_debug.h live in the hooks build
import numpy as np
import torch
import rust_tracer_amd as rta


def timeit(dev, opts, regs_c, out, stream, n=30):
    ts = []
    for r in range(n + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dev.render_frame_device(opts, regs_c, out.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        if r >= 3:
            ts.append(e0.elapsed_time(e1) / 5 * 1e3)
    return float(np.median(ts))


def main():
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
    T = int(os.environ.get("LPT_TILE", "16"))
    scene = rta.Scene.default(8)
    dev = scene.device(0)
    opts = (w, h, 1)
    out = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    big = [tuple(r) for r in rta.buckets(rta.RenderOptions(*opts))]
    dev.render_frame_device(opts, dev._regions(big), out.data_ptr(), stream)
    torch.cuda.synchronize()
    ref = out.cpu().numpy().copy()
    print("64x64 buckets, raster order: %.1f us" % timeit(dev, opts, dev._regions(big), out, stream))
    small = [(x, min(y + T, h), min(x + T, w), y) for y in range(0, h, T) for x in range(0, w, T)]
    cost = []
    tmp = torch.zeros(T * T * 4, dtype=torch.uint8, device="cuda")
    for reg in small:
        st = dev.render_tiles_device(opts, [reg], tmp.data_ptr(), stream, rta.RT_TRAVERSAL_SKIP, want_stats=True)
        cost.append(st["sphere_tests"] + st["bound_tests"])
    cost = np.array(cost)
    print("%d tiles of %dx%d, cost min %d median %d max %d" % (len(small), T, T, cost.min(), np.median(cost), cost.max()))
    for name, order in (("raster", np.arange(len(small))), ("descending cost", np.argsort(-cost, kind="stable")),
                        ("ascending cost", np.argsort(cost, kind="stable"))):
        regs_c = dev._regions([small[i] for i in order])
        out.zero_()
        dev.render_frame_device(opts, regs_c, out.data_ptr(), stream)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), ref)
        print("%dx%d tiles, %s: %.1f us" % (T, T, name, timeit(dev, opts, regs_c, out, stream)))


if __name__ == "__main__":
    main()
