#!/usr/bin/env python3
"""Kernel time of one workload in f32 and f64 (BASELINE config 3: the RFloat alias swap), us per launch, device output.
usage: f64_time.py [w h spp level]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RTRACE_HIP_LIBRARY", os.path.join(ROOT, "tests", "c", "librtrace_hip_test.so"))      # the controls of csrc/rt_debug.h live in the hooks build
import numpy as np
import torch
import rust_tracer_amd as rta


def main():
    w, h, spp, level = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (1920, 1080, 1, 8)
    opts = (w, h, spp)
    out = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for name, prec in (("f32", rta.RT_F32), ("f64", rta.RT_F64)):
        for tname, trav in (("skip", rta.RT_TRAVERSAL_SKIP), ("flat", rta.RT_TRAVERSAL_FLAT)):
            dev = rta.Scene.default(level, prec).device(0)
            regs_c = dev._regions([tuple(r) for r in rta.buckets(rta.RenderOptions(*opts))])
            n = 20 if tname == "skip" else 3
            for _ in range(300 if tname == "skip" else 2):       # (a new list's dispatch orders arrive from the background and are tried: let both settle)
                dev.render_frame_device(opts, regs_c, out.data_ptr(), stream, trav)
                torch.cuda.synchronize()
            ts = []
            for r in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    dev.render_frame_device(opts, regs_c, out.data_ptr(), stream, trav)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / n * 1e3)
            print("%dx%d spp %d L%d %s %s: %.1f us" % (w, h, spp, level, name, tname, min(ts[1:])))


if __name__ == "__main__":
    main()
