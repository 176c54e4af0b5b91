#!/usr/bin/env python3
"""Kernel time of one workload in f32 and f64 (BASELINE config 3: the RFloat alias swap), us per launch, device output.
usage: f64_time.py [w h spp level]"""
import os
import sys
import time

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
            # (a new list's dispatch orders arrive from the background thread and are then tried against each other: wait for them, then
            # let the trial settle -- 300 launches with a synchronisation each are over before the worker has made the orders)
            t_end = time.time() + (3.0 if tname == "skip" else 0.0)
            k = 0
            while k < 2 or (time.time() < t_end and not (k > 600 and "ordered" in rta.capi.last_launch())):
                dev.render_frame_device(opts, regs_c, out.data_ptr(), stream, trav)
                k += 1
                if k % 50 == 0:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            flags = rta.capi.last_launch()
            ts = []
            for r in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    dev.render_frame_device(opts, regs_c, out.data_ptr(), stream, trav)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / n * 1e3)
            print("%dx%d spp %d L%d %s %s: %.1f us   [%s]" % (w, h, spp, level, name, tname, min(ts[1:]), ",".join(sorted(flags))))


if __name__ == "__main__":
    main()
