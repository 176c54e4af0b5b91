#!/usr/bin/env python3
"""How much of a 1080p frame is its tail?  The same launches on ONE stream (frame k + 1 starts when frame k's last wave has ended:
what bench.py times) and dealt round-robin over 2 / 3 / 4 streams (frame k + 1's waves fill the SIMDs frame k's tail leaves idle).
usage: overlap_probe.py [w h spp level]   (GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RTRACE_HIP_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "c", "librtrace_hip_test.so"))      # rt_debug.h's controls: the hooks build
import numpy as np
import torch
import rust_tracer_amd as rta

w, h, spp, level = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (1920, 1080, 1, 8)
dev = rta.Scene.default(level).device(0)
opts = (w, h, spp)
regs = dev._regions([tuple(r) for r in rta.buckets(rta.RenderOptions(*opts))])
for n_streams in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    outs = [torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda") for _ in range(n_streams)]
    for k in range(n_streams * 4):                                  # warm-up: tables, contexts
        dev.render_frame_device(opts, regs, outs[k % n_streams].data_ptr(), streams[k % n_streams].cuda_stream)
    torch.cuda.synchronize()
    best = None
    for rep in range(7):
        frames = 240
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(torch.cuda.current_stream())
        for s in streams:
            s.wait_event(e0)
        for k in range(frames):
            dev.render_frame_device(opts, regs, outs[k % n_streams].data_ptr(), streams[k % n_streams].cuda_stream)
        for s in streams:
            ev = torch.cuda.Event()
            ev.record(s)
            torch.cuda.current_stream().wait_event(ev)
        e1.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / frames * 1e3
        best = us if best is None else min(best, us)
    same = all(bool(torch.equal(outs[0], o)) for o in outs[1:])
    print("%d stream(s): %.1f us per frame (best of 7 x %d frames), frames identical: %s" % (n_streams, best, frames, same), flush=True)
