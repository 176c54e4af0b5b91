#!/usr/bin/env python3
"""Diagnostic: render-kernel time against one control of csrc/rt_debug.h that is read when a tile list is first seen (a fresh
device scene per value), frames compared.   usage: knob_sweep.py KEY v1,v2,... [w h spp level launches]
e.g.  knob_sweep.py narrow_max 0,32,64,128      knob_sweep.py wg_policy 0,1,2,4 4096 4096 4 9 1"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RTRACE_HIP_LIBRARY", os.path.join(ROOT, "tests", "c", "librtrace_hip_test.so"))      # the controls of csrc/rt_debug.h live in the hooks build
import numpy as np
import torch
import rust_tracer_amd as rta
from rust_tracer_amd import capi

key = getattr(capi, "DEBUG_" + sys.argv[1].upper())
values = [int(v) for v in sys.argv[2].split(",")]
w, h, spp, level, launches = (int(a) for a in sys.argv[3:8]) if len(sys.argv) > 7 else (1920, 1080, 1, 8, 5)
stream = torch.cuda.current_stream().cuda_stream
ref = None
out = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
devs = {}
for v in values:
    capi.debug_set(key, v)
    devs[v] = rta.Scene.default(level).device(0)
    regs = devs[v]._regions([tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, spp))])
    out.zero_()
    devs[v].render_frame_device((w, h, spp), regs, out.data_ptr(), stream)      # builds this value's dispatch table
    torch.cuda.synchronize()
    f = out.cpu().numpy().copy()
    ref = f if ref is None else ref
    assert np.array_equal(f, ref), "value %d changes pixels" % v
times = {v: [] for v in values}
for r in range(10):
    for v in values:                                    # interleaved: one process, same clocks
        capi.debug_set(key, v)                          # (the controls are part of a dispatch table's identity: a launch finds ITS table only with the control set)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            devs[v].render_frame_device((w, h, spp), regs, out.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            times[v].append(e0.elapsed_time(e1) / launches * 1e3)
capi.debug_set(key, -1)
print("%s on %dx%d spp %d L%d:" % (sys.argv[1], w, h, spp, level), "  ".join("%d: %.1f us" % (v, float(np.median(t))) for v, t in times.items()))
