#!/usr/bin/env python3
"""Diagnostic: node-step statistics of one counted launch (rt_debug.h, RT_DEBUG_PRINT_STEPS): wave-level node visits, how many of them at ITEM
nodes, the busiest wave.   usage: debug_steps.py [w h spp level]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RTRACE_HIP_LIBRARY", os.path.join(ROOT, "tests", "c", "librtrace_hip_test.so"))      # the controls of csrc/rt_debug.h live in the hooks build
import torch
import rust_tracer_amd as rta

rta.capi.debug_set(rta.capi.DEBUG_PRINT_STEPS, 1)
w, h, spp, level = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (1920, 1080, 1, 8)
d = rta.Scene.default(level).device(0)
regs = d._regions([tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, spp))])
out = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
st = d.render_frame_device((w, h, spp), regs, out.data_ptr(), torch.cuda.current_stream().cuda_stream, rta.RT_TRAVERSAL_SKIP, want_stats=True)
print(st)
