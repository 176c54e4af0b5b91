#!/usr/bin/env python3
"""The narrow-workgroup cap (rt_capi.hip block_order) on the tile lists a rank of an N-GPU run renders: buckets i % N == 0 of the frame.
usage: shard_narrow_sweep.py [w h spp level]   (GPU box; a fresh device scene per value: the dispatch table is built when a list is first seen)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RTRACE_HIP_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "c", "librtrace_hip_test.so"))      # rt_debug.h's controls: the hooks build
import numpy as np
import torch
import rust_tracer_amd as rta
from rust_tracer_amd import capi

w, h, spp, level = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (1920, 1080, 1, 8)
stream = torch.cuda.current_stream().cuda_stream
allb = [tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, spp))]
values = [int(v) for v in os.environ.get("SHARD_SWEEP_VALUES", "-1,0,8,16,32,64,128").split(",")]
for n in [int(v) for v in os.environ.get("SHARD_SWEEP_N", "1,2,4,8,16").split(",")]:
    mine = allb[0::n]
    px = sum((r[2] - r[0]) * (r[1] - r[3]) for r in mine)
    out = torch.zeros(px * 4, dtype=torch.uint8, device="cuda")
    devs, regs, ref = {}, {}, None
    for v in values:
        capi.debug_set(capi.DEBUG_NARROW_MAX, v)
        devs[v] = rta.Scene.default(level).device(0)
        regs[v] = devs[v]._regions(mine)
        out.zero_()
        devs[v].render_tiles_device((w, h, spp), regs[v], out.data_ptr(), stream)
        torch.cuda.synchronize()
        f = out.cpu().numpy().copy()
        ref = f if ref is None else ref
        assert np.array_equal(f, ref)
    times = {v: [] for v in values}
    for r in range(10):
        for v in values:
            capi.debug_set(capi.DEBUG_NARROW_MAX, v)       # (the controls are part of a dispatch table's identity: a launch finds ITS table only with the control set)
            if r == 0:
                for _ in range(120):                         # the trial of the list's candidate orders
                    devs[v].render_tiles_device((w, h, spp), regs[v], out.data_ptr(), stream)
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                devs[v].render_tiles_device((w, h, spp), regs[v], out.data_ptr(), stream)
            e1.record()
            torch.cuda.synchronize()
            if r >= 2:
                times[v].append(e0.elapsed_time(e1) / 10 * 1e3)
    capi.debug_set(capi.DEBUG_NARROW_MAX, -1)
    print("N = %2d (%3d buckets):" % (n, len(mine)), "  ".join("%s: %.1f" % ("default" if v < 0 else v, float(np.median(t))) for v, t in times.items()), "us", flush=True)
    for d in devs.values():
        d.close()
