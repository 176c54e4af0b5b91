#!/usr/bin/env python3
"""The lane-cooperative walk (csrc/rt_coop.hpp) and the narrow-workgroup tier against the plain dispatch, over their controls (rt_debug.h):
rays per cooperative wave (RT_DEBUG_COOP_LEVEL: 1 = 16, 2 = 4, 3 = 1), the cost-map value from which a quad goes cooperative
(RT_DEBUG_COOP_THR), how many blocks are narrowed (RT_DEBUG_NARROW_MAX) and how many of those to 2x2-pixel waves (RT_DEBUG_NARROW_L2).
Variants are interleaved in one process and their frames compared byte for byte.
usage: coop_sweep.py w h level "<variant>;<variant>;..." [shard_of_N]     variant: comma-separated key=value, keys coop, thr, lvl, nmax, nl2, cmax, rest
   e.g. coop_sweep.py 800 600 8 "coop=0;thr=200;thr=200,nmax=64,nl2=0"        (GPU box; shard_of_N: only the buckets i % N == 0)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RTRACE_HIP_LIBRARY", os.path.join(ROOT, "tests", "c", "librtrace_hip_test.so"))      # the controls of csrc/rt_debug.h live in the hooks build
import numpy as np
import torch
import rust_tracer_amd as rta
from rust_tracer_amd import capi

KEYS = {"coop": capi.DEBUG_COOP, "thr": capi.DEBUG_COOP_THR, "lvl": capi.DEBUG_COOP_LEVEL, "nmax": capi.DEBUG_NARROW_MAX, "nl2": capi.DEBUG_NARROW_L2,
        "cmax": capi.DEBUG_COOP_MAX, "rest": capi.DEBUG_COOP_REST}


def main():
    w, h, level = (int(a) for a in sys.argv[1:4])
    specs = [v for v in sys.argv[4].split(";") if v]
    shard = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    launches, rounds = 20, 8            # (the two warm-up rounds also let the library finish timing its own candidates: rt_capi.hip pick_order)
    scene = rta.Scene.default(level)
    dev = scene.device(0)
    opts = (w, h, 1)
    regs = [tuple(r) for r in rta.buckets(rta.RenderOptions(*opts))][::shard]
    regs_c = dev._regions(regs)
    out = torch.zeros(sum((r - l) * (t - b) for (l, t, r, b) in regs) * 4, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    variants = [(sp, {} if sp == "auto" else {KEYS[k]: int(v) for k, v in (kv.split("=") for kv in sp.split(","))}) for sp in specs]       # auto: the library's choice

    def apply(knobs):
        for k in KEYS.values():
            capi.debug_set(k, knobs.get(k, -1))

    ref = None
    for name, knobs in variants:
        apply(knobs)
        out.zero_()
        dev.render_tiles_device(opts, regs_c, out.data_ptr(), stream, rta.RT_TRAVERSAL_SKIP)
        torch.cuda.synchronize()
        frame = out.cpu().numpy().copy()
        if ref is None:
            ref = frame
        assert np.array_equal(frame, ref), "%s changes pixels" % name
    times = {name: [] for name, _ in variants}
    for r in range(rounds + 2):
        for name, knobs in variants:
            apply(knobs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(launches):
                dev.render_tiles_device(opts, regs_c, out.data_ptr(), stream, rta.RT_TRAVERSAL_SKIP)
            e1.record()
            torch.cuda.synchronize()
            if r >= 2:
                times[name].append(e0.elapsed_time(e1) / launches * 1e3)
    apply({})
    print("%dx%d L%d, %d buckets, %d rounds x %d launches, us per launch (median / min)" % (w, h, level, len(regs), rounds, launches))
    for name, _ in variants:
        t = np.array(times[name])
        print("  %-34s %7.1f %7.1f" % (name, np.median(t), t.min()))


if __name__ == "__main__":
    main()
