#!/usr/bin/env python3
"""Diagnostic: render-kernel time against waves per SIMD, capped with dynamic LDS (csrc/rt_debug.h RT_DEBUG_LDS_BYTES); one process,
interleaved, frames compared.   usage: occupancy_sweep.py [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RTRACE_HIP_LIBRARY", os.path.join(ROOT, "tests", "c", "librtrace_hip_test.so"))      # the controls of csrc/rt_debug.h live in the hooks build
import numpy as np
import torch
import rust_tracer_amd as rta
from rust_tracer_amd import capi

WORK = {"config2": (800, 600, 1, 8, 20), "vga": (640, 480, 1, 8, 20), "1080p": (1920, 1080, 1, 8, 5), "make_image": (1024, 768, 4, 8, 3), "config5": (4096, 4096, 4, 9, 1), "4k": (3840, 2160, 1, 8, 3)}


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    names = sys.argv[2].split(",") if len(sys.argv) > 2 else list(WORK)
    stream = torch.cuda.current_stream().cuda_stream
    for name in names:
        w, h, spp, level, launches = WORK[name]
        dev = rta.Scene.default(level).device(0)
        regs = dev._regions([tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, spp))])
        out = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
        ref = None
        times = {}
        for r in range(rounds + 1):
            for waves in (8, 7, 6, 5, 4, 3):
                capi.debug_set(capi.DEBUG_LDS_BYTES, 0 if waves == 8 else (160 * 1024 // waves) & ~255)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(launches):
                    dev.render_frame_device((w, h, spp), regs, out.data_ptr(), stream)
                e1.record()
                torch.cuda.synchronize()
                if r == 0:
                    f = out.cpu().numpy().copy()
                    if ref is None:
                        ref = f
                    assert np.array_equal(f, ref)
                else:
                    times.setdefault(waves, []).append(e0.elapsed_time(e1) / launches * 1e3)
        print(name, " ".join("w%d: %.1f us" % (k, float(np.median(v))) for k, v in sorted(times.items(), reverse=True)), flush=True)
        capi.debug_set(capi.DEBUG_LDS_BYTES, -1)


if __name__ == "__main__":
    main()
