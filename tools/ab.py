#!/usr/bin/env python3
"""A/B timing of kernel tuning variants, interleaved in ONE process (cdna guide rule 24).  Checks every variant's frame
and counters against the first variant.  usage: ab.py [rounds] [w h spp level]
env: AB_TRAVERSAL=skip|flat  AB_KEY=skip_variant|block_order|narrow_max|packed_samples|host_copy|flat_kernels|skip_rays (rt_debug.h)  AB_VARIANTS=1,3,7 (skip: 1 C++ loops, 3 generated assembly
loops, 7 their fused flavour)  AB_LAUNCHES=5"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RTRACE_HIP_LIBRARY", os.path.join(ROOT, "tests", "c", "librtrace_hip_test.so"))      # the controls of csrc/rt_debug.h live in the hooks build
import numpy as np
import torch
import rust_tracer_amd as rta


TRAV = rta.RT_TRAVERSAL_FLAT if os.environ.get("AB_TRAVERSAL", "skip") == "flat" else rta.RT_TRAVERSAL_SKIP
KEY = getattr(rta.capi, "DEBUG_" + os.environ.get("AB_KEY", "skip_variant").upper())
LAUNCHES = int(os.environ.get("AB_LAUNCHES", "5"))


def setvar(v):
    rta.capi.debug_set(KEY, v)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    w, h, spp, level = (int(a) for a in sys.argv[2:6]) if len(sys.argv) > 5 else (1920, 1080, 1, 8)
    variants = [int(v) for v in os.environ.get("AB_VARIANTS", "1,3,7").split(",")]
    scene = rta.Scene.default(level, rta.RT_F64 if os.environ.get("AB_PRECISION", "f32") == "f64" else rta.RT_F32)
    dev = scene.device(0)
    opts = (w, h, spp)
    regs = [tuple(r) for r in rta.buckets(rta.RenderOptions(*opts))]
    regs_c = dev._regions(regs)
    out = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ref = None
    for v in variants:
        setvar(v)
        st = dev.render_tiles_device(opts, regs_c, out.data_ptr(), stream, TRAV, want_stats=True)
        torch.cuda.synchronize()
        frame = out.cpu().numpy().copy()
        out.zero_()
        dev.render_tiles_device(opts, regs_c, out.data_ptr(), stream, TRAV)          # the launch the timing uses (no counters)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), frame), "variant %d: the launch without counters renders different pixels" % v
        key = (st["primary"], st["hits"], st["shadow"], st["occluded"], st["sphere_tests"], st["bound_tests"])
        if ref is None:
            ref = (frame, key)
        else:
            assert np.array_equal(frame, ref[0]), "variant %d changes pixels" % v
            assert key == ref[1], "variant %d changes counters %r vs %r" % (v, key, ref[1])
    times = {v: [] for v in variants}
    for r in range(rounds + 2):
        for v in variants:
            setvar(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(LAUNCHES):
                dev.render_tiles_device(opts, regs_c, out.data_ptr(), stream, TRAV)
            e1.record()
            torch.cuda.synchronize()
            if r >= 2:
                times[v].append(e0.elapsed_time(e1) / LAUNCHES * 1e3)
    print("%dx%d spp %d L%d, %d rounds x %d launches, us per launch" % (w, h, spp, level, rounds, LAUNCHES))
    for v in variants:
        t = np.array(times[v])
        print("variant %d: median %.1f  min %.1f  max %.1f" % (v, np.median(t), t.min(), t.max()))


if __name__ == "__main__":
    main()
