#!/usr/bin/env python3
"""Diagnostic: filtered loops (variant 19 / 23) against the counting launch on the default scene."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RTRACE_HIP_LIBRARY", os.path.join(ROOT, "tests", "c", "librtrace_hip_test.so"))      # the controls of csrc/rt_debug.h live in the hooks build
import numpy as np
import rust_tracer_amd as rta

w, h, spp, level = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (256, 192, 1, 6)
scene = rta.Scene.default(level)
dev = scene.device(0)
regs = [tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, spp))]
counted, st = dev.render_tiles((w, h, spp), regs, rta.RT_TRAVERSAL_SKIP, want_stats=True)
print("stats", {k: st[k] for k in ("primary", "hits", "shadow", "occluded")})
print("filter unsure/pass", rta.capi.debug_count(2), "violations", rta.capi.debug_count(3))
for v in (3, 7, 19, 23):
    with rta.capi.debug(rta.capi.DEBUG_SKIP_VARIANT, v):
        plain, _ = dev.render_tiles((w, h, spp), regs, rta.RT_TRAVERSAL_SKIP, want_stats=False)
    d = (plain.reshape(-1, 4) != counted.reshape(-1, 4)).any(axis=1)
    print("variant", v, "differing pixels", int(d.sum()), "of", d.size)
    if d.any():
        idx = np.nonzero(d)[0][:5]
        for i in idx:
            print("   px", i, plain.reshape(-1, 4)[i], counted.reshape(-1, 4)[i])
