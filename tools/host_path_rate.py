#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point rt_render_tiles: the same 1080p frame, but
the RGBA bytes are copied back to (pageable) host memory inside the call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_tracer_amd as rta

s = rta.Scene.default()
d = s.device()
o = (1920, 1080, 1)
regs = [tuple(r) for r in rta.buckets(rta.RenderOptions(*o))]
_, st = d.render_tiles(o, regs, rta.RT_TRAVERSAL_SKIP)
rays = st["primary"] + st["shadow"]
for _ in range(5):
    d.render_tiles(o, regs, rta.RT_TRAVERSAL_SKIP, want_stats=False)
n = 50
t0 = time.perf_counter()
for _ in range(n):
    d.render_tiles(o, regs, rta.RT_TRAVERSAL_SKIP, want_stats=False)
dt = (time.perf_counter() - t0) / n
print("rt_render_tiles (host output, 8.29 MB D2H per frame): %.3f ms/frame, %.1f Mrays/s" % (dt * 1e3, rays / dt / 1e6))
