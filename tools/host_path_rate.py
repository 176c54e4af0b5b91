#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer entry points, 1080p default scene:

  rt_render_tiles  whole frame (510 buckets, one call) into pageable memory / into rt_host_alloc'd memory, every copy strategy
                   (csrc/rt_debug.h RT_DEBUG_HOST_COPY: 1 one D2H to the caller's pointer, 2 pinned staging + CPU copy, 3 the
                   kernel stores into the pinned host buffer)
  rt_render_region 510 calls per frame (the literal render.rs:283-294 shape) from 1 thread and from T threads, with and
                   without merging concurrent callers into shared passes

usage: host_path_rate.py [frames] [threads]      prints one JSON object"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RTRACE_HIP_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "c", "librtrace_hip_test.so"))      # rt_debug.h's controls: the hooks build
import numpy as np
import rust_tracer_amd as rta
from rust_tracer_amd import capi


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else min(64, os.cpu_count() or 1)
    s = rta.Scene.default()
    d = s.device()
    o = (1920, 1080, 1)
    regs = [tuple(r) for r in rta.buckets(rta.RenderOptions(*o))]
    arr = d._regions(regs)
    ref, st = d.render_tiles(o, arr, rta.RT_TRAVERSAL_SKIP)
    ref = ref.copy()
    rays = st["primary"] + st["shadow"]
    nbytes = ref.size
    out = {"workload": "1920x1080 L8 spp 1, skip traversal", "rays": rays, "bytes_per_frame": int(nbytes), "frames_timed": frames}

    def timed(fn, n=frames, warm=3):
        for _ in range(warm):
            fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        return (time.perf_counter() - t0) / n * 1e3

    pageable = np.empty(nbytes, dtype=np.uint8)
    pinned = capi.HostBuffer(int(nbytes))
    tiles = {}
    for name, buf, modes in (("pageable", pageable, (0, 1, 2)), ("pinned", pinned.array, (0, 1, 3))):
        for mode in modes:
            capi.debug_set(capi.DEBUG_HOST_COPY, mode if mode else -1)
            buf[:] = 0
            ms = timed(lambda: d.render_tiles(o, arr, rta.RT_TRAVERSAL_SKIP, want_stats=False, out=buf))
            assert np.array_equal(buf, ref), (name, mode)
            tiles["%s_mode%d" % (name, mode)] = {"ms_per_frame": round(ms, 4), "Mrays_per_s": round(rays / ms / 1e3, 1)}
    capi.debug_set(capi.DEBUG_HOST_COPY, -1)
    out["rt_render_tiles"] = tiles

    # rt_render_region: every bucket its own call, each into its own RGBABuffer-sized slice
    offs = np.cumsum([0] + [(r - l) * (t - b) * 4 for (l, t, r, b) in regs])
    region = {}
    for coalesce in (1, 0):
        capi.debug_set(capi.DEBUG_COALESCE, coalesce)
        for nt in (1, threads):
            frame = np.zeros(nbytes, dtype=np.uint8)

            def one_frame():
                if nt == 1:
                    for i, reg in enumerate(regs):
                        d.render_region(o, reg, rta.RT_TRAVERSAL_SKIP, out=frame[offs[i]:offs[i + 1]])
                    return
                nxt = [0]
                lock = threading.Lock()

                def work():
                    while True:
                        with lock:
                            i = nxt[0]
                            nxt[0] += 1
                        if i >= len(regs):
                            return
                        d.render_region(o, regs[i], rta.RT_TRAVERSAL_SKIP, out=frame[offs[i]:offs[i + 1]])
                th = [threading.Thread(target=work) for _ in range(nt)]
                [t.start() for t in th]
                [t.join() for t in th]

            ms = timed(one_frame, n=max(3, frames // 6), warm=1)
            assert np.array_equal(frame, ref), (coalesce, nt)
            region["coalesce%d_threads%d" % (coalesce, nt)] = {"ms_per_frame": round(ms, 3), "Mrays_per_s": round(rays / ms / 1e3, 1)}
    capi.debug_set(capi.DEBUG_COALESCE, -1)
    out["rt_render_region_510_calls"] = region
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
