#!/usr/bin/env python3
"""What a rank's shard of an N-GPU frame takes to render, measured on ONE GPU: the tile list rank 0 of an N-GPU `tiles` run gets (buckets
i % N == 0), the library left to its own choices (dispatch orders tried and settled during the warm-up) -> profiles/expected_shard_render.json,
which bench.py quotes beside the first hardware curve (`expected_shard_render_us`).  usage: shard_expect.py   (GPU box)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RTRACE_HIP_LIBRARY", os.path.join(ROOT, "tests", "c", "librtrace_hip_test.so"))      # the controls of csrc/rt_debug.h live in the hooks build
import numpy as np
import torch
import rust_tracer_amd as rta
from bench import git_head, kernel_src_sha


def shard_us(w, h, spp, level, n, launches, rounds=6):
    scene = rta.Scene.default(level)
    dev = scene.device(0)
    regs = [tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, spp))][::n]
    regs_c = dev._regions(regs)
    out = torch.zeros(sum((r - l) * (t - b) for (l, t, r, b) in regs) * 4, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    times = []
    for r in range(rounds + 12):                # (the library tries its dispatch orders over a list's first ~150 launches: measure after that)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            dev.render_tiles_device((w, h, spp), regs_c, out.data_ptr(), stream, rta.RT_TRAVERSAL_SKIP)
        e1.record()
        torch.cuda.synchronize()
        if r >= 12:
            times.append(e0.elapsed_time(e1) / launches * 1e3)
    return round(float(np.median(times)), 1)


def blit_us(w, h, level=8, launches=20, rounds=6):
    """k_blit_tiles over the whole bucket list (what the root runs on the gathered shards): median us per launch."""
    dev = rta.Scene.default(level).device(0)
    regs = dev._regions([tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, 1))])       # (the C array once: building it per call costs the host 0.2 ms)
    src = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    dst = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    times = []
    for r in range(rounds + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            dev.blit_tiles_device((w, h, 1), regs, src.data_ptr(), dst.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            times.append(e0.elapsed_time(e1) / launches * 1e3)
    return round(float(np.median(times)), 1)


def collective_us(nbytes, calls=50, rounds=6):
    """One RCCL gather call of `nbytes` per rank in a ONE-rank group on this box (torch.distributed, backend nccl): stream time per call and
    what the call costs the host thread.  No wire: the floor of what a frame pays for its collective at any N."""
    import time
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    src = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    dst = [torch.zeros(nbytes, dtype=torch.uint8, device="cuda")]
    dev_t, host_t = [], []
    for r in range(rounds + 2):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(calls):
            dist.gather(src, dst, dst=0)
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        if r >= 2:
            dev_t.append(e0.elapsed_time(e1) / calls * 1e3)
            host_t.append((t1 - t0) / calls * 1e6)
    dist.destroy_process_group()
    return round(float(np.median(dev_t)), 1), round(float(np.median(host_t)), 1)


def main():
    res = {"source": "tools/shard_expect.py: rank 0's tile list (buckets i % N == 0) rendered on one MI355X, median us per launch",
           "git_head": git_head(), "kernel_src_sha": kernel_src_sha(), "1080p": {}, "config5": {}}
    for n in (1, 2, 4, 8, 16):
        res["1080p"][str(n)] = shard_us(1920, 1080, 1, 8, n, 20)
        res["config5"][str(n)] = shard_us(4096, 4096, 4, 9, n, 3, rounds=3)
        print("N = %2d: 1080p %.1f us   config 5 %.1f us" % (n, res["1080p"][str(n)], res["config5"][str(n)]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for d in ("profiles", "gpurun_out"):           # (gpurun only carries gpurun_out/ back: copy it into profiles/ afterwards)
        json.dump(res, open(os.path.join(ROOT, d, "expected_shard_render.json"), "w"), indent=1)
    # What ONE frame of BASELINE config 4 will take at N = 2 / 4 / 8 before any curve exists: rank 0's shard render (above)
    # + one collective call (a one-rank RCCL group here: no wire, the floor) + the root's blit, next to the N = 1 frame, which is rendered
    # straight into the row-major frame and has neither.
    blit = blit_us(1920, 1080)
    lat = {"source": "tools/shard_expect.py on one MI355X: shard render = rank 0's tile list (buckets i % N == 0), median us per launch; collective = one torch.distributed "
                     "gather call of the padded shard in a ONE-rank RCCL group (stream time; no wire: a floor -- over xGMI the root also receives (N - 1) shards, "
                     "~7 us per MiB and link at 153 GB/s); blit = k_blit_tiles of the whole 1080p bucket list on the root",
           "git_head": git_head(), "kernel_src_sha": kernel_src_sha(), "blit_us": blit, "n": {}}
    for n in (2, 4, 8):
        shard_bytes = ((510 + n - 1) // n) * 64 * 64 * 4
        dev_us, host_us = collective_us(shard_bytes)
        lat["n"][str(n)] = {"shard_render_us": res["1080p"][str(n)], "collective_call_stream_us": dev_us, "collective_call_host_us": host_us, "blit_us": blit,
                            "expected_frame_latency_us": round(res["1080p"][str(n)] + dev_us + blit, 1), "shard_bytes": shard_bytes}
    lat["n"]["1"] = {"frame_us": res["1080p"]["1"], "note": "rendered straight into the row-major frame: no collective, no blit"}
    print(json.dumps(lat, indent=1))
    for d in ("profiles", "gpurun_out"):
        json.dump(lat, open(os.path.join(ROOT, d, "expected_frame_latency.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
