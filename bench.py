#!/usr/bin/env python3
"""Headline benchmark: Mrays/s + ms/frame on the 1920x1080, 21,845-sphere default scene (BASELINE.json).

A "step" is one frame: every 64x64 bucket of the frame through the HIP hot path (primary + shadow rays), the
RCCL gather of the u8 shards to rank 0 when N > 1, and the blit into the row-major frame.  The scene is already
resident in HBM when the timed region starts.  Rank 0 prints ONE JSON line.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WIDTH, HEIGHT, SPP, LEVEL = 1920, 1080, 1, 8
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_OPS = 256 * 4 * 32 * 2.4e9   # un-fused f32 lane-ops/s: 256 CU x 4 SIMD-32 x 2.4 GHz (FMA is forbidden by parity)
BYTES_PER_TEST = 16              # one ray x one item = one {cx,cy,cz,r} f32 record


def cpu_baseline(budget_s=12.0):
    """The oracle (CPU restatement of the reference's hierarchical path) timed on this host, same workload."""
    import oracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    o = oracle.Scene.default(oracle.F32, LEVEL)
    t0 = time.perf_counter()
    _, st, _ = o.render(WIDTH, HEIGHT, SPP, nthreads=1)
    t_single = time.perf_counter() - t0
    rays = st["primary"] + st["shadow"]
    best, frames, spent = None, 0, 0.0
    while frames < 3 or (spent < budget_s * 0.5 and frames < 20):
        t0 = time.perf_counter()
        o.render(WIDTH, HEIGHT, SPP, nthreads=cores)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        frames += 1
        spent += dt
    return {"value": round(rays / best / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d full frames of the same %dx%d spp %d L%d workload, hierarchical traversal, best frame; "
                      "1 core: %.3f Mrays/s" % (frames, WIDTH, HEIGHT, SPP, LEVEL, rays / t_single / 1e6),
            "ms_per_frame": round(best * 1e3, 2), "single_core_value": round(rays / t_single / 1e6, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import rust_tracer_amd as rta
    from rust_tracer_amd.dist import FrameSharder

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    scene = rta.Scene.default(LEVEL, rta.RT_F32)
    opts = rta.RenderOptions(WIDTH, HEIGHT, SPP)
    fs = FrameSharder(scene, opts, rank, world, local)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ray counters of this rank's shard (must equal the oracle's; checked by tests) -> rays per frame
    st = fs.render_shard(want_stats=True)
    cnt = torch.tensor([st["primary"], st["shadow"], st["sphere_tests"]], dtype=torch.int64, device="cuda")
    if dist is not None:
        dist.all_reduce(cnt)
    primary, shadow, tests = (int(v) for v in cnt.tolist())
    my_tests = st["sphere_tests"]

    for _ in range(args.warmup):
        fs.step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        # HIP events on the stream the kernel is launched on (torch's current stream is passed through the C ABI)
        ev[i][0].record()
        fs.render_shard()
        ev[i][1].record()
        fs.finish()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps

    tt = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed, kern_ms_max = (float(v) for v in tt.tolist())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        rays = primary + shadow
        value = rays / (ms_per_step * 1e-3) / 1e6
        alg_bytes = my_tests * BYTES_PER_TEST                    # this rank's launch
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch_n%d" % world)
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/sec + ms/frame, 1920x1080 20k-sphere scene",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (the reference's deterministic default scene: pyramid level 8)",
            "config": {"workload": "1920x1080, 21845 spheres (pyramid L8), spp 1, f32, flat DFS scan, 510 64x64 buckets "
                                   "round-robin over %d GPU(s), RCCL gather + device blit to rank 0" % world,
                       "width": WIDTH, "height": HEIGHT, "samples_per_pixel": SPP, "n_spheres": 21845,
                       "primary_rays": primary, "shadow_rays": shadow, "parallelism": "tiles/%d" % world},
            "mprimary_per_s": round(primary / (ms_per_step * 1e-3) / 1e6, 3),
            "kernel_ms": round(kern_ms, 4),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "note": "logical scan rate: 16 B x (primary+shadow rays) x 21845 items of rank 0's launch / its "
                                 "hipEvent duration; items are re-used from LDS/L2 so this may exceed 1 (SURVEY.md H3); "
                                 "the binding limit is un-fused f32 VALU issue, see valu"},
            "valu": {"ops_per_test": 17, "achieved_Tops": round(my_tests * 17 / (kern_ms * 1e-3) / 1e12, 2),
                     "peak_Tops": round(VALU_PEAK_OPS / 1e12, 1),
                     "frac": round(my_tests * 17 / (kern_ms * 1e-3) / VALU_PEAK_OPS, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
