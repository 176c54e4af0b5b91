#!/usr/bin/env python3
"""Headline benchmark: Mrays/s + ms/frame on the 1920x1080, 21,845-sphere default scene (BASELINE.json).

A "step" is one frame PER GPU: every 64x64 bucket of the frame through the HIP hot path (primary + shadow rays), rendered
straight into the row-major frame; for N > 1 the finished u8 frames are gathered to rank 0 over RCCL (the only collective;
the gather of step k overlaps the render of step k+1).  Per-GPU work is fixed as N grows: weak scaling, `value` counts the
rays of all N frames.  `--multi tiles` instead deals the buckets of ONE frame over the GPUs (BASELINE config 4, strong
scaling; byte-identical output, but a 0.11 ms frame bounded by its heaviest wave cannot get faster that way).  The scene is already
resident in HBM when the timed region starts.  Rank 0 prints ONE JSON line.

The headline uses the product's default traversal, RT_TRAVERSAL_SKIP (the reference's own bounding-sphere
hierarchy walked as a skip-pointer stream, bit-identical pixels and identical per-ray test counts).  The
north-star "linear scan" kernel (RT_TRAVERSAL_FLAT) is timed beside it in the same run and reported under "flat".

    python bench.py --gpus 1 --steps 100 --warmup 10
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

# name -> (width, height, samples_per_pixel, pyramid level).  The default is the configuration BASELINE.json's metric is
# quoted on; the others are BASELINE's neighbouring configs, for scaling / sizing experiments (never the headline).
WORKLOADS = {"1080p": (1920, 1080, 1, 8), "config2": (800, 600, 1, 8), "make_image": (1024, 768, 4, 8),
             "config5": (4096, 4096, 4, 9)}
WIDTH, HEIGHT, SPP, LEVEL = WORKLOADS["1080p"]
N_ITEMS = 21845
HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_OPS = 256 * 4 * 16 * 2 * 2.4e9  # un-fused f32 lane-ops/s with packed v_pk_mul/add: 256 CU x 4 SIMD x 16 lanes x 2 x
                                          # 2.4 GHz = 78.6 T (the 157 TF headline needs FMA, which parity forbids)
BYTES_PER_TEST = 16                      # one ray x one sphere = one {cx,cy,cz,r} f32 record


def cpu_baseline(budget_s=12.0):
    """The oracle (CPU restatement of the reference's hierarchical path) timed on this host, same workload."""
    import oracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    o = oracle.Scene.default(oracle.F32, LEVEL)
    t0 = time.perf_counter()
    # one single-threaded frame (the reference's default RTRACEMAXPROCS=1), unless the workload is far too big for that
    _, st, _ = o.render(WIDTH, HEIGHT, SPP, nthreads=1 if WIDTH * HEIGHT * SPP * SPP <= 2.5e7 else cores)
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
            "sample": "%d full frames of the same %dx%d spp %d L%d workload on %d threads (64x64 buckets), reference "
                      "hierarchical traversal, best frame; 1 thread: %.3f Mrays/s"
                      % (frames, WIDTH, HEIGHT, SPP, LEVEL, cores, rays / t_single / 1e6),
            "ms_per_frame": round(best * 1e3, 2), "single_core_value": round(rays / t_single / 1e6, 3)}


def flat_valu(m):
    """Useful un-fused f32 ops of the flat scan (8 per primary test with the pre-formed terms, 16 per shadow test,
    primitive.rs:56-58) over the kernel time, against the packed un-fused VALU peak."""
    st = m["my_stats"]
    shadow_tests = st["tests_executed"] - st["primary"] * N_ITEMS       # what the any-hit passes really ran
    ops = st["primary"] * N_ITEMS * 8 + shadow_tests * 16
    t = m["kern_ms"] * 1e-3
    return {"ops_per_test": "8 primary / 16 shadow", "tests_executed": st["tests_executed"],
            "achieved_Tops": round(ops / t / 1e12, 2), "peak_Tops": round(VALU_PEAK_OPS / 1e12, 1),
            "frac": round(ops / t / VALU_PEAK_OPS, 4),
            "note": "un-fused f32 ops of the tests the pipeline executed (queue lengths x pass lengths) over the time of "
                    "all its kernels, against the packed (v_pk_mul/add) un-fused VALU peak"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-flat", action="store_true", help="skip the secondary flat-scan measurement")
    ap.add_argument("--traversal", choices=("skip", "flat"), default="skip", help="traversal of the headline measurement")
    ap.add_argument("--force-collective", action="store_true",
                    help="diagnostic: take the shard -> RCCL gather -> blit path even at N = 1 (needs torch.distributed.run)")
    ap.add_argument("--multi", choices=("frames", "tiles"), default="frames",
                    help="N > 1: 'frames' = every GPU renders whole frames, N per step, gathered to rank 0 (weak scaling); "
                         "'tiles' = BASELINE config 4, the buckets of ONE frame dealt over the GPUs (strong scaling)")
    ap.add_argument("--frames-per-gather", type=int, default=4,
                    help="N > 1, 'frames' layout: frames a rank renders per RCCL gather (fewer, larger collectives)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="1080p",
                    help="1080p = the headline workload; the others are BASELINE's neighbouring configs")
    args = ap.parse_args()
    global WIDTH, HEIGHT, SPP, LEVEL, N_ITEMS
    WIDTH, HEIGHT, SPP, LEVEL = WORKLOADS[args.workload]
    N_ITEMS = (4 ** LEVEL - 1) // 3

    # Exactly ONE line may reach stdout.  RCCL prints a version banner on stdout (NCCL_DEBUG=VERSION is set on the GPU
    # boxes) whenever a communicator exists, so fd 1 is pointed at stderr for the whole run and the JSON line is written
    # to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import rust_tracer_amd as rta
    from rust_tracer_amd.dist import FrameSharder

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or args.force_collective:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    scene = rta.Scene.default(LEVEL, rta.RT_F32)
    opts = rta.RenderOptions(WIDTH, HEIGHT, SPP)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(traversal, steps, warmup):
        """-> dict with whole-job ms/step (max over ranks), this rank's kernel ms (HIP events), ray/test counters."""
        fs = FrameSharder(scene, opts, rank, world, local, traversal, force_collective=args.force_collective, mode=args.multi,
                          frames_per_gather=args.frames_per_gather)
        st = fs.render_shard(want_stats=True)          # counters of this rank's shard (equal the oracle's; tests)
        cnt = torch.tensor([st["primary"], st["shadow"]], dtype=torch.int64, device="cuda")
        if dist is not None:
            dist.all_reduce(cnt)
        primary, shadow = (int(v) for v in cnt.tolist())
        fs.run(warmup)
        barrier()
        # N = 1: the timed region is `steps` launches of the render kernel back to back on torch's current stream, which is the
        # stream handed to the C ABI -- two HIP events on that stream bracket exactly those launches, inside the timed region
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if not fs.collective:
            k0.record()
        fs.run(steps)                                  # N > 1: gather(k) on RCCL's stream overlaps render(k+1)
        if not fs.collective:
            k1.record()
        barrier()
        elapsed = time.perf_counter() - t0
        if fs.collective:
            # N > 1: the main stream also carries the waits on RCCL's stream, so the kernel alone is timed right after the
            # timed region: `steps` launches back to back between two HIP events (agrees with rocprofv3 --kernel-trace --stats)
            k0.record()
            for _ in range(steps):
                fs.render_shard()
            k1.record()
            torch.cuda.synchronize()
        kern_ms = k0.elapsed_time(k1) / steps
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        if dist is not None:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return {"elapsed": float(tt.item()), "kern_ms": kern_ms, "primary": primary, "shadow": shadow,
                "my_tests": st["sphere_tests"] + st["bound_tests"], "my_stats": st}     # algorithmic tests

    def roofline(m, kernel, note):
        alg = m["my_tests"] * BYTES_PER_TEST
        ach = alg / (m["kern_ms"] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("%s_n%d" % (kernel, world))
            except Exception:
                traffic = None
        out = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel": kernel,
               "kernel_ms": round(m["kern_ms"], 4), "tests_per_launch": m["my_tests"], "note": note}
        if traffic:
            # the physical figure beside the logical one: measured HBM bytes per launch over the live kernel time
            out["traffic_GBs"] = round(traffic / (m["kern_ms"] * 1e-3) / 1e9, 1)
            out["traffic_frac_of_peak"] = round(traffic / (m["kern_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        spath = os.path.join(ROOT, "profiles", "roofline_sq.json")
        if os.path.exists(spath):
            try:
                sq = json.load(open(spath)).get("%s_n%d" % (kernel, world))
            except Exception:
                sq = None
            if sq and sq.get("valu_insts"):
                # what actually limits the kernel: VALU issue (one wave64 VALU instruction occupies a SIMD for 4 cycles)
                cyc = m["kern_ms"] * 1e-3 * 2.4e9
                out["valu_issue"] = {"valu_insts_per_launch": sq["valu_insts"], "salu_insts_per_launch": sq.get("salu_insts"),
                                     "frac": round(sq["valu_insts"] * 4 / 1024 / cyc, 4),
                                     "note": "SQ_INSTS_VALU per launch (profiles/, rocprofv3 --pmc) x 4 cycles / 1024 SIMDs "
                                             "over the live kernel time at 2.4 GHz"}
        return out

    head_trav = rta.RT_TRAVERSAL_SKIP if args.traversal == "skip" else rta.RT_TRAVERSAL_FLAT
    m = measure(head_trav, args.steps, args.warmup)
    flat = None
    if args.traversal == "skip" and not args.no_flat:
        flat = measure(rta.RT_TRAVERSAL_FLAT, max(2, min(5, args.steps)), 1)

    if rank == 0:
        ms_per_step = m["elapsed"] / args.steps * 1e3
        rays = m["primary"] + m["shadow"]
        skip_note = ("algorithmic bytes = 16 B x (item + bound tests the reference's traversal makes for rank 0's rays, "
                     "counted by the kernel and equal to the CPU path's) / hipEvent duration of k_render_skip; records "
                     "arrive through the scalar cache / L2 (the whole scene is < 1 MB), so this is a logical rate, not "
                     "HBM traffic (SURVEY.md H3)")
        flat_note = ("algorithmic bytes = 16 B x (primary + shadow rays) x n_spheres items of rank 0's pass / hipEvent "
                     "duration of its kernels (k_flat_primary + 2 x k_flat_shadow + k_resolve_samples); every record staged "
                     "to LDS is re-used by all rays of a workgroup, so the logical rate exceeds the HBM peak; the binding "
                     "limit is un-fused f32 VALU issue (see valu)")
        if world == 1 and not args.force_collective:
            layout = "1 GPU, buckets rendered straight into the row-major frame"
        elif args.multi == "frames":
            layout = ("%d GPUs, every GPU renders a whole frame per step (%d frames per step), u8 frames gathered to rank 0 "
                      "over RCCL %d frames per rank at a time, each gather overlapped with the render of the next frames"
                      % (world, world, args.frames_per_gather))
        else:
            layout = ("%d GPUs, the buckets of ONE frame dealt round-robin (tile_id %% N), u8 shards gathered to rank 0 over "
                      "RCCL and blitted into the frame there, pipelined across frames" % world)
        out = {
            "metric": "Mrays/sec + ms/frame, 1920x1080 20k-sphere scene" if args.workload == "1080p" else
                      "Mrays/sec + ms/frame, %dx%d spp %d L%d (non-headline workload %s)" % (WIDTH, HEIGHT, SPP, LEVEL, args.workload),
            "value": round(rays / (ms_per_step * 1e-3) / 1e6, 3), "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak" if args.multi == "frames" else "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (the reference's deterministic default scene: pyramid level %d)" % LEVEL,
            "config": {"workload": "%dx%d, %d spheres (pyramid L%d), spp %d, f32, %s traversal, %d 64x64 buckets per frame; %s"
                                   % (WIDTH, HEIGHT, N_ITEMS, LEVEL, SPP, args.traversal, -(-WIDTH // 64) * -(-HEIGHT // 64), layout),
                       "width": WIDTH, "height": HEIGHT, "samples_per_pixel": SPP, "n_spheres": N_ITEMS,
                       "primary_rays": m["primary"], "shadow_rays": m["shadow"], "traversal": args.traversal,
                       "frames_per_step": world if (args.multi == "frames") else 1,
                       "parallelism": ("frames x %d" if args.multi == "frames" else "tiles/%d") % world},
            "mprimary_per_s": round(m["primary"] / (ms_per_step * 1e-3) / 1e6, 3),
            "roofline": roofline(m, "k_render_skip" if args.traversal == "skip" else "k_flat_primary",
                                 skip_note if args.traversal == "skip" else flat_note),
        }
        if flat is not None:
            fms = flat["elapsed"] / max(2, min(5, args.steps)) * 1e3
            out["flat"] = {"ms_per_step": round(fms, 4), "value": round(rays / (fms * 1e-3) / 1e6, 3), "unit": "Mrays/s",
                           "roofline": roofline(flat, "k_flat_primary", flat_note),
                           "valu": flat_valu(flat)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
