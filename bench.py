#!/usr/bin/env python3
"""Headline benchmark: Mrays/s + ms/frame on the 1920x1080, 21,845-sphere default scene (BASELINE.json).

A "step" is ONE frame: every 64x64 bucket of the 1080p frame through the HIP hot path (primary + shadow rays).
  N = 1   the buckets are rendered straight into the row-major frame in HBM.
  N > 1   BASELINE config 4: the buckets of that ONE frame are dealt round-robin (`tile_id % N`) over one process per GPU,
          every rank renders its shard tile-major, ONE RCCL gather brings the u8 shards to rank 0 over xGMI and rank 0
          blits them into the frame (strong scaling: total work fixed, `"scaling": "strong"`).  Consecutive frames are
          software-pipelined (gather(k) overlaps render(k+1)).  Two more layouts ride along under their own keys:
          `weak_frames` (every GPU renders whole frames) and `config5_tiles` (4096^2, 87,381 spheres, spp 4: the
          workload where sharding pays).
The scene is resident in HBM when the timed region starts.  Rank 0 prints ONE JSON line.

Timed region: `--repeats` (5) repetitions of: barrier + synchronize, EXACTLY `--steps` steps, barrier + synchronize; the max
over ranks of each repetition; `ms_per_step` is the median repetition (min / max beside it).  The frame buffers are zeroed
after the warm-up, so the frame left behind was produced by the timed launches: its CRC32 must equal the committed oracle
vector (tests/golden/oracle_vectors.json) or the run fails.

`roofline` (rank 0's render kernel): bound = VALU issue -- this path is un-fused f32 arithmetic on records that live in the
scalar cache / L2, not an HBM stream.  achieved = SURVEY.md 8(d)'s 17 flops per ray x record test x the tests of one launch
(counted by the kernel in this run, equal to the CPU path's) / the kernel's launch duration (HIP events on the launch
stream, in this run).  Two peaks are printed: `peak` = the guide's nominal 1,024 SIMDs x 64 lanes x 2.4 GHz / 2 cycles per wave64
VALU op = 78.6 T un-fused lane-ops/s (`frac` is against this one), and `peak_probe` = the same with the cycles a SIMD was MEASURED to
need per wave64 VOP2 instruction at 8 waves per SIMD (profiles/r02_valu_issue_probe.json, tools/valu_issue_probe.hip: 2.22 -> 70.9 T;
`frac_probe`).  `path_arithmetic_frac`: what the path's arithmetic needs with the ray-independent terms pre-formed -- 8 lane-ops per
primary test, 16 per shadow test (the kernel counts the two kinds separately) -- over the same time and nominal peak; the filtered
loops (rt_skip.hpp, VAR 16) rule most tests out with 4 - 6 fused operations instead, so this is work the reference needs, not work the
kernel issues.  `valu_lane_utilisation` = those lane-ops / (SQ_INSTS_VALU x 64), when profiles/ holds counters of these sources.
Figures that need rocprofv3 counters (instruction issue, HBM traffic) are quoted from profiles/ under `from_profiles`, stamped with
the kernel sources they were collected on, and dropped when that stamp is not the sources' of this run.
The timed region is at least 0.2 s: when --steps x --repeats frames take less, more repetitions are run (a step stays one frame).

`configs` (N = 1): BASELINE's neighbouring configurations (config 2 = 800x600, `make image`, config 5) in one short repetition each after
the headline -- ms per frame, the same 17-flop fraction of the nominal peak, the kernel the library chose, the frame CRC against the
committed oracle vector; never `value`.  `first_frame_ms`: a fresh Scene's first 1080p frame (dispatch tables, cost map and all).
N > 1 adds `frame_latency_ms` (ONE frame, render -> gather -> blit, nothing batched, nothing pipelined) beside the batched, pipelined
`ms_per_step`, `frames_per_gather`, the host cost of one collective call measured in this run, and the builder's expectation of a
rank's shard render time from one GPU (`expected_shard_render_us`).

`seam` (N = 1): the boundary the reference binds, timed from native threads by rust-tracer_amd/seam_bench (child process):
host_tiles / host_region / end_to_end (render + D2H + PPM write).  `flat`: the north-star linear scan, same run.
`cpu_baseline`: the oracle on this host's cores.

    python bench.py --gpus 1 --steps 100 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# name -> (width, height, samples_per_pixel, pyramid level, golden case in tests/golden/oracle_vectors.json).  The default is the
# configuration BASELINE.json's metric is quoted on; the others are BASELINE's neighbouring configs (never the headline).
WORKLOADS = {"1080p": (1920, 1080, 1, 8, "config3_1920x1080_f32"), "config2": (800, 600, 1, 8, "config2_800x600"),
             "make_image": (1024, 768, 4, 8, "make_image_1024x768_spp4"), "config5": (4096, 4096, 4, 9, "config5_4096x4096_spp4_L9"),
             # BASELINE config 5 with EXACTLY 100,000 spheres: an arbitrary list (tests/scenes.py) with an automatically built hierarchy
             "config5_100k": (4096, 4096, 4, "100k", "config5_100k_4096x4096_spp4")}
HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SIMD, LANES, CLOCK_HZ = 1024, 64, 2.4e9
BYTES_PER_TEST, FLOPS_PER_TEST = 16, 17  # one ray x one {cx,cy,cz,r} record; 3 sub + 8 mul + 6 add/sub to the reject test
PROBE = os.path.join(ROOT, "profiles", "r02_valu_issue_probe.json")


def kernel_src_sha():
    """Stamp of the device code: sha1 over the kernel sources (profiles/ summaries carry the stamp they were collected on)."""
    d = os.path.join(ROOT, "rust-tracer_amd", "csrc")
    h = hashlib.sha1()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def physical_cores():
    """Distinct (package, core) pairs of /proc/cpuinfo among the CPUs this process may run on (None when it cannot be told)."""
    try:
        allowed = os.sched_getaffinity(0)
        seen, cpu, phys = set(), None, None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k = k.strip()
            if k == "processor":
                cpu, phys = int(v), None
            elif k == "physical id":
                phys = int(v)
            elif k == "core id" and cpu in allowed:
                seen.add((phys, int(v)))
        return len(seen) or None
    except Exception:
        return None


def probe_cycles():
    """-> {(kind prefix, waves per SIMD): cycles per instruction per SIMD} from the committed VALU-issue probe."""
    out = {}
    try:
        for r in json.load(open(PROBE))["results"]:
            out[(r["kind"], r["waves_per_simd"])] = r
    except Exception:
        pass
    return out


def golden_case(name):
    try:
        for c in json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_vectors.json")))["cases"]:
            if c["name"] == name:
                return c
    except Exception:
        pass
    return None


def cpu_baseline(width, height, spp, level, budget_s=12.0):
    """The oracle (CPU restatement of the reference's hierarchical path) timed on this host, same workload."""
    import oracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if level == "100k":
        import numpy as np
        from rust_tracer_amd.scene import build_hierarchy
        from tests.scenes import hundred_thousand_spheres
        items, bounds, ranges, _ = build_hierarchy(hundred_thousand_spheres(), eye=(0.0, 0.0, -4.0))      # (as Scene.from_spheres_auto builds it)
        o = oracle.Scene.from_ranges(items.astype(np.float64), bounds.astype(np.float64), ranges)
    else:
        o = oracle.Scene.default(oracle.F32, level)
    t0 = time.perf_counter()
    # one single-threaded frame (the reference's default RTRACEMAXPROCS=1), unless the workload is far too big for that
    _, st, _ = o.render(width, height, spp, nthreads=1 if width * height * spp * spp <= 2.5e7 else cores)
    t_single = time.perf_counter() - t0
    rays = st["primary"] + st["shadow"]
    best, frames, spent = None, 0, 0.0
    while frames < 3 or (spent < budget_s * 0.5 and frames < 20):
        t0 = time.perf_counter()
        o.render(width, height, spp, nthreads=cores)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        frames += 1
        spent += dt
    return {"value": round(rays / best / 1e6, 3), "unit": "Mrays/s", "cores": cores, "cores_are": "hardware threads used (one pool thread per logical CPU of this process's affinity mask)",
            "physical_cores": physical_cores(), "cpu_model": cpu_model(), "kind": "port",
            "sample": "%d full frames of the same %dx%d spp %d %s workload on %d threads (64x64 buckets), reference "
                      "hierarchical traversal, best frame; 1 thread: %.3f Mrays/s"
                      % (frames, width, height, spp, "100,000 spheres" if level == "100k" else "L%d" % level, cores, rays / t_single / 1e6),
            "ms_per_frame": round(best * 1e3, 2), "single_core_value": round(rays / t_single / 1e6, 3)}


def run_seam(threads):
    """rust-tracer_amd/seam_bench as a child process (started before this process touches the GPU)."""
    exe = os.path.join(ROOT, "rust-tracer_amd", "seam_bench")
    if not os.path.exists(exe):
        return {"error": "rust-tracer_amd/seam_bench is not built"}
    try:
        r = subprocess.run([exe, "--frames", "20", "--threads", str(threads)], capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            return {"error": "seam_bench exit %d: %s" % (r.returncode, r.stderr[-300:])}
        d = json.loads(r.stdout)
        d["how"] = ("native threads through the C ABI (rust-tracer_amd/csrc/host/seam_bench.cpp), PCIe-inclusive, never `value`: host_tiles = "
                    "rt_render_tiles, all buckets in one call; host_region = one rt_render_region call per bucket (render.rs:283-294) from 1 and "
                    "from T pool threads, concurrent calls merged into shared passes; end_to_end = render + D2H + PPM encode + file write")
        return d
    except Exception as e:          # noqa: BLE001  (a failed side leg must not take the headline down)
        return {"error": repr(e)}


def native_gang(n_devices, golden):
    """rust-tracer_amd/gang_bench as a child process: the 1080p frame through rt_gang_render_frames on devices 0..N-1 of ONE process."""
    exe = os.path.join(ROOT, "rust-tracer_amd", "gang_bench")
    if not os.path.exists(exe):
        return {"error": "rust-tracer_amd/gang_bench is not built"}
    try:
        time.sleep(1.0)                     # the other ranks' processes are exiting
        env = {k: v for k, v in os.environ.items() if k != "RTRACE_HIP_LIBRARY"}
        r = subprocess.run([exe, "--devices", str(n_devices), "--frames", "200"], capture_output=True, text=True, timeout=120, env=env)
        if r.returncode != 0:
            return {"error": "gang_bench exit %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
        d = json.loads(r.stdout.strip().splitlines()[-1])
        if golden:
            d["crc_ok"] = d.get("frame_crc32") == golden["frame_crc32"]
        d["how"] = ("never `value`: ONE process, rt_gang_create over %d devices (ncclCommInitAll), rt_gang_render_frames pipelines render(f + 1) over "
                    "gather(f) + blit(f); ms_per_frame = host wall time of a 200-frame call / 200; frame_latency_ms = one rt_gang_render_frame call" % n_devices)
        return d
    except Exception as e:          # noqa: BLE001
        return {"error": repr(e)}


def make_image_wall(cores):
    """`make image` as the caller sees it (/root/reference/Makefile:6-7: `time ./target/release/rtrace --samples-per-pixel=4 --width=1024
    --height=768 out.tga`): a FRESH rtrace process, process start -> exit, output on /dev/shm; the parts rtrace --timings reports; the
    md5 of the file against the reference image's; and the CPU port (oracle/rtrace_cpu) as a process beside it.  Child
    processes, started before this process touches the GPU."""
    exe = os.path.join(ROOT, "rust-tracer_amd", "rtrace")
    cpu = os.path.join(ROOT, "oracle", "rtrace_cpu")
    tmp = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    out = os.path.join(tmp, "bench_make_image_%d.tga" % os.getpid())
    args = ["--samples-per-pixel=4", "--width=1024", "--height=768"]
    res = {"command": "rtrace --samples-per-pixel=4 --width=1024 --height=768 out.tga", "reference_md5": "63e866ff6d39850bbcf0fcea87024d19"}
    env = {k: v for k, v in os.environ.items() if k not in ("RTRACEMAXPROCS", "RTRACE_HIP_LIBRARY")}

    def md5(path):
        try:
            return hashlib.md5(open(path, "rb").read()).hexdigest()
        except OSError:
            return None

    try:
        walls, parts = [], None
        for i in range(4):                    # the first run also pays the page cache (binary, libraries, code object)
            if os.path.exists(out):
                os.remove(out)
            t0 = time.perf_counter()
            r = subprocess.run([exe, "--timings"] + args + [out], capture_output=True, text=True, timeout=300, env=env)
            walls.append((time.perf_counter() - t0) * 1e3)
            if r.returncode != 0:
                return dict(res, error="rtrace exit %d: %s" % (r.returncode, r.stderr[-300:]))
            for line in r.stderr.splitlines():
                if line.startswith("{"):
                    parts = json.loads(line)
        res.update({"make_image_wall_ms": round(min(walls[1:]), 2), "wall_ms_each": [round(w, 2) for w in walls], "first_run_wall_ms": round(walls[0], 2),
                    "parts_ms": parts, "md5_ok": md5(out) == res["reference_md5"],
                    "parts_note": "from main() of the last run: args; host_scene = Scene::default on the host; runtime_init = the first HIP call; device_context = the first call "
                                  "that needs the device (a pinned page); device_scene = rt_scene_create, of which first_queue = creating its stream -- in a process "
                                  "without one that is where the RUNTIME makes its first hardware queue (~19 ms, whoever creates the first stream or launches the "
                                  "first kernel: tools/init_probe.hip, profiles/r06_init_probe.log) -- and device_scene_own = the rest: the library's allocations, uploads "
                                  "by kernel (no copy engine) and stream derivation, its code object loaded by a helper thread meanwhile; the cost map and the "
                                  "dispatch orders wait for a second frame; render_and_first_write = Renderer::render up to the Drop; "
                                  "drop_write = the writer's final write; wall - main_to_here = exec, dynamic linking, static initialisers, exit / runtime teardown"})
    except Exception as e:          # noqa: BLE001
        res["error"] = repr(e)
    if os.path.exists(cpu):
        for name, nthreads in (("cpu_port_wall_ms_1_thread", 1), ("cpu_port_wall_ms_all_threads", cores)):
            try:
                best = None
                for _ in range(1 if nthreads == 1 else 2):
                    t0 = time.perf_counter()
                    r = subprocess.run([cpu] + args + [out], capture_output=True, timeout=600, env=dict(env, RTRACEMAXPROCS=str(nthreads)))
                    dt = (time.perf_counter() - t0) * 1e3
                    best = dt if best is None else min(best, dt)
                res[name] = round(best, 2) if r.returncode == 0 else None
                res["cpu_port_md5_ok"] = md5(out) == res["reference_md5"]
            except Exception as e:  # noqa: BLE001
                res[name] = repr(e)
        res["cpu_port_threads"] = cores
    if os.path.exists(out):
        os.remove(out)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=5, help="repetitions of the timed --steps loop (median reported)")
    ap.add_argument("--min-timed-region", type=float, default=0.2,
                    help="seconds: more repetitions of the --steps loop are run until the timed region is at least this long")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-flat", action="store_true", help="skip the secondary flat-scan measurement")
    ap.add_argument("--no-seam", action="store_true", help="skip the host-boundary legs (seam_bench child process)")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements (N > 1: weak_frames and config5_tiles; N = 1: frames_in_flight)")
    ap.add_argument("--no-configs", action="store_true", help="N = 1: skip BASELINE's neighbouring configurations (`configs`) and `first_frame_ms`")
    ap.add_argument("--traversal", choices=("skip", "flat"), default="skip", help="traversal of the headline measurement")
    ap.add_argument("--force-collective", action="store_true",
                    help="diagnostic: take the shard -> RCCL gather -> blit path even at N = 1 (needs torch.distributed.run)")
    ap.add_argument("--multi", choices=("tiles", "frames"), default="tiles",
                    help="N > 1 headline layout: 'tiles' = BASELINE config 4, the buckets of ONE frame dealt over the GPUs (strong scaling); "
                         "'frames' = every GPU renders whole frames, N per step (weak scaling)")
    ap.add_argument("--frames-per-gather", type=int, default=4,
                    help="N > 1: frames (or, 'tiles' layout, shards) a rank renders per RCCL gather (fewer, larger collectives)")
    ap.add_argument("--process-group-backend", choices=("nccl", "gloo"), default="nccl", help="N > 1: torch.distributed backend (nccl = RCCL over xGMI)")
    ap.add_argument("--sharder", default="rust_tracer_amd.dist:FrameSharder",
                    help="module:Class of the FrameSharder the ranks run (the product's; tests substitute one whose collective can share a GPU)")
    ap.add_argument("--no-make-image", action="store_true", help="N = 1: skip the `make image` process wall-time leg")
    ap.add_argument("--no-native-gang", action="store_true", help="N > 1: skip the single-process rt_gang side measurement (gang_bench child)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="1080p",
                    help="1080p = the headline workload; the others are BASELINE's neighbouring configs")
    args = ap.parse_args()
    width, height, spp, level, golden_name = WORKLOADS[args.workload]
    n_items = 100000 if level == "100k" else (4 ** level - 1) // 3

    # Exactly ONE line may reach stdout.  RCCL prints a version banner on stdout (NCCL_DEBUG=VERSION is set on the GPU
    # boxes) whenever a communicator exists, so fd 1 is pointed at stderr for the whole run and the JSON line is written
    # to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))

    seam, image_wall = None, None
    if world == 1 and not args.no_seam and not args.force_collective and args.workload == "1080p":
        seam = run_seam(min(64, os.cpu_count() or 1))       # a child process, before anything here initialises the GPU
    if world == 1 and not args.no_make_image and not args.force_collective and args.workload == "1080p":
        image_wall = make_image_wall(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))

    import numpy as np
    import torch
    import rust_tracer_amd as rta
    import importlib
    _mod, _cls = args.sharder.split(":")
    FrameSharder = getattr(importlib.import_module(_mod), _cls)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or args.force_collective:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.process_group_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    comm_dev = getattr(FrameSharder, "comm_device", "cuda") if dist is not None else "cuda"          # where the few scalars the ranks exchange live
    real_rccl = dist is None or args.process_group_backend == "nccl"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    scenes = {}

    def scene_of(lv):
        if lv not in scenes:
            if lv == "100k":
                from tests.scenes import hundred_thousand_spheres
                scenes[lv] = rta.Scene.from_spheres_auto(hundred_thousand_spheres())
            else:
                scenes[lv] = rta.Scene.default(lv, rta.RT_F32)
        return scenes[lv]

    def measure(wl, traversal, steps, warmup, repeats, mode, min_region_s=0.0):
        """-> dict: whole-job ms per step of each repetition (max over ranks), this rank's kernel ms (HIP events), counters,
        and the CRC of the frame the timed launches left on rank 0."""
        w, h, k, lv, gname = WORKLOADS[wl]
        opts = rta.RenderOptions(w, h, k)
        fs = FrameSharder(scene_of(lv), opts, rank, world, local, traversal, force_collective=args.force_collective, mode=mode,
                          frames_per_gather=args.frames_per_gather)
        st = fs.render_shard(want_stats=True)          # counters of this rank's shard (equal the oracle's; tests)
        if traversal != rta.RT_TRAVERSAL_SKIP:
            st = dict(st, primary_tests=None)
        cnt = torch.tensor([st["primary"], st["shadow"]], dtype=torch.int64, device=comm_dev)
        if dist is not None:
            dist.all_reduce(cnt)
        primary, shadow = (int(v) for v in cnt.tolist())
        fs.run(warmup)
        barrier()
        # what the timed launches leave behind is checked afterwards: start from zeroed buffers
        for b in fs.shards:
            b.zero_()
        if fs.frame is not None:
            fs.frame.zero_()
        if getattr(fs, "gathered_flat", None):
            for g in fs.gathered_flat:
                g.zero_()
        barrier()
        reps, kern = [], []
        rep = 0
        while rep < repeats or (min_region_s and sum(r * steps for r in reps) / 1e3 < min_region_s and rep < 4000):
            rep += 1
            # N = 1: the timed region is `steps` launches of the render kernel back to back on torch's current stream, which is the
            # stream handed to the C ABI -- two HIP events on that stream bracket exactly those launches, inside the timed region
            k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            barrier()
            t0 = time.perf_counter()
            if not fs.collective:
                k0.record()
            fs.run(steps)                              # N > 1: gather(k) on RCCL's stream overlaps render(k+1)
            if not fs.collective:
                k1.record()
            barrier()
            elapsed = time.perf_counter() - t0
            tt = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
            if dist is not None:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            reps.append(float(tt.item()) / steps * 1e3)
            if not fs.collective:
                kern.append(k0.elapsed_time(k1) / steps)
        launched = rta.capi.last_launch()               # what this thread's last (timed) render call launched
        # The clock the card holds under this load: one more counting launch right behind the timed ones -- its longest
        # wave's s_memtime (shader clock) over s_memrealtime (constant 100 MHz) span; counting launches are a diagnostic flavour, the
        # timed kernels carry no stamp.
        clk = None
        if traversal == rta.RT_TRAVERSAL_SKIP:
            st2 = fs.render_shard(want_stats=True)
            if st2.get("longest_wave_ref100mhz"):
                clk = {"mhz": 100.0 * st2["longest_wave_cycles"] / st2["longest_wave_ref100mhz"], "over_us": st2["longest_wave_ref100mhz"] / 100.0}
        crc, crc_ok = None, None
        if rank == 0:
            if mode == "frames" and fs.collective:
                frame = fs.frame_host(slot=0, of_rank=world - 1, index=0)        # a frame that crossed the wire
            else:
                frame = fs.frame_host()
            crc = zlib.crc32(np.ascontiguousarray(frame).tobytes()) & 0xFFFFFFFF
            g = golden_case(gname)
            crc_ok = (crc == g["frame_crc32"]) if g else None
        if fs.collective:
            # N > 1: the main stream also carries the waits on RCCL's stream, so the kernel alone is timed right after the
            # timed region: `steps` launches back to back between two HIP events (agrees with rocprofv3 --kernel-trace --stats)
            k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            k0.record()
            for _ in range(steps):
                fs.render_shard()
            k1.record()
            torch.cuda.synchronize()
            kern.append(k0.elapsed_time(k1) / steps)
        srt = sorted(reps)
        return {"ms_reps": reps, "ms_per_step": srt[len(srt) // 2], "kern_ms": sorted(kern)[len(kern) // 2], "primary": primary, "shadow": shadow,
                "my_tests": st["sphere_tests"] + st["bound_tests"], "my_stats": st, "crc": crc, "crc_ok": crc_ok, "launched": launched,
                "timed_region_s": sum(r * steps for r in reps) / 1e3, "frames_per_step": world if (mode == "frames" and fs.collective) else 1, "clock": clk}

    def measure_in_flight(wl, steps, n_streams=2):
        """N = 1 only, never `value`: the same `steps` launches dealt round-robin over n_streams HIP streams (each its own frame
        buffer), so that frame k + 1's waves fill the SIMDs that frame k's last, longest waves leave idle.  What a caller of the
        asynchronous entry points gets by keeping two frames in flight; the difference to `ms_per_step` is the share of a frame
        that is its tail."""
        w, h, k, lv, gname = WORKLOADS[wl]
        dev = scene_of(lv).device(local)
        regs = dev._regions([tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, k))])
        streams = [torch.cuda.Stream() for _ in range(n_streams)]
        outs = [torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda") for _ in range(n_streams)]
        for i in range(2 * n_streams):
            dev.render_frame_device((w, h, k), regs, outs[i % n_streams].data_ptr(), streams[i % n_streams].cuda_stream)
        torch.cuda.synchronize()
        for o in outs:
            o.zero_()
        reps = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                dev.render_frame_device((w, h, k), regs, outs[i % n_streams].data_ptr(), streams[i % n_streams].cuda_stream)
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t0) / steps * 1e3)
        g = golden_case(gname)
        crcs = [zlib.crc32(o.cpu().numpy().tobytes()) & 0xFFFFFFFF for o in outs]
        return {"streams": n_streams, "ms_per_step": sorted(reps)[len(reps) // 2], "ms_reps": reps,
                "crc_ok": (all(c == g["frame_crc32"] for c in crcs) if g else None)}

    probe = probe_cycles()
    src_sha = kernel_src_sha()

    def from_profiles(kernel, kern_ms):
        """Counter figures of rank 0's kernel quoted from profiles/ (rocprofv3 --pmc, separate passes) -- only when they were
        collected on exactly these kernel sources."""
        out = {}
        for name in ("roofline_sq.json", "roofline_traffic.json", "roofline_waves.json"):
            p = os.path.join(ROOT, "profiles", name)
            try:
                d = json.load(open(p))
            except Exception:
                continue
            ok = d.get("kernel_src_sha") == src_sha
            out[name] = {"tag": d.get("tag"), "git_head": d.get("git_head"), "kernel_src_sha": d.get("kernel_src_sha"), "matches_this_run": ok}
            if not ok:
                continue
            v = d.get("%s_n%d" % (kernel, world)) or (d.get("%s_n1" % kernel) if world == 1 else None)
            if v is None:
                continue
            t = kern_ms * 1e-3
            if name == "roofline_waves.json":
                # tools/wave_timeline.py: every wave's start / end recorded by the launch itself (the trace flavour of the loops, hooks build)
                out["wave_timeline"] = dict(v, note="one traced launch of the same kernel (tools/wave_timeline.py, profiles/): span_us = first wave's start to last "
                                                    "wave's end; longest_wave_us = the longest single wave -- a frame cannot end before its longest wave does")
            elif name == "roofline_traffic.json":
                out["hbm_traffic"] = {"bytes_per_launch": v, "GBs": round(v / t / 1e9, 1), "frac_of_hbm_peak": round(v / t / 1e9 / HBM_PEAK_GBS, 4),
                                      "note": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch (MI355X_MICROARCH.md gfx950 correction) over this run's kernel time"}
            else:
                insts = sum(v.get(k) or 0 for k in ("valu_insts", "salu_insts", "smem_insts", "branch_insts", "other_insts"))
                ref = {name: (probe.get((kind, 8)) or {}).get("cycles_per_instruction_per_simd") for name, kind in (
                    ("vop2", "v_mul_f32 (SGPR x VGPR, independent)"), ("vop2_new_sgpr_operand", "v_mul_f32 (a DIFFERENT SGPR x VGPR each instruction, independent)"),
                    ("vop3_cmp_e64", "v_cmp_lt_f32_e64 -> SGPR pair"), ("packed", "v_pk_fma_f32 (VGPR pairs, independent)"),
                    ("step_mix_10_valu_12_salu", "traversal-step mix: 10 VALU + 12 SALU per 22"))}
                if v.get("valu_active_quad_cycles"):
                    # NOT an issue rate: SQ_ACTIVE_INST_VALU sums, per wave, the quad-cycles in which a vector instruction of that wave is in
                    # progress -- eight resident waves overlap, so it reads ~0.9 on a launch whose waves wait half their cycles
                    out["valu_inflight_quadcycle_share"] = round(v["valu_active_quad_cycles"] * 4 / (N_SIMD * CLOCK_HZ * t), 4)
                if v.get("valu_insts"):
                    # the share of the SIMDs' nominal issue slots the launch's vector instructions fill: SQ_INSTS_VALU x 2 cycles per wave64
                    # instruction / (1,024 SIMDs x 2.4 GHz x kernel time)
                    out["valu_issue_slot_frac"] = round(v["valu_insts"] * 2.0 / (N_SIMD * CLOCK_HZ * t), 4)
                if v.get("wave_quad_cycles") and v.get("wait_any_quad_cycles") is not None:
                    wc = v["wave_quad_cycles"]
                    out["wave_cycles"] = {
                        "waiting": round(v["wait_any_quad_cycles"] / wc, 4), "issuing": round((v.get("active_inst_any_quad_cycles") or 0) / wc, 4),
                        "stalled_at_issue": round((v.get("wait_inst_any_quad_cycles") or 0) / wc, 4),
                        "note": "SQ_WAIT_ANY / SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES: the share of its cycles a wave is parked at "
                                "s_waitcnt (here: for node records from the scalar cache), issuing, or ready but not issued"}
                if v.get("sqc_dcache_req"):
                    rq = v["sqc_dcache_req"]
                    out["scalar_cache"] = {"requests_per_launch": rq, "hit": round((v.get("sqc_dcache_hits") or 0) / rq, 4), "miss": round((v.get("sqc_dcache_misses") or 0) / rq, 4),
                                           "miss_on_a_line_already_requested": round((v.get("sqc_dcache_misses_duplicate") or 0) / rq, 4)}
                if insts:
                    out["instruction_issue"] = {
                        "wave_instructions_per_launch": insts, "valu": v.get("valu_insts"), "salu": v.get("salu_insts"), "smem": v.get("smem_insts"),
                        "achieved_Ginst_s": round(insts / t / 1e9, 1),
                        "simd_cycles_per_wave_instruction": round(N_SIMD * CLOCK_HZ * t / insts, 3),
                        "probe_cycles_per_instruction": ref,
                        "note": "wave-instructions per launch (SQ_INSTS_*) and the SIMD cycles this run's kernel time leaves for each "
                                "(1,024 SIMDs x 2.4 GHz x kernel time / instructions), beside what the probe measured a SIMD needs per "
                                "instruction of each class at 8 waves per SIMD: a launch whose average sits between those figures keeps its "
                                "SIMDs issuing for its whole duration"}
        return out

    def roofline(m, kernel, flops_per_test, note):
        t = m["kern_ms"] * 1e-3
        vop2 = probe.get(("v_mul_f32 (SGPR x VGPR, independent)", 8))
        cyc = vop2["cycles_per_instruction_per_simd"] if vop2 else None
        peak = N_SIMD * LANES * CLOCK_HZ / 2.0 / 1e12              # MI355X_MICROARCH.md: SIMD-32, 2 cycles per wave64 VALU op
        peak_probe = N_SIMD * LANES * CLOCK_HZ / cyc / 1e12 if cyc else None
        ach = m["my_tests"] * flops_per_test / t / 1e12
        logical = m["my_tests"] * BYTES_PER_TEST / t / 1e9
        out = {"bound": "valu_issue", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s (un-fused f32 lane ops)",
               "frac": round(ach / peak, 4), "traffic": None, "kernel": kernel, "kernel_ms": round(m["kern_ms"], 4),
               "kernel_flavour": ("k_render_skip_fast" + ("_coop" if "cooperative" in m["launched"] else "") + " (rt_skip_fast.hpp)") if "fast_kernel" in m["launched"] else None,
               "tests_per_launch": m["my_tests"], "flops_per_test": flops_per_test,
               "peak_source": "MI355X_MICROARCH.md nominal: 1,024 SIMDs x 64 lanes x 2.4 GHz / 2 cycles per wave64 VALU op (un-fused: one lane-op per lane)",
               "peak_probe": round(peak_probe, 1) if peak_probe else None, "frac_probe": round(ach / peak_probe, 4) if peak_probe else None,
               "peak_probe_source": ("profiles/r02_valu_issue_probe.json: %.3f cycles per wave64 VOP2 instruction per SIMD measured at 8 waves per SIMD" % cyc) if cyc else None,
               "hbm_logical": {"GBs": round(logical, 1), "bytes_per_test": BYTES_PER_TEST,
                               "note": "16 B x tests / kernel time: a LOGICAL record rate -- the records come from the scalar cache / L2 / LDS, "
                                       "not from HBM, so it is not a fraction of the HBM peak (SURVEY.md H3)"},
               "note": note}
        if m.get("clock"):
            mhz = m["clock"]["mhz"]
            out["clock_mhz_measured"] = round(mhz, 1)
            out["frac_at_measured_clock"] = round(ach / (peak * mhz * 1e6 / CLOCK_HZ), 4)
            out["clock_source"] = ("s_memtime / s_memrealtime x 100 MHz over the longest wave (%.0f us) of a counting launch issued right behind the timed "
                                   "launches; `frac` stays priced at the nominal %.1f GHz" % (m["clock"]["over_us"], CLOCK_HZ / 1e9))
        pt = m["my_stats"].get("primary_tests")
        if pt is not None and kernel in ("k_render_skip", "k_render_skip2"):
            ops = 8 * pt + 16 * (m["my_tests"] - pt)
            out["path_arithmetic"] = {"primary_tests": pt, "shadow_tests": m["my_tests"] - pt, "lane_ops": ops,
                                      "frac": round(ops / t / 1e12 / peak, 4),
                                      "note": "8 lane-ops per primary test (terms pre-formed against the shared eye), 16 per shadow test -- the arithmetic the reference's "
                                              "tests need -- over the kernel time and the nominal peak.  The filtered loops rule most tests out with a cheaper proven "
                                              "bound (4 - 6 fused operations), so the kernel issues fewer operations than this"}
        fp = from_profiles(kernel, m["kern_ms"])
        out["from_profiles"] = fp
        if "hbm_traffic" in fp:
            out["traffic"] = fp["hbm_traffic"]["bytes_per_launch"]
        for k in ("valu_issue_slot_frac", "valu_inflight_quadcycle_share"):
            if k in fp:
                out[k] = fp[k]
        if "wave_timeline" in fp:
            out["longest_wave_us"] = fp["wave_timeline"].get("longest_wave_us")
            out["span_us"] = fp["wave_timeline"].get("span_us")
        for k in ("wave_cycles", "scalar_cache"):
            if k in fp:
                out[k] = fp[k]
        if "wave_cycles" in fp:
            out["what_binds"] = ("`frac` is the contract's compute roofline (the reference's 17 flops per test over the nominal vector peak); "
                                 "`valu_issue_slot_frac` is the share of the vector unit's issue slots the launch really fills.  What the launch waits for is the "
                                 "scalar data cache -- `wave_cycles.waiting` of a wave's cycles at s_waitcnt, `scalar_cache.miss` + `miss_on_a_line_already_requested` "
                                 "of its node fetches an L2 round trip -- and its longest wave (`longest_wave_us` of `span_us`): DESIGN.md 4.1")
        valu = (fp.get("instruction_issue") or {}).get("valu")
        if valu and "path_arithmetic" in out:
            out["valu_lane_utilisation"] = round(out["path_arithmetic"]["lane_ops"] / (valu * LANES), 4)
        return out

    head_trav = rta.RT_TRAVERSAL_SKIP if args.traversal == "skip" else rta.RT_TRAVERSAL_FLAT
    multi = args.multi if (world > 1 or args.force_collective) else "tiles"
    m = measure(args.workload, head_trav, args.steps, args.warmup, max(1, args.repeats), multi, min_region_s=args.min_timed_region)
    # which kernel the library chose for this workload (rt_capi.hip skip2_by_default): large frames walk two rays per lane
    skip_kernel = "k_render_skip2" if "two_rays" in m["launched"] else "k_render_skip"
    flat = None
    if args.traversal == "skip" and not args.no_flat and world == 1:
        flat = measure(args.workload, rta.RT_TRAVERSAL_FLAT, max(2, min(5, args.steps)), 1, 3, multi)
    in_flight = None
    if world == 1 and not args.force_collective and not args.no_extras and args.traversal == "skip":
        try:
            in_flight = measure_in_flight(args.workload, max(args.steps, 100))
        except Exception as e:                       # a side measurement must not take the headline down with it
            sys.stderr.write("bench.py: frames_in_flight skipped: %r\n" % (e,))
    peak_nominal = N_SIMD * LANES * CLOCK_HZ / 2.0 / 1e12
    configs, first_frame = None, None
    if world == 1 and not args.force_collective and not args.no_configs and args.workload == "1080p" and args.traversal == "skip":
        # BASELINE's neighbouring configurations, one short leg each (never `value`): what the driver's record otherwise never sees
        configs = {}
        for wl, steps_c, warm_c in (("config2", 200, 60), ("make_image", 50, 10), ("config5", 4, 2)):
            try:
                e = measure(wl, head_trav, steps_c, warm_c, 3, "tiles")
                kern = "k_render_skip2" if "two_rays" in e["launched"] else "k_render_skip"
                if "cooperative" in e["launched"]:
                    kern += " (+ lane-cooperative quads, rt_coop.hpp)"
                w_c, h_c, k_c, lv_c, _ = WORKLOADS[wl]
                if k_c > 1:
                    kern += " + k_resolve_words"
                t_c = e["kern_ms"] * 1e-3
                configs[wl] = {"workload": "%dx%d spp %d L%d" % (w_c, h_c, k_c, lv_c), "ms": round(e["ms_per_step"], 4), "kernel_ms": round(e["kern_ms"], 4),
                               "frac": round(e["my_tests"] * FLOPS_PER_TEST / t_c / 1e12 / peak_nominal, 4), "kernel": kern,
                               "value": round((e["primary"] + e["shadow"]) / (e["ms_per_step"] * 1e-3) / 1e6, 1), "unit": "Mrays/s",
                               "frame_crc_ok": e["crc_ok"]}
            except Exception as ex:                  # a side leg must not take the headline down with it
                configs[wl] = {"error": repr(ex)}
        scenes.clear()                               # (config 5's per-sample buffers)
        try:
            # a fresh Scene's first frame: everything a one-shot caller (`make image`) pays that the steady state does not
            fresh = rta.Scene.default(8, rta.RT_F32)
            t0 = time.perf_counter()
            dev = fresh.device(local)
            torch.cuda.synchronize()
            t_create = time.perf_counter() - t0
            regs_f = dev._regions([tuple(r) for r in rta.buckets(rta.RenderOptions(1920, 1080, 1))])
            buf = torch.zeros(1920 * 1080 * 4, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dev.render_frame_device((1920, 1080, 1), regs_f, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            t_first = time.perf_counter() - t0
            t0 = time.perf_counter()
            dev.render_frame_device((1920, 1080, 1), regs_f, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            t_second = time.perf_counter() - t0
            g = golden_case("config3_1920x1080_f32")
            first_frame = {"first_frame_ms": round(t_first * 1e3, 4), "second_frame_ms": round(t_second * 1e3, 4), "scene_create_ms": round(t_create * 1e3, 3),
                           "frame_crc_ok": ((zlib.crc32(buf.cpu().numpy().tobytes()) & 0xFFFFFFFF) == g["frame_crc32"]) if g else None,
                           "note": "host wall time, call to completion, of a fresh Scene's first 1920x1080 frame (tile table, dispatch orders and the "
                                   "scene's cost map are made here) and of the one after it; never `value`"}
        except Exception as ex:
            first_frame = {"error": repr(ex)}
    latency = None
    if (world > 1 or args.force_collective) and args.workload == "1080p" and args.traversal == "skip":
        # ONE frame end to end, nothing batched, nothing pipelined: render -> gather -> blit, synchronised -- what a caller who wants
        # THIS frame waits for.  And the host cost of a collective call (what batching several frames per gather amortises).
        fs1 = FrameSharder(scene_of(8), rta.RenderOptions(1920, 1080, 1), rank, world, local, head_trav, force_collective=args.force_collective, mode="tiles",
                           frames_per_gather=1)
        for _ in range(5):
            fs1.step()
        barrier()
        lat, host = [], []
        for _ in range(20):
            barrier()
            t0 = time.perf_counter()
            fs1.render_shard(slot=0)
            h0 = time.perf_counter()
            fs1.gather(slot=0, count=1)
            host.append(time.perf_counter() - h0)
            fs1.blit(slot=0, count=1)
            torch.cuda.synchronize()
            tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            lat.append(float(tt.item()) * 1e3)
        latency = {"frame_latency_ms": round(sorted(lat)[len(lat) // 2], 4), "collective_call_host_ms": round(sorted(host)[len(host) // 2] * 1e3, 4)}
    extras = {}
    if world > 1 and not args.no_extras and args.workload == "1080p" and args.traversal == "skip":
        other = "frames" if multi == "tiles" else "tiles"
        e = measure("1080p", head_trav, args.steps, max(2, args.warmup // 2), 3, other)
        extras["weak_frames" if other == "frames" else "strong_tiles"] = e
        extras["config5_tiles"] = measure("config5", head_trav, 3, 1, 3, "tiles")

    def summary(e, wl):
        w, h, k, lv, _ = WORKLOADS[wl]
        fr = e["frames_per_step"]
        rays = (e["primary"] + e["shadow"])
        srt = sorted(e["ms_reps"])
        return {"workload": "%dx%d spp %d %s" % (w, h, k, "100k spheres" if lv == "100k" else "L%d" % lv), "ms_per_step": round(e["ms_per_step"], 4), "ms_per_step_min_max": [round(srt[0], 4), round(srt[-1], 4)],
                "frames_per_step": fr, "value": round(rays / (e["ms_per_step"] * 1e-3) / 1e6, 3), "unit": "Mrays/s",
                "rank0_kernel_ms": round(e["kern_ms"], 4), "frame_crc32": e["crc"], "frame_crc_ok": e["crc_ok"]}

    if rank == 0:
        ms_per_step = m["ms_per_step"]
        rays = m["primary"] + m["shadow"]
        skip_note = ("achieved = 17 flops x (item + bound tests the reference's traversal makes for rank 0's rays, counted by the kernel in "
                     "this run and equal to the CPU path's) / HIP-event duration of k_render_skip in this run.  The records arrive "
                     "through the scalar cache / L2 (the whole scene is < 1 MB): VALU issue binds, not HBM")
        flat_note = ("achieved = the reference's un-fused flops for the tests the flat pipeline covered (8 per primary test with the pre-formed "
                     "terms, 16 per shadow test; queue lengths x pass lengths) / HIP-event duration of its kernels (k_flat_primary_sc + 2 x "
                     "k_flat_shadow_sc + k_resolve_samples).  The items are wave-uniform scalars (s_load -> SGPR-pair operands of packed "
                     "instructions, two rays per lane); a conservative bound of the discriminant (4 / 11 packed FMAs per item and ray pair, "
                     "margin proven and checked exhaustively: rt_debug_flat_filter_check) rejects items, and the reference's individually "
                     "rounded operations run only for the survivors -- so the peak it is priced against is the packed-FMA issue rate")
        if world == 1 and not args.force_collective:
            layout = "1 GPU, buckets rendered straight into the row-major frame"
        elif multi == "frames":
            layout = ("%d GPUs, every GPU renders a whole frame per step (%d frames per step), u8 frames gathered to rank 0 "
                      "over RCCL %d frames per rank at a time, each gather overlapped with the render of the next frames"
                      % (world, world, args.frames_per_gather))
        else:
            layout = ("%d GPUs, the buckets of ONE frame dealt round-robin (tile_id %% N), u8 shards gathered to rank 0 over "
                      "RCCL (the shards of %d consecutive frames per collective) and blitted into the frame there, pipelined across frames: "
                      "two render streams per rank, the gather of a batch under the render of the next" % (world, args.frames_per_gather))
        srt = sorted(m["ms_reps"])
        out = {
            "metric": "Mrays/sec + ms/frame, 1920x1080 20k-sphere scene" if args.workload == "1080p" else
                      "Mrays/sec + ms/frame, %dx%d spp %d %s (non-headline workload %s)" % (width, height, spp, "100k spheres" if level == "100k" else "L%d" % level, args.workload),
            "value": round(rays / (ms_per_step * 1e-3) / 1e6, 3),      # rays of every frame of a step (summed over ranks) / step time
            "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak" if (multi == "frames" and world > 1) else "strong", "vs_baseline": None, "dtype": "f32",
            "data": ("synthetic (100,000 seeded random spheres, tests/scenes.py; hierarchy built by the host)" if level == "100k" else
                     "synthetic (the reference's deterministic default scene: pyramid level %d)" % level),
            "config": {"workload": "%dx%d, %d spheres (%s), spp %d, f32, %s traversal, %d 64x64 buckets per frame; %s"
                                   % (width, height, n_items, "random list, auto hierarchy" if level == "100k" else "pyramid L%d" % level, spp, args.traversal,
                                      -(-width // 64) * -(-height // 64), layout),
                       "width": width, "height": height, "samples_per_pixel": spp, "n_spheres": n_items,
                       "primary_rays": m["primary"], "shadow_rays": m["shadow"], "traversal": args.traversal,
                       "frames_per_step": m["frames_per_step"],
                       "parallelism": ("frames x %d" if (multi == "frames" and world > 1) else "tiles/%d") % world,
                       "rccl_world_size": (dist.get_world_size() if dist is not None else None)},
            "repeats": {"n": len(m["ms_reps"]), "ms_per_step_each": [round(r, 4) for r in m["ms_reps"]], "median": round(ms_per_step, 4),
                        "min": round(srt[0], 4), "max": round(srt[-1], 4), "timed_region_s": round(m["timed_region_s"], 4)},
            "frame_crc32": m["crc"], "frame_crc_ok": m["crc_ok"],
            "frame_crc_source": "tests/golden/oracle_vectors.json:%s.frame_crc32 (frame buffers zeroed after the warm-up; read back after the timed loop)" % golden_name,
            "mprimary_per_s": round(m["primary"] / (ms_per_step * 1e-3) / 1e6, 3),
            "git_head": git_head(), "kernel_src_sha": src_sha,
            "roofline": roofline(m, skip_kernel if args.traversal == "skip" else "k_flat_primary", FLOPS_PER_TEST if args.traversal == "skip" else 8,
                                 skip_note if args.traversal == "skip" else flat_note),
        }
        if flat is not None:
            fst = flat["my_stats"]
            shadow_tests = fst["tests_executed"] - fst["primary"] * n_items       # what the any-hit passes really ran
            ops = fst["primary"] * n_items * 8 + shadow_tests * 16
            pkf = probe.get(("v_pk_fma_f32 (VGPR pairs, independent)", 8))
            pk_cyc = pkf["cycles_per_instruction_per_simd"] if pkf else 4.0
            pk_peak = N_SIMD * LANES * 4 * CLOCK_HZ / pk_cyc       # a packed FMA: 2 rays x 2 flops per lane
            t = flat["kern_ms"] * 1e-3
            fp_flat = from_profiles("k_flat_pipeline", flat["kern_ms"])
            issue = fp_flat.get("instruction_issue") or {}
            survey_ops = FLOPS_PER_TEST * (fst["primary"] + fst["shadow"]) * n_items        # 17 flops x every ray x every item
            out["flat"] = {"ms_per_step": round(flat["ms_per_step"], 4), "value": round(rays / (flat["ms_per_step"] * 1e-3) / 1e6, 3), "unit": "Mrays/s",
                           "frame_crc_ok": flat["crc_ok"],
                           "roofline": {"bound": "valu_issue",
                                        "frac": round(ops / t / pk_peak, 4),
                                        "frac_is": "reference_flops_over_packed_fma_peak: the un-fused flops of the tests the pipeline covered (8 per primary, 16 per "
                                                   "shadow test) over the packed-FMA issue peak.  survey_8d_frac below is a LOGICAL figure (it exceeds 1): SURVEY 8(d)'s "
                                                   "17 flops x rays x items counts operations the conservative filter provably makes unnecessary",
                                        "simd_cycles_per_wave_instruction": issue.get("simd_cycles_per_wave_instruction"),
                                        "probe_cycles_per_packed_instruction": round(pk_cyc, 3),
                                        "survey_8d_frac_logical": round(survey_ops / t / 1e12 / (N_SIMD * LANES * CLOCK_HZ / 2.0 / 1e12), 3),
                                        "reference_flops_over_packed_fma_peak": round(ops / t / pk_peak, 4),
                                        "kernel": "k_flat_primary_sc + k_flat_shadow_sc", "kernel_ms": round(flat["kern_ms"], 4),
                                        "tests_executed": fst["tests_executed"],
                                        "note": "What bounds the scan is instruction issue: the counters (when profiles/ holds them for these sources) give the SIMD cycles "
                                                "the launch leaves per wave-instruction, to be read against the probe's cost of a packed instruction -- equal means the "
                                                "SIMDs issue back to back.  survey_8d_frac_logical = SURVEY 8(d)'s 17 flops x rays x items / time / the nominal un-fused peak: "
                                                "it exceeds 1 because the kernel provably does not perform most of those operations -- a conservative bound of the "
                                                "discriminant (4 - 6 packed FMAs per item and ray pair; margin proven, checked exhaustively by rt_debug_flat_filter_check) "
                                                "rejects items and the reference's individually rounded operations run only for the survivors.  " + flat_note,
                                        "from_profiles": fp_flat}}
        for k, e in extras.items():
            out[k] = summary(e, "config5" if k == "config5_tiles" else "1080p")
            out[k]["scaling"] = "weak" if k == "weak_frames" else "strong"
        if in_flight is not None:
            out["frames_in_flight"] = {
                "streams": in_flight["streams"], "ms_per_step": round(in_flight["ms_per_step"], 4),
                "value": round(rays / (in_flight["ms_per_step"] * 1e-3) / 1e6, 3), "unit": "Mrays/s",
                "ms_per_step_each": [round(v, 4) for v in in_flight["ms_reps"]], "frame_crc_ok": in_flight["crc_ok"],
                "note": "NOT `value`: the same launches dealt round-robin over two HIP streams through the asynchronous entry point (rt_render_frame_device), "
                        "each stream its own frame buffer, host wall time over the steps.  `value` / `ms_per_step` time the launches in order on ONE stream, "
                        "where a frame ends with its few longest waves on an otherwise idle chip and its waves spend half their cycles waiting for node "
                        "records (`roofline.wave_cycles`); a second frame in flight fills both.  The difference is the frame's tail and part of its waiting."}
            ok_in_flight = in_flight["crc_ok"] is not False
        else:
            ok_in_flight = True
        if configs is not None:
            out["configs"] = configs
        if first_frame is not None:
            out["first_frame"] = first_frame
            if "first_frame_ms" in first_frame:
                out["first_frame_ms"] = first_frame["first_frame_ms"]
        if latency is not None:
            out.update(latency)
            out["frames_per_gather"] = args.frames_per_gather
            out["n_gt_1_note"] = ("ms_per_step / value are a THROUGHPUT figure for N > 1: consecutive frames are software-pipelined (two render streams per rank, the "
                                  "gather of a batch under the render of the next) and %d frames share one collective.  frame_latency_ms is ONE frame, render -> "
                                  "gather -> blit, nothing batched or pipelined (max over ranks, median of 20); collective_call_host_ms is what one "
                                  "torch.distributed gather call costs this rank's host thread." % args.frames_per_gather)
            try:
                exp = json.load(open(os.path.join(ROOT, "profiles", "expected_shard_render.json")))
                out["expected_shard_render_us"] = {"n_%d" % world: (exp.get("1080p") or {}).get(str(world)), "source": exp.get("source")}
            except Exception:
                pass
        if args.workload == "1080p" and args.traversal == "skip":
            try:
                # what the builder's one-GPU measurements add up to for ONE frame of BASELINE config 4 at N = 2 / 4 / 8 (rank 0's shard render + one
                # collective call + the root's blit) beside the N = 1 frame: written down before any curve exists (tools/shard_expect.py)
                lat = json.load(open(os.path.join(ROOT, "profiles", "expected_frame_latency.json")))
                n = lat.get("n") or {}
                out["expected_frame_latency_us"] = {"n_1": (n.get("1") or {}).get("frame_us"), "source": lat.get("source"), "kernel_src_sha": lat.get("kernel_src_sha"),
                                                    "note": "ONE frame, nothing pipelined: at every N the gather + blit cost more than the shards save -- N GPUs buy throughput "
                                                            "(ms_per_step at N > 1 pipelines frames) and larger frames, not the latency of a 1080p spp-1 frame"}
                for k in ("2", "4", "8"):
                    if k in n:
                        out["expected_frame_latency_us"]["n_" + k] = n[k]
            except Exception:
                pass
        if seam is not None:
            out["seam"] = seam
        if image_wall is not None:
            out["make_image"] = image_wall
            if "make_image_wall_ms" in image_wall:
                out["make_image_wall_ms"] = image_wall["make_image_wall_ms"]
        out["library"] = {"path": os.path.relpath(rta.capi.LIB_PATH, ROOT), "build": rta.capi.build_info(), "test_hooks": rta.capi.HAVE_TEST_HOOKS}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(width, height, spp, level)
        ok = m["crc_ok"] is not False and (flat is None or flat["crc_ok"] is not False) and all(e["crc_ok"] is not False for e in extras.values()) and ok_in_flight
        ok = ok and all(c.get("frame_crc_ok") is not False for c in (configs or {}).values()) and (first_frame or {}).get("frame_crc_ok") is not False
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world > 1 and not args.no_native_gang and real_rccl and args.workload == "1080p" and args.traversal == "skip":
            # The other ranks are on their way out and their GPUs are free: ONE process over all N devices (rt_gang_*: ncclCommInitAll, one
            # ncclGather per frame, no torch.distributed call on the path).  A child process with a time limit: whatever happens to it, the
            # headline above stands.
            out["native_gang"] = native_gang(world, golden_case(golden_name))
        os.write(json_fd, (json.dumps(out) + "\n").encode())
        if not ok:
            sys.stderr.write("bench.py: the frame left by the timed launches does not match the committed oracle vector\n")
            raise SystemExit(3)


if __name__ == "__main__":
    main()
