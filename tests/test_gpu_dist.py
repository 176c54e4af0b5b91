"""The multi-GPU code path on ONE GPU: a one-rank RCCL process group drives exactly what `bench.py --gpus N` runs
(double-buffered shards, asynchronous gather overlapped with the next render, device blit on rank 0)."""
import os
import socket

import numpy as np
import pytest

import rust_tracer_amd as rta
from rust_tracer_amd.dist import FrameSharder
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def one_rank_rccl():
    import torch
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("size", [(1920, 1080, 1), (800, 600, 2)])
def test_pipelined_gather_path_produces_the_oracle_frame(one_rank_rccl, size):
    import torch
    w, h, spp = size
    s, o = util.scene_pair_default()
    ref, _, _ = o.render(w, h, spp, nthreads=os.cpu_count() or 1)
    fs = FrameSharder(s, (w, h, spp), 0, 1, 0, rta.RT_TRAVERSAL_SKIP, force_collective=True)
    for sh in fs.shards:
        sh.fill_(0xAB)                      # poison: every byte the blit reads must have been rendered + gathered
    for g in fs.gathered_flat:
        g.fill_(0xCD)
    fs.step()
    np.testing.assert_array_equal(fs.frame_host(), ref)
    for steps in (1, 2, 5):
        fs.frame.zero_()
        torch.cuda.synchronize()
        fs.run(steps)
        np.testing.assert_array_equal(fs.frame_host(), ref)


def test_fused_single_gpu_path_equals_collective_path(one_rank_rccl):
    s, _ = util.scene_pair_default()
    a = FrameSharder(s, (1024, 768, 1), 0, 1, 0, rta.RT_TRAVERSAL_SKIP)
    b = FrameSharder(s, (1024, 768, 1), 0, 1, 0, rta.RT_TRAVERSAL_SKIP, force_collective=True)
    a.run(3)
    b.run(3)
    np.testing.assert_array_equal(a.frame_host(), b.frame_host())


def test_frames_mode_gathers_whole_frames(one_rank_rccl):
    # bench.py's N > 1 default (weak scaling): every rank renders a whole frame straight into row-major order and the
    # finished frames are gathered to rank 0 -- here with one rank, both pipeline slots
    import torch
    s, o = util.scene_pair_default()
    ref, _, _ = o.render(1920, 1080, 1, nthreads=os.cpu_count() or 1)
    fs = FrameSharder(s, (1920, 1080, 1), 0, 1, 0, rta.RT_TRAVERSAL_SKIP, force_collective=True, mode="frames")
    for g in fs.gathered_flat:
        g.fill_(0xCD)
    fs.run(5)
    for slot in (0, 1):
        np.testing.assert_array_equal(fs.frame_host(slot=slot, of_rank=0), ref)
    st = fs.render_shard(want_stats=True)
    assert st["primary"] == 1920 * 1080


def test_frames_mode_batched_gather_delivers_every_frame(one_rank_rccl):
    # bench.py gathers several frames per collective for N > 1: 7 frames in batches of 3 (3 + 3 + 1, the last one a partial
    # gather) -- every frame slot that was gathered must hold the oracle's frame, the untouched tail of the last batch's
    # buffer must still hold the poison
    import torch
    s, o = util.scene_pair_default()
    ref, _, _ = o.render(640, 360, 1, nthreads=os.cpu_count() or 1)
    fs = FrameSharder(s, (640, 360, 1), 0, 1, 0, rta.RT_TRAVERSAL_SKIP, force_collective=True, mode="frames", frames_per_gather=3)
    for g in fs.gathered_flat:
        g.fill_(0xCD)
    fs.run(7)
    torch.cuda.synchronize()
    # batches land in slots 0, 1, 0: slot 1 holds batch 2 (3 frames), slot 0 holds the final partial batch (1 frame) over
    # batch 1's frames 2 and 3
    for index in range(3):
        np.testing.assert_array_equal(fs.frame_host(slot=1, of_rank=0, index=index), ref)
        np.testing.assert_array_equal(fs.frame_host(slot=0, of_rank=0, index=index), ref)
    fs2 = FrameSharder(s, (640, 360, 1), 0, 1, 0, rta.RT_TRAVERSAL_SKIP, force_collective=True, mode="frames", frames_per_gather=4)
    for g in fs2.gathered_flat:
        g.fill_(0xCD)
    fs2.run(1)                                   # one partial batch: frames 1..3 of slot 0 were never gathered
    np.testing.assert_array_equal(fs2.frame_host(slot=0, index=0), ref)
    assert int(fs2.frame_host(slot=0, index=1).min()) == 0xCD and int(fs2.frame_host(slot=0, index=3).max()) == 0xCD


def test_tiles_mode_batched_gather_blits_every_frame_of_a_batch(one_rank_rccl):
    # "tiles" (BASELINE config 4) with several frames' shards per collective, as bench.py runs it for N > 1: 7 frames in batches of
    # 3 (3 + 3 + 1, the last one a partial gather), two render streams.  The frame on rank 0 must be the oracle's after the run, a
    # single step() must work with the batched buffers too, and every gathered slot must hold valid shards where it was written.
    import torch
    s, o = util.scene_pair_default()
    ref, _, _ = o.render(640, 360, 1, nthreads=os.cpu_count() or 1)
    fs = FrameSharder(s, (640, 360, 1), 0, 1, 0, rta.RT_TRAVERSAL_SKIP, force_collective=True, mode="tiles", frames_per_gather=3)
    for g in fs.gathered_flat:
        g.fill_(0xCD)
    fs.frame.fill_(0xEE)
    fs.run(7)
    np.testing.assert_array_equal(fs.frame_host(), ref)
    flat = [g.cpu().numpy() for g in fs.gathered_flat]
    unit = fs.unit_bytes
    for j in range(3):                                            # slot 1: batch 2, three shards; slot 0: batch 1 then the partial batch 3
        assert np.array_equal(flat[1][j * unit:(j + 1) * unit], flat[1][:unit])
        assert np.array_equal(flat[0][j * unit:(j + 1) * unit], flat[1][:unit])
    fs.frame.fill_(0xEE)
    fs.step()
    np.testing.assert_array_equal(fs.frame_host(), ref)
    fs.frame.fill_(0xEE)
    fs.run(2)                                                     # one partial batch
    np.testing.assert_array_equal(fs.frame_host(), ref)


@pytest.mark.parametrize("ndev", [1, 2, 4, 8])
@pytest.mark.parametrize("size", [(1920, 1080, 1), (200, 130, 2)])
def test_native_gang_rccl_gather_is_byte_identical(ndev, size):
    # rt_gang_*: one process, N ranks, bucket i -> rank i % N, one ncclGather to the root, blit, frame to the host.  Byte-identical
    # to the oracle for N in {1, 2, 4, 8} (8 = BASELINE config 4's own rank count: 510 buckets -> 64 / 63 per rank, the short shards padded), counters summed over the ranks equal the CPU path's.  With fewer GPUs than ranks the ranks share
    # GPU 0 and the gather goes through the stand-in for librccl.so (tests/c/fake_rccl.cpp, rt_debug_rccl_library): the N > 1 code --
    # sharding, equal-length padded shards, the grouped gather, the blit of the device-major gathered buffer -- runs either way.
    import contextlib
    import oracle
    w, h, spp = size
    s = rta.Scene.default(8 if w > 1000 else 6)
    o = oracle.Scene.default(level=8 if w > 1000 else 6)
    shared = rta.device_count() < ndev
    with (rta.capi.rccl_stand_in() if shared else contextlib.nullcontext()):
        g = rta.Gang(s, [0] * ndev if shared else list(range(ndev)))
    assert g.size() == ndev
    regs = [tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, spp))]
    ref, rst, _ = o.render(w, h, spp, os.cpu_count() or 1, oracle.MODE_HIERARCHY | oracle.MODE_ANYHIT_EXIT)
    # poison: every byte the caller gets must have been rendered, gathered and blitted by this call
    out = np.full(w * h * 4, 0xAB, dtype=np.uint8)
    frame, st = g.render_frame((w, h, spp), regs, want_stats=True, out=out)
    np.testing.assert_array_equal(frame, ref)
    for k in ("primary", "hits", "shadow", "occluded", "sphere_tests", "bound_tests"):
        assert st[k] == rst[k], k
    frame2, _ = g.render_frame((w, h, spp), regs[::-1])          # the deal follows the list order; the frame does not care
    np.testing.assert_array_equal(frame2, ref)
    if ndev > 1:
        # an odd number of buckets: the shards differ in length and the short ones travel padded
        odd = regs[:len(regs) - (1 if len(regs) % ndev == 0 else 0)]
        part, _ = g.render_frame((w, h, spp), odd)
        for (l, t, r, b) in odd:
            np.testing.assert_array_equal(part[b:t, l:r], ref[b:t, l:r])
    g.close()


def test_the_stand_in_library_is_only_for_gangs_created_under_it():
    # outside rt_debug_rccl_library a device may not be listed twice (RCCL wants one GPU per rank)
    s = rta.Scene.default(5)
    with pytest.raises(rta.RtError) as e:
        rta.Gang(s, [0, 0])
    assert e.value.status == rta.capi.RT_ERR_INVALID_ARGUMENT


@pytest.mark.parametrize("multi", ["tiles", "frames"])
def test_bench_collective_path_checks_its_own_frame(multi):
    # bench.py is what the driver runs at N = 1, 2, 4, 8: here the N > 1 code (shards, RCCL gather, blit, software pipeline) runs as
    # a one-rank job in a child process, and the line it prints must carry a frame CRC equal to the committed oracle vector -- the
    # same self-check every N > 1 run performs on rank 0.
    import json
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env.pop("RTRACE_HIP_LIBRARY", None)                            # bench.py runs the product library
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-collective", "--multi", multi, "--steps", "4",
                        "--warmup", "2", "--repeats", "2", "--min-timed-region", "0", "--no-cpu-baseline", "--no-flat", "--no-seam"], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line"
    d = json.loads(lines[0])
    assert d["frame_crc_ok"] is True and d["frame_crc32"] == 4168428064
    assert d["config"]["rccl_world_size"] == 1 and d["n_gpus"] == 1
    assert d["scaling"] == "strong" and d["roofline"]["bound"] == "valu_issue" and 0 < d["roofline"]["frac"] <= 1
    assert d["repeats"]["n"] == 2 and d["value"] > 1000
    assert d["library"]["test_hooks"] is False and d["library"]["path"].endswith("librtrace_hip.so")


@pytest.mark.parametrize("multi,extra", [("tiles", []), ("frames", ["--no-extras"])])
def test_two_real_bench_ranks_share_gpu_0(multi, extra):
    # First contact for the code the driver's scaling run executes: TWO bench.py processes (RANK 0 / 1, WORLD_SIZE 2), fresh children, both on
    # GPU 0.  RCCL refuses two ranks on one device, so the collective is the test-only host-staged one (tests/host_staged.py: shard ->
    # pinned host memory -> CPU gather -> rank 0's device buffer); everything else is what an RCCL job runs on real kernels: rank 1's
    # render / gather / buffer-reuse ordering (dist.py op_*), partial batches (7 steps in batches of 3 = 3 + 3 + 1), rank 0's blits of
    # two ranks' shards, the N > 1 keys of the JSON line (frame_latency_ms ...), weak_frames and config5_tiles riding along.  Rank 0's
    # frame CRC must be the committed oracle vector's (/root/reference/src/rust/render.rs:273-307 is what the deal + gather replaces).
    import json
    import subprocess
    import sys
    import rust_tracer_amd  # noqa: F401
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for rank in (0, 1):
        env = util.product_env(RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--process-group-backend", "gloo", "--sharder", "tests.host_staged:HostStagedFrameSharder", "--multi", multi,
                                       "--frames-per-gather", "3", "--steps", "7", "--warmup", "3", "--repeats", "2", "--min-timed-region", "0",
                                       "--no-cpu-baseline"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rank, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d: %s" % (rank, se[-3000:])
    assert outs[1][0].strip() == "", "only rank 0 prints the line"
    lines = [l for l in outs[0][0].splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_world_size"] == 2 and d["steps"] == 7
    assert d["frame_crc_ok"] is True and d["value"] > 100 and d["library"]["test_hooks"] is False
    if multi == "tiles":
        assert d["frame_crc32"] == 4168428064 and d["scaling"] == "strong" and d["config"]["parallelism"] == "tiles/2"
        assert d["frame_latency_ms"] > 0 and d["frames_per_gather"] == 3
        assert d["weak_frames"]["frame_crc_ok"] is True and d["weak_frames"]["frames_per_step"] == 2
        assert d["config5_tiles"]["frame_crc_ok"] is True
        assert d["config"]["primary_rays"] == 2073600 and d["config"]["shadow_rays"] == 1337403      # summed over the two ranks
    else:
        assert d["scaling"] == "weak" and d["config"]["frames_per_step"] == 2 and d["config"]["primary_rays"] == 2 * 2073600


@pytest.mark.parametrize("ndev", [1, 8])
def test_native_gang_bench(ndev):
    # bench.py's `native_gang` side key (rust-tracer_amd/gang_bench: ONE process, rt_gang_render_frames over N devices, no torch.distributed
    # on the path).  N = 1: the product binary on a one-rank RCCL communicator.  N = 8, BASELINE config 4's own rank count: the -DRT_TEST_HOOKS
    # twin with all eight ranks on GPU 0 through the stand-in for librccl.so -- 510 buckets -> 64 / 63 per rank, padded shards, pipelined.
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if ndev == 1:
        cmd = [os.path.join(root, "rust-tracer_amd", "gang_bench"), "--devices", "1", "--frames", "30"]
    else:
        cmd = [os.path.join(root, "tests", "c", "gang_bench_test"), "--devices", str(ndev), "--frames", "30", "--rccl-stand-in", rta.capi.FAKE_RCCL]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["devices"] == ndev and d["frame_crc32"] == 4168428064 and d["both_buffers_equal"] is True
    assert d["primary"] == 2073600 and d["shadow"] == 1337403 and d["ms_per_frame"] > 0 and d["frame_latency_ms"] > 0
