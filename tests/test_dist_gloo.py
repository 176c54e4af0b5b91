"""The N > 1 path on CPU: world_size-2 gloo processes exercise the same shard layout, the same gather call and the
same frame assembly the GPU path uses (rust-tracer_amd/dist.py); the per-rank tile bytes come from the oracle here
because there is no GPU -- this tests the sharding / gather / blit logic, not the kernels."""
import os
import socket
import sys

import numpy as np
import pytest

import rust_tracer_amd as rta
from rust_tracer_amd import dist as rdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("size", [(1920, 1080), (800, 600), (1024, 768), (64, 64)])
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_shard_layout_partitions_the_bucket_list(size, world):
    o = rta.RenderOptions(size[0], size[1], 1)
    bl, per_rank, shard_px = rdist.shard_layout(o, world)
    seen = sorted(i for idx, _, _ in per_rank for i in idx)
    assert seen == list(range(len(bl)))                            # every bucket exactly once
    for r, (idx, offs, px) in enumerate(per_rank):
        assert idx == list(range(r, len(bl), world))               # round-robin tile_id % world
        assert px <= shard_px and px == sum(bl[i].area() for i in idx)
        assert offs == [sum(bl[j].area() for j in idx[:k]) for k in range(len(idx))]
    regions, offsets, spx = rdist.gathered_tile_table(o, world)
    assert spx == shard_px and len(regions) == len(bl)
    assert sum((r - l) * (t - b) for (l, t, r, b) in regions) == size[0] * size[1]


def test_1080p_shard_sizes_match_the_survey():
    # 510 tiles, 8 GPUs -> 64 tiles/GPU padded, ~1 MiB per shard
    _, per_rank, shard_px = rdist.shard_layout(rta.RenderOptions(1920, 1080, 1), 8)
    assert max(len(idx) for idx, _, _ in per_rank) == 64
    assert shard_px * 4 <= 64 * 64 * 64 * 4


def _worker(rank, world, port, w, h, spp, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import oracle
    import rust_tracer_amd as rta
    from rust_tracer_amd import dist as rdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        opts = rta.RenderOptions(w, h, spp)
        bl, per_rank, shard_px = rdist.shard_layout(opts, world)
        idx, offs, px = per_rank[rank]
        o = oracle.Scene.default()
        shard = np.zeros(shard_px * 4, dtype=np.uint8)               # padded to equal length like the GPU shard
        for i, off in zip(idx, offs):
            tile, _ = o.render_region(w, h, spp, *bl[i])
            shard[off * 4:off * 4 + tile.size] = tile.reshape(-1)
        t = torch.from_numpy(shard)
        gl = [torch.zeros_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, gl, dst=0)                                    # the same collective FrameSharder.finish issues
        if rank == 0:
            frame = rdist.assemble_host(opts, world, torch.stack(gl).numpy())
            ref, _, _ = o.render(w, h, spp, nthreads=2)
            q.put(bool(np.array_equal(frame, ref)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


class _CpuOps:
    """The five operations run_pipeline() schedules, on CPU tensors over gloo.  Every frame k is rendered with a
    different pattern (the oracle tile XOR k), so a slot mix-up in the schedule shows up as a wrong frame."""

    def __init__(self, rank, world, opts, oracle_scene, log):
        import torch
        self.torch, self.rank, self.world, self.opts, self.o, self.log = torch, rank, world, opts, oracle_scene, log
        self.bl, self.per_rank, self.shard_px = rdist.shard_layout(opts, world)
        self.shards = [torch.zeros(self.shard_px * 4, dtype=torch.uint8) for _ in range(2)]
        self.gathered = [[torch.zeros(self.shard_px * 4, dtype=torch.uint8) for _ in range(world)] for _ in range(2)]
        self.frames = []
        self.k_render = 0
        self.slot_frame = [None, None]
        self.tiles = {}

    def op_render(self, slot):
        k = self.k_render
        self.k_render += 1
        self.slot_frame[slot] = k
        idx, offs, _ = self.per_rank[self.rank]
        sh = np.zeros(self.shard_px * 4, dtype=np.uint8)
        for i, off in zip(idx, offs):
            if i not in self.tiles:
                self.tiles[i] = self.o.render_region(self.opts.width, self.opts.height, self.opts.samples_per_pixel, *self.bl[i])[0].reshape(-1)
            sh[off * 4:off * 4 + self.tiles[i].size] = self.tiles[i] ^ np.uint8(k & 0xFF)
        self.shards[slot].copy_(self.torch.from_numpy(sh))
        self.log.append(("render", k, slot))

    def op_gather_async(self, slot):
        import torch.distributed as dist
        self.log.append(("gather", self.slot_frame[slot], slot))
        return dist.gather(self.shards[slot], self.gathered[slot] if self.rank == 0 else None, dst=0, async_op=True)

    def op_blit_after(self, work, slot):
        work.wait()
        if self.rank == 0:
            k = self.slot_frame[slot] if False else None
            self.frames.append(rdist.assemble_host(self.opts, self.world, self.torch.stack(self.gathered[slot]).numpy()))
        self.log.append(("blit", slot))

    def op_before_reuse(self, slot):
        pass

    def op_drain(self):
        self.log.append(("drain",))


def _pipeline_worker(rank, world, port, steps, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import oracle
    import rust_tracer_amd as rta
    from rust_tracer_amd import dist as rdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        opts = rta.RenderOptions(256, 192, 1)
        o = oracle.Scene.default()
        log = []
        ops = _CpuOps(rank, world, opts, o, log)
        rdist.run_pipeline(steps, ops)
        ok = True
        if rank == 0:
            ref, _, _ = o.render(256, 192, 1, nthreads=2)
            ok = len(ops.frames) == steps and all(np.array_equal(f, ref ^ np.uint8(k & 0xFF)) for k, f in enumerate(ops.frames))
        n_render = sum(1 for e in log if e[0] == "render")
        n_blit = sum(1 for e in log if e[0] == "blit")
        q.put((rank, bool(ok), n_render, n_blit, log[-1] == ("drain",)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("steps", [1, 2, 5])
def test_world_size_2_pipelined_schedule_delivers_every_frame_in_order(steps):
    # the schedule bench.py runs for N > 1 (FrameSharder.run -> run_pipeline), with CPU stand-ins for the device operations:
    # both ranks execute it, frames differ from each other, rank 0 must assemble frame k from frame k's shards
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, steps, q)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(180) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    res = sorted(q.get(timeout=5) for _ in range(2))
    for rank, ok, n_render, n_blit, drained in res:
        assert ok and n_render == steps and n_blit == steps and drained, (rank, ok, n_render, n_blit, drained)


@pytest.mark.parametrize("size", [(320, 200, 1), (192, 128, 2)])
def test_world_size_2_gather_assembles_the_identical_frame(size):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, size[0], size[1], size[2], q)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert q.get(timeout=5) is True
