"""TEST INFRASTRUCTURE (moved out of rust-tracer_amd/dist.py in round 6): a FrameSharder whose one collective goes through pinned host memory and a
CPU (gloo) gather.  RCCL refuses two ranks on one device; with this sharder two ranks can execute, on ONE GPU and on real kernels, everything
else an RCCL job runs: the sharding, the render / gather / blit ordering of run_pipeline on a sender and on the root, partial batches, buffer
reuse (tests/test_gpu_dist.py starts bench.py with `--sharder tests.host_staged:HostStagedFrameSharder --process-group-backend gloo`)."""
from rust_tracer_amd.dist import FrameSharder


class _HostStagedWork:
    """What HostStagedFrameSharder.gather returns: wait() has the semantics of a device collective's Work.wait() -- the current stream is ordered
    behind the gathered data (rank 0: the gathered CPU rows are copied into the device buffer on it)."""

    def __init__(self, sharder, work, slot, nbytes):
        self.sharder, self.work, self.slot, self.nbytes = sharder, work, slot, nbytes

    def wait(self):
        fs = self.sharder
        if self.work is not None:
            self.work.wait()                                   # gloo: blocks the host until this rank's part of the gather is done
        if fs.rank == 0:
            for r in range(fs.world):                          # pinned -> device, on the current stream (the blit that follows is behind it)
                fs.gathered[self.slot][r][:self.nbytes].copy_(fs.h_gathered[self.slot][r][:self.nbytes], non_blocking=True)
            fs._h2d_done[self.slot] = fs.torch.cuda.Event()
            fs._h2d_done[self.slot].record()
        return True


class HostStagedFrameSharder(FrameSharder):
    comm_device = "cpu"                                        # where bench.py keeps the few scalars the ranks exchange

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        torch = self.torch
        if self.collective:
            self.h_shards = [torch.zeros(self.batch * self.unit_bytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
            self._h2d_done = [None, None]
            if self.rank == 0:
                self.h_gathered = [[torch.zeros(self.batch * self.unit_bytes, dtype=torch.uint8).pin_memory() for _ in range(self.world)] for _ in range(2)]

    def gather(self, slot=0, async_op=False, count=None):
        import torch.distributed as dist
        if not self.collective:
            return super().gather(slot, async_op, count)
        count = self.batch if count is None else count
        nb = count * self.unit_bytes
        if self.rank == 0 and self._h2d_done[slot] is not None:
            self._h2d_done[slot].synchronize()              # the copy out of h_gathered[slot] of the slot's previous gather
        self.h_shards[slot][:nb].copy_(self.shards[slot][:nb], non_blocking=True)
        self.torch.cuda.current_stream(self.device).synchronize()      # the shard is rendered and in host memory
        work = dist.gather(self.h_shards[slot][:nb], [g[:nb] for g in self.h_gathered[slot]] if self.rank == 0 else None, dst=0, async_op=async_op)
        staged = _HostStagedWork(self, work if async_op else None, slot, nb)
        if async_op:
            return staged
        staged.wait()
        return None
