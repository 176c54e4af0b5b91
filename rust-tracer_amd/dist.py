"""Tile sharding across the GPUs of one node and frame assembly.

The reference's only parallelism is "independent 64x64 buckets on a thread pool" joined by a bounded channel
(render.rs:264-307).  Here the same buckets are dealt round-robin over one process per GPU; each rank renders
its shard tile-major into device memory and a single RCCL gather of the u8 shards over xGMI brings them to
rank 0, which blits them into the row-major frame on the device.  No other collective touches the data path.

Two ways to use N GPUs, both with the gather as the only collective:
  * mode "tiles" (BASELINE config 4, strong scaling; bench.py's headline for N > 1): the buckets of ONE frame are dealt round-robin
    over the ranks, the tile-major shards are gathered and blitted into the frame on rank 0.  Byte-identical for every N.  A 1080p
    frame takes ~0.046 ms on one GPU and a rank's shard of it is still a few tens of microseconds (profiles/: 24 us at N = 8 with
    the lane-cooperative walk of its heaviest quads), less than one torch.distributed gather call costs the host (~0.03 ms) -- so
    `frames_per_gather` shards share a collective and consecutive frames are pipelined; bench.py also prints the un-batched,
    un-pipelined latency of one frame.
  * mode "frames" (weak scaling, rides along as `weak_frames`): every GPU renders whole frames -- all 510 buckets of its own frame
    per step, N frames per step in total -- and the finished u8 frames are gathered to rank 0, where the writer lives.

The sharding arithmetic is plain Python (testable on CPU with gloo); only `FrameSharder.step` touches the GPU.

Load order: PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as the system one librtrace_hip.so links),
so a process that uses both must `import torch` BEFORE `import rust_tracer_amd` (bench.py and tests/conftest.py do)."""
import numpy as np

from . import capi
from .render import buckets, RenderOptions


def shard_indices(n_tiles, rank, world):
    """Round-robin `tile_id % world == rank` in the scheduler's row-major bucket order (render.rs:273-298)."""
    return list(range(rank, n_tiles, world))


def shard_layout(options, world):
    """For every rank: (tile indices, px offset of each tile inside the rank's shard, shard px count)."""
    bl = buckets(options)
    per_rank = []
    for r in range(world):
        idx = shard_indices(len(bl), r, world)
        offs, px = [], 0
        for i in idx:
            offs.append(px)
            px += bl[i].area()
        per_rank.append((idx, offs, px))
    shard_px = max(p for _, _, p in per_rank)          # shards are padded to equal length for the gather
    return bl, per_rank, shard_px


def gathered_tile_table(options, world):
    """Tile list and source pixel offsets for blitting the gathered [world, shard_px] buffer into the frame."""
    bl, per_rank, shard_px = shard_layout(options, world)
    regions, offsets = [], []
    for r, (idx, offs, _) in enumerate(per_rank):
        for i, o in zip(idx, offs):
            regions.append(tuple(bl[i]))
            offsets.append(r * shard_px + o)
    return regions, np.asarray(offsets, dtype=np.uint32), shard_px


def assemble_host(options, world, gathered):
    """CPU model of the root's blit (used by the gloo tests): gathered uint8[world, shard_px*4] -> frame[h, w, 4]."""
    regions, offsets, shard_px = gathered_tile_table(options, world)
    flat = np.asarray(gathered, dtype=np.uint8).reshape(-1)
    frame = np.zeros((options.height, options.width, 4), dtype=np.uint8)
    for (l, t, r, b), o in zip(regions, offsets):
        n = (r - l) * (t - b) * 4
        frame[b:t, l:r] = flat[int(o) * 4:int(o) * 4 + n].reshape(t - b, r - l, 4)
    return frame


def run_pipeline(steps, ops):
    """The frame schedule of the multi-GPU path, double-buffered: render(k) -> async gather(k) -> [gather(k-1) done: blit(k-1)].
    Frame k uses buffer slot k & 1; a slot is only re-rendered after the gather that read it has completed, and only
    re-gathered after the blit that read its gathered copy has been ordered (op_before_reuse)."""
    works = [None, None]

    def finish(slot):
        w, works[slot] = works[slot], None
        ops.op_blit_after(w, slot)

    for k in range(steps):
        slot = k & 1
        if works[slot] is not None:              # frame k-2 used this slot and nobody has waited for its gather yet
            finish(slot)
        ops.op_before_reuse(slot)
        ops.op_render(slot)
        works[slot] = ops.op_gather_async(slot)
        prev = slot ^ 1
        if k > 0 and works[prev] is not None:
            finish(prev)
    for slot in ((steps - 2) & 1, (steps - 1) & 1):
        if steps > 0 and works[slot] is not None:
            finish(slot)
    ops.op_drain()


class FrameSharder:
    """One per process (= per GPU).  world == 1: a step renders the buckets straight into the row-major frame.
    world > 1: a step renders this rank's buckets tile-major, one RCCL gather brings the u8 shards to rank 0, and rank 0
    blits them into the frame.  run() pipelines consecutive frames: the gather of frame k (RCCL's own stream) overlaps
    the render of frame k+1, so a sequence of frames costs max(render, gather + blit) per frame instead of their sum."""

    def __init__(self, scene, options, rank=0, world=1, device=0, traversal=capi.RT_TRAVERSAL_SKIP, force_collective=False,
                 mode="tiles", frames_per_gather=1):
        import torch
        self.torch = torch
        if mode not in ("tiles", "frames"):
            raise ValueError("mode must be 'tiles' or 'frames'")
        self.mode = mode
        # both modes may batch: a rank renders `batch` consecutive frames (their shards, in "tiles" mode) into one buffer and ONE gather
        # moves them all (fewer, larger collectives: the per-call host cost of a gather -- some 30 us through torch.distributed, more
        # than a 1080p shard takes to render -- is paid once per batch, the wire time is the same)
        self.batch = max(1, int(frames_per_gather))
        self._counts, self._next, self._slot_count = [], 0, [self.batch, self.batch]
        # force_collective: take the shard -> gather -> blit path even for world == 1 (a one-rank RCCL gather); lets a
        # single-GPU test drive exactly the code the 8-GPU run executes
        self.collective = world > 1 or force_collective
        self.options = RenderOptions(*options)
        self.rank, self.world, self.device = rank, world, device
        self.traversal = traversal
        self.dev = scene.device(device)
        # "frames": this rank owns a whole frame (all buckets, laid out like a 1-rank shard = the row-major frame itself)
        bl, per_rank, self.shard_px = shard_layout(self.options, world if mode == "tiles" else 1)
        if mode == "frames":
            self.shard_px = self.options.width * self.options.height
        self.my_regions = [tuple(bl[i]) for i in per_rank[rank if mode == "tiles" else 0][0]]
        self.my_regions_c = self.dev._regions(self.my_regions)
        tdev = torch.device("cuda", device)
        self.unit_bytes = self.shard_px * 4                      # one frame ("frames") / one shard ("tiles")
        # two shard buffers: frame k+1 is rendered while frame k's shard is still being gathered
        self.shards = [torch.zeros(self.batch * self.unit_bytes, dtype=torch.uint8, device=tdev) for _ in range(2)]
        self.shard = self.shards[0]
        self.frame = None
        self.side = torch.cuda.Stream(device=tdev) if (rank == 0 and self.collective) else None   # rank 0's blits run here
        # one render stream per shard buffer: frame k + 1 is rendered on the other stream while frame k's few longest waves are
        # still finishing (a frame -- and a shard of it -- ends with its heaviest pixels' chains on an otherwise idle chip,
        # DESIGN.md 4.1), and the gather of a slot is ordered behind that slot's stream only
        self.render_streams = [torch.cuda.Stream(device=tdev) for _ in range(2)] if self.collective else None
        if rank == 0:
            self.frame = torch.zeros(self.options.height * self.options.width * 4, dtype=torch.uint8, device=tdev)
            regions, offsets, _ = gathered_tile_table(self.options, world if mode == "tiles" else 1)
            self.all_regions_c = self.dev._regions(regions)
            self.all_offsets = offsets
            # "tiles" with a batch: the gathered buffer is [rank][frame of the batch][shard], so frame j's tiles of rank r start
            # (r * batch + j) shards in instead of r
            per_rank = shard_layout(self.options, world if mode == "tiles" else 1)[1]
            rank_of = np.concatenate([np.full(len(idx), r, dtype=np.int64) for r, (idx, _, _) in enumerate(per_rank)])
            self.batch_offsets = [(offsets.astype(np.int64) + (rank_of * (self.batch - 1) + j) * self.shard_px).astype(np.uint32)
                                  for j in range(self.batch)]
            if self.collective:
                # one buffer per in-flight frame for the blit; the gather lists are views of its rows
                self.gathered_flat = [torch.zeros(world * self.batch * self.unit_bytes, dtype=torch.uint8, device=tdev) for _ in range(2)]
                self.gathered = [list(g.view(world, self.batch * self.unit_bytes).unbind(0)) for g in self.gathered_flat]

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def render_shard(self, want_stats=False, slot=0, index=0):
        """This rank's buckets into shard buffer `slot` (enqueued on torch's current stream): tile-major in "tiles" mode,
        straight into row-major frame order in "frames" mode (the shard IS this rank's frame; `index` picks the frame of
        the batch)."""
        if self.mode == "frames":
            return self.dev.render_frame_device(tuple(self.options), self.my_regions_c,
                                                self.shards[slot].data_ptr() + index * self.unit_bytes, self._stream(),
                                                self.traversal, want_stats)
        return self.dev.render_tiles_device(tuple(self.options), self.my_regions_c, self.shards[slot].data_ptr() + index * self.unit_bytes,
                                            self._stream(), self.traversal, want_stats)

    def render_frame(self, want_stats=False):
        """world == 1: the buckets straight into the row-major frame (rt_render_frame_device = render + blit fused)."""
        return self.dev.render_frame_device(tuple(self.options), self.my_regions_c, self.frame.data_ptr(), self._stream(),
                                            self.traversal, want_stats)

    def gather(self, slot=0, async_op=False, count=None):
        """The one collective on the data path: equal-length u8 shards to rank 0 over RCCL (`count` frames of a batch).  (A subclass may
        move the bytes another way -- tests/host_staged.py does, to run two ranks on one GPU --: what it returns for async_op must wait() like
        a device collective's Work, the current stream ordered behind the gathered data.)"""
        import torch.distributed as dist
        count = self.batch if count is None else count
        if count == self.batch:
            return dist.gather(self.shards[slot], self.gathered[slot] if self.rank == 0 else None, dst=0, async_op=async_op)
        nb = count * self.unit_bytes                              # the last, partial batch of a run
        return dist.gather(self.shards[slot][:nb], [g[:nb] for g in self.gathered[slot]] if self.rank == 0 else None, dst=0,
                           async_op=async_op)

    def blit(self, slot=0, count=None):
        """Rank 0: gathered shards -> row-major frame (set_pixels_from_buffer on the device).  "frames" mode gathers finished
        frames (gathered[slot][r] is rank r's frame), there is nothing to assemble."""
        if self.rank == 0 and self.mode == "tiles":
            if not self.collective:
                self.dev.blit_tiles_device(tuple(self.options), self.all_regions_c, self.shards[slot].data_ptr(), self.frame.data_ptr(),
                                           self._stream(), self.all_offsets)
                return
            for j in range(count if count is not None else self.batch):      # every frame of the batch, in order, into the frame
                self.dev.blit_tiles_device(tuple(self.options), self.all_regions_c, self.gathered_flat[slot].data_ptr(), self.frame.data_ptr(),
                                           self._stream(), self.batch_offsets[j])

    def step(self):
        """One complete frame, nothing left in flight."""
        if not self.collective:
            self.render_frame()
        else:
            self.render_shard(slot=0)
            self.gather(slot=0, count=1)
            self.blit(slot=0, count=1)

    # ---- the operations run_pipeline() schedules (a CPU stand-in with the same five methods is used by the gloo tests) ----
    def op_render(self, slot):
        count = self._counts[self._next] if self._next < len(self._counts) else self.batch
        self._next += 1
        self._slot_count[slot] = count
        with self.torch.cuda.stream(self.render_streams[slot]):
            for j in range(count):
                self.render_shard(slot=slot, index=j)

    def op_gather_async(self, slot):
        # issued under the slot's render stream: the collective waits for that stream's work (the render of this slot) only
        with self.torch.cuda.stream(self.render_streams[slot]):
            return self.gather(slot=slot, async_op=True, count=self._slot_count[slot])

    def op_blit_after(self, work, slot):
        """Frame in `slot` has been gathered once `work` is done: blit it.  Rank 0 does that on its side stream, so the blit
        overlaps the next render instead of queueing behind it (Work.wait() only orders the stream it is called on)."""
        if self.side is not None:
            with self.torch.cuda.stream(self.side):
                work.wait()
                self.blit(slot=slot, count=self._slot_count[slot])
        else:
            with self.torch.cuda.stream(self.render_streams[slot]):      # the slot's next render must not overwrite a shard still being sent
                work.wait()

    def op_before_reuse(self, slot):
        """shards[slot] is about to be re-rendered and gathered[slot] overwritten by the next gather: the gather that read the
        shard has been waited for on the side stream (op_blit_after) and the blit that read its gathered copy runs there -- the
        slot's render stream is ordered behind both."""
        if self.side is not None:
            self.render_streams[slot].wait_stream(self.side)

    def op_drain(self):
        main = self.torch.cuda.current_stream(self.device)                        # the caller synchronises the main stream only
        if self.side is not None:
            main.wait_stream(self.side)
        for rs in self.render_streams:
            main.wait_stream(rs)

    def run(self, steps):
        """`steps` complete frames.  world > 1: software-pipelined, gather(k) overlaps render(k+1); every frame has been
        blitted on rank 0 when this returns (the caller still synchronises the device)."""
        if not self.collective:
            for _ in range(steps):
                self.render_frame()
            return
        # `steps` frames in batches of self.batch (the last one may be partial); one pipeline step per batch
        self._counts = [self.batch] * (steps // self.batch) + ([steps % self.batch] if steps % self.batch else [])
        self._next = 0
        main = self.torch.cuda.current_stream(self.device)
        for rs in self.render_streams:                           # whatever the caller enqueued before (zeroed buffers, an earlier run)
            rs.wait_stream(main)
        if self.side is not None:
            self.side.wait_stream(main)
        run_pipeline(len(self._counts), self)

    def frame_host(self, slot=0, of_rank=0, index=0):
        """Rank 0: the assembled frame ("tiles"), or frame `index` of the batch gathered from `of_rank` in buffer `slot`
        ("frames")."""
        self.torch.cuda.synchronize(self.device)
        if self.mode == "tiles" or not self.collective:
            src = self.frame
        else:
            src = self.gathered[slot][of_rank][index * self.unit_bytes:(index + 1) * self.unit_bytes]
        return src.cpu().numpy().reshape(self.options.height, self.options.width, 4)
