"""Host-side mirror of the reference's renderer surface (render.rs) over the C ABI:

    RenderOptions, ImageRegion, RGBABuffer, RGBABufferWriter, PPMStdoutRGBABufferWriter, Renderer

Same names, argument meaning and error behaviour (the reference panics; here ValueError / RtError).  The one
deliberate difference: edge buckets are clipped instead of asserting w % 64 == 0 && h % 64 == 0
(render.rs:265-266), because BASELINE's 800x600 and 1920x1080 configs are not multiples of 64;
`strict_64=True` brings the assertion back.  Pixels come only from the HIP kernels; this module moves bytes."""
import queue
import sys
import threading
import time
from collections import namedtuple
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import capi

CHUNK_SIZE = 64                                                    # render.rs:264

RenderOptions = namedtuple("RenderOptions", "width height samples_per_pixel")   # render.rs:33-38


class ImageRegion(namedtuple("ImageRegion", "l t r b")):           # render.rs:42-72
    __slots__ = ()

    def width(self):
        return self.r - self.l

    def height(self):
        return self.t - self.b

    def area(self):
        return self.width() * self.height()

    def contains(self, o):
        return o.l >= self.l and o.b >= self.b and o.t <= self.t and o.r <= self.r

    def buffer_offset(self, x, y):
        return (y - self.b) * self.width() + (x - self.l)


class RGBABuffer:                                                  # render.rs:74-135
    def __init__(self, region, buf=None):
        self.reg = region
        self.buf = np.zeros((region.height(), region.width(), 4), dtype=np.uint8) if buf is None else \
            np.asarray(buf, dtype=np.uint8).reshape(region.height(), region.width(), 4)

    def set_pixels_from_buffer(self, b):                           # render.rs:112-126
        if not self.reg.contains(b.reg):
            raise ValueError("buffer must be contained in our rectangle")
        y0, x0 = b.reg.b - self.reg.b, b.reg.l - self.reg.l
        self.buf[y0:y0 + b.reg.height(), x0:x0 + b.reg.width()] = b.buf

    def buffer(self):
        return self.buf

    def region(self):
        return self.reg


class RGBABufferWriter:                                            # trait, render.rs:20-30
    def begin(self, x, y):
        raise NotImplementedError

    def write_rgba_buffer(self, buffer):
        raise NotImplementedError


class PPMStdoutRGBABufferWriter(RGBABufferWriter):                 # render.rs:319-434
    """P6 (rgb=True) or P5 writer.  `out` is a path (file sink: progressive whole-file rewrite at most once per
    second, render.rs:427-432), or '-' / a binary file object (written once at close, like the Drop impl).
    `clock` (seconds, monotonic) is injectable so that a test can watch the progressive rewrites."""

    def __init__(self, write_rgb, out, clock=time.monotonic):
        self.rgb = write_rgb
        self.out = out
        self.clock = clock
        self.width = self.height = None
        self.image = None
        self.last_written_at = None
        self.buffer_dirty = False

    def output_is_file(self):
        return isinstance(self.out, str) and self.out != "-"

    def begin(self, x, y):
        self.width, self.height = x, y
        self.image = RGBABuffer(ImageRegion(0, y, x, 0))

    def write_rgba_buffer(self, buffer):
        self.image.set_pixels_from_buffer(buffer)
        self.buffer_dirty = True
        now = self.clock()
        if self.output_is_file() and (self.last_written_at is None or self.last_written_at + 1.0 <= now):      # render.rs:427-432
            self.last_written_at = now
            self.write_buffer_with_header()

    def encode(self):
        if self.width is None:
            raise RuntimeError("begin() called")                   # .expect("begin() called") render.rs:380
        head = ("%s\n%d %d\n255\n" % ("P6" if self.rgb else "P5", self.width, self.height)).encode()
        px = self.image.buf.reshape(-1, 4)
        if self.rgb:
            body = np.ascontiguousarray(px[:, :3]).tobytes()
        else:                                                      # render.rs:399
            s = px[:, 0].astype(np.float32) + px[:, 1].astype(np.float32) + px[:, 2].astype(np.float32)
            body = (s / np.float32(3.0)).astype(np.uint8).tobytes()
        return head + body

    def write_buffer_with_header(self):                            # render.rs:359-407
        if not self.buffer_dirty:
            return
        data = self.encode()
        if self.output_is_file():
            with open(self.out, "wb") as f:                        # set_len(0) + seek(0) + write
                f.write(data)
        else:
            f = sys.stdout.buffer if self.out == "-" else self.out
            f.write(data)
            f.flush()
        self.buffer_dirty = False

    def close(self):                                               # Drop, render.rs:331-335
        self.write_buffer_with_header()


def buckets(options, chunk=CHUNK_SIZE):
    """The scheduler's bucket list, row-major, y outer (render.rs:273-298); edge buckets clipped (H5)."""
    out = []
    for y in range(0, options.height, chunk):
        for x in range(0, options.width, chunk):
            out.append(ImageRegion(x, min(y + chunk, options.height), min(x + chunk, options.width), y))
    return out


# Buckets handed to the device per call.  Large enough to fill 256 CUs (64 buckets = 1,024 workgroups), small enough that
# finished buckets keep reaching the writer while the rest of a long render is still running -- the reference's observable
# behaviour (tiles arrive in completion order, the file is rewritten once per second, render.rs:301-307, 427-432) --
# even with RTRACEMAXPROCS = 1.
MAX_BUCKETS_PER_CALL = 64


class Renderer:
    @staticmethod
    def render_region(o, scene, buf, device=0, traversal=None):
        """Renderer::render_region(o, scene, buf) render.rs:218 -- fills buf for buf.region() on the GPU.
        traversal: None = the reference's hierarchy walk when the scene has bounds (else the flat scan)."""
        # straight into the RGBABuffer's storage when that is one contiguous block; a buffer that wraps a strided view (a
        # sub-rectangle of a frame, say) is filled through a temporary -- reshape(-1) of such a view would be a silent copy
        direct = buf.buf.dtype == np.uint8 and buf.buf.flags.c_contiguous
        data, stats = scene.device(device).render_region(tuple(o), tuple(buf.region()), traversal, want_stats=True,
                                                         out=buf.buf.reshape(-1) if direct else None)
        if not direct:
            buf.buf[...] = data.reshape(buf.buf.shape)
        return stats

    @staticmethod
    def render(o, scene, writer, pool=1, device=0, traversal=None, tiles_per_call=None, strict_64=False):
        """Renderer::render(o, scene, writer, pool) render.rs:260-310.

        `pool` keeps the meaning of the reference's ThreadPool size (RTRACEMAXPROCS / --num-cores): the number of
        host scheduler threads.  Each thread hands the device a BATCH of buckets per call (one launch per 64x64
        bucket would starve 256 CUs, H4), at most MAX_BUCKETS_PER_CALL; finished buckets reach the writer through a
        bounded queue of 4 (sync_channel(4), render.rs:271) in completion order.  Returns accumulated ray statistics.
        strict_64: reproduce `assert!(w % 64 == 0 && h % 64 == 0)` (render.rs:265-266) instead of clipping edge buckets."""
        if strict_64 and (o.width % CHUNK_SIZE or o.height % CHUNK_SIZE):
            raise ValueError("TODO: handle chunk sizes")                 # the assert!s custom message
        bl = buckets(o)
        writer.begin(o.width, o.height)
        dev = scene.device(device)
        traversal = dev.default_traversal() if traversal is None else traversal
        pool = max(1, int(pool))
        per = tiles_per_call or max(1, min(MAX_BUCKETS_PER_CALL, -(-len(bl) // pool)))
        batches = [bl[i:i + per] for i in range(0, len(bl), per)]
        q = queue.Queue(maxsize=4)
        total = {"primary": 0, "hits": 0, "shadow": 0, "occluded": 0, "sphere_tests": 0, "device_ms": 0.0}
        lock = threading.Lock()
        stop = threading.Event()

        def work(batch):
            if stop.is_set():
                return
            data, st = dev.render_tiles(tuple(o), [tuple(r) for r in batch], traversal)
            with lock:
                for k in total:
                    total[k] += st[k]
            off = 0
            for r in batch:
                n = r.area() * 4
                item = RGBABuffer(r, data[off:off + n])
                while not stop.is_set():                           # a failed render must not leave workers blocked in put()
                    try:
                        q.put(item, timeout=0.05)
                        break
                    except queue.Full:
                        pass
                off += n

        count = len(bl)
        failure = None
        with ThreadPoolExecutor(max_workers=pool) as ex:
            futs = [ex.submit(work, b) for b in batches]
            try:
                while count:
                    try:
                        writer.write_rgba_buffer(q.get(timeout=0.05))
                        count -= 1
                    except queue.Empty:
                        for f in futs:
                            if f.done() and f.exception() is not None:
                                raise f.exception()
            except BaseException as e:                             # worker or writer failure: release the workers, then re-raise
                failure = e
                stop.set()
        if failure is not None:
            raise failure
        for f in futs:
            f.result()
        assert count == 0, "We really should have processed all chunks here"      # render.rs:308-309
        return total
