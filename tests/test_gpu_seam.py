"""The host-facing seam added in round 3, through the C ABI on an MI355X: delivery in completion order (rt_render_tiles_stream),
the frame-ahead of rt_render_region, pipelined gang frames, pinned-buffer lifetime and the `--scene` path of the native host."""
import gc
import os
import subprocess
import threading

import numpy as np
import pytest

import oracle
import rust_tracer_amd as rta
from tests import util

pytestmark = pytest.mark.gpu
SKIP, FLAT = rta.RT_TRAVERSAL_SKIP, rta.RT_TRAVERSAL_FLAT
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bucket_list(w, h, spp=1):
    return [tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, spp))]


@pytest.mark.parametrize("size,trav", [((800, 600, 1), SKIP), ((333, 200, 2), SKIP), ((256, 192, 1), FLAT), ((2048, 1600, 1), SKIP)])
def test_stream_call_delivers_every_bucket_once_with_the_reference_bytes(size, trav):
    # rt_render_tiles_stream: batches enqueued at once, buckets handed to the callback batch by batch in list order
    s, o = util.scene_pair_default()
    w, h, spp = size
    regs = bucket_list(w, h, spp)
    ref, _, _ = o.render(w, h, spp, nthreads=os.cpu_count() or 1)
    frame = np.zeros((h, w, 4), dtype=np.uint8)
    seen = []

    def on_tile(i, reg, px):
        assert reg == regs[i]
        l, t, r, b = reg
        frame[b:t, l:r] = px
        seen.append(i)

    s.device().render_tiles_stream(size, regs, on_tile, trav)
    assert seen == list(range(len(regs)))
    np.testing.assert_array_equal(frame, ref)


def test_stream_call_propagates_a_failing_consumer_and_bad_regions():
    s, _ = util.scene_pair_default()
    regs = bucket_list(256, 192)

    def boom(i, reg, px):
        if i == 3:
            raise KeyError("writer failed")

    with pytest.raises(KeyError):
        s.device().render_tiles_stream((256, 192, 1), regs, boom, SKIP)
    with pytest.raises(rta.RtError) as e:
        s.device().render_tiles_stream((256, 192, 1), regs + [(0, 300, 64, 236)], lambda *a: None, SKIP)
    assert e.value.status == rta.capi.RT_ERR_INVALID_REGION
    # the context is usable afterwards
    data, _ = s.device().render_tiles((256, 192, 1), regs, SKIP, want_stats=False)
    assert data.size == 256 * 192 * 4


def test_render_region_frame_ahead_serves_a_frame_from_one_pass():
    # render.rs:283-294 calls render_region once per bucket.  From one thread that used to be one device pass per call; now the first
    # call of a frame renders the whole bucket grid and the others are copies -- and a bucket is only handed out once per pass: asking
    # for it again (the next frame) renders again.
    s, o = util.scene_pair_default()
    d = s.device()
    w, h = 800, 600
    regs = bucket_list(w, h)
    ref, _, _ = o.render(w, h, 1, nthreads=os.cpu_count() or 1)
    count = lambda: rta.capi.debug_count(rta.capi.DEBUG_COUNT_FRAME_AHEAD_PASSES)
    # default: the pass for the next frame is started while this frame's buckets are handed out (frames 2 and 3 below come out of such
    # passes); RT_DEBUG_FRAME_AHEAD = 1: every pass is rendered when its frame is first asked for.  One pass per frame either way.
    for mode in (-1, 1):
        with rta.capi.debug(rta.capi.DEBUG_FRAME_AHEAD, mode):
            c0 = count()
            for frame in range(3):
                for (l, t, r, b) in regs:
                    got, _ = d.render_region((w, h, 1), (l, t, r, b), SKIP)
                    np.testing.assert_array_equal(got, ref[b:t, l:r])
            assert count() - c0 == 3
    # other options: a new grid; a region that is not a bucket of the grid: its own pass; with stats: the counting path
    got, _ = d.render_region((w, h, 2), regs[7], SKIP)
    ref2, _ = o.render_region(w, h, 2, *regs[7])
    np.testing.assert_array_equal(got, ref2)
    c1 = count()
    got, _ = d.render_region((w, h, 1), (10, 90, 70, 30), SKIP)
    np.testing.assert_array_equal(got, ref[30:90, 10:70])
    got, st = d.render_region((w, h, 1), regs[0], SKIP, want_stats=True)
    assert st["primary"] == 64 * 64
    assert count() == c1
    # switched off: every call its own pass
    with rta.capi.debug(rta.capi.DEBUG_FRAME_AHEAD, 0):
        got, _ = d.render_region((w, h, 1), regs[5], SKIP)
        np.testing.assert_array_equal(got, ref[regs[5][3]:regs[5][1], regs[5][0]:regs[5][2]])
    assert count() == c1


def test_render_region_lone_requests_stay_cheap_and_device_pointers_are_refused():
    # a lone request for a grid bucket (a partial redraw, a tool) is rendered on its own -- no whole-grid pass, nothing
    # started ahead; the frame-ahead engages with the second distinct bucket of the same frame.  A device pointer is refused by
    # rt_render_region exactly as rt_render_tiles refuses it, whatever path would have served the call.
    import ctypes as C
    import torch
    s, o = util.scene_pair_default()
    d = s.device()
    w, h = 1024, 768
    regs = bucket_list(w, h)
    count = lambda: rta.capi.debug_count(rta.capi.DEBUG_COUNT_FRAME_AHEAD_PASSES)
    c0 = count()
    for _ in range(3):                                           # the same bucket again and again: never a whole-grid pass
        got, _ = d.render_region((w, h, 1), regs[17], SKIP)
    ref17, _ = o.render_region(w, h, 1, *regs[17])
    np.testing.assert_array_equal(got, ref17)
    assert count() == c0
    got, _ = d.render_region((w, h, 1), regs[18], SKIP)          # a second bucket of that frame: now the grid is rendered
    ref18, _ = o.render_region(w, h, 1, *regs[18])
    np.testing.assert_array_equal(got, ref18)
    assert count() == c0 + 1
    dev = torch.zeros(64 * 64 * 4, dtype=torch.uint8, device="cuda")
    opts = rta.capi.Options(w, h, 1)
    reg = rta.capi.Region(*regs[3])
    rc = rta.capi.lib.rt_render_region(d._h, C.byref(opts), SKIP, C.byref(reg), C.c_void_p(dev.data_ptr()), None)
    assert rc == rta.capi.RT_ERR_INVALID_ARGUMENT and b"device memory" in rta.capi.lib.rt_last_error_message()


def test_render_region_pass_started_ahead_survives_destroy_and_option_changes():
    # The pass for the next frame is started while the current one is handed out: a scene destroyed with that pass in flight must wait
    # for it, and a caller that alternates between options (each change discards the pass that was started ahead) still gets the right
    # bytes every time.
    s, o = util.scene_pair_default()
    for _ in range(8):
        d = s.device()
        got, _ = d.render_region((640, 448, 1), (0, 64, 64, 0), SKIP)          # one bucket only: the pass for frame 2 is in flight now
        d.close()
        s._device.clear()
    d = s.device()
    refs = {}
    for k in range(12):
        w, h, spp = ((256, 192, 1), (320, 256, 1), (256, 192, 2))[k % 3]
        regs = bucket_list(w, h, spp)
        i = (5 * k) % len(regs)
        got, _ = d.render_region((w, h, spp), regs[i], SKIP)
        if (w, h, spp, i) not in refs:
            refs[(w, h, spp, i)], _ = o.render_region(w, h, spp, *regs[i])
        np.testing.assert_array_equal(got, refs[(w, h, spp, i)])


def test_render_region_frame_ahead_under_concurrent_callers():
    s, o = util.scene_pair_default()
    d = s.device()
    w, h = 512, 384
    regs = bucket_list(w, h)
    ref, _, _ = o.render(w, h, 1, nthreads=os.cpu_count() or 1)
    out = {}
    jobs = [(f, i) for f in range(4) for i in range(len(regs))]
    nxt, lock = [0], threading.Lock()

    def work():
        while True:
            with lock:
                k = nxt[0]
                nxt[0] += 1
            if k >= len(jobs):
                return
            out[jobs[k]] = d.render_region((w, h, 1), regs[jobs[k][1]], SKIP)[0]

    th = [threading.Thread(target=work) for _ in range(12)]
    [t.start() for t in th]
    [t.join() for t in th]
    for (f, i), got in out.items():
        l, t, r, b = regs[i]
        np.testing.assert_array_equal(got, ref[b:t, l:r])


def test_render_region_validates_out_and_renderer_fills_strided_buffers():
    s, o = util.scene_pair_default()
    d = s.device()
    with pytest.raises(ValueError):
        d.render_region((128, 128, 1), (0, 64, 64, 0), SKIP, out=np.zeros(100, dtype=np.uint8))
    with pytest.raises(ValueError):
        d.render_region((128, 128, 1), (0, 64, 64, 0), SKIP, out=np.zeros(64 * 64 * 4, dtype=np.float32))
    ref, _ = o.render_region(128, 128, 1, 0, 64, 64, 0)
    # an RGBABuffer over a sub-rectangle of a frame (a strided view): filled through a temporary, not silently skipped
    frame = np.zeros((128, 128, 4), dtype=np.uint8)
    buf = rta.RGBABuffer(rta.ImageRegion(0, 64, 64, 0))
    buf.buf = frame[0:64, 0:64]
    assert not buf.buf.flags.c_contiguous
    rta.Renderer.render_region(rta.RenderOptions(128, 128, 1), s, buf)
    np.testing.assert_array_equal(frame[0:64, 0:64], ref)


def test_a_temporary_host_buffer_lives_as_long_as_its_array():
    s, o = util.scene_pair_default()
    regs = bucket_list(320, 200)
    ref, _, _ = o.render(320, 200, 1, nthreads=4)
    data, _ = s.device().render_tiles((320, 200, 1), regs, SKIP, want_stats=False, out=rta.capi.HostBuffer(320 * 200 * 4).array)
    gc.collect()
    junk = [rta.capi.HostBuffer(320 * 200 * 4) for _ in range(4)]       # would reuse the freed range
    for j in junk:
        j.array[:] = 0xAB
    np.testing.assert_array_equal(util.stitch((320, 200), regs, data), ref)


_GANG_SCRIPT = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import oracle
import rust_tracer_amd as rta
from tests import util

s, o = util.scene_pair_default()
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 1
try:
    if ranks > 1:
        # several ranks on this one GPU through the stand-in for librccl.so (tests/c/fake_rccl.cpp, rt_debug_rccl_library)
        with rta.capi.rccl_stand_in():
            g = rta.Gang(s, [0] * ranks)
    else:
        g = rta.Gang(s, [0])
except rta.RtError as e:
    if e.status == rta.capi.RT_ERR_UNSUPPORTED:
        print("SKIP no RCCL")
        sys.exit(0)
    raise
assert g.size() == ranks
w, h = 800, 600
regs = [tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, 1))]
ref, rst, _ = o.render(w, h, 1, nthreads=os.cpu_count() or 1)
frames, st = g.render_frames((w, h, 1), regs, 5, want_stats=True)
for f in frames:
    np.testing.assert_array_equal(f, ref)
assert util.ray_stats(st) == util.ray_stats(rst)
pinned = [rta.capi.HostBuffer(w * h * 4) for _ in range(3)]
frames, _ = g.render_frames((w, h, 1), regs, 3, out=[p.array for p in pinned])
for f in frames:
    np.testing.assert_array_equal(f, ref)
# a partial tile list leaves the other pixels of a fresh frame zero, never stale device memory
part, _ = g.render_frame((w, h, 1), regs[:7])
assert not part[64:].any() and np.array_equal(part[0:64, 0:448], ref[0:64, 0:448])
g.close()
print("OK")
"""


@pytest.mark.parametrize("ranks", [2, 4])
def test_gang_pipelined_frames_several_ranks_on_one_gpu(tmp_path, ranks):
    # the same with 2 and 4 ranks: rt_gang_render_frames' N > 1 code (ncclGroupStart / one ncclGather per rank / ncclGroupEnd, shards double-
    # buffered, gather(f) under render(f + 1), blit on the root) EXECUTED on this box's one GPU -- every rank a communicator of the
    # stand-in library on device 0 (tests/c/fake_rccl.cpp; /root/reference/src/rust/render.rs:271,293,301 is what the gather replaces)
    import sys
    if not rta.capi.HAVE_TEST_HOOKS:
        pytest.skip("several ranks on one GPU need the stand-in for librccl.so (csrc/rt_debug.h): not in the library that ships")
    script = tmp_path / "gang_frames.py"
    script.write_text(_GANG_SCRIPT)
    r = subprocess.run([sys.executable, str(script), ROOT, str(ranks)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "OK" in r.stdout and "SKIP" not in r.stdout


def test_gang_pipelined_frames_one_rank(tmp_path):
    # rt_gang_render_frames on a one-rank communicator (what a 1-GPU box can run): gather(f) under render(f + 1), double-buffered
    # shards; pinned destinations are written by the root's blit kernel, pageable ones copied.  In a process of its own: the gang
    # creates its communicator with ncclCommInitAll inside whatever RCCL instance the process has loaded, and this session's other
    # tests have created and destroyed torch.distributed process groups in it.
    import sys
    script = tmp_path / "gang_frames.py"
    script.write_text(_GANG_SCRIPT)
    r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=300, env=util.product_env())      # (the product library: no control needed)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    if "SKIP" in r.stdout:
        pytest.skip("no RCCL")
    assert "OK" in r.stdout


def test_rtrace_scene_file_with_an_automatically_built_hierarchy(tmp_path):
    # SURVEY.md 8f.4 on the native host: `rtrace --scene <file>` builds the bounding-sphere hierarchy itself (csrc/host/scene.hpp);
    # the oracle renders the Python mirror's hierarchy of the same list (both builders agree bit for bit: test_host_and_abi.py)
    rng = np.random.default_rng(9)
    n = 2000
    sp = np.concatenate([rng.uniform([-2, -1.5, 0], [2, 1.5, 4], (n, 3)), rng.uniform(0.02, 0.12, (n, 1))], axis=1).astype(np.float32).astype(np.float64)
    path = tmp_path / "spheres.txt"
    with open(path, "w") as f:
        f.write("eye 0.1 -0.2 -4.5\nlight -1 -2 1.5\n")
        for q in sp:
            f.write("%r %r %r %r\n" % tuple(float(v) for v in q))
    out = str(tmp_path / "out.tga")
    exe = os.path.join(ROOT, "rust-tracer_amd", "rtrace")
    r = subprocess.run([exe, "--width=400", "--height=300", "--samples-per-pixel=2", "--scene=" + str(path), out], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    from rust_tracer_amd.scene import build_hierarchy
    items, bounds, ranges, _ = build_hierarchy(sp, eye=(0.1, -0.2, -4.5))
    o = oracle.Scene.from_ranges(items.astype(np.float64), bounds.astype(np.float64), ranges, (-1.0, -2.0, 1.5), (0.1, -0.2, -4.5))
    img, st, _ = o.render(400, 300, 2, nthreads=os.cpu_count() or 1)
    assert st["hits"] > 10000
    ref = str(tmp_path / "ref.ppm")
    oracle.write_ppm(ref, img)
    assert open(out, "rb").read() == open(ref, "rb").read()


@pytest.mark.parametrize("scale", [1.0 - 1.9e-3, 1.0 + 1.9e-3])
def test_filter_bounds_hold_at_the_edges_of_the_accepted_light_length(scale):
    # rt_scene_create accepts a light_unit whose squared length is within 2e-3 of 1; the filtered loops' shadow bounds carry that
    # deviation (eta).  At both edges: the counting launch evaluates the bounds for every test it makes (violations are asserted 0
    # by the autouse fixture) and the filtered assembly loops render the same bytes as the reference loops.
    items, bounds, ranges = util.random_nested_scene(35, depth=3, fan=3, leaf_items=2, concentric=True)
    light = rta.normalized((-1.0, -3.0, 2.0), rta.RT_F32).astype(np.float64) * np.sqrt(scale)
    s = rta.Scene(items, light, (0.05, -0.1, -3.2), bounds, ranges, rta.RT_F32)
    regs = bucket_list(192, 160, 2)
    counted, st = s.device().render_tiles((192, 160, 2), regs, SKIP, want_stats=True)
    assert st["shadow"] > 1000
    for variant in (3, 7, 19, 23):
        with util.loop_flavour(variant):
            plain, _ = s.device().render_tiles((192, 160, 2), regs, SKIP, want_stats=False)
        np.testing.assert_array_equal(plain, counted)


_REF_FRAMES = {}


def _ref_frame(w, h, spp):
    if (w, h, spp) not in _REF_FRAMES:
        _, o = util.scene_pair_default()
        _REF_FRAMES[(w, h, spp)] = o.render(w, h, spp, nthreads=os.cpu_count() or 1)[0]
    return _REF_FRAMES[(w, h, spp)]


def _encode(ref, fmt):
    """The writer's conversion on the CPU (render.rs:392-399): P6 keeps R, G, B; P5 is ((r + g + b) as f32 / 3.0) as u8."""
    if fmt == rta.capi.RT_FRAME_RGBA:
        return ref
    if fmt == rta.capi.RT_FRAME_RGB:
        return ref[..., :3]
    f = ref[..., 0].astype(np.float32) + ref[..., 1].astype(np.float32) + ref[..., 2].astype(np.float32)
    return (f / np.float32(3.0)).astype(np.uint8)[..., None]


@pytest.mark.parametrize("pinned", [True, False], ids=["pinned", "pageable"])
@pytest.mark.parametrize("fmt", [rta.capi.RT_FRAME_RGBA, rta.capi.RT_FRAME_RGB, rta.capi.RT_FRAME_GREY], ids=["rgba", "rgb", "grey"])
@pytest.mark.parametrize("size", [(1920, 1080, 1), (201, 131, 2), (800, 600, 1)])
def test_device_encoded_frame_is_the_writers_image(size, fmt, pinned):
    # rt_render_frame_stream: the buckets rendered, converted ON THE DEVICE to the file's pixel format and put in their place in the caller's
    # row-major image (render.rs:373-401 is the writer's CPU loop this replaces; 112-126 the blit) -- bit for bit what the CPU conversion of
    # the oracle's frame gives, for aligned and odd widths (201 x 3 bytes per row: every row segment starts and ends inside a 32-bit word),
    # into memory the device writes directly and into pageable memory; batches reported once each, in order.
    w, h, spp = size
    s, _ = util.scene_pair_default()
    want = _encode(_ref_frame(w, h, spp), fmt)
    bpp = want.shape[2]
    regs = [tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, spp))]
    keep = rta.capi.HostBuffer(w * h * bpp) if pinned else None
    out = keep.array if pinned else np.empty(w * h * bpp, dtype=np.uint8)
    out[:] = 0xAB
    seen = []
    s.device().render_frame_stream((w, h, spp), regs, fmt, out, on_batch=lambda first, n: seen.append((first, n)))
    np.testing.assert_array_equal(out.reshape(h, w, bpp), want)
    assert [f for f, _ in seen] == sorted(f for f, _ in seen) and sum(n for _, n in seen) == len(regs) and seen[0][0] == 0
    # a partial list: the bytes of the other buckets are left alone (row segments of neighbours share 32-bit words at odd widths)
    out[:] = 0xAB
    part = regs[::2]
    s.device().render_frame_stream((w, h, spp), part, fmt, out)
    got = out.reshape(h, w, bpp)
    mask = np.zeros((h, w), dtype=bool)
    for (l, t, r, b) in part:
        mask[b:t, l:r] = True
        np.testing.assert_array_equal(got[b:t, l:r], want[b:t, l:r])
    assert (got[~mask] == 0xAB).all()


def test_device_encoded_frame_ragged_regions_and_flat_traversal():
    # regions that are no buckets: odd origins and widths, one pixel wide, one row high; and the flat traversal's pipeline in front of the encoder
    w, h = 201, 131
    s, _ = util.scene_pair_default()
    ref = _ref_frame(w, h, 1)
    regs = [(3, 50, 40, 7), (41, 131, 200, 60), (200, 131, 201, 0), (0, 1, 199, 0), (100, 59, 101, 58)]
    for fmt in (rta.capi.RT_FRAME_RGB, rta.capi.RT_FRAME_GREY):
        want = _encode(ref, fmt)
        bpp = want.shape[2]
        for trav in (rta.RT_TRAVERSAL_SKIP, rta.RT_TRAVERSAL_FLAT):
            buf = rta.capi.HostBuffer(w * h * bpp)
            buf.array[:] = 0x5A
            s.device().render_frame_stream((w, h, 1), regs, fmt, buf.array, traversal=trav)
            got = buf.array.reshape(h, w, bpp)
            mask = np.zeros((h, w), dtype=bool)
            for (l, t, r, b) in regs:
                mask[b:t, l:r] = True
                np.testing.assert_array_equal(got[b:t, l:r], want[b:t, l:r])
            assert (got[~mask] == 0x5A).all()
    import ctypes
    o = rta.capi.Options(w, h, 1)
    arr = s.device()._regions(regs)
    junk = np.zeros(w * h * 4, dtype=np.uint8)
    cb = rta.capi.BATCH_CALLBACK(lambda *_: None)
    assert rta.capi.lib.rt_render_frame_stream(s.device()._h, ctypes.byref(o), rta.RT_TRAVERSAL_SKIP, arr, len(arr), 7, junk.ctypes.data, cb, None) == rta.capi.RT_ERR_INVALID_ARGUMENT
    assert rta.capi.lib.rt_render_frame_stream(s.device()._h, ctypes.byref(o), rta.RT_TRAVERSAL_SKIP, arr, len(arr), 1, junk.ctypes.data + 1, cb, None) == rta.capi.RT_ERR_INVALID_ARGUMENT


def test_device_encoded_frame_f64_and_a_large_scene():
    # the encoder sits behind every render kernel: the f64 walk (BASELINE config 3's type-alias swap) and the two-ray kernel's sample-packed
    # pass over the 87,381-sphere pyramid (BASELINE config 5's scene), P6 payload against the oracle's frame
    for precision, level, (w, h, spp) in ((rta.RT_F64, 8, (320, 256, 1)), (rta.RT_F32, 9, (512, 384, 4))):
        s, o = util.scene_pair_default(precision, level)
        ref, _, _ = o.render(w, h, spp, nthreads=os.cpu_count() or 1)
        regs = [tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, spp))]
        buf = rta.capi.HostBuffer(w * h * 3)
        buf.array[:] = 0x11
        s.device().render_frame_stream((w, h, spp), regs, rta.capi.RT_FRAME_RGB, buf.array)
        np.testing.assert_array_equal(buf.array.reshape(h, w, 3), ref[..., :3])


def test_scene_setup_cost_reports_the_create_call():
    # rt_scene_setup_cost: what rt_scene_create took, and the stream's share of it (in a fresh process the runtime's first hardware queue)
    d = rta.Scene.default().device()
    total, stream = d.setup_cost()
    assert 0.0 < stream <= total < 5000.0, (total, stream)
    d.close()
