"""CPU-only checks: the C-ABI library loads and exports every symbol include/rtrace_hip.h declares (no compute
calls without a GPU), the host-side mirror behaves like the reference's (render.rs tests), and the host scene
builder equals the oracle's."""
import ctypes
import os
import re

import numpy as np
import pytest

import oracle
import rust_tracer_amd as rta
from rust_tracer_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rtrace_hip.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z_]+)\s*\(", hdr))
    assert declared == set(capi.SYMBOLS)
    import subprocess
    for path in (capi.PRODUCT_LIB_PATH, capi.TEST_LIB_PATH):
        lib = ctypes.CDLL(path)
        for name in declared:
            assert getattr(lib, name) is not None
        assert lib.rt_abi_version() == capi.ABI_VERSION
    # the PRODUCT exports the ABI and nothing else: no rt_debug_* hook, no kernel stub, no C++ symbol (csrc/exports.map)
    exported = {l.split()[-1] for l in subprocess.run(["nm", "-D", "--defined-only", capi.PRODUCT_LIB_PATH], capture_output=True, text=True, check=True).stdout.splitlines() if l.strip()}
    assert exported == declared, sorted(exported ^ declared)


def test_the_tests_run_on_the_hooks_build_and_the_product_has_no_hooks():
    # tests/conftest.py points the binding at tests/c/librtrace_hip_test.so (same sources, -DRT_TEST_HOOKS); the product is what rtrace,
    # `make image`, bench.py and smoke() load, and it carries no diagnostic entry point and no stand-in for librccl.so
    assert capi.HAVE_TEST_HOOKS and os.path.samefile(capi.LIB_PATH, capi.TEST_LIB_PATH)
    assert "RT_TEST_HOOKS" in capi.build_info()
    product = ctypes.CDLL(capi.PRODUCT_LIB_PATH)
    product.rt_build_info.restype = ctypes.c_char_p
    info = product.rt_build_info().decode()
    assert "RT_TEST_HOOKS" not in info and info == capi.build_info().replace(" | RT_TEST_HOOKS", "")      # the same sources, the same toolchain
    for name in capi.DEBUG_SYMBOLS:
        assert not hasattr(product, name), name
        assert hasattr(capi.lib, name), name
    blob = open(capi.PRODUCT_LIB_PATH, "rb").read()
    assert b"rt_debug" not in blob and b"rccl-stand-in" not in blob and b"fake_rccl" not in blob
    rtrace = open(os.path.join(ROOT, "rust-tracer_amd", "rtrace"), "rb").read()
    assert b"rccl-stand-in" not in rtrace and b"rt_debug" not in rtrace


def test_product_never_touches_the_oracle():
    # the oracle is test infrastructure: no file of the product package may mention it
    pkg = os.path.join(ROOT, "rust-tracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, f)


def test_strerror_and_sizes():
    assert capi.lib.rt_strerror(capi.RT_OK) == b"ok"
    assert b"region" in capi.lib.rt_strerror(capi.RT_ERR_INVALID_REGION)
    regs = (capi.Region * 2)(capi.Region(0, 64, 64, 0), capi.Region(64, 24, 96, 0))
    assert capi.lib.rt_tiles_rgba_bytes(regs, 2) == (64 * 64 + 32 * 24) * 4
    bad = (capi.Region * 1)(capi.Region(5, 1, 5, 0))
    assert capi.lib.rt_tiles_rgba_bytes(bad, 1) == 0


@pytest.mark.skipif(rta.device_count() > 0, reason="checks the no-device behaviour")
def test_no_device_fails_loudly_no_cpu_fallback():
    s = rta.Scene.three_spheres()
    with pytest.raises(rta.RtError) as e:
        s.device()
    assert e.value.status == capi.RT_ERR_NO_DEVICE


def test_scene_create_rejects_bad_arguments_before_touching_a_device():
    h = ctypes.c_void_p()
    items = np.array([[0, 0, 0, 1]], dtype=np.float32)
    v = np.zeros(3, dtype=np.float32)
    rc = capi.lib.rt_scene_create(0, capi.RT_F32, None, 1, v.ctypes.data, v.ctypes.data, None, None, 0, ctypes.byref(h))
    assert rc == capi.RT_ERR_INVALID_ARGUMENT
    rc = capi.lib.rt_scene_create(0, capi.RT_F32, items.ctypes.data, 0, v.ctypes.data, v.ctypes.data, None, None, 0, ctypes.byref(h))
    assert rc == capi.RT_ERR_INVALID_ARGUMENT
    bad = np.array([[0, 0, 0, 0]], dtype=np.float32)       # radius must be > 0
    rc = capi.lib.rt_scene_create(0, capi.RT_F32, bad.ctypes.data, 1, v.ctypes.data, v.ctypes.data, None, None, 0, ctypes.byref(h))
    assert rc == capi.RT_ERR_INVALID_ARGUMENT
    assert capi.lib.rt_last_error_message() != b""


@pytest.mark.parametrize("precision,oprec", [(rta.RT_F32, oracle.F32), (rta.RT_F64, oracle.F64)])
@pytest.mark.parametrize("level", [2, 5, 8])
def test_host_scene_builder_equals_oracle(precision, oprec, level):
    s = rta.Scene.default(level, precision)
    o = oracle.Scene.default(oprec, level)
    np.testing.assert_array_equal(s.items, o.flatten())
    b, r = o.bounds()
    np.testing.assert_array_equal(s.bounds, b)
    np.testing.assert_array_equal(s.ranges, r)
    l, e = o.light_eye()
    np.testing.assert_array_equal(s.directional_light, l)
    np.testing.assert_array_equal(s.eye, e)


def test_pyramid_counts_and_level_assert():
    # group::tests::pyramid group.rs:172-184; assert!(level > 1) group.rs:59
    items, bounds, ranges = rta.pyramid(8, (1.0, -1.0, 0.0), 1.0)
    assert (bounds.shape[0], items.shape[0]) == (5461, 21845)
    with pytest.raises(ValueError):
        rta.pyramid(1, (0, 0, 0), 1.0)


def test_image_region():
    # render::tests::image_region render.rs:483-499
    r = rta.ImageRegion(l=2, t=18, r=34, b=2)
    assert r.width() == 32 and r.height() == 16 and r.area() == 16 * 32
    assert r.contains(r)
    l = r._replace(l=1)
    assert l.contains(r) and not r.contains(l)
    assert r.buffer_offset(3, 3) == 33


def test_bucket_list_matches_the_scheduler():
    # render.rs:273-298 row-major, y outer; 64x128 -> 2 buckets (basic_rendering); clipped edges (H5)
    assert [tuple(b) for b in rta.buckets(rta.RenderOptions(64, 128, 2))] == [(0, 64, 64, 0), (0, 128, 64, 64)]
    b = rta.buckets(rta.RenderOptions(800, 600, 1))
    assert len(b) == 130 and tuple(b[12]) == (768, 64, 800, 0) and tuple(b[-1]) == (768, 600, 800, 576)
    assert len(rta.buckets(rta.RenderOptions(1920, 1080, 1))) == 510
    assert sum(x.area() for x in b) == 800 * 600


def test_ppm_writer_bytes(tmp_path):
    # render.rs:373-401
    w = rta.PPMStdoutRGBABufferWriter(True, str(tmp_path / "a.tga"))
    w.begin(3, 2)
    buf = rta.RGBABuffer(rta.ImageRegion(0, 2, 3, 0))
    buf.buf[..., 0] = 10; buf.buf[..., 1] = 20; buf.buf[..., 2] = 33; buf.buf[..., 3] = 255
    w.write_rgba_buffer(buf)
    w.close()
    assert open(str(tmp_path / "a.tga"), "rb").read() == b"P6\n3 2\n255\n" + bytes([10, 20, 33] * 6)
    g = rta.PPMStdoutRGBABufferWriter(False, str(tmp_path / "g.tga"))
    g.begin(3, 2)
    g.write_rgba_buffer(buf)
    g.close()
    assert open(str(tmp_path / "g.tga"), "rb").read() == b"P5\n3 2\n255\n" + bytes([21] * 6)


def test_rgba_buffer_blit_requires_containment():
    big = rta.RGBABuffer(rta.ImageRegion(0, 64, 64, 0))
    with pytest.raises(ValueError):
        big.set_pixels_from_buffer(rta.RGBABuffer(rta.ImageRegion(32, 96, 96, 32)))


def test_auto_hierarchy_is_a_valid_enclosing_nesting():
    # arbitrary sphere lists.  The built bounds must enclose their subtrees (in REAL) and nest, so the
    # oracle accepts the same description and flat == hierarchy for an eye outside the root bound (H2).
    rng = np.random.default_rng(3)
    sp = np.concatenate([rng.uniform([-2, -1.5, 0], [2, 1.5, 4], (3000, 3)), rng.uniform(0.02, 0.08, (3000, 1))], axis=1)
    sp = sp.astype(np.float32).astype(np.float64)
    items, bounds, ranges, order = rta.build_hierarchy(sp, leaf_size=4)
    np.testing.assert_array_equal(items, sp[order].astype(np.float32))
    assert sorted(order.tolist()) == list(range(3000))
    assert ranges[0].tolist() == [0, 3000]
    for gi in range(len(bounds)):
        f, c = ranges[gi]
        it, b = items[f:f + c].astype(np.float64), bounds[gi].astype(np.float64)
        assert (np.linalg.norm(it[:, :3] - b[:3], axis=1) + it[:, 3] <= b[3]).all(), gi
    o = oracle.Scene.from_ranges(items.astype(np.float64), bounds.astype(np.float64), ranges)
    assert o.counts() == (len(bounds), 3000)
    a, _ = o.render_region(160, 120, 1, 0, 120, 160, 0, oracle.MODE_HIERARCHY)
    b, _ = o.render_region(160, 120, 1, 0, 120, 160, 0, oracle.MODE_FLAT)
    np.testing.assert_array_equal(a, b)


def test_header_is_plain_c_and_the_library_links_from_c(tmp_path):
    # the boundary is a C ABI: a C99 translation unit (no C++, no HIP headers) includes the header and links the library
    import subprocess
    exe = str(tmp_path / "abi_check")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_check.c"), "-o", exe, "-L", os.path.dirname(capi.PRODUCT_LIB_PATH),
                           "-lrtrace_hip", "-Wl,-rpath," + os.path.dirname(capi.PRODUCT_LIB_PATH)])      # the product library
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "sizeof rt_options 6 rt_region 8 rt_range 8 rt_stats %d" % ctypes.sizeof(capi.Stats) in out.stdout
    # the ctypes mirror has the same field offsets as the C struct
    offs = " ".join("%s %d" % (n, getattr(capi.Stats, n).offset) for n, _ in capi.Stats._fields_)
    assert "offsets stats: " + offs in out.stdout
    assert ctypes.sizeof(capi.Options) == 6 and ctypes.sizeof(capi.Region) == 8 and ctypes.sizeof(capi.Range) == 8


@pytest.mark.parametrize("name,header", [("gen_skip_asm", "rt_skip_rot.hpp"), ("gen_skip2_asm", "rt_skip2_rot.hpp"), ("gen_flat_asm", "rt_flat_rot.hpp")])
def test_generated_traversal_loops_are_up_to_date(tmp_path, name, header):
    # csrc/rt_skip_rot.hpp / rt_flat_rot.hpp are the output of tools/gen_*_asm.py: the committed header must be what the generator writes
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    committed = open(gen.OUT).read()
    assert gen.OUT.endswith(header)
    gen.OUT = str(tmp_path / header)
    gen.main()
    assert open(gen.OUT).read() == committed


def test_library_reads_no_environment_variable_and_keeps_diagnostics_out_of_the_abi():
    # diagnostic switches live behind rt_debug_set (csrc/rt_debug.h, not in include/): the shipped library neither imports
    # getenv nor carries RT_* switch names a stray environment variable could trigger
    import subprocess
    for path in (capi.PRODUCT_LIB_PATH, capi.TEST_LIB_PATH):
        syms = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True, check=True).stdout
        assert "getenv" not in syms
    blob = open(capi.PRODUCT_LIB_PATH, "rb").read()
    assert re.findall(rb"RT_[A-Z_]{3,}", blob) == []
    hdr = open(os.path.join(ROOT, "include", "rtrace_hip.h")).read()
    assert "rt_debug" not in hdr
    assert capi.lib.rt_debug_set(999, 1) == capi.RT_ERR_INVALID_ARGUMENT
    for key in range(13):
        assert capi.lib.rt_debug_set(key, -1) == capi.RT_OK
    # every function csrc/rt_debug.h declares is exported; the filter check refuses a null scene without touching a device
    dbg = open(os.path.join(ROOT, "rust-tracer_amd", "csrc", "rt_debug.h")).read()
    for name in re.findall(r"\b(rt_debug_\w+)\s*\(", dbg):
        assert hasattr(capi.lib, name), name
    import ctypes
    assert capi.lib.rt_debug_flat_filter_check(None, 4, 4, 1, ctypes.byref((ctypes.c_ulonglong * 6)())) == capi.RT_ERR_INVALID_ARGUMENT


def test_strict_64_and_writer_clock_on_the_host_mirror(tmp_path):
    # f.2 progressive output (render.rs:422-433): with a file sink the whole image is rewritten when the first bucket arrives
    # and then at most once per second -- observed here mid-render through the injectable clock: the file on disk is a
    # syntactically valid P6 whose finished buckets hold their pixels and whose unfinished ones are still zero
    now = [100.0]
    path = str(tmp_path / "p.tga")
    w = rta.PPMStdoutRGBABufferWriter(True, path, clock=lambda: now[0])
    w.begin(128, 128)
    tiles = [rta.ImageRegion(x, y + 64, x + 64, y) for y in (0, 64) for x in (0, 64)]
    fill = [17, 34, 51, 68]

    def on_disk():
        data = open(path, "rb").read()
        assert data.startswith(b"P6\n128 128\n255\n") and len(data) == 15 + 128 * 128 * 3
        return np.frombuffer(data[15:], dtype=np.uint8).reshape(128, 128, 3)

    w.write_rgba_buffer(rta.RGBABuffer(tiles[0], np.full((64, 64, 4), fill[0], dtype=np.uint8)))      # first bucket: written at once
    img = on_disk()
    assert (img[0:64, 0:64] == fill[0]).all() and (img[64:, :] == 0).all() and (img[0:64, 64:] == 0).all()
    now[0] += 0.5
    w.write_rgba_buffer(rta.RGBABuffer(tiles[1], np.full((64, 64, 4), fill[1], dtype=np.uint8)))      # within the second: not yet
    assert (on_disk()[0:64, 64:] == 0).all()
    now[0] += 0.6
    w.write_rgba_buffer(rta.RGBABuffer(tiles[2], np.full((64, 64, 4), fill[2], dtype=np.uint8)))      # a second has passed: rewritten
    img = on_disk()
    assert (img[0:64, 64:] == fill[1]).all() and (img[64:, 0:64] == fill[2]).all() and (img[64:, 64:] == 0).all()
    w.write_rgba_buffer(rta.RGBABuffer(tiles[3], np.full((64, 64, 4), fill[3], dtype=np.uint8)))
    assert (on_disk()[64:, 64:] == 0).all()
    w.close()                                                                                              # Drop writes the final image
    assert (on_disk()[64:, 64:] == fill[3]).all()


@pytest.mark.parametrize("size", [(1920, 1080), (800, 600), (4096, 4096), (64, 64)])
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_native_gang_sharding_equals_the_python_shard_layout(size, world):
    # rt_gang_render_frame(s) and dist.FrameSharder must deal a frame's buckets identically (bucket i -> GPU i % N, tile-major inside
    # the shard, shards padded to the longest): the native side's arithmetic is a pure function (rt_debug_gang_layout, no device)
    from rust_tracer_amd import dist
    o = rta.RenderOptions(size[0], size[1], 1)
    regs = [tuple(r) for r in rta.buckets(o)]
    dev, off, px, padded = capi.gang_layout(regs, world)
    bl, per_rank, shard_px = dist.shard_layout(o, world)
    assert padded == shard_px
    for r, (idx, offs, p) in enumerate(per_rank):
        assert list(dev[idx]) == [r] * len(idx)
        assert list(off[idx]) == offs
        assert int(px[r]) == p
    # the gathered table the root blits from
    regions, offsets, _ = dist.gathered_tile_table(o, world)
    mine = sorted(range(len(regs)), key=lambda i: (int(dev[i]), int(off[i])))
    assert [regs[i] for i in mine] == regions
    assert [int(dev[i]) * padded + int(off[i]) for i in mine] == [int(v) for v in offsets]


@pytest.mark.parametrize("precision", [rta.RT_F32, rta.RT_F64], ids=["f32", "f64"])
@pytest.mark.parametrize("eye", [None, (0.3, -0.2, -4.0)], ids=["split-order", "nearer-first"])
def test_the_hierarchy_builder_equals_its_numpy_restatement(precision, eye):
    # csrc/host/hierarchy.hpp is the ONE builder both hosts run (the Python host through the library's rt_build_hierarchy); scene.py keeps a plain
    # numpy restatement of its arithmetic -- near-minimal enclosing spheres by Badoiu-Clarkson steps, the nearer half first -- and the two
    # must agree bit for bit: items, bounds, ranges, order
    from rust_tracer_amd.scene import build_hierarchy, build_hierarchy_reference
    rng = np.random.default_rng(91)
    for n, leaf in ((1, 4), (5, 4), (777, 3), (3000, 4)):
        sp = np.concatenate([rng.uniform([-3, -2, 0], [3, 2, 6], (n, 3)), rng.uniform(0.01, 0.2, (n, 1))], axis=1).astype(np.float32).astype(np.float64)
        a = build_hierarchy(sp, leaf, precision, eye=eye)
        b = build_hierarchy_reference(sp, leaf, precision, eye=eye)
        for x, y in zip(a, b):
            assert x.dtype == y.dtype and np.array_equal(x, y), (n, leaf)
    # tighter than the box-centre spheres of round 5 (steps = 0 is that builder)
    tight = build_hierarchy_reference(sp, 4, precision, eye=eye)[1][:, 3].astype(np.float64)
    loose = build_hierarchy_reference(sp, 4, precision, eye=eye, steps=0)[1][:, 3].astype(np.float64)
    assert (tight ** 2).sum() < 0.95 * (loose ** 2).sum()            # (cross-sections: what a ray meets)


def test_native_and_python_hierarchy_builders_agree_bit_for_bit(tmp_path):
    # SURVEY.md 8f.4 on both hosts: `rtrace --scene <file>` (csrc/host/scene.hpp, Scene::from_file) and scene.py's build_hierarchy
    # must produce the same items, bounds and ranges -- host_tests --hierarchy prints CRCs of the C++ side's flattened arrays
    import subprocess
    import zlib
    rng = np.random.default_rng(77)
    n = 5000
    sp = np.concatenate([rng.uniform([-3, -2, 0], [3, 2, 6], (n, 3)), rng.uniform(0.01, 0.2, (n, 1))], axis=1).astype(np.float32).astype(np.float64)
    path = tmp_path / "scene.txt"
    with open(path, "w") as f:
        f.write("# %d random spheres\nlight -1 -3 2\n" % n)
        for s in sp:
            f.write("%r %r %r %r\n" % tuple(float(v) for v in s))
    exe = os.path.join(ROOT, "rust-tracer_amd", "host_tests")
    out = subprocess.run([exe, "--hierarchy", str(path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    from rust_tracer_amd.scene import build_hierarchy
    items, bounds, ranges, _ = build_hierarchy(sp, eye=(0.0, 0.0, -4.0))      # (Scene::from_file's default eye: the nearer half of a group first)
    want = "%d %d %d %d %d" % (items.shape[0], bounds.shape[0], zlib.crc32(items.tobytes()), zlib.crc32(bounds.tobytes()), zlib.crc32(ranges.tobytes()))
    assert out.stdout.strip() == want
    bad = tmp_path / "bad.txt"
    bad.write_text("1 2 3\n")
    assert subprocess.run([exe, "--hierarchy", str(bad)], capture_output=True, text=True, timeout=60).returncode == 2


def test_the_rccl_stand_in_exports_what_the_library_binds():
    # tests/c/fake_rccl.cpp (test infrastructure: several ranks of rt_gang_* on one GPU) must offer every entry point rt_capi.hip binds
    # from librccl.so, and the library must take it (and give it up again) through rt_debug_rccl_library without touching a device
    import ctypes
    import rust_tracer_amd as rta
    fake = ctypes.CDLL(rta.capi.FAKE_RCCL)
    for name in ("ncclCommInitAll", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclGather", "ncclGetErrorString", "ncclGetVersion"):
        assert hasattr(fake, name), name
    v = ctypes.c_int(-1)
    assert fake.ncclGetVersion(ctypes.byref(v)) == 0 and v.value == 0          # 0: not a real RCCL
    with rta.capi.rccl_stand_in():
        pass
    assert rta.capi.lib.rt_debug_rccl_library(b"/nonexistent/librccl.so") == rta.capi.RT_ERR_INVALID_ARGUMENT


def test_scene_setup_cost_rejects_null_arguments_without_a_device():
    # rt_scene_setup_cost (ABI 4, diagnostic): like rt_scene_traits it reads the handle only -- NULL is an argument error, not a crash
    import ctypes as C
    from rust_tracer_amd import capi
    total, stream = C.c_double(-1.0), C.c_double(-1.0)
    assert capi.lib.rt_scene_setup_cost(None, C.byref(total), C.byref(stream)) == capi.RT_ERR_INVALID_ARGUMENT
    assert b"NULL" in capi.lib.rt_last_error_message()
