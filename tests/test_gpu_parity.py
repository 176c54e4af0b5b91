"""Parity of the HIP path (through the C ABI) against the CPU oracle: bit-exact RGBA bytes and identical ray
counters, on the BASELINE configs, the reference's golden image, and the domain's edge cases.  Needs an MI355X."""
import json
import os
import threading
import zlib

import numpy as np
import pytest

import oracle
import rust_tracer_amd as rta
from tests import util

pytestmark = pytest.mark.gpu

FLAT = rta.RT_TRAVERSAL_FLAT


def full(w, h):
    return [(0, h, w, 0)]


def bucket_list(w, h, spp=1):
    return [tuple(r) for r in rta.buckets(rta.RenderOptions(w, h, spp))]


def test_device_present():
    assert rta.device_count() >= 1


@pytest.mark.parametrize("precision", [rta.RT_F32, rta.RT_F64])
def test_config1_three_spheres_single_tile(precision):
    # BASELINE config 1: single 64x64 tile, 3 spheres, 1 light
    s, o = util.scene_pair_spheres(util.THREE_SPHERES, util.THREE_BOUND, precision)
    data, st = s.device().render_tiles((64, 64, 1), [(0, 64, 64, 0)], traversal=FLAT)
    for mode in (oracle.MODE_HIERARCHY, oracle.MODE_FLAT):
        ref, rst = o.render_region(64, 64, 1, 0, 64, 64, 0, mode)
        np.testing.assert_array_equal(data.reshape(64, 64, 4), ref)
        assert util.ray_stats(st) == util.ray_stats(rst)
    assert st["sphere_tests"] == (st["primary"] + st["shadow"]) * 3


def test_default_scene_single_tile_and_counters():
    # SURVEY 8(d) row 1 companion: default scene 64x64 spp 1 -> 4096 / 2487 / 1923 / 909
    s, o = util.scene_pair_default()
    data, st = s.device().render_tiles((64, 64, 1), [(0, 64, 64, 0)], traversal=FLAT)
    ref, rst = o.render_region(64, 64, 1, 0, 64, 64, 0)
    np.testing.assert_array_equal(data.reshape(64, 64, 4), ref)
    assert util.ray_stats(st) == (4096, 2487, 1923, 909) == util.ray_stats(rst)


def test_reference_test_shape_64x128_spp2_two_buckets():
    # render::tests::basic_rendering render.rs:466-481: 64x128, spp 2 -> exactly 2 buckets
    s, o = util.scene_pair_default()
    regs = bucket_list(64, 128, 2)
    assert len(regs) == 2
    data, st = s.device().render_tiles((64, 128, 2), regs, traversal=FLAT)
    ref, rst, n = o.render(64, 128, 2, nthreads=2)
    assert n == 2
    np.testing.assert_array_equal(util.stitch((64, 128), regs, data), ref)
    assert util.ray_stats(st) == util.ray_stats(rst)


def test_golden_make_image_1024x768_spp4(golden_dir):
    # the reference's own output image (src/img/rtrace-output.png), RGB bit-exact; alpha against the oracle
    ref_rgb = np.load(os.path.join(golden_dir, "make_image_1024x768_spp4_rgb.npz"))["rgb"]
    meta = json.load(open(os.path.join(golden_dir, "make_image_1024x768_spp4_tiles.json")))
    s, o = util.scene_pair_default()
    regs = bucket_list(1024, 768, 4)
    assert len(regs) == 192
    data, st = s.device().render_tiles((1024, 768, 4), regs, traversal=FLAT)
    frame = util.stitch((1024, 768), regs, data)
    assert int((frame[:, :, :3] != ref_rgb).any(axis=2).sum()) == 0
    off = 0
    for i, (l, t, r, b) in enumerate(regs):
        tile = data[off:off + 64 * 64 * 4].reshape(64, 64, 4)
        assert zlib.crc32(np.ascontiguousarray(tile[:, :, :3]).tobytes()) & 0xFFFFFFFF == meta["tile_crc32"][i], i
        off += 64 * 64 * 4
    assert util.ray_stats(st) == (12582912, 9430527, 7211901, 3586443)
    oref, _, _ = o.render(1024, 768, 4, nthreads=os.cpu_count() or 1)
    np.testing.assert_array_equal(frame, oref)          # includes the alpha channel


def test_config2_800x600_clipped_edge_buckets():
    # BASELINE config 2: 13x10 = 130 buckets, edge buckets 32 wide / 24 tall (the reference itself asserts, H5)
    s, o = util.scene_pair_default()
    regs = bucket_list(800, 600)
    assert len(regs) == 130
    data, st = s.device().render_tiles((800, 600, 1), regs, traversal=FLAT)
    ref, rst, _ = o.render(800, 600, 1, nthreads=os.cpu_count() or 1)
    np.testing.assert_array_equal(util.stitch((800, 600), regs, data), ref)
    assert util.ray_stats(st) == (480000, 359528, 275032, 136797) == util.ray_stats(rst)


def test_config3_1920x1080_f32():
    # BASELINE config 3 (f32 side) / the bench workload: 510 buckets, last row 56 px
    s, o = util.scene_pair_default()
    regs = bucket_list(1920, 1080)
    assert len(regs) == 510
    data, st = s.device().render_tiles((1920, 1080, 1), regs, traversal=FLAT)
    ref, rst, _ = o.render(1920, 1080, 1, nthreads=os.cpu_count() or 1)
    np.testing.assert_array_equal(util.stitch((1920, 1080), regs, data), ref)
    assert util.ray_stats(st) == (2073600, 1777280, 1337403, 730313) == util.ray_stats(rst)
    assert st["sphere_tests"] == (2073600 + 1337403) * 21845


def test_config3_1920x1080_f64_type_alias_swap():
    # f64 is a separate golden (P7), pinned only by the oracle's own f64 instantiation ("parity unpinned")
    s, o = util.scene_pair_default(rta.RT_F64)
    regs = bucket_list(1920, 1080)
    data, st = s.device().render_tiles((1920, 1080, 1), regs, traversal=FLAT)
    ref, rst, _ = o.render(1920, 1080, 1, nthreads=os.cpu_count() or 1)
    np.testing.assert_array_equal(util.stitch((1920, 1080), regs, data), ref)
    assert util.ray_stats(st) == util.ray_stats(rst)


def test_whole_frame_as_one_region_equals_buckets():
    # a pixel does not depend on its tile (render.rs:217): one region == 130 stitched buckets
    s, _ = util.scene_pair_default()
    d = s.device()
    one, st1 = d.render_tiles((800, 600, 1), full(800, 600), traversal=FLAT)
    regs = bucket_list(800, 600)
    many, st2 = d.render_tiles((800, 600, 1), regs, traversal=FLAT)
    np.testing.assert_array_equal(one.reshape(600, 800, 4), util.stitch((800, 600), regs, many))
    assert util.ray_stats(st1) == util.ray_stats(st2)


def test_tile_order_and_duplicates_are_honoured():
    s, o = util.scene_pair_default()
    regs = [(64, 128, 128, 64), (0, 64, 64, 0), (64, 128, 128, 64), (130, 77, 131, 76)]
    data, st = s.device().render_tiles((256, 192, 2), regs, traversal=FLAT)
    off = 0
    for (l, t, r, b) in regs:
        ref, _ = o.render_region(256, 192, 2, l, t, r, b)
        n = ref.size
        np.testing.assert_array_equal(data[off:off + n].reshape(ref.shape), ref)
        off += n
    assert off == data.size
    assert st["primary"] == (64 * 64 * 3 + 1) * 4


@pytest.mark.parametrize("region", [(3, 19, 35, 2), (17, 18, 18, 17), (0, 5, 200, 0), (199, 150, 200, 0)])
def test_ragged_regions(region):
    # ImageRegion{l:2,t:18,r:34,b:2}-like odd rectangles, 1x1, single row, single column
    s, o = util.scene_pair_default()
    l, t, r, b = region
    data, st = s.device().render_tiles((200, 150, 1), [region], traversal=FLAT)
    ref, rst = o.render_region(200, 150, 1, l, t, r, b)
    np.testing.assert_array_equal(data.reshape(ref.shape), ref)
    assert util.ray_stats(st) == util.ray_stats(rst)


@pytest.mark.parametrize("precision", [rta.RT_F32, rta.RT_F64])
def test_tie_break_first_in_dfs_order_wins(precision):
    # primitive.rs:79 strict `>=` reject: of two items at exactly equal distance the first one keeps the hit
    for spheres in (util.TIE_SPHERES, util.TIE_SPHERES[::-1]):
        s, o = util.scene_pair_spheres(spheres, util.TIE_BOUND, precision)
        data, st = s.device().render_tiles((64, 64, 1), [(0, 64, 64, 0)], traversal=FLAT)
        ref, rst = o.render_region(64, 64, 1, 0, 64, 64, 0)
        np.testing.assert_array_equal(data.reshape(64, 64, 4), ref)
        assert util.ray_stats(st) == util.ray_stats(rst)
    a, _ = util.scene_pair_spheres(util.TIE_SPHERES, util.TIE_BOUND, precision)
    b, _ = util.scene_pair_spheres(util.TIE_SPHERES[::-1], util.TIE_BOUND, precision)
    da, _ = a.device().render_tiles((64, 64, 1), [(0, 64, 64, 0)], traversal=FLAT)
    db, _ = b.device().render_tiles((64, 64, 1), [(0, 64, 64, 0)], traversal=FLAT)
    col_a, col_b = da.reshape(64, 64, 4)[:, 32], db.reshape(64, 64, 4)[:, 32]
    assert (col_a != col_b).any(), "the tie column must depend on item order, or the scene does not exercise the rule"


def test_pyramid_level9_87381_items_chunk_boundaries():
    # config 5's scene size (L9 = 87,381 items = 85 LDS chunks + 341) on a small image
    s, o = util.scene_pair_default(level=9)
    assert s.items.shape[0] == 87381
    regs = bucket_list(192, 128, 2)
    data, st = s.device().render_tiles((192, 128, 2), regs, traversal=FLAT)
    ref, rst, _ = o.render(192, 128, 2, nthreads=os.cpu_count() or 1)
    np.testing.assert_array_equal(util.stitch((192, 128), regs, data), ref)
    assert util.ray_stats(st) == util.ray_stats(rst)


@pytest.mark.parametrize("n_items", [1, 2, 1023, 1024, 1025])
def test_item_counts_around_the_lds_chunk(n_items):
    rng = np.random.default_rng(n_items)
    sp = np.concatenate([rng.uniform(-1.5, 1.5, (n_items, 3)), rng.uniform(0.02, 0.12, (n_items, 1))], axis=1)
    sp = sp.astype(np.float32).astype(np.float64)
    s, o = util.scene_pair_spheres(sp, (0, 0, 0, 3.0))
    data, st = s.device().render_tiles((96, 80, 1), [(0, 80, 96, 0)], traversal=FLAT)
    ref, rst = o.render_region(96, 80, 1, 0, 80, 96, 0, oracle.MODE_FLAT)
    np.testing.assert_array_equal(data.reshape(ref.shape), ref)
    assert util.ray_stats(st) == util.ray_stats(rst)


def test_idempotent_and_stats_optional():
    s, _ = util.scene_pair_default()
    d = s.device()
    a, _ = d.render_tiles((320, 200, 1), full(320, 200), traversal=FLAT)
    b, none = d.render_tiles((320, 200, 1), full(320, 200), want_stats=False, traversal=FLAT)
    assert none is None
    np.testing.assert_array_equal(a, b)


def test_concurrent_callers_share_one_scene():
    # render_region is invoked concurrently from up to RTRACEMAXPROCS pool threads on the same scene (render.rs:283)
    s, o = util.scene_pair_default()
    d = s.device()
    regs = bucket_list(256, 256)
    out = [None] * len(regs)

    def work(i):
        out[i], _ = d.render_tiles((256, 256, 1), [regs[i]], traversal=FLAT)

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(regs))]
    [t.start() for t in th]
    [t.join() for t in th]
    ref, _, _ = o.render(256, 256, 1, nthreads=4)
    np.testing.assert_array_equal(util.stitch((256, 256), regs, np.concatenate(out)), ref)


def test_renderer_surface_writes_reference_ppm(tmp_path):
    # Renderer::render + PPMStdoutRGBABufferWriter: bytes of the file == oracle's PPM of the oracle frame
    s, o = util.scene_pair_default()
    opts = rta.RenderOptions(256, 192, 2)
    path = str(tmp_path / "out.tga")
    w = rta.PPMStdoutRGBABufferWriter(True, path)
    rta.Renderer.render(opts, s, w, pool=3, traversal=FLAT)
    w.close()
    ref, _, _ = o.render(256, 192, 2, nthreads=4)
    oracle.write_ppm(str(tmp_path / "ref.ppm"), ref)
    assert open(path, "rb").read() == open(str(tmp_path / "ref.ppm"), "rb").read()


def test_error_codes_instead_of_panics():
    s, _ = util.scene_pair_default()
    d = s.device()
    with pytest.raises(rta.RtError) as e:
        d.render_tiles((64, 64, 1), [(0, 65, 64, 0)], traversal=FLAT)          # outside the image
    assert e.value.status == rta.capi.RT_ERR_INVALID_REGION
    with pytest.raises(rta.RtError) as e:
        d.render_tiles((64, 64, 1), [(10, 20, 10, 0)], traversal=FLAT)         # empty
    assert e.value.status == rta.capi.RT_ERR_INVALID_REGION
    with pytest.raises(rta.RtError) as e:
        d.render_tiles((0, 64, 1), [(0, 64, 64, 0)], traversal=FLAT)           # an image without pixels (spp == 0 is the reference's black frame)
    assert e.value.status == rta.capi.RT_ERR_INVALID_ARGUMENT
    with pytest.raises(rta.RtError) as e:
        rta.Scene.from_spheres([(0, 0, 0, -1.0)], (0, 0, 0, 3.0)).device()
    assert e.value.status == rta.capi.RT_ERR_INVALID_ARGUMENT
    with pytest.raises(rta.RtError) as e:
        rta.Scene.from_spheres([(float("nan"), 0, 0, 1.0)], (0, 0, 0, 3.0)).device()
    assert e.value.status == rta.capi.RT_ERR_INVALID_ARGUMENT


# ----------------------------------------------------------------------------------------------------------
# RT_TRAVERSAL_SKIP: the reference's own hierarchy (group.rs:72-83) as a skip-pointer stream.  Pixels must be
# the hierarchy's, and every lane must perform exactly the reference's tests: the item/bound test counters equal
# the oracle's (with shadow rays stopping at their first hit, which cannot change a pixel).
# ----------------------------------------------------------------------------------------------------------
SKIP = rta.RT_TRAVERSAL_SKIP
HIER_EXIT = oracle.MODE_HIERARCHY | oracle.MODE_ANYHIT_EXIT


@pytest.mark.parametrize("precision", [rta.RT_F32, rta.RT_F64])
def test_skip_config1_three_spheres(precision):
    s, o = util.scene_pair_spheres(util.THREE_SPHERES, util.THREE_BOUND, precision)
    data, st = s.device().render_tiles((64, 64, 1), [(0, 64, 64, 0)], SKIP)
    ref, rst = o.render_region(64, 64, 1, 0, 64, 64, 0, HIER_EXIT)
    np.testing.assert_array_equal(data.reshape(64, 64, 4), ref)
    assert util.all_stats(st) == util.all_stats(rst)


def test_skip_golden_make_image_1024x768_spp4(golden_dir):
    ref_rgb = np.load(os.path.join(golden_dir, "make_image_1024x768_spp4_rgb.npz"))["rgb"]
    s, o = util.scene_pair_default()
    regs = bucket_list(1024, 768, 4)
    data, st = s.device().render_tiles((1024, 768, 4), regs, SKIP)
    frame = util.stitch((1024, 768), regs, data)
    assert int((frame[:, :, :3] != ref_rgb).any(axis=2).sum()) == 0
    oref, rst, _ = o.render(1024, 768, 4, os.cpu_count() or 1, HIER_EXIT)
    np.testing.assert_array_equal(frame, oref)
    assert util.all_stats(st) == util.all_stats(rst)


@pytest.mark.parametrize("precision", [rta.RT_F32, rta.RT_F64])
def test_skip_config3_1920x1080(precision):
    s, o = util.scene_pair_default(precision)
    regs = bucket_list(1920, 1080)
    data, st = s.device().render_tiles((1920, 1080, 1), regs, SKIP)
    ref, rst, _ = o.render(1920, 1080, 1, os.cpu_count() or 1, HIER_EXIT)
    np.testing.assert_array_equal(util.stitch((1920, 1080), regs, data), ref)
    assert util.all_stats(st) == util.all_stats(rst)


def test_skip_config2_800x600_and_flat_agree():
    s, o = util.scene_pair_default()
    regs = bucket_list(800, 600)
    a, sa = s.device().render_tiles((800, 600, 1), regs, SKIP)
    b, sb = s.device().render_tiles((800, 600, 1), regs, FLAT)
    np.testing.assert_array_equal(a, b)                       # SURVEY.md P3 on the GPU
    assert util.ray_stats(sa) == util.ray_stats(sb) == (480000, 359528, 275032, 136797)
    _, rst, _ = o.render(800, 600, 1, os.cpu_count() or 1, HIER_EXIT)
    assert util.all_stats(sa) == util.all_stats(rst)


def test_skip_pyramid_level9():
    s, o = util.scene_pair_default(level=9)
    regs = bucket_list(320, 256, 2)
    data, st = s.device().render_tiles((320, 256, 2), regs, SKIP)
    ref, rst, _ = o.render(320, 256, 2, os.cpu_count() or 1, HIER_EXIT)
    np.testing.assert_array_equal(util.stitch((320, 256), regs, data), ref)
    assert util.all_stats(st) == util.all_stats(rst)


@pytest.mark.parametrize("precision", [rta.RT_F32, rta.RT_F64])
def test_skip_reproduces_the_inside_bound_cull_flat_does_not(precision):
    # the reference's result is DEFINED by the hierarchy; SKIP follows it, FLAT is the flat semantics
    s, o = util.scene_pair_ranges(util.INSIDE_ITEMS, util.INSIDE_BOUNDS, util.INSIDE_RANGES, precision)
    hier, hst = o.render_region(64, 64, 1, 0, 64, 64, 0, HIER_EXIT)
    flat, _ = o.render_region(64, 64, 1, 0, 64, 64, 0, oracle.MODE_FLAT)
    assert (hier != flat).any()
    a, sa = s.device().render_tiles((64, 64, 1), [(0, 64, 64, 0)], SKIP)
    b, _ = s.device().render_tiles((64, 64, 1), [(0, 64, 64, 0)], FLAT)
    np.testing.assert_array_equal(a.reshape(64, 64, 4), hier)
    np.testing.assert_array_equal(b.reshape(64, 64, 4), flat)
    assert util.all_stats(sa) == util.all_stats(hst)


@pytest.mark.parametrize("seed", range(6))
def test_skip_random_nested_scenes_with_loose_and_tight_bounds(seed):
    # items and sub-groups interleaved, bounds that do not enclose their subtree: culling changes pixels, so this
    # checks the per-lane cull rule, the resume masking and the wave-level jump against the recursive oracle
    items, bounds, ranges = util.random_nested_scene(seed)
    s, o = util.scene_pair_ranges(items, bounds, ranges, eye=(0.1, 0.2, -4.0))
    regs = [(0, 96, 128, 0), (5, 70, 69, 6)]
    data, st = s.device().render_tiles((128, 96, 2), regs, SKIP)
    off = 0
    tot = None
    for (l, t, r, b) in regs:
        ref, rst = o.render_region(128, 96, 2, l, t, r, b, HIER_EXIT)
        np.testing.assert_array_equal(data[off:off + ref.size].reshape(ref.shape), ref)
        off += ref.size
        tot = rst if tot is None else {k: tot[k] + rst[k] for k in tot}
    assert util.all_stats(st) == util.all_stats(tot)


def test_skip_tie_break_and_ragged():
    for spheres in (util.TIE_SPHERES, util.TIE_SPHERES[::-1]):
        s, o = util.scene_pair_spheres(spheres, util.TIE_BOUND)
        data, st = s.device().render_tiles((64, 64, 1), [(3, 61, 64, 1)], SKIP)
        ref, rst = o.render_region(64, 64, 1, 3, 61, 64, 1, HIER_EXIT)
        np.testing.assert_array_equal(data.reshape(ref.shape), ref)
        assert util.all_stats(st) == util.all_stats(rst)


def test_skip_needs_bounds_and_rejects_bad_nesting():
    s = rta.Scene(np.array([[0, 0, 0, 1.0]]), rta.normalized((-1, -3, 2)), (0, 0, -4))
    with pytest.raises(rta.RtError) as e:
        s.device().render_tiles((64, 64, 1), [(0, 64, 64, 0)], SKIP)
    assert e.value.status == rta.capi.RT_ERR_UNSUPPORTED
    bad = rta.Scene(np.array([[0, 0, 0, 1.0], [1, 0, 0, 1.0], [2, 0, 0, 1.0]]), rta.normalized((-1, -3, 2)), (0, 0, -4),
                    np.array([[0, 0, 0, 9.0], [0, 0, 0, 9.0], [0, 0, 0, 9.0]]), np.array([[0, 3], [0, 2], [1, 2]], dtype=np.int32))
    with pytest.raises(rta.RtError) as e:
        bad.device()
    assert e.value.status == rta.capi.RT_ERR_INVALID_ARGUMENT


def test_lean_sqrt_is_correctly_rounded_for_every_f32():
    # the traversal loops use a lean correctly-rounded sqrt; it must equal the IEEE sqrt bit for bit on all 2^32 inputs
    assert rta.Scene.default(4).device().traits() == rta.capi.RT_SCENE_HAS_BOUNDS | rta.capi.RT_SCENE_CONCENTRIC
    bad, first = rta.capi.selftest_sqrt(0)
    assert bad == 0, "first differing input bits: 0x%08x" % first


def test_lean_reciprocal_is_correctly_rounded_for_every_f32():
    # Vector::normalized multiplies by len.recip(): the kernels' 3-instruction reciprocal must equal 1.0f / x on all 2^32 inputs
    bad, first = rta.capi.selftest_rcp(0)
    assert bad == 0, "first differing input bits: 0x%08x" % first


# ----------------------------------------------------------------------------------------------------------
# Committed vectors (tests/golden/oracle_vectors.json): the expected bytes are data in the repository, so these
# checks do not depend on running the oracle at test time.
# ----------------------------------------------------------------------------------------------------------
def _vector_cases():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.json")
    return json.load(open(path))["cases"]


def _scene_for(case):
    prec = rta.RT_F64 if case["precision"] == "f64" else rta.RT_F32
    if case["scene"] == "default8":
        return rta.Scene.default(8, prec)
    if case["scene"] == "default9":
        return rta.Scene.default(9, prec)
    if case["scene"] == "three_spheres":
        return rta.Scene.three_spheres(prec)
    if case["scene"] == "tie":
        return rta.Scene.from_spheres(util.TIE_SPHERES, util.TIE_BOUND, precision=prec)
    if case["scene"] == "inside":
        return util.scene_pair_ranges(util.INSIDE_ITEMS, util.INSIDE_BOUNDS, util.INSIDE_RANGES, prec)[0]
    if case["scene"] == "hundred_thousand_spheres":
        from tests.scenes import hundred_thousand_spheres
        return rta.Scene.from_spheres_auto(hundred_thousand_spheres(), precision=prec)
    raise KeyError(case["scene"])


@pytest.mark.parametrize("case", _vector_cases(), ids=lambda c: c["name"])
@pytest.mark.parametrize("trav", [SKIP, FLAT], ids=["skip", "flat"])
def test_committed_vectors(case, trav):
    if case["scene"] == "inside" and (trav == FLAT) != bool(case.get("flat")):
        pytest.skip("the inside-bound scene has one golden per traversal semantics")
    if case["scene"] == "hundred_thousand_spheres" and trav == FLAT:
        pytest.skip("tight automatic bounds: the flat semantics differ from the hierarchy's by a byte (test_100k_arbitrary_spheres...)")
    s = _scene_for(case)
    w, h, spp = case["width"], case["height"], case["spp"]
    regs = bucket_list(w, h, spp)
    assert len(regs) == case["buckets"]
    data, st = s.device().render_tiles((w, h, spp), regs, trav)
    off = 0
    for i, (l, t, r, b) in enumerate(regs):
        n = (r - l) * (t - b) * 4
        assert zlib.crc32(data[off:off + n].tobytes()) & 0xFFFFFFFF == case["tile_crc32"][i], (case["name"], i)
        off += n
    for k in ("primary", "hits", "shadow", "occluded"):
        assert st[k] == case["stats"][k], k
    if trav == SKIP:
        assert (st["sphere_tests"], st["bound_tests"]) == (case["stats"]["sphere_tests"], case["stats"]["bound_tests"])


@pytest.mark.parametrize("trav", [SKIP, FLAT], ids=["skip", "flat"])
@pytest.mark.parametrize("size", [(800, 600, 1), (320, 200, 2)])
def test_render_frame_device_equals_tiles_plus_blit(trav, size):
    # rt_render_frame_device == rt_render_tiles_device + rt_blit_tiles_device == host stitch of rt_render_tiles
    import torch
    w, h, spp = size
    s, _ = util.scene_pair_default()
    d = s.device()
    regs = bucket_list(w, h, spp)
    host, _ = d.render_tiles((w, h, spp), regs, trav)
    expect = util.stitch((w, h), regs, host)
    stream = torch.cuda.current_stream().cuda_stream
    frame = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    st = d.render_frame_device((w, h, spp), regs, frame.data_ptr(), stream, trav, want_stats=True)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(frame.cpu().numpy().reshape(h, w, 4), expect)
    assert st["primary"] == w * h * spp * spp
    shard = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    frame2 = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    d.render_tiles_device((w, h, spp), regs, shard.data_ptr(), stream, trav)
    d.blit_tiles_device((w, h, spp), regs, shard.data_ptr(), frame2.data_ptr(), stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(frame2.cpu().numpy().reshape(h, w, 4), expect)


def test_the_largest_frame_the_reference_accepts():
    # RenderOptions holds u16 sizes that must be multiples of 64 (render.rs:35-36, 265-266): 65,472 x 65,472 = 4,286,582,784 pixels, a 17 GB
    # frame whose byte offsets need 33 bits and whose pixel count still fits 32.  A pixel depends on (x, y, width, height) alone, so every
    # bucket of the big frame must equal the same bucket rendered by itself (rt_render_tiles, one region), and that one the oracle's
    # render_region at this frame size.  Sixty buckets: the corners, the rows in which the byte offset passes 4 / 8 / 12 GiB, the pyramid, anywhere.
    import torch
    W = H = 65472
    if torch.cuda.mem_get_info()[0] < (W * H * 4) * 1.2:
        pytest.skip("needs 20 GB of free device memory")
    s, o = util.scene_pair_default()
    d = s.device()
    opts = (W, H, 1)
    regs = bucket_list(W, H)
    assert len(regs) == 1023 * 1023
    frame = torch.full((W * H * 4,), 7, dtype=torch.uint8, device="cuda")
    st = d.render_frame_device(opts, d._regions(regs), frame.data_ptr(), torch.cuda.current_stream().cuda_stream, SKIP, want_stats=True)
    torch.cuda.synchronize()
    assert st["primary"] == W * H
    f2 = frame.view(H, W, 4)
    rng = np.random.default_rng(3)
    nb = W // 64
    picks = {(0, 0), (nb - 1, 0), (0, nb - 1), (nb - 1, nb - 1), (nb // 2, nb // 2), (nb // 2 - 1, nb // 2), (nb // 2, nb // 2 + 1)}
    row_4g = ((1 << 30) // W) // 64
    for yy in (row_4g - 1, row_4g, row_4g + 1, 2 * row_4g, 3 * row_4g):
        picks.add((int(rng.integers(0, nb)), yy)); picks.add((nb // 2, yy))
    while len(picks) < 60:
        if rng.random() < 0.6:
            picks.add((int(nb // 2 + rng.integers(-nb // 6, nb // 6 + 1)), int(nb // 2 + rng.integers(-nb // 6, nb // 6 + 1))))
        else:
            picks.add((int(rng.integers(0, nb)), int(rng.integers(0, nb))))
    shows_geometry = 0
    for (bx, by) in sorted(picks):
        l, b = bx * 64, by * 64
        tile, _ = d.render_tiles(opts, [(l, b + 64, l + 64, b)], SKIP, want_stats=False)
        ref = tile.reshape(64, 64, 4)
        np.testing.assert_array_equal(f2[b:b + 64, l:l + 64].cpu().numpy(), ref, err_msg="bucket %d, %d" % (bx, by))
        shows_geometry += int((ref[..., :3] != ref[0, 0, :3]).any())
        # ... and the bucket by itself is the CPU path's bucket at this frame size (a 64x64 bucket costs the oracle milliseconds)
        oref, _ = o.render_region(W, H, 1, l, b + 64, l + 64, b, HIER_EXIT)
        np.testing.assert_array_equal(ref, oref, err_msg="bucket %d, %d against the oracle" % (bx, by))
    assert shows_geometry >= 10
    del f2, frame
    torch.cuda.empty_cache()


def test_frame_mode_leaves_unlisted_buckets_alone():
    import torch
    s, _ = util.scene_pair_default()
    d = s.device()
    regs = bucket_list(256, 192)
    frame = torch.full((256 * 192 * 4,), 7, dtype=torch.uint8, device="cuda")
    d.render_frame_device((256, 192, 1), regs[::2], frame.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = frame.cpu().numpy().reshape(192, 256, 4)
    host, _ = d.render_tiles((256, 192, 1), regs)
    full = util.stitch((256, 192), regs, host)
    for i, (l, t, r, b) in enumerate(regs):
        if i % 2 == 0:
            np.testing.assert_array_equal(got[b:t, l:r], full[b:t, l:r])
        else:
            assert (got[b:t, l:r] == 7).all()


def test_many_async_calls_in_flight_reuse_nothing_they_should_not():
    # back-to-back asynchronous passes with different tile lists on one stream: results must not interfere
    import torch
    s, o = util.scene_pair_default()
    d = s.device()
    stream = torch.cuda.current_stream().cuda_stream
    jobs = []
    for k in range(12):
        regs = bucket_list(128 + 64 * (k % 3), 128)
        buf = torch.zeros(sum((r - l) * (t - b) for l, t, r, b in regs) * 4, dtype=torch.uint8, device="cuda")
        d.render_tiles_device((128 + 64 * (k % 3), 128, 1 + (k % 2)), regs, buf.data_ptr(), stream, SKIP)
        jobs.append((128 + 64 * (k % 3), 1 + (k % 2), regs, buf))
    torch.cuda.synchronize()
    for w, spp, regs, buf in jobs:
        ref, _, _ = o.render(w, 128, spp, nthreads=4)
        np.testing.assert_array_equal(util.stitch((w, 128), regs, buf.cpu().numpy()), ref)


def test_repeated_and_concurrent_launches_are_bitwise_stable():
    # the assembly traversal loops must give the same bytes on every launch, also with several frames in flight on
    # different streams (different co-residency / timing): guards against an instruction-hazard slip in the assembly
    import torch
    s, o = util.scene_pair_default()
    d = s.device()
    regs = bucket_list(1920, 1080)
    ref, _ = d.render_tiles((1920, 1080, 1), regs, SKIP)
    want = zlib.crc32(ref.tobytes())
    streams = [torch.cuda.Stream() for _ in range(4)]
    bufs = [torch.zeros(1920 * 1080 * 4, dtype=torch.uint8, device="cuda") for _ in range(4)]
    for rnd in range(12):
        for st, buf in zip(streams, bufs):
            buf.zero_()
        torch.cuda.synchronize()
        for st, buf in zip(streams, bufs):
            d.render_tiles_device((1920, 1080, 1), regs, buf.data_ptr(), st.cuda_stream, SKIP)
        torch.cuda.synchronize()
        for buf in bufs:
            assert zlib.crc32(buf.cpu().numpy().tobytes()) == want, rnd


def test_config5_4096x4096_level9_spp4_selected_buckets():
    # BASELINE config 5's shape (87,381 spheres, 16 samples/px): the whole 4096x4096 frame is rendered in one pass
    # (268 M samples through the sample-parallel path); a spread of buckets is compared with the oracle
    s, o = util.scene_pair_default(level=9)
    opts = (4096, 4096, 4)
    regs = bucket_list(4096, 4096, 4)
    assert len(regs) == 4096
    data, st = s.device().render_tiles(opts, regs, SKIP, want_stats=False)
    assert st is None and data.size == 4096 * 4096 * 4
    for i in (0, 31 * 64 + 31, 32 * 64 + 32, 20 * 64 + 33, 45 * 64 + 30, 63 * 64 + 63, 40 * 64 + 12):
        l, t, r, b = regs[i]
        ref, _ = o.render_region(4096, 4096, 4, l, t, r, b)
        np.testing.assert_array_equal(data[i * 16384:(i + 1) * 16384].reshape(64, 64, 4), ref)
    # property at full size: the same frame as 16 regions of 1024x1024 gives the same pixels
    big = [(x, y + 1024, x + 1024, y) for y in range(0, 4096, 1024) for x in range(0, 4096, 1024)]
    data2, _ = s.device().render_tiles(opts, big, SKIP, want_stats=False)
    np.testing.assert_array_equal(util.stitch((4096, 4096), regs, data), util.stitch((4096, 4096), big, data2))


@pytest.mark.parametrize("precision", [rta.RT_F32, rta.RT_F64])
@pytest.mark.parametrize("trav", [SKIP, FLAT], ids=["skip", "flat"])
def test_tangent_ray_zero_discriminant(precision, trav):
    # pixel (32, 32) of a 64x64 image looks exactly along +z; for a sphere at (0.5, 0, -2) r 0.5 and the eye at (0, 0, -4)
    # the discriminant is EXACTLY 0 (4 - 4.25 + 0.25): the tangent hit goes through the small-input branch of the exact
    # square root (the scaled path of the assembly loops)
    spheres = [(0.5, 0.0, -2.0, 0.5), (-3.0, 2.0, 1.0, 0.75)]
    s, o = util.scene_pair_spheres(spheres, (0.0, 0.0, 0.0, 6.0), precision)
    d, pos = o.intersect((0, 0, -4, 0, 0, 1))
    assert d == 2.0                                         # the oracle agrees this ray grazes the first sphere at t = 2
    data, st = s.device().render_tiles((64, 64, 1), [(0, 64, 64, 0)], trav)
    ref, rst = o.render_region(64, 64, 1, 0, 64, 64, 0, HIER_EXIT if trav == SKIP else oracle.MODE_FLAT)
    np.testing.assert_array_equal(data.reshape(64, 64, 4), ref)
    assert util.ray_stats(st) == util.ray_stats(rst)
    assert data.reshape(64, 64, 4)[32, 32, 3] in (0, 255)   # the pixel is a hit (lit or shadowed), not background
    assert tuple(data.reshape(64, 64, 4)[32, 32, :3]) != (34, 10, 10)


@pytest.mark.parametrize("trav", [SKIP, FLAT], ids=["skip", "flat"])
def test_more_distinct_tile_lists_than_the_table_cache_holds(trav):
    # the reference calls render_region once per bucket: 48 different single-bucket lists on one scene must all work
    # (the device tile-table cache holds 32; beyond that tables are uploaded per call)
    s, o = util.scene_pair_default()
    d = s.device()
    regs = bucket_list(512, 384)
    assert len(regs) == 48
    ref, _, _ = o.render(512, 384, 1, nthreads=4)
    for (l, t, r, b) in regs + regs[:8]:
        data, _ = d.render_tiles((512, 384, 1), [(l, t, r, b)], trav, want_stats=False)
        np.testing.assert_array_equal(data.reshape(t - b, r - l, 4), ref[b:t, l:r])


@pytest.mark.parametrize("variant", [0, 1, 3, 7, 19, 23])
@pytest.mark.parametrize("concentric", [False, True], ids=["nested", "concentric"])
@pytest.mark.parametrize("precision", [rta.RT_F32, rta.RT_F64], ids=["f32", "f64"])
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_every_skip_loop_flavour_on_deep_random_scenes(variant, concentric, precision, seed):
    # rt_debug_set(RT_DEBUG_SKIP_VARIANT) picks the traversal-loop flavour explicitly (0/1 C++ loops, 3 generated assembly loops, 7 their fused
    # flavour -- used for concentric scenes only, the library drops the bit otherwise; the default is 7).  Deeper, wider
    # random trees with loose bounds and an eye inside some bounds: culling decides pixels, many lanes retire at different
    # items in the shadow walk.
    items, bounds, ranges = util.random_nested_scene(seed, depth=4, fan=4, leaf_items=2, concentric=concentric)
    s, o = util.scene_pair_ranges(items, bounds, ranges, precision, eye=(0.05, -0.1, -2.2))
    assert s.device().traits() == rta.capi.RT_SCENE_HAS_BOUNDS | (rta.capi.RT_SCENE_CONCENTRIC if concentric else 0)
    regs = bucket_list(192, 160, 2)
    ref, rst, _ = o.render(192, 160, 2, os.cpu_count() or 1, HIER_EXIT)
    with util.loop_flavour(variant):
        plain, _ = s.device().render_tiles((192, 160, 2), regs, SKIP, want_stats=False)
        counted, st = s.device().render_tiles((192, 160, 2), regs, SKIP, want_stats=True)
    np.testing.assert_array_equal(util.stitch((192, 160), regs, plain), ref)
    np.testing.assert_array_equal(counted, plain)
    assert util.all_stats(st) == util.all_stats(rst)


@pytest.mark.parametrize("size", [(1920, 1080, 1), (200, 1000, 1), (333, 77, 2)])
def test_block_dispatch_order_never_changes_a_pixel(size):
    # The library dispatches a pass's 16x16 blocks most-expensive-first (cost map rendered once per scene).  Same pixels
    # and same counters with the ordering switched off, for landscape, portrait (rows beyond the square cost map are
    # clamped) and ragged sizes, and for tile lists given in a scrambled order.
    w, h, spp = size
    scene = rta.Scene.default(6)
    d = scene.device()
    regs = bucket_list(w, h, spp)
    rng = np.random.default_rng(5)
    scrambled = [regs[i] for i in rng.permutation(len(regs))]
    out = {}
    for flag in (b"0", b"1"):
        with util.control(rta.capi.DEBUG_BLOCK_ORDER, int(flag)):
            a, sa = d.render_tiles((w, h, spp), regs, SKIP, want_stats=True)
            b, _ = d.render_tiles((w, h, spp), scrambled, SKIP, want_stats=False)
        out[flag] = (util.stitch((w, h), regs, a), util.stitch((w, h), scrambled, b), util.all_stats(sa))
    np.testing.assert_array_equal(out[b"0"][0], out[b"1"][0])
    np.testing.assert_array_equal(out[b"0"][1], out[b"1"][1])
    np.testing.assert_array_equal(out[b"1"][0], out[b"1"][1])
    assert out[b"0"][2] == out[b"1"][2]
    o = oracle.Scene.default(level=6)
    ref, _, _ = o.render(w, h, spp, os.cpu_count() or 1, HIER_EXIT)
    np.testing.assert_array_equal(out[b"1"][0], ref)


def test_100k_arbitrary_spheres_with_auto_built_hierarchy():
    # BASELINE config 5 asks for "100k spheres"; the pyramid only offers 87,381 -- SURVEY.md 8f.4: an arbitrary list of
    # exactly 100,000 spheres with an automatically built bounding-sphere hierarchy (65,535 groups, depth 16)
    rng = np.random.default_rng(5)
    n = 100000
    sp = np.concatenate([rng.uniform([-3, -2, 0], [3, 2, 6], (n, 3)), rng.uniform(0.01, 0.03, (n, 1))], axis=1)
    sp = sp.astype(np.float32).astype(np.float64)
    s = rta.Scene.from_spheres_auto(sp)
    assert s.items.shape[0] == n
    o = oracle.Scene.from_ranges(s.items.astype(np.float64), s.bounds.astype(np.float64), s.ranges)
    regs = bucket_list(512, 384)
    skip, st = s.device().render_tiles((512, 384, 1), regs, SKIP)
    ref, rst, _ = o.render(512, 384, 1, os.cpu_count() or 1, HIER_EXIT)
    np.testing.assert_array_equal(util.stitch((512, 384), regs, skip), ref)
    assert util.all_stats(st) == util.all_stats(rst)
    # FLAT has the flat semantics (every item, no culling).  With bounds this tight it is NOT bitwise the hierarchy: the
    # f32 cull test `bound.distance >= hit.distance` can reject a group whose item is nearer by less than the rounding of
    # the two distances (1 byte of 786,432 differs on this scene).  SKIP above is the reference; FLAT must equal the flat
    # restatement.
    flat, fst = s.device().render_tiles((512, 384, 1), regs, FLAT)
    frame = util.stitch((512, 384), regs, flat)
    assert int((frame != ref).sum()) <= 8
    c = rta.capi.flat_filter_check(s.device()._h, 160, 120, 1)            # 100,000 items x every ray of a small frame, pair by pair
    assert c[0] > 10_000 and c[2] == 0 and c[5] == 0, c
    for (l, t, r, b) in ((192, 256, 256, 192), (320, 128, 384, 64)):
        fref, _ = o.render_region(512, 384, 1, l, t, r, b, oracle.MODE_FLAT)
        np.testing.assert_array_equal(frame[b:t, l:r], fref)


@pytest.mark.parametrize("narrow_max", [0, 64])
def test_narrow_blocks_on_ragged_tiles(narrow_max):
    # The most expensive 16x16 blocks of a pass go out as four (or, in a small pass, sixteen) narrow workgroups.  Here the
    # expensive part of the image is covered by ragged 50x50 tiles (their last block row / column is clipped to 2 pixels), and
    # up to 64 of the 256 blocks are narrowed: every pixel and every counter must equal the CPU path's, and the un-narrowed
    # launch's.
    s, o = rta.Scene.default(), oracle.Scene.default()      # a fresh device scene: its table cache has not seen this tile list
    w, h = 1920, 1080
    regs = [(x, y + 50, x + 50, y) for y in range(440, 640, 50) for x in range(860, 1060, 50)]
    with util.control(rta.capi.DEBUG_NARROW_MAX, narrow_max):
        d = s.device()
        plain, _ = d.render_tiles((w, h, 1), regs, SKIP, want_stats=False)
        counted, st = d.render_tiles((w, h, 1), regs, SKIP, want_stats=True)
    np.testing.assert_array_equal(counted, plain)
    off, tot = 0, None
    for (l, t, r, b) in regs:
        ref, rst = o.render_region(w, h, 1, l, t, r, b, HIER_EXIT)
        n = (r - l) * (t - b) * 4
        np.testing.assert_array_equal(plain[off:off + n].reshape(t - b, r - l, 4), ref)
        off += n
        tot = util.all_stats(rst) if tot is None else tuple(a + c for a, c in zip(tot, util.all_stats(rst)))
    assert util.all_stats(st) == tot


@pytest.mark.parametrize("precision", [rta.RT_F32, rta.RT_F64], ids=["f32", "f64"])
@pytest.mark.parametrize("spp", [2, 3, 4, 5, 8])
def test_every_sample_count_on_a_ragged_image(spp, precision):
    # spp 2 / 4 / 8 take the packed sample-parallel mapping (a wave = the samples of a few neighbouring pixels, samples stored
    # [pixel][sample]); 3 and 5 the plain one (a wave = one sample of 8x8 pixels).  150x70 leaves clipped buckets, blocks and
    # sub-blocks on both edges.  Pixels, alpha and every counter against the CPU path; switching the packed mapping off (rt_debug.h) must not
    # change a byte.
    s, o = util.scene_pair_default(precision)
    w, h = 150, 70
    regs = bucket_list(w, h, spp)
    ref, rst, _ = o.render(w, h, spp, os.cpu_count() or 1, HIER_EXIT)
    data, st = s.device().render_tiles((w, h, spp), regs, SKIP)
    np.testing.assert_array_equal(util.stitch((w, h), regs, data), ref)
    assert util.all_stats(st) == util.all_stats(rst)
    with util.control(rta.capi.DEBUG_PACKED_SAMPLES, 0):
        plain, _ = s.device().render_tiles((w, h, spp), regs, SKIP, want_stats=False)
    np.testing.assert_array_equal(plain, data)


@pytest.mark.parametrize("traversal", [SKIP, FLAT], ids=["skip", "flat"])
@pytest.mark.parametrize("spp,precision", [(16, rta.RT_F32), (255, rta.RT_F32), (256, rta.RT_F32), (300, rta.RT_F32), (16, rta.RT_F64), (256, rta.RT_F64)],
                         ids=["16-f32", "255-f32", "256-f32", "300-f32", "16-f64", "256-f64"])
def test_very_many_samples_per_pixel(spp, precision, traversal):
    # samples_per_pixel is a u16 (render.rs:37).  Up to 255 (spp^2 <= 65,535) a launch is sample-parallel with one grid row per sample; from
    # 256 on every thread loops over its pixel's samples in the reference's order.  A 12x6 region across the pyramid's silhouette in a 64x64
    # frame of the level-4 pyramid (so that samples hit, miss and are shadowed inside single pixels): pixels, alpha and every counter.
    if traversal == FLAT and spp >= 256:
        pytest.skip("the flat scan takes at most 65,535 samples per pixel and says so (test_flat_scan_sample_limit)")
    s, o = util.scene_pair_default(precision, 4)
    w = h = 64
    l, t, r, b = 26, 40, 38, 34
    ref, rst = o.render_region(w, h, spp, l, t, r, b, HIER_EXIT if traversal == SKIP else oracle.MODE_FLAT)
    assert len(np.unique(ref.reshape(-1, 4), axis=0)) > 8          # not a flat patch
    data, st = s.device().render_tiles((w, h, spp), [(l, t, r, b)], traversal)
    np.testing.assert_array_equal(data.reshape(t - b, r - l, 4), ref)
    if traversal == SKIP:
        assert util.all_stats(st) == util.all_stats(rst)
    else:
        assert (st["primary"], st["hits"], st["shadow"], st["occluded"]) == (rst["primary"], rst["hits"], rst["shadow"], rst["occluded"])


@pytest.mark.parametrize("precision", [rta.RT_F32, rta.RT_F64], ids=["f32", "f64"])
@pytest.mark.parametrize("traversal", [SKIP, FLAT], ids=["skip", "flat"])
def test_zero_samples_per_pixel_is_the_references_black_frame(traversal, precision):
    # render.rs:219-250 with samples_per_pixel = 0: the sample loops do not run, g = 0 * inf = NaN, alpha likewise, and set_pixel_from_vector's
    # `r as u8` turns NaN into 0 -- every listed pixel {0, 0, 0, 0}, no ray cast.  Through every entry point that takes RenderOptions.
    import torch
    s, o = util.scene_pair_default(precision, 4)
    d = s.device()
    w, h = 150, 70
    regs = bucket_list(w, h)
    ref, rst, _ = o.render(w, h, 0, 2, HIER_EXIT if traversal == SKIP else oracle.MODE_FLAT)
    assert not ref.any() and rst["primary"] == 0
    buf = np.full(w * h * 4, 9, dtype=np.uint8)
    data, st = d.render_tiles((w, h, 0), regs, traversal, out=buf)
    assert not data.any() and util.all_stats(st) == (0, 0, 0, 0, 0, 0)
    one, _ = d.render_region((w, h, 0), regs[1], traversal)
    assert not one.any() and one.shape == (regs[1][1] - regs[1][3], regs[1][2] - regs[1][0], 4)
    # frame mode: only the listed buckets are touched
    frame = torch.full((w * h * 4,), 7, dtype=torch.uint8, device="cuda")
    d.render_frame_device((w, h, 0), regs[::2], frame.data_ptr(), torch.cuda.current_stream().cuda_stream, traversal)
    torch.cuda.synchronize()
    got = frame.cpu().numpy().reshape(h, w, 4)
    for i, (l, t, r, b) in enumerate(regs):
        assert (got[b:t, l:r] == (0 if i % 2 == 0 else 7)).all()
    seen = []
    d.render_tiles_stream((w, h, 0), regs, lambda i, reg, rgba: seen.append((i, bool(np.asarray(rgba).any()))), traversal)
    assert sorted(seen) == [(i, False) for i in range(len(regs))]
    p6 = np.full(w * h * 3, 5, dtype=np.uint8)
    d.render_frame_stream((w, h, 0), regs, rta.capi.RT_FRAME_RGB, p6, None, traversal)
    assert not p6.any()


def test_python_host_degenerate_options(tmp_path):
    # render.py, the thin mirror: no samples -> every bucket arrives black (render.rs:219-250); no pixels -> no bucket, a writer that never
    # gets dirty writes nothing (render.rs:361-363)
    s, _ = util.scene_pair_default(rta.RT_F32, 4)
    path = str(tmp_path / "black.tga")
    w = rta.PPMStdoutRGBABufferWriter(True, path)
    rta.Renderer.render(rta.RenderOptions(192, 128, 0), s, w, pool=2)
    w.close()
    assert open(path, "rb").read() == b"P6\n192 128\n255\n" + bytes(192 * 128 * 3)
    for wh in ((0, 64), (64, 0)):
        path = str(tmp_path / "empty.tga")
        w = rta.PPMStdoutRGBABufferWriter(True, path)
        rta.Renderer.render(rta.RenderOptions(wh[0], wh[1], 1), s, w)
        w.close()
        assert not os.path.exists(path)          # (the Python writer opens its file when it first writes; main.rs creates it before: the CLI test sees it empty)


def test_flat_scan_sample_limit():
    s, _ = util.scene_pair_default(rta.RT_F32, 4)
    with pytest.raises(rta.capi.RtError, match="too many samples"):
        s.device().render_tiles((64, 64, 256), [(0, 64, 64, 0)], FLAT)


# ---------------------------------------------------------------- host-buffer boundary (round 2)
def test_host_buffers_pinned_registered_and_pageable_deliver_the_same_bytes():
    # rt_render_tiles recognises rt_host_alloc'd / rt_host_register'd memory by address and lets the kernel store into it
    # directly; pageable memory takes a copy.  Every copy strategy (csrc/rt_debug.h RT_DEBUG_HOST_COPY) must deliver the same bytes.
    s, o = util.scene_pair_default()
    d = s.device()
    w, h = 320, 200
    regs = bucket_list(w, h)
    ref, _, _ = o.render(w, h, 1, nthreads=4)
    n = w * h * 4
    pinned = rta.capi.HostBuffer(n)
    registered = np.zeros(n + 8192, dtype=np.uint8)
    page = (registered.ctypes.data + 4095) & ~4095          # register whole pages
    reg_view = registered[page - registered.ctypes.data:][:n]
    rta.capi.check(rta.capi.lib.rt_host_register(reg_view.ctypes.data, n), "rt_host_register")
    try:
        for buf, modes in ((np.zeros(n, dtype=np.uint8), (0, 1, 2)), (pinned.array, (0, 1, 3)), (reg_view, (0, 1, 3))):
            for mode in modes:
                buf[:] = 0
                with rta.capi.debug(rta.capi.DEBUG_HOST_COPY, mode if mode else -1):
                    data, _ = d.render_tiles((w, h, 1), regs, SKIP, want_stats=False, out=buf)
                np.testing.assert_array_equal(util.stitch((w, h), regs, data), ref)
        # an interior pointer of a pinned range is recognised too
        pinned.array[:] = 0
        data, _ = d.render_tiles((w, h, 1), regs[1:], SKIP, want_stats=False, out=pinned.array[64 * 64 * 4:])
        np.testing.assert_array_equal(data, util_tile_major(ref, regs[1:]))
    finally:
        rta.capi.check(rta.capi.lib.rt_host_unregister(reg_view.ctypes.data), "rt_host_unregister")
        pinned.close()


def util_tile_major(frame, regs):
    return np.concatenate([np.ascontiguousarray(frame[b:t, l:r]).reshape(-1) for (l, t, r, b) in regs])


def test_render_tiles_rejects_a_device_pointer():
    import torch
    s, _ = util.scene_pair_default()
    dev = torch.zeros(64 * 64 * 4, dtype=torch.uint8, device="cuda")
    o = rta.capi.Options(64, 64, 1)
    reg = s.device()._regions([(0, 64, 64, 0)])
    rc = rta.capi.lib.rt_render_tiles(s.device()._h, o, SKIP, reg, 1, dev.data_ptr(), None)
    assert rc == rta.capi.RT_ERR_INVALID_ARGUMENT
    assert b"rt_render_tiles_device" in rta.capi.lib.rt_last_error_message()


@pytest.mark.parametrize("leaders", [0, 1, 2, 3])
def test_concurrent_render_region_calls_are_merged_and_stay_exact(leaders):
    # the literal render.rs:283-294 shape: one rt_render_region call per bucket from many pool threads at once.  The library
    # merges concurrent callers into shared passes (leaders = passes in flight; 0 = no merging): every caller must get exactly
    # its bucket's bytes, whatever it was merged with -- here two different images are rendered concurrently by 24 threads.
    s, o = util.scene_pair_default()
    d = s.device()
    jobs = [((256, 192, 1), r) for r in bucket_list(256, 192)] + [((200, 130, 2), r) for r in bucket_list(200, 130, 2)]
    refs = {(256, 192, 1): o.render(256, 192, 1, nthreads=4)[0], (200, 130, 2): o.render(200, 130, 2, nthreads=4)[0]}
    out = [None] * len(jobs)
    calls0 = rta.capi.debug_count(0)
    nxt, lock = [0], threading.Lock()

    def work():
        while True:
            with lock:
                i = nxt[0]
                nxt[0] += 1
            if i >= len(jobs):
                return
            out[i], _ = d.render_region(jobs[i][0], jobs[i][1], SKIP)

    with rta.capi.debug(rta.capi.DEBUG_COALESCE, leaders), rta.capi.debug(rta.capi.DEBUG_FRAME_AHEAD, 0):     # (the merging path, not the frame-ahead)
        th = [threading.Thread(target=work) for _ in range(24)]
        [t.start() for t in th]
        [t.join() for t in th]
    for (opts, (l, t, r, b)), got in zip(jobs, out):
        np.testing.assert_array_equal(got, refs[opts][b:t, l:r])
    assert rta.capi.debug_count(0) - calls0 == (len(jobs) if leaders else 0)


def test_a_bad_region_fails_alone_in_a_merged_pass():
    s, o = util.scene_pair_default()
    d = s.device()
    regs = bucket_list(256, 192) + [(0, 200, 64, 136)]          # the last one sticks out of the 256x192 image
    res = [None] * len(regs)

    def work(i):
        try:
            res[i] = d.render_region((256, 192, 1), regs[i], SKIP)[0]
        except rta.RtError as e:
            res[i] = e

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(regs))]
    [t.start() for t in th]
    [t.join() for t in th]
    ref = o.render(256, 192, 1, nthreads=4)[0]
    for (l, t, r, b), got in zip(regs[:-1], res[:-1]):
        np.testing.assert_array_equal(got, ref[b:t, l:r])
    assert isinstance(res[-1], rta.RtError) and res[-1].status == rta.capi.RT_ERR_INVALID_REGION


def test_renderer_defaults_to_the_hierarchy_and_strict_64_reproduces_the_assertion(tmp_path):
    # Renderer.render(o, scene, writer) with no traversal given is the reference's hierarchy walk (render.rs:218 -> group.rs:72);
    # strict_64 brings back assert!(w % 64 == 0 && h % 64 == 0) (render.rs:265-266)
    s, o = util.scene_pair_ranges(util.INSIDE_ITEMS, util.INSIDE_BOUNDS, util.INSIDE_RANGES)
    assert s.device().default_traversal() == SKIP
    opts = rta.RenderOptions(128, 64, 1)
    w = rta.PPMStdoutRGBABufferWriter(True, str(tmp_path / "a.tga"))
    rta.Renderer.render(opts, s, w)
    w.close()
    ref, _, _ = o.render(128, 64, 1, 2, HIER_EXIT)
    oracle.write_ppm(str(tmp_path / "ref.ppm"), ref)
    assert open(str(tmp_path / "a.tga"), "rb").read() == open(str(tmp_path / "ref.ppm"), "rb").read()
    buf = rta.RGBABuffer(rta.ImageRegion(0, 64, 64, 0))
    rta.Renderer.render_region(opts, s, buf)
    np.testing.assert_array_equal(buf.buf, ref[0:64, 0:64])
    with pytest.raises(ValueError, match="TODO: handle chunk sizes"):
        rta.Renderer.render(rta.RenderOptions(800, 600, 1), s, rta.PPMStdoutRGBABufferWriter(True, str(tmp_path / "b.tga")), strict_64=True)
    assert rta.Scene.from_spheres(util.THREE_SPHERES, util.THREE_BOUND).device().default_traversal() == SKIP
    assert rta.Scene(np.array(util.THREE_SPHERES), rta.normalized((-1, -3, 2), rta.RT_F32), (0, 0, -4)).device().default_traversal() == FLAT


def test_progressive_output_partial_ppm_on_disk_mid_render(tmp_path):
    # f.2 (render.rs:301-307, 422-433): buckets reach the writer in completion order while the rest of the frame is still to
    # be rendered, and the file sink rewrites the whole image as they come (at most once per second: the clock is injected and
    # jumps 2 s per bucket).  With RTRACEMAXPROCS = 1 the device gets 8 buckets per call, so when bucket k arrives the file must
    # be a valid P6 holding exactly the first k buckets (bit-exact) and zeros everywhere else.
    s, o = util.scene_pair_default()
    opts = rta.RenderOptions(512, 384, 1)
    ref, _, _ = o.render(512, 384, 1, nthreads=os.cpu_count() or 1)
    path = str(tmp_path / "progressive.tga")
    now = [0.0]
    seen = []

    class Spy(rta.PPMStdoutRGBABufferWriter):
        def write_rgba_buffer(self, buffer):
            now[0] += 2.0
            super().write_rgba_buffer(buffer)
            seen.append(tuple(buffer.region()))
            data = open(path, "rb").read()
            assert data.startswith(b"P6\n512 384\n255\n") and len(data) == 15 + 512 * 384 * 3
            img = np.frombuffer(data[15:], dtype=np.uint8).reshape(384, 512, 3)
            done = np.zeros((384, 512), dtype=bool)
            for (l, t, r, b) in seen:
                done[b:t, l:r] = True
            assert np.array_equal(img[done], ref[:, :, :3][done])
            assert (img[~done] == 0).all()

    w = Spy(True, path, clock=lambda: now[0])
    rta.Renderer.render(opts, s, w, pool=1, tiles_per_call=8)
    assert len(seen) == 48 and len(set(seen)) == 48
    w.close()
    oracle.write_ppm(str(tmp_path / "ref.ppm"), ref)
    assert open(path, "rb").read() == open(str(tmp_path / "ref.ppm"), "rb").read()


@pytest.mark.parametrize("policy", [0, 1, 4])
@pytest.mark.parametrize("size", [(1920, 1080, 1), (333, 277, 4), (640, 480, 2)])
def test_dealing_blocks_to_workgroups_never_changes_a_pixel(size, policy):
    # Past 32,768 workgroups a pass's blocks are dealt to fewer workgroups on the host (several descriptors per workgroup,
    # longest first); csrc/rt_debug.h RT_DEBUG_WG_POLICY forces it for any pass.  Same bytes and counters as one block per workgroup.
    w, h, spp = size
    s, o = rta.Scene.default(7), oracle.Scene.default(level=7)
    regs = bucket_list(w, h, spp)
    ref, rst, _ = o.render(w, h, spp, os.cpu_count() or 1, HIER_EXIT)
    with rta.capi.debug(rta.capi.DEBUG_WG_POLICY, policy):
        d = s.device()
        plain, _ = d.render_tiles((w, h, spp), regs, SKIP, want_stats=False)
        counted, st = d.render_tiles((w, h, spp), regs, SKIP, want_stats=True)
    np.testing.assert_array_equal(util.stitch((w, h), regs, plain), ref)
    np.testing.assert_array_equal(counted, plain)
    assert util.all_stats(st) == util.all_stats(rst)


@pytest.mark.parametrize("n_items", [1, 2, 3, 4, 5, 6, 7, 8, 9, 1023, 1024, 1025, 1026, 1027, 2053])
def test_scalar_fed_flat_scan_item_counts_around_the_group_and_pass_sizes(n_items):
    # f32 RT_TRAVERSAL_FLAT runs the scalar-fed scan (rt_flat_sc.hpp + generated rt_flat_rot.hpp): groups of three items, two groups
    # per loop iteration, the first shadow pass covers 342 groups = 1,026 items.  Item counts around every one of those boundaries,
    # against the oracle's flat semantics, and against round 1's LDS kernels (csrc/rt_debug.h RT_DEBUG_FLAT_KERNELS = 0).
    rng = np.random.default_rng(n_items)
    sp = np.concatenate([rng.uniform(-1.5, 1.5, (n_items, 3)), rng.uniform(0.02, 0.12, (n_items, 1))], axis=1)
    sp = sp.astype(np.float32).astype(np.float64)
    s, o = util.scene_pair_spheres(sp, (0, 0, 0, 3.0))
    regs = bucket_list(150, 70, 2)
    ref, rst, _ = o.render(150, 70, 2, os.cpu_count() or 1, oracle.MODE_FLAT)
    data, st = s.device().render_tiles((150, 70, 2), regs, FLAT)
    np.testing.assert_array_equal(util.stitch((150, 70), regs, data), ref)
    assert util.ray_stats(st) == util.ray_stats(rst)
    with rta.capi.debug(rta.capi.DEBUG_FLAT_KERNELS, 0):
        lds, lst = s.device().render_tiles((150, 70, 2), regs, FLAT)
    np.testing.assert_array_equal(lds, data)
    assert util.ray_stats(lst) == util.ray_stats(st)


@pytest.mark.parametrize("scale", [1e-20, 1e-10, 1.0, 1e6, 5e13])
def test_flat_scan_filter_never_rejects_a_candidate(scale):
    # The f32 flat scan rejects items with a conservative bound of the discriminant (FMA chains, rt_flat_sc.hpp flat_filter_constant /
    # flat_shadow_filter_rr; the error analysis is in tools/gen_flat_asm.py) and runs the reference's exact operations only on the
    # survivors.  rt_debug_flat_filter_check evaluates both for every ray x item pair of a frame: no pair may have disc >= 0 with a
    # negative bound -- on the default scene at 1080p (4.5e10 pairs) and on nested scenes scaled from the subnormal range to the
    # validation bound -- and the bound must not be so loose that the exact path runs everywhere.
    if scale == 1.0:
        s = rta.Scene.default(8)
        c = rta.capi.flat_filter_check(s.device()._h, 1920, 1080, 1)
        assert c[0] > 5_000_000 and c[3] > 1_000_000
        assert c[1] < 2 * c[0] and c[4] < 2 * c[3]
        assert c[2] == 0 and c[5] == 0
    for seed in (31, 32, 33):
        items, bounds, ranges = util.random_nested_scene(seed, depth=3, fan=3, leaf_items=2, concentric=seed == 32)
        sc = lambda a: (np.asarray(a, dtype=np.float64) * scale).astype(np.float32).astype(np.float64)
        eye = tuple(float(v) for v in sc((0.07, -0.12, -3.1)))
        s, _ = util.scene_pair_ranges(sc(items), sc(bounds), ranges, rta.RT_F32, eye=eye)
        c = rta.capi.flat_filter_check(s.device()._h, 320, 240, 2)
        assert c[0] > 50_000 and c[3] > 10_000, c
        assert c[2] == 0 and c[5] == 0, c
    if scale == 1.0:
        # a scene far from the coordinate origin: the shadow filter takes centres and origins relative to the scene's centroid, so
        # its margin (relative to |c'|^2 + |o'|^2) stays tight; frames equal the oracle's flat semantics
        items, bounds, ranges = util.random_nested_scene(34, depth=3, fan=3, leaf_items=2)
        shift = np.array([3000.0, -2000.0, 5000.0])
        mv = lambda a: np.concatenate([np.asarray(a, dtype=np.float64)[:, :3] + shift, np.asarray(a, dtype=np.float64)[:, 3:]], axis=1).astype(np.float32).astype(np.float64)
        eye = tuple(float(v) for v in (np.array([0.07, -0.12, -3.1]) + shift).astype(np.float32))
        s, o = util.scene_pair_ranges(mv(items), mv(bounds), ranges, rta.RT_F32, eye=eye)
        c = rta.capi.flat_filter_check(s.device()._h, 320, 240, 2)
        assert c[0] > 30_000 and c[3] > 10_000 and c[2] == 0 and c[5] == 0, c
        assert c[1] < 2 * c[0] and c[4] < 2 * c[3], c
        regs = bucket_list(160, 120, 2)
        ref, rst, _ = o.render(160, 120, 2, os.cpu_count() or 1, oracle.MODE_FLAT)
        data, st = s.device().render_tiles((160, 120, 2), regs, FLAT)
        np.testing.assert_array_equal(util.stitch((160, 120), regs, data), ref)
        assert util.ray_stats(st) == util.ray_stats(rst)


@pytest.mark.parametrize("scale", [1.0, 1e-20, 5e13])
def test_f64_flat_scan_filter_never_rejects_a_candidate_and_matches_the_unfiltered_kernels(scale):
    # The f64 flat scan (rt_flat_f64.hpp) puts the same kind of conservative bound in front of its exact test (eps = 2^-53).
    # rt_debug_flat_filter_check on the f64 arrays: no ray x item pair with disc >= 0 and a negative bound, on the default scene at
    # 1080p and on nested scenes scaled to both ends of the validated range and moved far from the origin; and the filtered kernels
    # render the bytes and counters of round 1's unfiltered LDS kernels (RT_DEBUG_FLAT_KERNELS = 0), which the other f64 tests hold
    # against the oracle.
    scenes = []
    if scale == 1.0:
        s = rta.Scene.default(8, rta.RT_F64)
        c = rta.capi.flat_filter_check(s.device()._h, 1920, 1080, 1)
        assert c[0] > 5_000_000 and c[3] > 1_000_000
        assert c[1] < 2 * c[0] and c[4] < 2 * c[3]
        assert c[2] == 0 and c[5] == 0
        scenes.append((rta.Scene.default(6, rta.RT_F64), (333, 217, 1)))
    for seed in (31, 32, 33):
        items, bounds, ranges = util.random_nested_scene(seed, depth=3, fan=3, leaf_items=2, concentric=seed == 32)
        sc = lambda a: np.asarray(a, dtype=np.float64) * scale
        eye = tuple(float(v) for v in sc((0.07, -0.12, -3.1)))
        s, _ = util.scene_pair_ranges(sc(items), sc(bounds), ranges, rta.RT_F64, eye=eye)
        c = rta.capi.flat_filter_check(s.device()._h, 320, 240, 2)
        assert c[0] > 50_000 and c[3] > 10_000, c
        assert c[2] == 0 and c[5] == 0, c
        scenes.append((s, (160, 120, 2)))
    if scale == 1.0:
        items, bounds, ranges = util.random_nested_scene(34, depth=3, fan=3, leaf_items=2)
        shift = np.array([3000.0, -2000.0, 5000.0])
        mv = lambda a: np.concatenate([np.asarray(a, dtype=np.float64)[:, :3] + shift, np.asarray(a, dtype=np.float64)[:, 3:]], axis=1)
        eye = tuple(float(v) for v in (np.array([0.07, -0.12, -3.1]) + shift))
        s, _ = util.scene_pair_ranges(mv(items), mv(bounds), ranges, rta.RT_F64, eye=eye)
        c = rta.capi.flat_filter_check(s.device()._h, 320, 240, 2)
        assert c[0] > 30_000 and c[3] > 10_000 and c[2] == 0 and c[5] == 0, c
        assert c[1] < 2 * c[0] and c[4] < 2 * c[3], c
        scenes.append((s, (160, 120, 2)))
    for s, (w, h, spp) in scenes:
        regs = bucket_list(w, h, spp)
        new, st_new = s.device().render_tiles((w, h, spp), regs, FLAT)
        new = new.copy()
        with rta.capi.debug(rta.capi.DEBUG_FLAT_KERNELS, 0):
            old, st_old = s.device().render_tiles((w, h, spp), regs, FLAT)
        assert np.array_equal(new, old)
        assert util.ray_stats(st_new) == util.ray_stats(st_old)


def test_scalar_fed_flat_scan_on_tiny_discriminants():
    # the exact path of the flat scan's assembly (root of a denormal / zero discriminant: the scaled `tiny` branch) on a scene
    # scaled by 1e-20, and the default scene at 1080p against the golden frame CRC
    items, bounds, ranges = util.random_nested_scene(41, depth=2, fan=3, leaf_items=3)
    sc = lambda a: (np.asarray(a, dtype=np.float64) * 1e-20).astype(np.float32).astype(np.float64)
    eye = tuple(float(v) for v in sc((0.07, -0.12, -3.1)))
    s, o = util.scene_pair_ranges(sc(items), sc(bounds), ranges, rta.RT_F32, eye=eye)
    regs = bucket_list(96, 72, 2)
    ref, rst, _ = o.render(96, 72, 2, os.cpu_count() or 1, oracle.MODE_FLAT)
    assert rst["hits"] > 500
    data, st = s.device().render_tiles((96, 72, 2), regs, FLAT)
    np.testing.assert_array_equal(util.stitch((96, 72), regs, data), ref)
    assert util.ray_stats(st) == util.ray_stats(rst)
    s, _ = util.scene_pair_default()
    regs = bucket_list(1920, 1080)
    data, st = s.device().render_tiles((1920, 1080, 1), regs, FLAT, want_stats=False)
    case = next(c for c in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_vectors.json")))["cases"] if c["name"] == "config3_1920x1080_f32")
    assert zlib.crc32(util.stitch((1920, 1080), regs, data).tobytes()) & 0xFFFFFFFF == case["frame_crc32"]


@pytest.mark.parametrize("w,h,spp,level", [(1920, 1080, 1, 8), (1024, 768, 4, 8), (800, 600, 2, 8), (97, 64, 8, 4), (333, 217, 1, 6),
                                           (2048, 2048, 4, 7), (2560, 1664, 1, 8)])
def test_two_rays_per_lane_walk_on_the_default_scene(w, h, spp, level):
    # k_render_skip2 against k_render_skip, byte for byte, on the reference's pyramid: cost-ordered and narrowed descriptors
    # (1080p), the sample-packed modes, a ragged frame, a pass dealt out over workgroups (2048 x 2048 spp 4: 262,144 descriptors)
    # and the library's own choice -- spp-1 frames of 2.9 M pixels and more (2560 x 1664) and sample-packed frames of 6 M samples and
    # more (`make image`, 2048 x 2048 spp 4) take the two-ray kernel without being asked
    s = rta.Scene.default(level)
    regs = bucket_list(w, h, spp)
    d = s.device()
    with util.control(rta.capi.DEBUG_SKIP_RAYS, 1):
        one, _ = d.render_tiles((w, h, spp), regs, SKIP, want_stats=False)
        one = one.copy()
    with util.control(rta.capi.DEBUG_SKIP_RAYS, 2):
        two, _ = d.render_tiles((w, h, spp), regs, SKIP, want_stats=False)
        two = two.copy()
    auto, _ = d.render_tiles((w, h, spp), regs, SKIP, want_stats=False)
    assert np.array_equal(two, one) and np.array_equal(auto, one)
    if (w, h, spp, level) == (1920, 1080, 1, 8):
        case = next(c for c in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_vectors.json")))["cases"] if c["name"] == "config3_1920x1080_f32")
        assert zlib.crc32(util.stitch((w, h), regs, two).tobytes()) & 0xFFFFFFFF == case["frame_crc32"]


@pytest.mark.parametrize("percent", [0, 35, 70])
def test_shadow_origins_the_bounds_do_not_cover_run_the_reference_arithmetic(percent):
    # The filtered shadow walks cover ray origins within a radius Ro of the scene's centroid; an origin further out gets q1 = NaN, which
    # passes every outer bound and fails every sure test, so the ray falls back to the reference's arithmetic at every node.  No origin
    # of a real scene is out there -- RT_DEBUG_FILTER_RO_PERCENT shrinks Ro at scene creation so that all (0), most (35) or some (70 %)
    # of them are: both kernels (one ray per lane: per-lane NaN; two: one uncovered ray turns its whole wave) must still render the
    # unfiltered C++ loops' bytes, at spp 1 and in a sample-packed mode.
    with rta.capi.debug(rta.capi.DEBUG_FILTER_RO_PERCENT, percent):
        d = rta.Scene.default(6).device()
    for (w, h, spp) in ((333, 217, 1), (160, 120, 4)):
        regs = bucket_list(w, h, spp)
        with rta.capi.debug(rta.capi.DEBUG_SKIP_VARIANT, 1):
            ref, _ = d.render_tiles((w, h, spp), regs, SKIP, want_stats=True)      # also checks the bounds next to every test it makes
            ref = ref.copy()
        for rays in (1, 2):
            with util.control(rta.capi.DEBUG_SKIP_RAYS, rays):
                got, _ = d.render_tiles((w, h, spp), regs, SKIP, want_stats=False)
            assert np.array_equal(got, ref), (percent, w, h, spp, rays)
    d.close()


def test_only_memory_this_library_pinned_is_written_by_the_kernel():
    # rt_render_tiles lets the kernel store into host memory only inside ranges rt_host_alloc / rt_host_register recorded.  The
    # runtime's own view (hipPointerGetAttributes) also lists ranges it locked for an earlier pageable copy, and such a record can
    # outlive the buffer -- trusting it faulted intermittently on freshly allocated numpy buffers.  Fresh 8 MB buffers in a row,
    # memory pinned by somebody else (torch), and the error paths of free / unregister.
    import torch
    s, _ = util.scene_pair_default()
    d = s.device()
    regs = bucket_list(1920, 1080)
    ref, _ = d.render_tiles((1920, 1080, 1), regs, SKIP, want_stats=False)
    ref = ref.copy()
    for _ in range(12):
        buf = np.empty(ref.size, dtype=np.uint8)                 # a new mapping each time: large allocations are mmap'ed and unmapped
        data, _ = d.render_tiles((1920, 1080, 1), regs, SKIP, want_stats=False, out=buf)
        assert np.array_equal(data, ref)
        del buf, data
    t = torch.empty(ref.size, dtype=torch.uint8).pin_memory()    # pinned, but not by this library: treated as pageable
    data, _ = d.render_tiles((1920, 1080, 1), regs, SKIP, want_stats=False, out=t.numpy())
    assert np.array_equal(data, ref)
    assert rta.capi.lib.rt_host_free(t.data_ptr()) == rta.capi.RT_ERR_INVALID_ARGUMENT
    assert rta.capi.lib.rt_host_unregister(t.data_ptr()) == rta.capi.RT_ERR_INVALID_ARGUMENT
    hb = rta.capi.HostBuffer(ref.size + 4096)
    data, _ = d.render_tiles((1920, 1080, 1), regs, SKIP, want_stats=False, out=hb.array[4096:])      # interior pointer, exact room
    assert np.array_equal(data, ref)
    small = rta.capi.HostBuffer(65536)                            # too small for the pass from this offset: copied, not overrun
    with pytest.raises(ValueError):
        d.render_tiles((1920, 1080, 1), regs, SKIP, want_stats=False, out=small.array)
    hb.close(); small.close()
