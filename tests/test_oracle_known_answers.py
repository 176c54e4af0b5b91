"""Pins the CPU oracle on every known answer the reference's own in-module unit tests hold
(SURVEY.md section 4).  Each test names the reference test it restates.  CPU only."""
import math

import numpy as np
import pytest

import oracle

INF = float("inf")
PRECS = [oracle.F32, oracle.F64]


@pytest.mark.parametrize("prec", PRECS)
def test_vec_normalize(prec):
    # vec::tests::normalize  vec.rs:157-170
    n, len_in, len_out = oracle.vec_normalized((2.0, 0.0, 0.0), prec)
    assert len_in == 2.0
    assert len_out == 1.0
    assert n.tolist() == [1.0, 0.0, 0.0]


@pytest.mark.parametrize("prec", PRECS)
def test_sphere_intersect(prec):
    # sphere::intersect  primitive.rs:146-173 (setup_scene :123-143)
    s = (0.0, 0.0, 0.0, 1.0)
    r1 = (2.0, 0.0, 0.0, -1.0, 0.0, 0.0)
    r2 = (2.0, 0.0, 0.0, 1.0, 0.0, 0.0)
    assert oracle.sphere_distance_from_ray(s, r1, prec) == 1.0
    assert oracle.sphere_distance_from_ray(s, r2, prec) == INF

    d, pos = oracle.sphere_intersect(s, r1, 2.0, prec)
    assert d == 1.0 and pos[0] == 1.0
    d, _ = oracle.sphere_intersect(s, r1, 0.5, prec)
    assert d == 0.5, "Max Distance too short"
    d, _ = oracle.sphere_intersect(s, r2, 10.0, prec)
    assert d == 10.0, "r2 is shot the wrong way"


@pytest.mark.parametrize("prec", PRECS)
def test_sphere_strict_less_rule(prec):
    # primitive.rs:79 `if distance >= hit.distance return` : an equal distance does NOT replace the hit
    s = (0.0, 0.0, 0.0, 1.0)
    r1 = (2.0, 0.0, 0.0, -1.0, 0.0, 0.0)
    d, pos = oracle.sphere_intersect(s, r1, 1.0, prec)
    assert d == 1.0 and pos.tolist() == [0.0, 0.0, 0.0]


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("mode", [oracle.MODE_HIERARCHY, oracle.MODE_FLAT])
def test_group_intersect(prec, mode):
    # group::tests::intersect  group.rs:153-170 (setup_group :118-151)
    g = oracle.Scene.from_spheres([(0, 0, 0, 1.0), (0, 0, 2.0, 1.0)], (0, 0, 0, 3.0), prec=prec)
    assert g.counts() == (1, 2)
    r1 = (2.0, 0, 0, -1.0, 0, 0)
    r2 = (2.0, 0, 2.0, -1.0, 0, 0)
    r3 = (2.0, 0, 0, 1.0, 0, 0)
    for ray in (r1, r2):
        d, pos = g.intersect(ray, INF, mode)
        assert d == 1.0
        assert pos[0] == 1.0 and pos[2] == 0.0
    d, _ = g.intersect(r3, INF, mode)
    assert d == INF


@pytest.mark.parametrize("prec", PRECS)
def test_pyramid_counts(prec):
    # group::tests::pyramid  group.rs:172-184
    s = oracle.Scene.pyramid(8, (1.0, -1.0, 0.0), 1.0, prec=prec)
    assert s.counts() == (5461, 21845)
    bounds, ranges = s.bounds()
    assert ranges[0].tolist() == [0, 21845]
    # root has 5 children: its own sphere then four sub-pyramids of (21845-1)/4 items each
    assert ranges[1].tolist() == [1, 5461]
    assert bounds[0].tolist() == [1.0, -1.0, 0.0, 3.0]


def test_pyramid_level_must_exceed_one():
    # assert!(level > 1) group.rs:59
    with pytest.raises(ValueError):
        oracle.Scene.pyramid(1, (0, 0, 0), 1.0)


def test_pyramid_first_levels_f32():
    # group.rs:40-52 evaluated by hand in f32: rn = (3*r)/sqrt(12f32); children dz outer, dx inner
    s = oracle.Scene.pyramid(2, (0.0, -1.0, 0.0), 1.0)
    flat = s.flatten()
    rn = np.float32(3.0) * np.float32(1.0) / np.sqrt(np.float32(12.0))
    exp = [(0.0, -1.0, 0.0, 1.0)]
    for dz in (-1, 1):
        for dx in (-1, 1):
            exp.append((np.float32(dx) * rn, np.float32(-1.0) + rn, np.float32(dz) * rn, 0.5))
    assert flat.dtype == np.float32
    np.testing.assert_array_equal(flat, np.array(exp, dtype=np.float32))


def test_scene_default_light_eye():
    # render.rs:154-164
    s = oracle.Scene.default()
    light, eye = s.light_eye()
    v = np.array([-1.0, -3.0, 2.0], dtype=np.float32)
    ln = np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
    exp = v * (np.float32(1.0) / ln)
    np.testing.assert_array_equal(light, exp)
    assert eye.tolist() == [0.0, 0.0, -4.0]


def test_basic_rendering_tile_count():
    # render::tests::basic_rendering  render.rs:466-481: 64x128, spp 2, pool of 1 -> exactly 2 buckets
    s = oracle.Scene.default()
    img, st, n = s.render(64, 128, 2, nthreads=1)
    assert n == 2
    assert st["primary"] == 64 * 128 * 4
    assert img.shape == (128, 64, 4)


def test_default_scene_64x64_ray_statistics():
    # SURVEY.md 8(d) config 1 companion: default scene 64x64 spp 1 (P6)
    s = oracle.Scene.default()
    _, st = s.render_region(64, 64, 1, 0, 64, 64, 0)
    assert (st["primary"], st["hits"], st["shadow"], st["occluded"]) == (4096, 2487, 1923, 909)


def test_threads_do_not_change_pixels():
    s = oracle.Scene.default()
    a, sa, _ = s.render(256, 192, 2, nthreads=1)
    b, sb, _ = s.render(256, 192, 2, nthreads=8)
    np.testing.assert_array_equal(a, b)
    assert sa == sb


def test_flat_scan_equals_hierarchy_default_scene():
    # a flat DFS-order scan is byte-identical to the bounding-sphere traversal on the default scene
    s = oracle.Scene.default()
    a, sa = s.render_region(320, 256, 1, 96, 192, 224, 64, oracle.MODE_HIERARCHY)
    b, sb = s.render_region(320, 256, 1, 96, 192, 224, 64, oracle.MODE_FLAT)
    np.testing.assert_array_equal(a, b)
    for k in ("primary", "hits", "shadow", "occluded"):
        assert sa[k] == sb[k]
    assert sb["sphere_tests"] == (sb["primary"] + sb["shadow"]) * 21845


def test_ppm_writer(tmp_path):
    # render.rs:373-401: P6 header + RGB (alpha dropped); P5 = ((r+g+b) as f32 / 3.0) as u8
    f = np.zeros((2, 3, 4), dtype=np.uint8)
    f[..., 0] = 10; f[..., 1] = 20; f[..., 2] = 33; f[..., 3] = 255
    p = tmp_path / "a.ppm"
    oracle.write_ppm(str(p), f, rgb=True)
    data = p.read_bytes()
    assert data.startswith(b"P6\n3 2\n255\n")
    assert data[len(b"P6\n3 2\n255\n"):] == bytes([10, 20, 33] * 6)
    oracle.write_ppm(str(p), f, rgb=False)
    data = p.read_bytes()
    assert data == b"P5\n3 2\n255\n" + bytes([21] * 6)
