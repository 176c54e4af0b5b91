"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of oracle/librt_oracle.so.

CPU restatement of rust-tracer's hot path (see oracle/rt_oracle.h for the reference file:line map and the
pinning statement).  Importable only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
product package never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librt_oracle.so")

F32, F64 = 0, 1
MODE_HIERARCHY, MODE_FLAT = 0, 1
MODE_ANYHIT_EXIT = 2      # OR-able: shadow rays stop at the first hit (identical pixels, the GPU SKIP kernel's counters)


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("primary", "hits", "shadow", "occluded", "sphere_tests", "bound_tests")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(force=False):
    """Compile the C restatement (gcc, strict IEEE flags) if the .so is missing or stale."""
    srcs = [os.path.join(_HERE, f) for f in ("rt_oracle.c", "rt_oracle_impl.h", "rt_oracle.h", "Makefile")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "librt_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.orc_scene_default.restype = C.c_void_p
        L.orc_scene_default.argtypes = [C.c_int, C.c_uint]
        L.orc_scene_pyramid.restype = C.c_void_p
        L.orc_scene_pyramid.argtypes = [C.c_int, C.c_uint, dp, C.c_double, dp, dp]
        L.orc_scene_from_spheres.restype = C.c_void_p
        L.orc_scene_from_spheres.argtypes = [C.c_int, dp, C.c_int, dp, dp, dp]
        L.orc_scene_from_ranges.restype = C.c_void_p
        L.orc_scene_from_ranges.argtypes = [C.c_int, dp, C.c_int, dp, C.c_void_p, C.c_int, dp, dp]
        L.orc_scene_free.argtypes = [C.c_void_p]
        L.orc_scene_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_scene_flatten.restype = C.c_int
        L.orc_scene_flatten.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_scene_bounds.restype = C.c_int
        L.orc_scene_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_scene_light_eye.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_render_region.argtypes = [C.c_void_p, C.c_int] + [C.c_uint] * 7 + [C.c_void_p, C.POINTER(Stats)]
        L.orc_render.restype = C.c_int
        L.orc_render.argtypes = [C.c_void_p, C.c_int] + [C.c_uint] * 4 + [C.c_void_p, C.POINTER(Stats)]
        L.orc_write_ppm.restype = C.c_int
        L.orc_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_uint, C.c_uint, C.c_int]
        L.orc_sphere_distance_from_ray.restype = C.c_double
        L.orc_sphere_distance_from_ray.argtypes = [C.c_int, dp, dp]
        L.orc_sphere_intersect.argtypes = [C.c_int, dp, dp, C.c_double, dp]
        L.orc_scene_intersect.argtypes = [C.c_void_p, C.c_int, dp, C.c_double, dp]
        L.orc_vec_normalized.argtypes = [C.c_int, dp, dp, dp, dp]
        _lib = L
    return _lib


def _d(a):
    arr = np.ascontiguousarray(a, dtype=np.float64)
    return arr, arr.ctypes.data_as(C.POINTER(C.c_double))


class Scene:
    """Owns an orc_scene*.  prec: F32 (reference as shipped) or F64 (type-alias swap, parity unpinned)."""

    def __init__(self, handle, prec):
        if not handle:
            raise ValueError("oracle scene construction failed (level must be > 1, group.rs:59)")
        self._h = C.c_void_p(handle)
        self.prec = prec
        self.real = np.float32 if prec == F32 else np.float64

    @classmethod
    def default(cls, prec=F32, level=8):
        return cls(lib().orc_scene_default(prec, level), prec)

    @classmethod
    def pyramid(cls, level, origin, radius, light=(-1.0, -3.0, 2.0), eye=(0.0, 0.0, -4.0), prec=F32):
        _, o = _d(origin); _, l = _d(light); _, e = _d(eye)
        return cls(lib().orc_scene_pyramid(prec, level, o, float(radius), l, e), prec)

    @classmethod
    def from_spheres(cls, spheres4, bound4, light=(-1.0, -3.0, 2.0), eye=(0.0, 0.0, -4.0), prec=F32):
        sp, spp = _d(np.asarray(spheres4, dtype=np.float64).reshape(-1, 4))
        _, b = _d(bound4); _, l = _d(light); _, e = _d(eye)
        return cls(lib().orc_scene_from_spheres(prec, spp, sp.shape[0], b, l, e), prec)

    @classmethod
    def from_ranges(cls, items4, bounds4, ranges2, light=(-1.0, -3.0, 2.0), eye=(0.0, 0.0, -4.0), prec=F32):
        """Arbitrary nesting from DFS items + pre-order ranges (ranges[0] = root {0, n})."""
        it, itp = _d(np.asarray(items4, dtype=np.float64).reshape(-1, 4))
        bd, bdp = _d(np.asarray(bounds4, dtype=np.float64).reshape(-1, 4))
        rg = np.ascontiguousarray(ranges2, dtype=np.int32).reshape(-1, 2)
        _, l = _d(light); _, e = _d(eye)
        return cls(lib().orc_scene_from_ranges(prec, itp, it.shape[0], bdp, rg.ctypes.data, rg.shape[0], l, e), prec)

    def __del__(self):
        try:
            if self._h:
                lib().orc_scene_free(self._h)
                self._h = None
        except Exception:
            pass

    def counts(self):
        g, i = C.c_int(), C.c_int()
        lib().orc_scene_counts(self._h, C.byref(g), C.byref(i))
        return g.value, i.value

    def flatten(self):
        """DFS-ordered items as REAL[n,4] = cx,cy,cz,r (what the C-ABI's rt_scene_create takes)."""
        _, n = self.counts()
        out = np.empty((n, 4), dtype=self.real)
        lib().orc_scene_flatten(self._h, out.ctypes.data)
        return out

    def bounds(self):
        """(REAL[g,4] bound spheres, int32[g,2] first/count item ranges), DFS pre-order."""
        g, _ = self.counts()
        b = np.empty((g, 4), dtype=self.real)
        r = np.empty((g, 2), dtype=np.int32)
        lib().orc_scene_bounds(self._h, b.ctypes.data, r.ctypes.data)
        return b, r

    def light_eye(self):
        l = np.empty(3, dtype=self.real); e = np.empty(3, dtype=self.real)
        lib().orc_scene_light_eye(self._h, l.ctypes.data, e.ctypes.data)
        return l, e

    def render_region(self, w, h, spp, l, t, r, b, mode=MODE_HIERARCHY):
        """Renderer::render_region on ImageRegion{l,t,r,b}; returns (uint8[t-b, r-l, 4], stats dict)."""
        out = np.zeros((t - b, r - l, 4), dtype=np.uint8)
        st = Stats()
        lib().orc_render_region(self._h, mode, w, h, spp, l, t, r, b, out.ctypes.data, C.byref(st))
        return out, st.as_dict()

    def render(self, w, h, spp, nthreads=1, mode=MODE_HIERARCHY):
        """Renderer::render: 64x64 buckets on nthreads workers; returns (uint8[h,w,4], stats dict, n_buckets)."""
        out = np.zeros((h, w, 4), dtype=np.uint8)
        st = Stats()
        n = lib().orc_render(self._h, mode, w, h, spp, nthreads, out.ctypes.data, C.byref(st))
        return out, st.as_dict(), n

    def intersect(self, ray6, hit_distance_in=float("inf"), mode=MODE_HIERARCHY):
        _, r = _d(ray6)
        out = np.zeros(4, dtype=np.float64)
        lib().orc_scene_intersect(self._h, mode, r, float(hit_distance_in), out.ctypes.data_as(C.POINTER(C.c_double)))
        return out[0], out[1:]


def sphere_distance_from_ray(sphere4, ray6, prec=F32):
    _, s = _d(sphere4); _, r = _d(ray6)
    return lib().orc_sphere_distance_from_ray(prec, s, r)


def sphere_intersect(sphere4, ray6, hit_distance_in, prec=F32):
    _, s = _d(sphere4); _, r = _d(ray6)
    out = np.zeros(4, dtype=np.float64)
    lib().orc_sphere_intersect(prec, s, r, float(hit_distance_in), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out[0], out[1:]


def vec_normalized(v3, prec=F32):
    _, v = _d(v3)
    out = np.zeros(3, dtype=np.float64)
    li, lo = C.c_double(), C.c_double()
    lib().orc_vec_normalized(prec, v, out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(li), C.byref(lo))
    return out, li.value, lo.value


def write_ppm(path, frame_rgba, rgb=True):
    f = np.ascontiguousarray(frame_rgba, dtype=np.uint8)
    h, w = f.shape[:2]
    rc = lib().orc_write_ppm(os.fsencode(path), f.ctypes.data, w, h, 1 if rgb else 0)
    if rc != 0:
        raise OSError("orc_write_ppm failed for %r" % (path,))
