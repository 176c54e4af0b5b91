"""Host-side Scene of the drop-in: what `Scene`, `SphericalGroup::pyramid` and `Scene::default()` are in the
reference (render.rs:138-167, group.rs:27-66), kept as the flat DFS arrays the C ABI consumes.

All scene arithmetic is done in the scene's REAL type with numpy scalars (IEEE, one rounding per operation,
reference operation order) -- centres accumulate rounding level by level, so the builder replays the
recursion instead of using a closed form.  Rendering never happens here; see render.py / capi.py."""
import ctypes as C

import numpy as np

from . import capi


def _real(precision):
    return np.float32 if precision == capi.RT_F32 else np.float64


def pyramid(level, origin, radius, precision=capi.RT_F32):
    """SphericalGroup::pyramid (group.rs:58-65) -> (items REAL[n,4], bounds REAL[g,4], ranges int32[g,2]).

    items are in traversal order (own sphere first, then the four sub-pyramids, dz outer / dx inner,
    group.rs:39-53); bounds/ranges are in DFS pre-order with ranges[i] = (first item, item count) of the subtree."""
    if level <= 1:
        raise ValueError("Levels equal or smaller than one cause empty groups")      # group.rs:59-60
    R = _real(precision)
    three, half, s12 = R(3.0), R(0.5), np.sqrt(R(12.0))
    items, bounds, ranges = [], [], []

    def rec(lv, px, py, pz, r):
        if lv == 1:
            items.append((px, py, pz, r))
            return
        bi = len(bounds)
        bounds.append((px, py, pz, three * r))          # group.rs:40-41
        ranges.append(None)
        first = len(items)
        items.append((px, py, pz, r))                   # group.rs:39
        rn = three * r / s12                            # group.rs:43
        for dz in (-1, 1):
            for dx in (-1, 1):
                rec(lv - 1, px + R(dx) * rn, py + rn, pz + R(dz) * rn, r * half)
        ranges[bi] = (first, len(items) - first)

    rec(level, R(origin[0]), R(origin[1]), R(origin[2]), R(radius))
    return (np.array(items, dtype=R).reshape(-1, 4), np.array(bounds, dtype=R).reshape(-1, 4),
            np.array(ranges, dtype=np.int32).reshape(-1, 2))


def normalized(v, precision=capi.RT_F32):
    """Vector::normalized (vec.rs:92-95): v * (1/len), len = sqrt((x*x + y*y) + z*z)."""
    R = _real(precision)
    x, y, z = R(v[0]), R(v[1]), R(v[2])
    ln = np.sqrt((x * x + y * y) + z * z)
    rc = R(1.0) / ln
    return np.array([x * rc, y * rc, z * rc], dtype=R)


def build_hierarchy(spheres, leaf_size=4, precision=capi.RT_F32, eye=None):
    """Bounding-sphere hierarchy for an arbitrary sphere list (scenes other than the pyramid, e.g. BASELINE
    config 5 with exactly 100,000 spheres).  Not in the reference -- its only scene builder is `pyramid` -- but the result
    is an ordinary `TypedGroup` tree in the flat description the C ABI takes: median splits along the longest axis until
    at most `leaf_size` spheres remain; every group's bound is a near-minimal enclosing sphere of its whole subtree (inflated
    by 1e-4 so that it also encloses after rounding to REAL); with `eye`, a group's nearer half comes first.

    ONE implementation for both hosts: csrc/host/hierarchy.hpp, here through the library's rt_build_hierarchy (no device is
    touched).  build_hierarchy_reference below restates the same arithmetic in numpy; the tests hold one against the other.

    Returns (items REAL[n,4] in the tree's DFS order, bounds REAL[g,4], ranges int32[g,2], order int64[n]) where
    items == spheres[order]."""
    R = _real(precision)
    sp = np.ascontiguousarray(np.asarray(spheres, dtype=np.float64).reshape(-1, 4))
    n = sp.shape[0]
    if n == 0:
        raise ValueError("build_hierarchy needs at least one sphere")
    items = np.zeros((n, 4), dtype=R)
    bounds = np.zeros((2 * n, 4), dtype=R)
    ranges = np.zeros((2 * n, 2), dtype=np.int32)
    order = np.zeros(n, dtype=np.uint64)
    ng = C.c_uint32(0)
    e = None if eye is None else np.ascontiguousarray(np.asarray(eye, dtype=np.float64).reshape(3))
    capi.check(capi.lib.rt_build_hierarchy(sp.ctypes.data, n, int(leaf_size), None if e is None else e.ctypes.data, precision, items.ctypes.data,
                                           bounds.ctypes.data, ranges.ctypes.data, order.ctypes.data, C.byref(ng)), "rt_build_hierarchy")
    g = int(ng.value)
    return items, bounds[:g].copy(), ranges[:g].copy(), order.astype(np.int64)


def build_hierarchy_reference(spheres, leaf_size=4, precision=capi.RT_F32, eye=None, steps=20):
    """The arithmetic of csrc/host/hierarchy.hpp restated in numpy, operation for operation (test infrastructure: slow -- a few numpy calls
    per group and per Badoiu-Clarkson step).  Same return value as build_hierarchy."""
    R = _real(precision)
    sp = np.asarray(spheres, dtype=np.float64).reshape(-1, 4)
    if sp.shape[0] == 0:
        raise ValueError("build_hierarchy needs at least one sphere")
    order, bounds, ranges = [], [], []
    e = None if eye is None else np.asarray(eye, dtype=np.float64).reshape(3)

    def reach(c, r, centre):
        d = c - centre
        return np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]) + r

    def enclosing(idx):
        c, r = sp[idx, :3], sp[idx, 3]
        lo, hi = (c - r[:, None]).min(axis=0), (c + r[:, None]).max(axis=0)
        centre = (lo + hi) * 0.5
        for k in range(1, steps + 1):
            if idx.size <= 1:
                break
            j = int(np.argmax(reach(c, r, centre)))            # the first of the farthest
            v = c[j] - centre
            nrm = np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
            if nrm == 0.0:
                break
            far = c[j] + (v / nrm) * r[j]
            centre = centre + (far - centre) / float(k + 1)
        radius = float(reach(c, r, centre).max()) * (1.0 + 1e-4) + 1e-30
        return (float(centre[0]), float(centre[1]), float(centre[2]), radius)

    def build(idx, bound):
        gi = len(bounds)
        bounds.append(bound)
        ranges.append(None)
        first = len(order)
        if idx.size <= leaf_size:
            order.extend(int(i) for i in idx)
        else:
            c = sp[idx, :3]
            axis = int(np.argmax(c.max(axis=0) - c.min(axis=0)))
            srt = idx[np.argsort(c[:, axis], kind="stable")]
            half = srt.size // 2
            parts = [srt[:half], srt[half:]]
            b = [enclosing(parts[0]), enclosing(parts[1])]
            nearer = 0
            if e is not None:
                key = []
                for k in range(2):
                    dx, dy, dz = b[k][0] - e[0], b[k][1] - e[1], b[k][2] - e[2]
                    key.append((dx * dx + dy * dy) + dz * dz)
                if key[1] < key[0]:
                    nearer = 1
            build(parts[nearer], b[nearer])
            build(parts[1 - nearer], b[1 - nearer])
        ranges[gi] = (first, len(order) - first)

    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 10000))
    all_idx = np.arange(sp.shape[0])
    build(all_idx, enclosing(all_idx))
    order = np.asarray(order, dtype=np.int64)
    return (sp[order].astype(R), np.asarray(bounds, dtype=np.float64).astype(R),
            np.asarray(ranges, dtype=np.int32).reshape(-1, 2), order)


class Scene:
    """Scene{group, directional_light, eye} (render.rs:138-142) with the group flattened to DFS arrays."""

    def __init__(self, items, directional_light, eye, bounds=None, ranges=None, precision=capi.RT_F32):
        R = _real(precision)
        self.precision = precision
        self.items = np.ascontiguousarray(items, dtype=R).reshape(-1, 4)
        self.directional_light = np.ascontiguousarray(directional_light, dtype=R).reshape(3)
        self.eye = np.ascontiguousarray(eye, dtype=R).reshape(3)
        self.bounds = None if bounds is None else np.ascontiguousarray(bounds, dtype=R).reshape(-1, 4)
        self.ranges = None if ranges is None else np.ascontiguousarray(ranges, dtype=np.int32).reshape(-1, 2)
        self._device = {}

    @classmethod
    def default(cls, level=8, precision=capi.RT_F32):
        """Scene::default() render.rs:144-166 (level 8 there)."""
        items, bounds, ranges = pyramid(level, (0.0, -1.0, 0.0), 1.0, precision)
        return cls(items, normalized((-1.0, -3.0, 2.0), precision), (0.0, 0.0, -4.0), bounds, ranges, precision)

    @classmethod
    def from_spheres(cls, spheres, bound, light=(-1.0, -3.0, 2.0), eye=(0.0, 0.0, -4.0), precision=capi.RT_F32):
        """One group {bound, children = spheres as Items}; BASELINE config 1's "3 spheres, 1 light" shape."""
        items = np.asarray(spheres, dtype=np.float64).reshape(-1, 4)
        return cls(items, normalized(light, precision), eye, np.asarray(bound, dtype=np.float64).reshape(1, 4),
                   np.array([[0, items.shape[0]]], dtype=np.int32), precision)

    @classmethod
    def from_spheres_auto(cls, spheres, light=(-1.0, -3.0, 2.0), eye=(0.0, 0.0, -4.0), leaf_size=4, precision=capi.RT_F32):
        """Arbitrary sphere list with an automatically built bounding-sphere hierarchy (build_hierarchy)."""
        items, bounds, ranges, _ = build_hierarchy(spheres, leaf_size, precision, eye=eye)
        return cls(items, normalized(light, precision), eye, bounds, ranges, precision)

    @classmethod
    def three_spheres(cls, precision=capi.RT_F32):
        """The build-defined config-1 scene."""
        return cls.from_spheres([(0.0, -1.0, 0.0, 1.0), (-1.2, 0.2, 0.0, 0.5), (1.2, 0.2, 0.0, 0.5)],
                                (0.0, -1.0, 0.0, 3.0), precision=precision)

    def device(self, device=0):
        """Uploads once per device and caches the handle (replaces Arc<Scene> sharing, render.rs:279)."""
        if device not in self._device:
            self._device[device] = DeviceScene(self, device)
        return self._device[device]


class DeviceScene:
    """Owns an rt_scene* (device copies of a Scene)."""

    def __init__(self, scene, device=0):
        self.scene = scene
        self.device = device
        h = C.c_void_p()
        nb = 0 if scene.bounds is None else scene.bounds.shape[0]
        st = capi.lib.rt_scene_create(
            device, scene.precision, scene.items.ctypes.data, scene.items.shape[0],
            scene.directional_light.ctypes.data, scene.eye.ctypes.data,
            scene.bounds.ctypes.data if nb else None, scene.ranges.ctypes.data if nb else None, nb, C.byref(h))
        capi.check(st, "rt_scene_create")
        self._h = h

    def traits(self):
        """rt_scene_traits -> bit set of capi.RT_SCENE_HAS_BOUNDS / capi.RT_SCENE_CONCENTRIC."""
        t = C.c_uint32(0)
        capi.check(capi.lib.rt_scene_traits(self._h, C.byref(t)), "rt_scene_traits")
        return t.value

    def setup_cost(self):
        """rt_scene_setup_cost -> (total_ms, stream_ms) of the rt_scene_create call that made this scene."""
        total, stream = C.c_double(0), C.c_double(0)
        capi.check(capi.lib.rt_scene_setup_cost(self._h, C.byref(total), C.byref(stream)), "rt_scene_setup_cost")
        return total.value, stream.value

    def close(self):
        if getattr(self, "_h", None):
            capi.lib.rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _regions(regions):
        arr = (capi.Region * len(regions))()
        for i, (l, t, r, b) in enumerate(regions):
            arr[i] = capi.Region(l, t, r, b)
        return arr

    def render_tiles(self, options, regions, traversal=None, want_stats=True, out=None):
        """rt_render_tiles: regions = [(l, t, r, b), ...] -> (uint8[total_px*4] tile-major, stats dict | None).
        out: optional uint8 array to render into.  Pinned memory is written by the kernel directly: keep the allocation in a
        variable while its array is in use -- `hb = capi.HostBuffer(n); dev.render_tiles(..., out=hb.array)`."""
        traversal = self.default_traversal() if traversal is None else traversal
        arr = regions if isinstance(regions, C.Array) else self._regions(regions)
        nbytes = capi.lib.rt_tiles_rgba_bytes(arr, len(arr))
        if out is None:
            out = np.empty(max(int(nbytes), 1), dtype=np.uint8)
        elif out.dtype != np.uint8 or not out.flags.c_contiguous or out.size < nbytes:
            raise ValueError("out must be a contiguous uint8 array of at least %d bytes" % nbytes)
        st = capi.Stats()
        o = capi.Options(*options)
        rc = capi.lib.rt_render_tiles(self._h, C.byref(o), traversal, arr, len(arr), out.ctypes.data,
                                      C.byref(st) if want_stats else None)
        capi.check(rc, "rt_render_tiles")
        return out.reshape(-1)[:int(nbytes)], (st.as_dict() if want_stats else None)

    def render_region(self, options, region, traversal=None, want_stats=False, out=None):
        """rt_render_region: one bucket (l, t, r, b) -> (uint8[h, w, 4], stats dict | None); the literal render.rs:283-294 call."""
        traversal = self.default_traversal() if traversal is None else traversal
        l, t, r, b = region
        reg = capi.Region(l, t, r, b)
        nbytes = (t - b) * (r - l) * 4
        if out is None:
            out = np.empty(nbytes, dtype=np.uint8)
        elif out.dtype != np.uint8 or not out.flags.c_contiguous or out.size < nbytes:
            raise ValueError("out must be a contiguous uint8 array of at least %d bytes" % nbytes)
        st = capi.Stats()
        o = capi.Options(*options)
        rc = capi.lib.rt_render_region(self._h, C.byref(o), traversal, C.byref(reg), out.ctypes.data, C.byref(st) if want_stats else None)
        capi.check(rc, "rt_render_region")
        return out.reshape(-1)[:nbytes].reshape(t - b, r - l, 4), (st.as_dict() if want_stats else None)

    def render_tiles_stream(self, options, regions, on_tile, traversal=None):
        """rt_render_tiles_stream: on_tile(index, (l, t, r, b), uint8[h, w, 4] view valid during the call) for every bucket, in
        completion order, while later batches are still rendering (the reference's channel consumer, render.rs:301-307)."""
        traversal = self.default_traversal() if traversal is None else traversal
        arr = regions if isinstance(regions, C.Array) else self._regions(regions)
        err = []

        def cb(_user, index, region, rgba):
            if err:
                return
            try:
                reg = region.contents
                h, w = reg.t - reg.b, reg.r - reg.l
                on_tile(int(index), (reg.l, reg.t, reg.r, reg.b), np.ctypeslib.as_array(rgba, shape=(h * w * 4,)).reshape(h, w, 4))
            except BaseException as e:      # noqa: BLE001  (must not unwind through the C frames)
                err.append(e)

        o = capi.Options(*options)
        rc = capi.lib.rt_render_tiles_stream(self._h, C.byref(o), traversal, arr, len(arr), capi.TILE_CALLBACK(cb), None)
        if err:
            raise err[0]
        capi.check(rc, "rt_render_tiles_stream")

    def render_frame_stream(self, options, regions, frame_format, out, on_batch=None, traversal=None):
        """rt_render_frame_stream: the listed buckets, converted on the device, into their place in `out` -- a row-major uint8 frame in
        the file's pixel format (capi.RT_FRAME_RGBA / _RGB / _GREY: 4 / 3 / 1 bytes per pixel).  on_batch(first_tile, n_tiles) after each
        batch is in place.  A capi.HostBuffer array is written by the device directly."""
        traversal = self.default_traversal() if traversal is None else traversal
        arr = regions if isinstance(regions, C.Array) else self._regions(regions)
        bpp = {capi.RT_FRAME_RGBA: 4, capi.RT_FRAME_RGB: 3, capi.RT_FRAME_GREY: 1}[frame_format]
        if out.dtype != np.uint8 or out.size != options[0] * options[1] * bpp or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a C-contiguous uint8 array of width * height * %d bytes" % bpp)
        err = []

        def cb(_user, first, count):
            if err or on_batch is None:
                return
            try:
                on_batch(int(first), int(count))
            except BaseException as e:      # noqa: BLE001  (must not unwind through the C frames)
                err.append(e)

        o = capi.Options(*options)
        rc = capi.lib.rt_render_frame_stream(self._h, C.byref(o), traversal, arr, len(arr), frame_format, out.ctypes.data, capi.BATCH_CALLBACK(cb), None)
        if err:
            raise err[0]
        capi.check(rc, "rt_render_frame_stream")
        return out

    def default_traversal(self):
        """The reference's hierarchy walk whenever the scene has bounds; the flat scan otherwise."""
        return capi.RT_TRAVERSAL_SKIP if self.scene.bounds is not None and len(self.scene.bounds) else capi.RT_TRAVERSAL_FLAT

    def render_tiles_device(self, options, regions, out_ptr, stream=0, traversal=None, want_stats=False):
        """rt_render_tiles_device: enqueue on `stream` (hipStream_t as int), output to device pointer `out_ptr`."""
        traversal = self.default_traversal() if traversal is None else traversal
        arr = regions if isinstance(regions, C.Array) else self._regions(regions)
        st = capi.Stats()
        o = capi.Options(*options)
        rc = capi.lib.rt_render_tiles_device(self._h, C.byref(o), traversal, arr, len(arr), C.c_void_p(out_ptr),
                                             C.c_void_p(stream), C.byref(st) if want_stats else None)
        capi.check(rc, "rt_render_tiles_device")
        return st.as_dict() if want_stats else None

    def render_frame_device(self, options, regions, frame_ptr, stream=0, traversal=capi.RT_TRAVERSAL_SKIP, want_stats=False):
        """rt_render_frame_device: the buckets rendered straight into a row-major device frame (render + blit fused)."""
        arr = regions if isinstance(regions, C.Array) else self._regions(regions)
        st = capi.Stats()
        o = capi.Options(*options)
        rc = capi.lib.rt_render_frame_device(self._h, C.byref(o), traversal, arr, len(arr), C.c_void_p(frame_ptr),
                                             C.c_void_p(stream), C.byref(st) if want_stats else None)
        capi.check(rc, "rt_render_frame_device")
        return st.as_dict() if want_stats else None

    def blit_tiles_device(self, options, regions, src_ptr, frame_ptr, stream=0, src_px_offset=None):
        """rt_blit_tiles_device: tile-major device tiles -> row-major device frame (set_pixels_from_buffer)."""
        arr = regions if isinstance(regions, C.Array) else self._regions(regions)
        offs = None
        if src_px_offset is not None:
            offs = np.ascontiguousarray(src_px_offset, dtype=np.uint32)
        o = capi.Options(*options)
        rc = capi.lib.rt_blit_tiles_device(self._h, C.byref(o), arr, len(arr), offs.ctypes.data if offs is not None else None,
                                           C.c_void_p(src_ptr), C.c_void_p(frame_ptr), C.c_void_p(stream))
        capi.check(rc, "rt_blit_tiles_device")


class Gang:
    """rt_gang: the Scene replicated on several GPUs of this node in ONE process; a frame's buckets are dealt round-robin over
    them and the u8 shards come back through one RCCL gather (the native twin of dist.FrameSharder's one-process-per-GPU path)."""

    def __init__(self, scene, devices):
        self.scene = scene
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        nb = 0 if scene.bounds is None else scene.bounds.shape[0]
        st = capi.lib.rt_gang_create(devs, len(devices), scene.precision, scene.items.ctypes.data, scene.items.shape[0],
                                     scene.directional_light.ctypes.data, scene.eye.ctypes.data,
                                     scene.bounds.ctypes.data if nb else None, scene.ranges.ctypes.data if nb else None, nb, C.byref(h))
        capi.check(st, "rt_gang_create")
        self._h = h

    def size(self):
        n = C.c_int(0)
        capi.check(capi.lib.rt_gang_size(self._h, C.byref(n)), "rt_gang_size")
        return n.value

    def render_frame(self, options, regions, traversal=capi.RT_TRAVERSAL_SKIP, want_stats=False, out=None):
        """rt_gang_render_frame -> (uint8[h, w, 4] row-major frame, stats dict | None)."""
        arr = regions if isinstance(regions, C.Array) else DeviceScene._regions(regions)
        o = capi.Options(*options)
        if out is None:
            out = np.zeros(o.width * o.height * 4, dtype=np.uint8)
        st = capi.Stats()
        rc = capi.lib.rt_gang_render_frame(self._h, C.byref(o), traversal, arr, len(arr), out.ctypes.data, C.byref(st) if want_stats else None)
        capi.check(rc, "rt_gang_render_frame")
        return out.reshape(o.height, o.width, 4), (st.as_dict() if want_stats else None)

    def render_frames(self, options, regions, n_frames, traversal=capi.RT_TRAVERSAL_SKIP, want_stats=False, out=None):
        """rt_gang_render_frames -> (list of n_frames uint8[h, w, 4] frames, stats dict | None): frame f's gather runs under the
        render of frame f + 1.  out: optional list of contiguous uint8 arrays (pinned ones are written by the root GPU directly)."""
        arr = regions if isinstance(regions, C.Array) else DeviceScene._regions(regions)
        o = capi.Options(*options)
        nbytes = o.width * o.height * 4
        if out is None:
            out = [np.zeros(nbytes, dtype=np.uint8) for _ in range(n_frames)]
        for a in out:
            if a.dtype != np.uint8 or not a.flags.c_contiguous or a.size < nbytes:
                raise ValueError("every frame must be a contiguous uint8 array of at least %d bytes" % nbytes)
        ptrs = (C.c_void_p * n_frames)(*[a.ctypes.data for a in out])
        st = capi.Stats()
        rc = capi.lib.rt_gang_render_frames(self._h, C.byref(o), traversal, arr, len(arr), ptrs, n_frames, C.byref(st) if want_stats else None)
        capi.check(rc, "rt_gang_render_frames")
        return [a.reshape(-1)[:nbytes].reshape(o.height, o.width, 4) for a in out], (st.as_dict() if want_stats else None)

    def close(self):
        if getattr(self, "_h", None):
            capi.lib.rt_gang_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
