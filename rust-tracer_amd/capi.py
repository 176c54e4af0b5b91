"""ctypes binding of include/rtrace_hip.h -- the same symbols a Rust `extern "C"` block would bind.

There is no CPU fallback here: if librtrace_hip.so is missing the import of this module raises, and if no
gfx950 device is visible every render call raises RtError(RT_ERR_NO_DEVICE).

Which library: rust-tracer_amd/librtrace_hip.so, the product (no rt_debug_* symbol in it).  The parity tests and tools/ set
RTRACE_HIP_LIBRARY to tests/c/librtrace_hip_test.so -- the same sources built with -DRT_TEST_HOOKS (csrc/rt_debug.h) -- before importing
this module; the diagnostic bindings at the end of this file exist only then (`HAVE_TEST_HOOKS`)."""
import contextlib
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB_PATH = os.path.join(HERE, "librtrace_hip.so")
TEST_LIB_PATH = os.path.join(os.path.dirname(HERE), "tests", "c", "librtrace_hip_test.so")
LIB_PATH = os.environ.get("RTRACE_HIP_LIBRARY") or PRODUCT_LIB_PATH

RT_OK, RT_ERR_INVALID_ARGUMENT, RT_ERR_INVALID_REGION, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_OUT_OF_MEMORY, RT_ERR_UNSUPPORTED = range(7)
RT_F32, RT_F64 = 0, 1
RT_TRAVERSAL_FLAT, RT_TRAVERSAL_SKIP = 0, 1
ABI_VERSION = 4

# every symbol include/rtrace_hip.h declares
SYMBOLS = ("rt_abi_version", "rt_build_hierarchy", "rt_device_count", "rt_scene_create", "rt_scene_destroy", "rt_scene_traits", "rt_scene_setup_cost", "rt_render_tiles",
           "rt_render_tiles_device", "rt_render_frame_device", "rt_render_region", "rt_blit_tiles_device", "rt_selftest_sqrt", "rt_selftest_rcp", "rt_tiles_rgba_bytes", "rt_strerror", "rt_last_error_message",
           "rt_host_alloc", "rt_host_free", "rt_host_register", "rt_host_unregister",
           "rt_gang_create", "rt_gang_destroy", "rt_gang_size", "rt_gang_render_frame", "rt_gang_render_frames", "rt_render_tiles_stream",
           "rt_last_launch_flags", "rt_build_info", "rt_render_frame_stream")
# csrc/rt_debug.h: only in the -DRT_TEST_HOOKS build
DEBUG_SYMBOLS = ("rt_debug_set", "rt_debug_count", "rt_debug_wave_trace", "rt_debug_flat_filter_check", "rt_debug_gang_layout", "rt_debug_rccl_library",
                 "rt_debug_shard_costs")


class Options(C.Structure):      # rt_options / RenderOptions render.rs:33-38
    _fields_ = [("width", C.c_uint16), ("height", C.c_uint16), ("samples_per_pixel", C.c_uint16)]


class Region(C.Structure):       # rt_region / ImageRegion render.rs:42-48
    _fields_ = [("l", C.c_uint16), ("t", C.c_uint16), ("r", C.c_uint16), ("b", C.c_uint16)]


class Range(C.Structure):
    _fields_ = [("first", C.c_int32), ("count", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("primary", C.c_uint64), ("hits", C.c_uint64), ("shadow", C.c_uint64), ("occluded", C.c_uint64),
                ("sphere_tests", C.c_uint64), ("bound_tests", C.c_uint64), ("tests_executed", C.c_uint64),
                ("primary_tests", C.c_uint64), ("device_ms", C.c_double), ("longest_wave_cycles", C.c_uint64), ("longest_wave_ref100mhz", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class RtError(RuntimeError):
    def __init__(self, status, what, detail):
        self.status = status
        super().__init__("%s: %s (status %d)%s" % (what, _strerror(status), status, (" -- " + detail) if detail else ""))


if not os.path.exists(LIB_PATH):
    raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                      "There is no CPU fallback." % LIB_PATH)

lib = C.CDLL(LIB_PATH)
lib.rt_abi_version.restype = C.c_int
lib.rt_build_hierarchy.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
lib.rt_device_count.argtypes = [C.POINTER(C.c_int)]
lib.rt_scene_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
lib.rt_scene_destroy.argtypes = [C.c_void_p]
lib.rt_render_tiles.argtypes = [C.c_void_p, C.POINTER(Options), C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(Stats)]
lib.rt_render_tiles_device.argtypes = [C.c_void_p, C.POINTER(Options), C.c_int, C.c_void_p, C.c_uint32, C.c_void_p,
                                       C.c_void_p, C.POINTER(Stats)]
lib.rt_render_frame_device.argtypes = lib.rt_render_tiles_device.argtypes
lib.rt_render_region.argtypes = [C.c_void_p, C.POINTER(Options), C.c_int, C.POINTER(Region), C.c_void_p, C.POINTER(Stats)]
lib.rt_blit_tiles_device.argtypes = [C.c_void_p, C.POINTER(Options), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]
lib.rt_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
lib.rt_host_free.argtypes = [C.c_void_p]
lib.rt_host_register.argtypes = [C.c_void_p, C.c_size_t]
lib.rt_host_unregister.argtypes = [C.c_void_p]
lib.rt_gang_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_uint32, C.POINTER(C.c_void_p)]
lib.rt_gang_destroy.argtypes = [C.c_void_p]
lib.rt_gang_size.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
lib.rt_gang_render_frame.argtypes = [C.c_void_p, C.POINTER(Options), C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(Stats)]
lib.rt_gang_render_frames.argtypes = [C.c_void_p, C.POINTER(Options), C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(Stats)]
TILE_CALLBACK = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.POINTER(Region), C.POINTER(C.c_uint8))      # rt_tile_callback
lib.rt_render_tiles_stream.argtypes = [C.c_void_p, C.POINTER(Options), C.c_int, C.c_void_p, C.c_uint32, TILE_CALLBACK, C.c_void_p]
BATCH_CALLBACK = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32)                                        # rt_batch_callback
RT_FRAME_RGBA, RT_FRAME_RGB, RT_FRAME_GREY = 0, 1, 2
lib.rt_render_frame_stream.argtypes = [C.c_void_p, C.POINTER(Options), C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, BATCH_CALLBACK, C.c_void_p]
lib.rt_selftest_sqrt.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
lib.rt_selftest_rcp.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
lib.rt_scene_traits.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
lib.rt_scene_setup_cost.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
RT_SCENE_HAS_BOUNDS, RT_SCENE_CONCENTRIC = 1, 2
lib.rt_tiles_rgba_bytes.restype = C.c_uint64
lib.rt_tiles_rgba_bytes.argtypes = [C.c_void_p, C.c_uint32]
lib.rt_strerror.restype = C.c_char_p
lib.rt_strerror.argtypes = [C.c_int]
lib.rt_last_error_message.restype = C.c_char_p
lib.rt_last_launch_flags.restype = C.c_uint32
lib.rt_build_info.restype = C.c_char_p
RT_LAUNCH_TWO_RAYS, RT_LAUNCH_COOPERATIVE, RT_LAUNCH_SAMPLE_PARALLEL, RT_LAUNCH_ORDERED, RT_LAUNCH_FLAT_PIPELINE, RT_LAUNCH_COUNTING, RT_LAUNCH_FAST_KERNEL = 1, 2, 4, 8, 16, 32, 64
HAVE_TEST_HOOKS = hasattr(lib, "rt_debug_set")

if lib.rt_abi_version() != ABI_VERSION:
    raise ImportError("librtrace_hip.so ABI %d != binding ABI %d: rebuild" % (lib.rt_abi_version(), ABI_VERSION))


def _strerror(status):
    return lib.rt_strerror(status).decode()


def check(status, what):
    if status != RT_OK:
        raise RtError(status, what, lib.rt_last_error_message().decode())


def build_info():
    """rt_build_info: the toolchain and kernel sources the loaded library was built from."""
    return lib.rt_build_info().decode()


def last_launch():
    """rt_last_launch_flags of the calling thread as a set of names."""
    f = lib.rt_last_launch_flags()
    names = (("two_rays", RT_LAUNCH_TWO_RAYS), ("cooperative", RT_LAUNCH_COOPERATIVE), ("sample_parallel", RT_LAUNCH_SAMPLE_PARALLEL),
             ("ordered", RT_LAUNCH_ORDERED), ("flat_pipeline", RT_LAUNCH_FLAT_PIPELINE), ("counting", RT_LAUNCH_COUNTING), ("fast_kernel", RT_LAUNCH_FAST_KERNEL))
    return {n for n, bit in names if f & bit}


def device_count():
    n = C.c_int(0)
    st = lib.rt_device_count(C.byref(n))
    return n.value if st == RT_OK else 0


class HostBuffer:
    """rt_host_alloc'd bytes as a numpy uint8 array (`.array`): RGBABuffer storage the render kernel writes directly.

    The allocation lives as long as ANY array derived from `.array` (views, slices, what render_tiles returns): the array's base is
    a ctypes block whose finalizer calls rt_host_free, so `render_tiles(..., out=capi.HostBuffer(n).array)` is safe.  close() drops
    this object's own reference; close(force=True) frees the pinned block NOW (a benchmark or a long-running service that must not
    accumulate frame-sized pinned buffers) -- every array still derived from it is invalid from then on."""

    def __init__(self, nbytes):
        import weakref
        import numpy as np
        p = C.c_void_p()
        check(lib.rt_host_alloc(nbytes, C.byref(p)), "rt_host_alloc")
        block = (C.c_uint8 * nbytes).from_address(p.value)
        self._finalizer = weakref.finalize(block, lib.rt_host_free, C.c_void_p(p.value))      # runs when the last array over the block is gone
        self._finalizer.atexit = False                                       # not during interpreter shutdown: the GPU runtime may be gone by then
        self.array = np.ctypeslib.as_array(block)                            # .base chain keeps `block` alive

    def close(self, force=False):
        self.array = None
        if force:
            self._finalizer()                                                # rt_host_free now (a no-op if it already ran)


def selftest_sqrt(device=0):
    """rt_selftest_sqrt -> (mismatches, first_bad_bits) over all 2^32 f32 bit patterns."""
    bad, first = C.c_uint64(0), C.c_uint32(0)
    check(lib.rt_selftest_sqrt(device, C.byref(bad), C.byref(first)), "rt_selftest_sqrt")
    return bad.value, first.value


def selftest_rcp(device=0):
    """rt_selftest_rcp -> (mismatches, first_bad_bits) over all 2^32 f32 bit patterns."""
    bad, first = C.c_uint64(0), C.c_uint32(0)
    check(lib.rt_selftest_rcp(device, C.byref(bad), C.byref(first)), "rt_selftest_rcp")
    return bad.value, first.value


# ---- csrc/rt_debug.h: diagnostic controls (not part of the drop-in ABI; the library reads no environment variable) ----
(DEBUG_SKIP_VARIANT, DEBUG_BLOCK_ORDER, DEBUG_NARROW_MAX, DEBUG_PACKED_SAMPLES, DEBUG_PRINT_STEPS, DEBUG_PRINT_COSTS,
 DEBUG_HOST_COPY, DEBUG_COALESCE, DEBUG_LDS_BYTES, DEBUG_WG_POLICY, DEBUG_NARROW_L2, DEBUG_FLAT_KERNELS, DEBUG_SKIP_RAYS,
 DEBUG_FRAME_AHEAD, DEBUG_FILTER_RO_PERCENT, DEBUG_COOP, DEBUG_COOP_THR, DEBUG_COOP_MAX, DEBUG_COOP_LEVEL, DEBUG_COOP_REST, DEBUG_ASYNC_ORDERS,
 DEBUG_FAST_KERNEL, DEBUG_EXACT_COSTS) = range(23)
if HAVE_TEST_HOOKS:
    lib.rt_debug_set.argtypes = [C.c_int, C.c_longlong]
    lib.rt_debug_wave_trace.argtypes = [C.c_char_p]
    lib.rt_debug_count.restype = C.c_longlong
    lib.rt_debug_count.argtypes = [C.c_int]
    lib.rt_debug_flat_filter_check.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_ulonglong * 6)]
    lib.rt_debug_gang_layout.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    lib.rt_debug_shard_costs.argtypes = [C.c_void_p, C.POINTER(Options), C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.rt_debug_rccl_library.argtypes = [C.c_char_p]


def _need_hooks(what):
    if not HAVE_TEST_HOOKS:
        raise RuntimeError("%s needs the -DRT_TEST_HOOKS build of the library: set RTRACE_HIP_LIBRARY=%s before importing rust_tracer_amd "
                           "(%s is the product and has no rt_debug_* entry point)" % (what, TEST_LIB_PATH, LIB_PATH))


def debug_count(counter):
    """rt_debug_count."""
    _need_hooks("rt_debug_count")
    return lib.rt_debug_count(counter)
(DEBUG_COUNT_REGION_CALLS, DEBUG_COUNT_REGION_PASSES, DEBUG_COUNT_FILTER_PASS, DEBUG_COUNT_FILTER_VIOLATIONS, DEBUG_COUNT_PRIMARY_TESTS,
 DEBUG_COUNT_FRAME_AHEAD_PASSES, DEBUG_COUNT_TWO_RAY_LAUNCHES, DEBUG_COUNT_COOP_LAUNCHES) = range(8)


def debug_set(key, value=-1):
    """rt_debug_set; value < 0 restores the default."""
    _need_hooks("rt_debug_set")
    check(lib.rt_debug_set(key, value), "rt_debug_set")


@contextlib.contextmanager
def debug(key, value):
    """with capi.debug(capi.DEBUG_SKIP_VARIANT, 3): ...  -- the control is back at its default afterwards."""
    debug_set(key, value)
    try:
        yield
    finally:
        debug_set(key, -1)



def flat_filter_check(scene_handle, width, height, spp):
    """rt_debug_flat_filter_check -> (pairs with disc >= 0, pairs with bound >= 0, pairs with disc >= 0 but bound < 0) for the primary
    filter, then the same three for the shadow filter."""
    _need_hooks("rt_debug_flat_filter_check")
    counts = (C.c_ulonglong * 6)()
    check(lib.rt_debug_flat_filter_check(scene_handle, width, height, spp, C.byref(counts)), "rt_debug_flat_filter_check")
    return tuple(int(c) for c in counts)



def gang_layout(regions, n_devices):
    """rt_debug_gang_layout (no device needed) -> (device of each bucket, its first pixel inside that device's shard, pixels of
    each device's shard, padded shard length in pixels)."""
    import numpy as np
    _need_hooks("rt_debug_gang_layout")
    arr = (Region * len(regions))(*[Region(*r) for r in regions])
    dev = np.zeros(len(regions), dtype=np.uint32)
    off = np.zeros(len(regions), dtype=np.uint32)
    px = np.zeros(n_devices, dtype=np.uint64)
    padded = C.c_uint64(0)
    check(lib.rt_debug_gang_layout(arr, len(regions), n_devices, dev.ctypes.data, off.ctypes.data, px.ctypes.data, C.byref(padded)), "rt_debug_gang_layout")
    return dev, off, px, int(padded.value)


def shard_costs(scene_handle, options, regions, n_devices):
    """rt_debug_shard_costs -> float64[n_devices]: the cost map's prediction for each device's shard (tests, arbitrary scale)."""
    import numpy as np
    _need_hooks("rt_debug_shard_costs")
    arr = (Region * len(regions))(*[Region(*r) for r in regions])
    cost = np.zeros(n_devices, dtype=np.float64)
    o = Options(*options)
    check(lib.rt_debug_shard_costs(scene_handle, C.byref(o), arr, len(regions), n_devices, cost.ctypes.data), "rt_debug_shard_costs")
    return cost


FAKE_RCCL = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "c", "libfake_rccl.so")


@contextlib.contextmanager
def rccl_stand_in(path=FAKE_RCCL):
    """TESTS ONLY: gangs created inside bind the stand-in library (tests/c/fake_rccl.cpp) instead of librccl.so and may put several ranks
    on one device -- the N > 1 code of rt_gang_* on a one-GPU box."""
    _need_hooks("rt_debug_rccl_library")
    check(lib.rt_debug_rccl_library(path.encode()), "rt_debug_rccl_library")
    try:
        yield
    finally:
        check(lib.rt_debug_rccl_library(None), "rt_debug_rccl_library")


def wave_trace(path):
    _need_hooks("rt_debug_wave_trace")
    check(lib.rt_debug_wave_trace(path.encode() if path else None), "rt_debug_wave_trace")
