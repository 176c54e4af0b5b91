"""rust-tracer_amd: MI355X (gfx950) backend for rust-tracer's per-pixel ray-sphere hot path.

Layout: csrc/ (hand-written HIP kernels + C ABI -> librtrace_hip.so), capi.py (ctypes binding of
include/rtrace_hip.h), scene.py / render.py (host-side mirror of the reference's Scene / Renderer surface),
dist.py (tile sharding across GPUs + RCCL gather).  Importing this package requires the built library."""
from . import capi
from .capi import RT_F32, RT_F64, RT_TRAVERSAL_FLAT, RT_TRAVERSAL_SKIP, RtError, device_count
from .scene import Scene, DeviceScene, Gang, pyramid, normalized, build_hierarchy
from .render import (RenderOptions, ImageRegion, RGBABuffer, RGBABufferWriter, PPMStdoutRGBABufferWriter,
                     Renderer, buckets, CHUNK_SIZE)

__all__ = ["capi", "RT_F32", "RT_F64", "RT_TRAVERSAL_FLAT", "RT_TRAVERSAL_SKIP", "RtError", "device_count",
           "Scene", "DeviceScene", "Gang", "pyramid", "normalized", "build_hierarchy", "RenderOptions", "ImageRegion", "RGBABuffer",
           "RGBABufferWriter", "PPMStdoutRGBABufferWriter", "Renderer", "buckets", "CHUNK_SIZE"]
