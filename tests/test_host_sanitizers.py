"""The C++ host side under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: the GPU pool has no sanitizer runs).
host_tests (vector / pyramid / region / bucket-list known answers, the hierarchy builder on a sphere list) and a driver of the PPM
writer (clipped buckets in a scrambled order, in-place rewrites, the SSSE3 conversion's 16-byte stores)."""
import os
import subprocess
import tempfile

import pytest

from tests import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rust-tracer_amd")
HOST = os.path.join(PKG, "csrc", "host")
FLAGS = ["-std=c++17", "-O1", "-g", "-ffp-contract=off", "-fno-fast-math", "-pthread", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
         "-fno-sanitize-recover=undefined"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0")


def build(out, sources):
    if not os.path.exists(os.path.join(PKG, "librtrace_hip.so")):
        pytest.skip("librtrace_hip.so is not built")
    r = subprocess.run(["g++"] + FLAGS + ["-I" + HOST, "-o", out] + sources + ["-L" + PKG, "-lrtrace_hip", "-Wl,-rpath," + PKG], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr and "cannot find" in r.stderr:
        pytest.skip("the sanitizer runtimes are not installed")
    assert r.returncode == 0, r.stderr[-2000:]


def clean(r):
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]


def test_host_tests_and_hierarchy_builder_under_sanitizers():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "host_tests_san")
        build(exe, [os.path.join(HOST, "host_tests.cpp"), os.path.join(HOST, "render.cpp")])
        r = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=300)
        clean(r)
        assert "host_tests ok" in r.stdout
        path = os.path.join(d, "spheres.txt")
        with open(path, "w") as f:
            for s in scenes.hundred_thousand_spheres(n=3000):
                f.write("%r %r %r %r\n" % tuple(float(v) for v in s))
        r = subprocess.run([exe, "--hierarchy", path], capture_output=True, text=True, env=ENV, timeout=300)
        clean(r)
        assert r.stdout.split()[0] == "3000"


def test_ppm_writer_under_sanitizers():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "writer_san")
        build(exe, [os.path.join(ROOT, "tests", "c", "writer_sanitize.cpp"), os.path.join(HOST, "render.cpp")])
        r = subprocess.run([exe, os.path.join(d, "out.ppm")], capture_output=True, text=True, env=ENV, timeout=300)
        clean(r)
        assert "P5 file bytes 26015" in r.stdout and "P6 file bytes 78015" in r.stdout
        # the device path's bookkeeping: complete rows in the partial rewrite, the spare image reused by the second writer
        assert "device path pass 0: partial file 78015 bytes, final 78015 bytes, header 15" in r.stdout
        assert "device path pass 1: partial file 78015 bytes, final 78015 bytes, header 15" in r.stdout
