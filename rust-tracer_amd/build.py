"""Builds librtrace_hip.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc, and tests/c/librtrace_hip_test.so (the same sources
with -DRT_TEST_HOOKS: csrc/rt_debug.h's controls for the parity tests and tools/).  Cross-compiles without a GPU."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librtrace_hip.so")
TEST_LIB = os.path.join(os.path.dirname(HERE), "tests", "c", "librtrace_hip_test.so")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build(force=False, verbose=False):
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".h", ".cpp")) or f == "Makefile"]
    srcs.append(os.path.join(os.path.dirname(HERE), "include", "rtrace_hip.h"))
    if force or _stale(LIB, srcs) or _stale(TEST_LIB, srcs):
        cmd = ["make", "-j2", "-C", CSRC] + ([] if verbose else ["-s"]) + (["-B"] if force else [])
        subprocess.check_call(cmd)
    for lib in (LIB, TEST_LIB):
        if not os.path.exists(lib):
            raise RuntimeError("hipcc did not produce %s" % lib)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
