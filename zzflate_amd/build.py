"""Builds libzzflate_amd.so (HIP kernels + C ABI + C++ shim) in-tree for gfx950 with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libzzflate_amd.so")
SOURCES = [os.path.join(CSRC, "zz_api.hip"), os.path.join(CSRC, "zz_cxx_shim.cpp")]


def _newest_source_mtime():
    m = 0.0
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def build(force=False, verbose=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_source_mtime():
        return LIB
    import fcntl
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # several ranks may import the package at once: one of them builds (into a temporary file, renamed into place),
    # the others wait on the lock and find the library up to date
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_source_mtime():
            return LIB
        tmp = f"{LIB}.{os.getpid()}.tmp"
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", tmp] + SOURCES
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        os.replace(tmp, LIB)
    return LIB


CAREFUL_LIB = os.path.join(HERE, "libzzflate_amd_careful.so")


def build_careful(force=False, verbose=False):
    """A test build with -DZZ_ST_ALWAYS_CAREFUL (the sequential level-2 stream takes its one-position-at-a-time form
    everywhere, zz_stream2.h): built here, by __graft_entry__.build(), so that the GPU test run does not spend its time in
    the compiler (tests/test_gpu_sequential.py). Never loaded by the package itself."""
    if not force and os.path.exists(CAREFUL_LIB) and os.path.getmtime(CAREFUL_LIB) >= _newest_source_mtime():
        return CAREFUL_LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tmp = f"{CAREFUL_LIB}.{os.getpid()}.tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DZZ_ST_ALWAYS_CAREFUL", "-o", tmp] + SOURCES
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(tmp, CAREFUL_LIB)
    return CAREFUL_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
