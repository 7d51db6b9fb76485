"""Builds libunetmi.so (hipcc, gfx950 only) in-tree next to this file."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
LIB = os.path.join(HERE, "libunetmi.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        [os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "unetmi.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=True):
    if not force and not stale():
        return LIB
    cmd = [HIPCC] + FLAGS + sources() + ["-o", LIB]
    if verbose:
        print("[umi.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
