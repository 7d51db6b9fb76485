"""Builds libunetmi.so (hipcc, gfx950 only) in-tree next to this file.

One object file per csrc/*.hip (compiled in parallel, only the stale ones), then one link.  The whole build runs under an
exclusive file lock and the library is moved into place with os.replace, so N ranks importing the package at the same time
never see (or write) a half-written .so."""
import concurrent.futures
import fcntl
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
LIB = os.path.join(HERE, "libunetmi.so")
OBJ = os.path.join(HERE, "_obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
          "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]
FLAGS = CFLAGS + ["-shared"]          # one-shot form (tools/ build experiment variants with it)


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + \
        [os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "unetmi.h")]


def _obj(src):
    return os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")


def _stale_objs():
    ht = max(os.path.getmtime(h) for h in _headers())
    out = []
    for s in sources():
        o = _obj(s)
        if not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), ht):
            out.append(s)
    return out


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in sources() + _headers())


def build_lib(force=False, verbose=True):
    if not force and not stale():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not stale():         # another process built it while we waited
            return LIB
        todo = sources() if force else _stale_objs()

        def cc(src):
            cmd = [HIPCC] + CFLAGS + ["-c", src, "-o", _obj(src)]
            if verbose:
                print("[umi.build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)

        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
            list(ex.map(cc, todo))
        known = {_obj(s) for s in sources()}
        for o in glob.glob(os.path.join(OBJ, "*.o")):      # objects of deleted sources must not be linked
            if o not in known:
                os.remove(o)
        tmp = LIB + ".tmp.%d" % os.getpid()
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"] + sorted(known) + ["-o", tmp]
        if verbose:
            print("[umi.build] link ->", LIB, flush=True)
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
