"""Cross-compiles a variant of libunetmi for same-box A/B timing: ONE source rebuilt with extra -D flags, linked with the
standard objects of the others.  usage: python tools/build_variant.py NAME SOURCE.hip [-DFLAG ...]  -> tools/_ab/libunetmi_NAME.so
(tools/_ab/ is git-ignored but travels to the GPU box with the snapshot)."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
from umi import build as B

name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build_lib(verbose=False)
out_dir = os.path.join(REPO, "tools", "_ab")
os.makedirs(out_dir, exist_ok=True)
src = os.path.join(B.CSRC, os.path.basename(src))
obj = os.path.join(out_dir, f"{name}_{os.path.basename(src)[:-4]}.o")
subprocess.check_call([B.HIPCC] + B.CFLAGS + flags + ["-c", src, "-o", obj])
objs = [B._obj(s) for s in B.sources() if s != src] + [obj]
lib = os.path.join(out_dir, f"libunetmi_{name}.so")
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + ["-o", lib])
print(lib)
