"""Timing-only experiment: builds conv_mfma variants (no staging / no MFMA) as separate .so files and times a few
layers.  Results are wrong by construction for the ablated builds; only durations matter."""
import ctypes, os, subprocess, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import torch
from umi import build as B

variants = {"base": [], "no_stage": ["-DUMI_EXP_NO_STAGE"], "no_mfma": ["-DUMI_EXP_NO_MFMA"]}
extra = [a for a in sys.argv[1:] if not a.startswith("VAR:")]      # extra -D flags applied to all
if any(a.startswith("VAR:") for a in sys.argv[1:]):               # VAR:name=-Dflag,-Dflag ... replaces the default variants
    variants = {"base": []}
    for a in sys.argv[1:]:
        if a.startswith("VAR:"):
            nm, fl = a[4:].split("=", 1)
            variants[nm] = fl.split(",")
libs = {}
for name, flags in variants.items():
    out = os.path.join(REPO, "gpurun_out", f"libexp_{name}.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call([B.HIPCC] + B.FLAGS + flags + extra + B.sources() + ["-o", out])
    libs[name] = ctypes.CDLL(out)

from umi import ops, lib as L
shapes = [("64->64@512", 16, 512, 512, 64, 64), ("128->64@512", 16, 512, 512, 128, 64), ("128->128@256", 16, 256, 256, 128, 128),
          ("256->256@128", 16, 128, 128, 256, 256), ("512->512@64", 16, 64, 64, 512, 512), ("1024->1024@32", 16, 32, 32, 1024, 1024)]
for nm, n, h, w, ci, co in shapes:
    x = torch.randn(n, h, w, ci, device="cuda").half()
    wgt = torch.randn(co, ci, 3, 3, device="cuda") * (2.0 / (9 * ci)) ** 0.5
    tx = ops.passthrough_tx(ci, "cuda"); tx[:, 3] = 0
    y = torch.empty(n, h, w, co, device="cuda", dtype=torch.float16)
    wp = ops.pack_conv_fwd(wgt, torch.float16, k8=True)
    rows = n * ((w + 31) // 32) * ((h + 7) // 8)
    part = torch.empty(rows * 2 * co * 2, device="cuda")
    line = [nm]
    for name, lb in libs.items():
        f = lb.umi_conv_fwd
        f.restype = ctypes.c_int
        f.argtypes = L.SIGNATURES["umi_conv_fwd"][1]
        args = (x.data_ptr(), ci, tx.data_ptr(), wp.data_ptr(), None, y.data_ptr(), co, part.data_ptr(), n, h, w, ci, co,
                3, 3, 1, 1, h, w, 0, 0, h, w, 1, 1, 0, torch.cuda.current_stream().cuda_stream)
        assert f(*args) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f(*args)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        line.append(f"{name}={ms:.3f}ms({2.0*n*h*w*9*ci*co/ms/1e9:.0f}TF)")
    print("  ".join(line), flush=True)
