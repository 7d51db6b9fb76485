"""Timing of the ConvTranspose2d(k=2,s=2) trio (forward into the concat buffer, data gradient, weight gradient) on the four
up-path levels of the bench's U-Net (reference Model.py:56-57), HIP events, 10 launches each, median of 3 rounds.  Per launch:
TFLOP/s and GB/s of the algorithmic bytes (inputs read once + outputs written once) -- the shallow levels are HBM-bound, the
deep ones MFMA-bound."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import torch
from umi import ops, lib as L

dev, dt = "cuda", torch.float16
N, F, HW = 16, 64, 512


def timeit(fn):
    ts = []
    for r in range(3):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    return sorted(ts)[1]


tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
TILES = os.environ.get("AB_TILES", "").split(",") if os.environ.get("AB_TILES") else []      # e.g. 256x64,128x128
for lvl in range(4):
    ci = F * 16 >> lvl
    co = ci // 2
    h = HW // 16 << lvl
    x = torch.randn(N, h, h, ci, device=dev).to(dt)
    tx = ops.passthrough_tx(ci, dev); tx[:, 3] = 0.0
    w = torch.randn(ci, co, 2, 2, device=dev) * 0.05
    b = torch.randn(co, device=dev)
    cat = torch.empty(N, 2 * h, 2 * h, 2 * co, device=dev, dtype=dt)
    dest = cat[..., co:]
    g = (torch.randn(N, 2 * h, 2 * h, 2 * co, device=dev) * 0.1).to(dt)[..., co:]
    dx = torch.empty(N, h, h, ci, device=dev, dtype=dt)
    gw = torch.empty(ci, co, 2, 2, device=dev)
    wf = ops.pack_convT_fwd(w, dt, k8=True)
    wd = ops.pack_convT_dgrad(w, dt, k8=True)
    fl = 2.0 * N * h * h * ci * co * 4
    by_f = (x.numel() + dest.numel()) * 2
    t = {}
    t["fwd"] = timeit(lambda: ops.conv_fwd(x, tx, lambda lay: wf, b, dest, 2, 2, 2, 0, flags=L.CONV_UPSAMPLE2))
    t["dgrad"] = timeit(lambda: ops.conv_fwd(g, None, lambda lay: wd, None, dx, 2, 2, 2, 0))
    t["wgrad"] = timeit(lambda: ops.conv_wgrad(g, None, x, tx, gw, co * 4, 4, 1, 1.0, 2, 2, 2, 0))
    line = f"L{lvl + 1} {ci:4d}->{co:4d} @{h:3d}->{2 * h:3d}:"
    for k in ("fwd", "dgrad", "wgrad"):
        tot[k] += t[k]
        line += f"  {k} {t[k]:.3f} ms {fl / t[k] / 1e9:5.0f} TF {by_f / t[k] / 1e6:5.0f} GB/s |"
    print(line, flush=True)
    for tl in TILES:                          # the pointwise kernel's tile override (conv1x1_mfma.hip, read per call)
        os.environ["UMI_C1_TILE"] = tl
        tf = timeit(lambda: ops.conv_fwd(x, tx, lambda lay: wf, b, dest, 2, 2, 2, 0, flags=L.CONV_UPSAMPLE2))
        td = timeit(lambda: ops.conv_fwd(g, None, lambda lay: wd, None, dx, 2, 2, 2, 0))
        del os.environ["UMI_C1_TILE"]
        print(f"      tile {tl:8s}: fwd {tf:.3f} ms  dgrad {td:.3f} ms", flush=True)
print("SUM  " + "  ".join(f"{k} {v:.3f} ms" for k, v in tot.items()))
