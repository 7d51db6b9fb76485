"""Runs the DoubleConv conv3x3 forward launches (and optionally wgrad) of the bench workload a few times each,
for rocprofv3 kernel-trace / PMC passes.  usage: python tools/prof_convs.py [fwd|wgrad|both] [reps]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import torch
import bench
from umi import ops

what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev, dt = "cuda", torch.float16
for name, n, h, w, ci, co in bench.double_conv_shapes(1, 64, 512, 512, 16):
    if ci < 16:
        continue
    x = torch.randn(n, h, w, ci, device=dev).to(dt)
    wgt = torch.randn(co, ci, 3, 3, device=dev) * (2.0 / (9 * ci)) ** 0.5
    tx = ops.passthrough_tx(ci, dev); tx[:, 3] = 0.0
    y = torch.empty(n, h, w, co, device=dev, dtype=dt)
    lay, _ = ops.conv_plan(x, y, 3, 3, 1, 1)
    wp = ops.pack_conv_fwd(wgt, dt, k8=bool(lay))
    dy = (torch.randn(n, h, w, co, device=dev) * 0.1).to(dt)
    gw = torch.empty(co, ci, 3, 3, device=dev)
    for _ in range(reps):
        if what in ("fwd", "both"):
            ops.conv_fwd(x, tx, lambda l: wp, None, y, 3, 3, 1, 1, want_stats=True)
        if what in ("wgrad", "both"):
            ops.conv_wgrad(x, tx, dy, None, gw, ci * 9, 9, 1, 1.0, 3, 3, 1, 1)
    torch.cuda.synchronize()
    print(name, n, h, w, ci, co, flush=True)
