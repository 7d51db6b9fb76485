"""Timing of the conv3x3 weight-gradient kernel on the bench's DoubleConv layers (HIP events, 10 launches each, median of 3
rounds); AB_ZERO=1: all-zero operands (the clock the chip holds without data-dependent power, MI355X_MICROARCH.md DVFS)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import torch
import bench
from umi import ops

dev, dt = "cuda", torch.float16
tot_t = tot_f = 0.0
for name, n, h, w, ci, co in bench.double_conv_shapes(1, 64, 512, 512, 16):
    if ci < 16:
        continue
    x = torch.randn(n, h, w, ci, device=dev).to(dt)
    tx = ops.passthrough_tx(ci, dev); tx[:, 3] = 0.0
    dy = (torch.randn(n, h, w, co, device=dev) * 0.1).to(dt)
    if os.environ.get("AB_ZERO") == "1":
        x.zero_(); dy.zero_()
    gw = torch.empty(co, ci, 3, 3, device=dev)
    ts = []
    for r in range(3):
        ops.conv_wgrad(x, tx, dy, None, gw, ci * 9, 9, 1, 1.0, 3, 3, 1, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv_wgrad(x, tx, dy, None, gw, ci * 9, 9, 1, 1.0, 3, 3, 1, 1)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ms = sorted(ts)[1]
    fl = 2.0 * n * h * w * 9 * ci * co
    tot_t += ms; tot_f += fl
    print(f"{name:9s} {ci:4d}->{co:4d}@{h:3d}  {ms:.3f} ms {fl / ms / 1e9:6.0f} TF", flush=True)
print(f"SUM {tot_t:.3f} ms = {tot_f / tot_t / 1e9:.0f} TF")
