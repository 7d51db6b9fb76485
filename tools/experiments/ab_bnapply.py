"""Per-layer A/B of umi_conv_wgrad_bnapply against umi_bn_bwd_apply + umi_conv_wgrad on the DoubleConv shapes of the bench
workload (batch 16, 512x512, feat 64): ms per layer, interleaved rounds, medians."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import torch
import bench
from umi import ops
DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(3)
tot = [0.0, 0.0]
for name, n, h, w, ci, co in bench.double_conv_shapes(1, 64, 512, 512, 16):
    if ci < 16:
        continue
    x = torch.randn(n, h, w, ci, device=DEV, generator=g).half()
    y = torch.randn(n, h, w, co, device=DEV, generator=g).half()
    da = (0.1 * torch.randn(n, h, w, co, device=DEV, generator=g)).half()
    txa = ops.passthrough_tx(ci, DEV); txa[:, 3] = 0
    tb = ops.passthrough_tx(co, DEV); tb[:, 3] = 0
    rstd = torch.ones(co, device=DEV)
    s1, s2 = ops.bn_bwd(da.clone(), y, tb, rstd, apply=False)
    gw = torch.empty(co, ci, 3, 3, device=DEV)
    dz = torch.empty_like(da)
    work = da.clone()

    def sep():
        ops.bn_bwd_apply(work, y, tb, rstd, s1, s2)
        ops.conv_wgrad(x, txa, work, None, gw, ci * 9, 9, 1, 1.0, 3, 3, 1, 1)

    def fused():
        assert ops.conv_wgrad_bnapply(x, txa, da, y, tb, rstd, s1, s2, dz, gw, ci * 9, 9, 1, 1.0, 3, 3, 1, 1)

    res = {}
    for fn in (sep, fused):
        fn()
    torch.cuda.synchronize()
    ts = {sep: [], fused: []}
    for r in range(5):
        for fn in (sep, fused):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts[fn].append(e0.elapsed_time(e1) / 5)
    a, b = sorted(ts[sep])[2], sorted(ts[fused])[2]
    tot[0] += a; tot[1] += b
    print(f"{name:9s} {ci:4d}->{co:4d}@{h:3d}  apply+wgrad {a:.3f} ms   fused {b:.3f} ms   {'FUSED' if b < a else ''}", flush=True)
print(f"SUM apply+wgrad {tot[0]:.3f}  fused {tot[1]:.3f}")
