"""Where does umi_conv_wgrad_bnapply differ from umi_bn_bwd_apply + umi_conv_wgrad?  (debug aid for the fused kernel)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import torch
from umi import ops
DEV = "cuda"
for shape in [(1, 13, 37, 72, 136), (1, 5, 70, 16, 24), (1, 8, 32, 64, 72)]:
    N, H, W, Ci, Co = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(N, H, W, Ci, generator=g).half().to(DEV)
    y = torch.randn(N, H, W, Co, generator=g).half().to(DEV)
    da = (0.1 * torch.randn(N, H, W, Co, generator=g)).half().to(DEV)
    txa = ops.passthrough_tx(Ci, DEV)
    tb = ops.passthrough_tx(Co, DEV)
    tb[:, 0] = 0.1 * torch.randn(Co, generator=g).to(DEV)
    tb[:, 1] = (0.5 + torch.rand(Co, generator=g)).to(DEV)
    tb[:, 3] = 0
    rstd = (0.5 + torch.rand(Co, generator=g)).to(DEV)
    ref = da.clone()
    s1, s2 = ops.bn_bwd(ref, y, tb, rstd)
    gwr = torch.empty(Co, Ci, 3, 3, device=DEV)
    ops.conv_wgrad(x, txa, ref, None, gwr, Ci * 9, 9, 1, 1.0, 3, 3, 1, 1)
    dz = torch.full((N, H, W, Co), float("nan"), device=DEV, dtype=torch.float16)
    gw = torch.empty(Co, Ci, 3, 3, device=DEV)
    ok = ops.conv_wgrad_bnapply(x, txa, da, y, tb, rstd, s1, s2, dz, gw, Ci * 9, 9, 1, 1.0, 3, 3, 1, 1)
    torch.cuda.synchronize()
    bad = (dz != ref) | dz.isnan()
    print(shape, "ok", ok, "dz bad", int(bad.sum()), "nan", int(dz.isnan().sum()), "gw equal", torch.equal(gw, gwr))
    if bad.any():
        idx = bad.nonzero()
        print(" first", idx[:5].tolist(), " last", idx[-3:].tolist())
        print(" bad channels", sorted(set(idx[:, 3].tolist()))[:40])
        print(" bad rows", sorted(set(idx[:, 1].tolist())), "bad cols", sorted(set(idx[:, 2].tolist()))[:40])
        i = tuple(idx[0].tolist())
        print(" got", dz[i].item(), "want", ref[i].item())
