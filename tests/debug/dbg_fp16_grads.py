import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import numpy as np, torch
import Model, loss as L
from oracle import recipe, ref_unet
from umi import graph as G
name = sys.argv[1] if len(sys.argv) > 1 else "unet_3_4_8"
g = np.load(os.path.join(REPO, "tests/golden", name + ".npz"))
cin, ncls, feat = int(g["cin"]), int(g["ncls"]), int(g["feat"])
B, H, W, seed = int(g["B"]), int(g["H"]), int(g["W"]), int(g["seed"])
ref = ref_unet.RefUNet(cin, ncls, feat, False); ref.load_state_dict(recipe.fill_state_dict(ref.state_dict(), seed=seed))
x, lab = recipe.synthetic_batch(B, cin, H, W, ncls, seed=seed)
L.CLASS_NUMBER = ncls
ref.train(); rl = ref_unet.dice_bce_mc(ref(x), lab, ncls); rl.backward()
for mode, scale_mul in (("fp32", 1), ("fp16", 1), ("fp16", 16), ("fp16", 1/16)):
    orig = G.default_loss_scale
    G.default_loss_scale = lambda dt, n, o=orig, s=scale_mul: o(dt, n) * (s if dt == torch.float16 else 1)
    m = Model.UNet(cin, ncls, feat, False, compute_dtype=mode); m.load_state_dict(recipe.fill_state_dict(m.state_dict(), seed=seed)); m.to("cuda").train()
    out = m(x.cuda()); l = L.calc_loss(out, lab.cuda(), loss_type="dice_bce_mc"); l.backward()
    G.default_loss_scale = orig
    print(mode, scale_mul, "loss", l.item(), rl.item())
    for (k, p), (_, rp) in zip(m.named_parameters(), ref.named_parameters()):
        e = ((p.grad.cpu() - rp.grad).norm() / (rp.grad.norm() + 1e-30)).item()
        if e > 0.02 or mode == "fp32" and e > 1e-3:
            print(f"   {k:50s} rel {e:.3e}  |ref| {rp.grad.norm().item():.3e}")
