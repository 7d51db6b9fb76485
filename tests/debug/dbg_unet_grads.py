"""Debug aid: per-parameter fp32 gradient error of Model.UNet against the oracle, labels from two seeds."""
import os, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "unet-torch_amd")]
from oracle import recipe, ref_unet
import Model, loss as L
cin, ncls, feat, B, H, W, seed = 1, 2, 8, 2, 64, 64, 8
lseed = int(sys.argv[1]) if len(sys.argv) > 1 else seed
ref = ref_unet.RefUNet(cin, ncls, feat, False)
ref.load_state_dict(recipe.fill_state_dict(ref.state_dict(), seed=seed))
x, _ = recipe.synthetic_batch(B, cin, H, W, ncls, seed=seed)
_, lab = recipe.synthetic_batch(B, cin, H, W, ncls, seed=lseed)
L.CLASS_NUMBER = ncls
m = Model.UNet(cin, ncls, feat, False, compute_dtype="fp32")
m.load_state_dict(ref.state_dict()); m.cuda().train(); ref.train()
o = m(x.cuda()); r = ref(x)
o.retain_grad(); r.retain_grad()
l = L.calc_loss(o, lab.cuda(), loss_type="dice_bce_mc"); rl = ref_unet.dice_bce_mc(r, lab, ncls)
l.backward(); rl.backward()
a, b = o.grad.cpu().double(), r.grad.double()
print("logits err", ((o.detach().cpu().double() - r.detach().double()).norm() / r.detach().double().norm()).item())
print("dlogits err", ((a - b).norm() / b.norm()).item(), "loss", l.item(), rl.item())
for (k, p), (_, rp) in zip(m.named_parameters(), ref.named_parameters()):
    a, b = p.grad.detach().cpu().double(), rp.grad.double()
    print(f"{k:50s} {((a-b).norm()/(b.norm()+1e-30)).item():.3e}  |ref|={b.norm().item():.3e}")
