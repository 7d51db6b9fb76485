"""Debug aid: per-parameter gradient error of Model.UNet_multitask against the oracle (fp32)."""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "unet-torch_amd")]
from oracle import recipe, ref_unet
import Model, loss as L
cin, ncls, feat, B, H, W, seed = 1, 2, 8, 2, 64, 64, int(os.environ.get("SEED", "8"))
ref = ref_unet.RefUNetMultitask(cin, ncls, feat, False)
ref.load_state_dict(recipe.fill_state_dict(ref.state_dict(), seed=seed))
x, lab1 = recipe.synthetic_batch(B, cin, H, W, ncls, seed=seed)
_, lab2 = recipe.synthetic_batch(B, cin, H, W, ncls, seed=seed + 100)
L.CLASS_NUMBER = ncls
if len(sys.argv) > 2 and sys.argv[2] == "W":          # decoder 1 gets decoder 2's weights
    sd = ref.state_dict()
    for k in list(sd):
        if "decod1" in k:
            sd[k] = sd[k.replace("decod1", "decod2")].clone()
    ref.load_state_dict(sd)
if len(sys.argv) > 2 and sys.argv[2] == "G":          # all BatchNorm gammas positive
    sd = ref.state_dict()
    for k in list(sd):
        if sd[k].dim() == 1 and k.endswith("weight"):
            sd[k] = sd[k].abs()
    ref.load_state_dict(sd)
m = Model.UNet_multitask(cin, ncls, feat, False, compute_dtype="fp32")
m.load_state_dict(ref.state_dict()); m.cuda().train(); ref.train()
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
variant = sys.argv[2] if len(sys.argv) > 2 else ""
if variant == "A":                                   # every decoder gets its own concat buffers
    orig = Model._build_decoder
    Model._build_decoder = lambda t, skips, ups, own_buffers=False: orig(t, skips, ups, own_buffers=True)
if variant == "D":                                   # decoder 1 own buffers, decoder 2 shares the encoder's
    orig = Model._build_decoder
    Model._build_decoder = lambda t, skips, ups, own_buffers=False: orig(t, skips, ups, own_buffers=not own_buffers)
o1, o2 = m(x.cuda()); r1, r2 = ref(x)
if mode == "both":
    l = L.calc_loss(o1, lab1.cuda(), loss_type="dice_bce_mc") + L.calc_loss(o2, lab2.cuda(), loss_type="dice_bce_mc")
    rl = ref_unet.dice_bce_mc(r1, lab1, ncls) + ref_unet.dice_bce_mc(r2, lab2, ncls)
elif mode == "1":
    l = L.calc_loss(o1, lab1.cuda(), loss_type="dice_bce_mc"); rl = ref_unet.dice_bce_mc(r1, lab1, ncls)
else:
    l = L.calc_loss(o2, lab2.cuda(), loss_type="dice_bce_mc"); rl = ref_unet.dice_bce_mc(r2, lab2, ncls)
l.backward(); rl.backward()
worst = 0.0
for (k, p), (_, rp) in zip(m.named_parameters(), ref.named_parameters()):
    if p.grad is not None and rp.grad is not None:
        worst = max(worst, ((p.grad.detach().cpu().double() - rp.grad.double()).norm() / (rp.grad.double().norm() + 1e-30)).item())
print(f"SEED {seed} WORST {worst:.3e}")
for (k, p), (_, rp) in zip(m.named_parameters(), ref.named_parameters()):
    if p.grad is None or rp.grad is None:
        print(f"{k:50s} none {p.grad is None} {rp.grad is None}"); continue
    a, b = p.grad.detach().cpu().double(), rp.grad.double()
    print(f"{k:50s} {((a-b).norm()/(b.norm()+1e-30)).item():.3e}  |ref|={b.norm().item():.3e}")
