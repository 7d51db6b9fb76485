"""Debug aid: per-parameter fp32 gradient error of Model.UNet_attention against the oracle."""
import os, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "unet-torch_amd")]
from oracle import recipe, ref_unet
import Model, loss as L
cin, ncls, feat, B, H, W, seed = 1, 2, 8, 2, 64, 64, int(os.environ.get("SEED", "15"))
ref = ref_unet.RefUNetAttention(cin, ncls, feat, False)
ref.load_state_dict(recipe.fill_state_dict(ref.state_dict(), seed=seed))
x, lab = recipe.synthetic_batch(B, cin, H, W, ncls, seed=seed)
L.CLASS_NUMBER = ncls
m = Model.UNet_attention(cin, ncls, feat, False, compute_dtype=os.environ.get("DT", "fp32"))
assert list(m.state_dict().keys()) == list(ref.state_dict().keys())
m.load_state_dict(ref.state_dict()); m.cuda().train(); ref.train()
o = m(x.cuda()); r = ref(x)
print("logits err", ((o.detach().cpu().double() - r.detach().double()).norm() / r.detach().double().norm()).item())
l = L.calc_loss(o, lab.cuda(), loss_type="dice_bce_mc"); rl = ref_unet.dice_bce_mc(r, lab, ncls)
l.backward(); rl.backward()
print("loss", l.item(), rl.item())
worst, rows = 0.0, []
for (k, p), (_, rp) in zip(m.named_parameters(), ref.named_parameters()):
    a, b = p.grad.detach().cpu().double(), rp.grad.double()
    if b.norm() < 1e-9:
        rows.append(f"{k:50s} abs {a.abs().max().item():.2e} (ref ~0: {b.abs().max().item():.1e})"); continue
    e = ((a - b).norm() / b.norm()).item(); worst = max(worst, e)
    rows.append(f"{k:50s} {e:.3e}  |ref|={b.norm().item():.3e}")
print(f"SEED {seed} WORST {worst:.3e}")
if os.environ.get("VERBOSE"):
    print("\n".join(rows))
for (k, v), (_, rv) in zip(m.state_dict().items(), ref.state_dict().items()):
    if "running" in k:
        e = (v.cpu().double() - rv.double()).abs().max().item()
        if e > 1e-5: print("running stat mismatch", k, e)
