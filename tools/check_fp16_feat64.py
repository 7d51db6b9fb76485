"""Whole-network parity of the BENCHMARKED path: UNet(cin, ncls, 64) in fp16 storage (every MFMA tile configuration of the
bench workload is live at these channel widths) against the CPU oracle, fp32 and with the same fp16 rounding points.

Runs as a child of tests/test_gpu_unet.py::test_unet_fp16_feat64_benchmark_widths with UMI_TRACE_GENERIC=1 (the library then
reports every convolution / weight gradient that falls back to the generic VALU kernels on stderr; the test asserts there
is none) and prints ONE JSON line with the measurements; `--out FILE` also writes it (profiles/r02_fp16_feat64_parity.json is
such a run).  usage: python tools/check_fp16_feat64.py CIN NCLS [--size 64] [--batch 2] [--out FILE]"""
import argparse, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import numpy as np
import torch
import Model
import loss as L
from oracle import recipe, ref_unet

ap = argparse.ArgumentParser()
ap.add_argument("cin", type=int)
ap.add_argument("ncls", type=int)
ap.add_argument("--size", type=int, default=64)
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--seed", type=int, default=64)
ap.add_argument("--out")
a = ap.parse_args()
DEV, F = "cuda", 64
L.CLASS_NUMBER = a.ncls
torch.set_num_threads(min(16, os.cpu_count() or 1))


def rel(x, y):
    x, y = x.detach().double().cpu(), y.detach().double().cpu()
    return ((x - y).norm() / (y.norm() + 1e-30)).item()


def cos(x, y):
    x, y = x.detach().double().cpu().flatten(), y.detach().double().cpu().flatten()
    return (x @ y / (x.norm() * y.norm() + 1e-30)).item()


ref = ref_unet.RefUNet(a.cin, a.ncls, F, False)
ref.load_state_dict(recipe.fill_state_dict(ref.state_dict(), seed=a.seed))
x, lab = recipe.synthetic_batch(a.batch, a.cin, a.size, a.size, a.ncls, seed=a.seed)

m = Model.UNet(a.cin, a.ncls, F, False, compute_dtype="fp16")
m.load_state_dict(ref.state_dict())
m.to(DEV).train()
logits = m(x.to(DEV))
loss = L.calc_loss(logits, lab.to(DEV), loss_type="dice_bce_mc")
loss.backward()
torch.cuda.synchronize()
logits = logits.detach().float().cpu()

ref.train()
rl = ref(x)
rloss = ref_unet.dice_bce_mc(rl, lab, a.ncls)
rloss.backward()
rl = rl.detach()


def quant_run(noise):
    q = ref_unet.RefUNet(a.cin, a.ncls, F, False, quant="fp16")
    q.load_state_dict(ref.state_dict())
    q.train()
    q.noise = noise
    ql = q(x)
    qloss = ref_unet.dice_bce_mc(ql, lab, a.ncls)
    qloss.backward()
    return q, ql.detach(), qloss.item()


q, ql, qloss = quant_run(0.0)
qn, _, _ = quant_run(1e-7)            # the quantised oracle against itself under summation-order-sized noise: the chaos floor

scale = rl.abs().max().item()
e32 = (logits - rl).abs().max().item() / scale
eq = (logits - ql).abs().max().item() / ql.abs().max().item()
# argmax masks away from near-ties of the fp32 oracle: a pixel counts when its top-2 logit margin exceeds 4x the measured
# worst logit error (a pixel below that margin can legitimately flip under fp16 storage)
top2 = rl.topk(2, dim=1).values
margin = (top2[:, 0] - top2[:, 1])
clear = margin > 4.0 * (logits - rl).abs().max().item()
am, ar = logits.argmax(1), rl.argmax(1)
mism_clear = int(((am != ar) & clear).sum())
mism_all = int((am != ar).sum())
names = [k for k, _ in m.named_parameters()]
g_q = {k: rel(p.grad, qp.grad) for (k, p), (_, qp) in zip(m.named_parameters(), q.named_parameters())}
g_cos = {k: cos(p.grad, rp.grad) for (k, p), (_, rp) in zip(m.named_parameters(), ref.named_parameters())}
g_32 = {k: rel(p.grad, rp.grad) for (k, p), (_, rp) in zip(m.named_parameters(), ref.named_parameters())}
floor = {k: rel(p.grad, qp.grad) for (k, p), (_, qp) in zip(qn.named_parameters(), q.named_parameters())}
worst = max(g_q, key=g_q.get)
out = {
    "model": f"UNet({a.cin},{a.ncls},64) fp16 storage, {a.batch}x{a.cin}x{a.size}x{a.size}, seed {a.seed}",
    "logits_max_err_over_scale_vs_fp32_oracle": e32, "logits_max_err_over_scale_vs_fp16_oracle": eq,
    "loss": float(loss.detach()), "loss_fp32_oracle": float(rloss.detach()), "loss_fp16_oracle": qloss,
    "pixels": int(am.numel()), "pixels_clear_of_near_ties": int(clear.sum()), "argmax_mismatch_clear": mism_clear,
    "argmax_mismatch_all": mism_all,
    "grad_rel_l2_vs_fp16_oracle": {"median": float(np.median(list(g_q.values()))), "worst": g_q[worst], "worst_tensor": worst},
    "grad_rel_l2_fp16_oracle_self_noise_floor": {"median": float(np.median(list(floor.values()))), "worst": max(floor.values())},
    "grad_rel_l2_vs_fp32_oracle": {"median": float(np.median(list(g_32.values()))), "worst": max(g_32.values())},
    "grad_cosine_vs_fp32_oracle": {"median": float(np.median(list(g_cos.values()))), "worst": min(g_cos.values()),
                                   "worst_tensor": min(g_cos, key=g_cos.get)},
    "grads_finite": bool(all(torch.isfinite(p.grad).all() for p in m.parameters())),
}
line = json.dumps(out)
print("FP16_FEAT64 " + line, flush=True)
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
