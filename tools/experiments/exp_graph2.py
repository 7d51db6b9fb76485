"""Bounded capture experiments: which part of the step can be captured in a HIP graph?"""
import os, sys, time, faulthandler
faulthandler.enable()
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import torch
import loss as L
import Model
L.CLASS_NUMBER = 2
torch.manual_seed(0)
stage = sys.argv[1]
F_, B_, S_ = (int(v) for v in (sys.argv[2:5] + ["8", "2", "64"][len(sys.argv) - 2:]))
m = Model.UNet(1, 2, F_, compute_dtype="fp16").cuda().train()
x = torch.randn(B_, 1, S_, S_, device="cuda")
lab = torch.randint(0, 2, (B_, S_, S_), device="cuda").float()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
def fwd():
    with torch.no_grad():
        return m(x)
def fwdbwd():
    out = m(x); l = L.calc_loss(out, lab, loss_type="dice_bce_mc"); opt.zero_grad(set_to_none=True); l.backward(); return l
def full():
    l = fwdbwd(); opt.step(); return l
fn = {"fwd": fwd, "fwdbwd": fwdbwd, "full": full}[stage]
with torch.cuda.stream(s):
    for _ in range(3): fn()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
print("warm ok", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    r = fn()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
print("replayed", stage, float(r.float().sum()), flush=True)
