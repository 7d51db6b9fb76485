"""Cause of the `hipStreamEndCapture` crash seen in round 1 (tools/experiments/exp_graph.py, first version): does a LIVE
autograd graph of an eager default-stream step (its loss tensor still referenced) make the captured backward run the
parameters' cached AccumulateGrad nodes on the default, non-capturing stream?

usage: exp_graph_accgrad.py keep|drop [features batch size]     (one mode per process; run each in a child)"""
import faulthandler, gc, os, sys, warnings
faulthandler.enable()
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import torch
import loss as L
import Model

mode = sys.argv[1]
F_, B_, S_ = (int(v) for v in (sys.argv[2:5] + ["16", "4", "128"][len(sys.argv) - 2:]))
L.CLASS_NUMBER = 2
torch.manual_seed(0)
m = Model.UNet(1, 2, F_, compute_dtype="fp16").cuda().train()
opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
x = torch.randn(B_, 1, S_, S_, device="cuda")
lab = torch.randint(0, 2, (B_, S_, S_), device="cuda").float()


def step():
    l = L.calc_loss(m(x), lab, loss_type="dice_bce_mc")
    opt.zero_grad(set_to_none=True)
    l.backward()
    opt.step()
    return l


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
kept = None
for _ in range(2):
    kept = step()                     # eager, default stream; `kept` holds the step's autograd graph
torch.cuda.synchronize()
if mode == "drop":
    kept = None
    gc.collect()
print("before capture, mode", mode, flush=True)
warnings.simplefilter("always")
g = torch.cuda.CUDAGraph()
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    with torch.cuda.graph(g):
        static_loss = step()
for ww in w:
    print("WARNING:", str(ww.message)[:300], flush=True)
print("captured", flush=True)
g.replay()
torch.cuda.synchronize()
print("replayed ok", float(static_loss), flush=True)
