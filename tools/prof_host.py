"""Host-side profile of the eager U-Net training step (where does the Python time of ~350 launches go?)."""
import cProfile, os, pstats, sys, io
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "unet-torch_amd")]
import torch
import Model, loss as L
from umi import optim as uo
L.CLASS_NUMBER = 2
m = Model.UNet(1, 2, 64, compute_dtype="fp16").cuda().train()
opt = uo.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
B = int(os.environ.get("B", "2")); S = int(os.environ.get("S", "256"))
x = torch.randn(B, 1, S, S, device="cuda"); y = torch.randint(0, 2, (B, S, S), device="cuda").float()
def step():
    loss = L.calc_loss(m(x), y, loss_type="dice_bce_mc")
    opt.zero_grad()
    loss.backward()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host issue time per step: {(t1 - t0) / 20 * 1e3:.2f} ms (B={B}, {S}x{S}: GPU time is small, so this is the host cost)")
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3500])
