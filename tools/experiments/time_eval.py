"""Times the U-Net forward at the bench shape: training-mode forward (tape recorded), no-grad training-mode forward, eval-mode
forward and umi.infer.predict_mask."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import torch
import Model
from umi import infer

torch.manual_seed(0)
m = Model.UNet(1, 2, 64, compute_dtype="fp16").to("cuda")
x = torch.randn(16, 1, 512, 512, device="cuda")


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


m.train()
print("train fwd (recorded) ms:", t(lambda: m(x)))
with torch.no_grad():
    print("train-mode fwd, no_grad ms:", t(lambda: m(x)))
m.eval()
with torch.no_grad():
    print("eval fwd ms:", t(lambda: m(x)))
print("predict_mask ms:", t(lambda: infer.predict_mask(m, x)))
