"""Timing of the first conv's kernels (Ci <= 4: stem3x3_fwd / stem3x3_wgrad) at the bench shape (16 x 1 x 512 x 512 -> 64)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import torch
from umi import ops

dev = "cuda"
N, H, W, Ci, Co = 16, 512, 512, 1, 64
x = torch.randn(N, H, W, Ci, device=dev).half()
w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.3
y = torch.empty(N, H, W, Co, device=dev, dtype=torch.float16)
dy = (torch.randn(N, H, W, Co, device=dev) * 0.1).half()
gw = torch.empty(Co, Ci, 3, 3, device=dev)
wp = None


def timeit(fn):
    ts = []
    for r in range(3):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    return sorted(ts)[1]


pk = {}
def fwd():
    ops.conv_fwd(x, None, lambda lay: pk.setdefault(lay, ops.pack_conv_fwd(w, torch.float16, k8=bool(lay))), None, y, 3, 3, 1, 1, want_stats=True)
def wg():
    ops.conv_wgrad(x, None, dy, None, gw, Ci * 9, 9, 1, 1.0, 3, 3, 1, 1)
tf, tw = timeit(fwd), timeit(wg)
by = N * H * W * Co * 2
print(f"stem fwd {tf:.3f} ms ({by / tf / 1e6:.0f} GB/s of the output)   stem wgrad {tw:.3f} ms ({by / tw / 1e6:.0f} GB/s of dy)")
