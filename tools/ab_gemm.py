"""Timing of the pointwise matrix-core kernel (umi_conv_fwd, R = S = 1) on the ViT linear shapes of TransUNet config 4
(4,704 tokens) per tile override (UMI_C1_TILE): us per launch and TFLOP/s.  usage: python tools/ab_gemm.py [tiles ...]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import torch
from umi import ops
tiles = sys.argv[1:] or ["", "128x64", "128x128", "256x64", "256x128"]
M = 4704
dev = "cuda"
for K, N in [(768, 768), (3072, 768), (768, 3072), (768, 2304), (2304, 768)]:
    x = torch.randn(1, 1, M, K, device=dev).half()
    w = torch.randn(N, K, 1, 1, device=dev) * K ** -0.5
    y = torch.empty(1, 1, M, N, device=dev, dtype=torch.float16)
    wp = ops.pack_conv_fwd(w, torch.float16, k8=True)
    line = [f"K={K:5d} N={N:5d}"]
    ref = None
    for t in tiles:
        if t:
            os.environ["UMI_C1_TILE"] = t
        else:
            os.environ.pop("UMI_C1_TILE", None)
        for _ in range(3):
            ops.conv_fwd(x, None, lambda lay: wp, None, y, 1, 1, 1, 0)
        torch.cuda.synchronize()
        if ref is None:
            ref = y.clone()
        else:
            assert torch.equal(ref, y), (K, N, t)
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.conv_fwd(x, None, lambda lay: wp, None, y, 1, 1, 1, 0)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        us = sorted(ts)[2]
        line.append(f"{t or 'auto':8s} {us:6.1f} us {2 * M * K * N / us / 1e6:5.0f} TF")
    print("  ".join(line), flush=True)
