"""How fast does the vendor BLAS (through torch.matmul / F.linear) run the ViT linear shapes of TransUNet config 4?
(Decision aid for routing the plain GEMMs to hipBLASLt; GPU-event timing of 50 back-to-back calls.)"""
import torch
import torch.nn.functional as F
M = 4704
for K, N in [(768, 768), (3072, 768), (768, 3072), (768, 2304), (2304, 768)]:
    x = torch.randn(M, K, device="cuda", dtype=torch.float16)
    w = torch.randn(N, K, device="cuda", dtype=torch.float16)
    b = torch.randn(N, device="cuda", dtype=torch.float16)
    dy = torch.randn(M, N, device="cuda", dtype=torch.float16)
    res = []
    for name, fn in (("fwd+bias", lambda: F.linear(x, w, b)), ("dgrad", lambda: dy @ w), ("wgrad", lambda: dy.t() @ x)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        res.append(f"{name} {us:6.1f} us {2 * M * K * N / us / 1e6:5.0f} TF")
    print(f"K={K:5d} N={N:5d}  " + "   ".join(res), flush=True)
