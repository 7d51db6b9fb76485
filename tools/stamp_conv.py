"""In-kernel s_memtime stamps of conv_mfma.hip on the bench's DoubleConv launches (diagnostic build:
`python tools/build_variant.py stamp conv_mfma.hip -DUMI_STAMP` here, then on the GPU box `python tools/stamp_conv.py [impl]`).
Per layer, medians over waves: cycles per chunk in each main-loop segment, and prologue / loop / epilogue per workgroup.
Shares, not absolute times: the stamps' fences forbid overlaps the real kernel has (+~11 % wave cycles)."""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import numpy as np
import torch
import bench
from umi import lib as L, ops

impl = int(sys.argv[1]) if len(sys.argv) > 1 else 2
lb = ctypes.CDLL(os.path.join(REPO, "tools", "_ab", "libunetmi_stamp.so"))
f = lb.umi_conv_fwd
f.restype = ctypes.c_int
f.argtypes = L.SIGNATURES["umi_conv_fwd"][1]
lb.umi_tune_conv3x3_impl(impl)
lb.umi_debug_read_stamps.argtypes = [ctypes.c_void_p]
for name, n, h, w, ci, co in bench.double_conv_shapes(1, 64, 512, 512, 16):
    if ci < 16:
        continue
    x = torch.randn(n, h, w, ci, device="cuda").half()
    wp = ops.pack_conv_fwd(torch.randn(co, ci, 3, 3, device="cuda") * 0.02, torch.float16, k8=True)
    tx = ops.passthrough_tx(ci, "cuda")
    tx[:, 3] = 0
    y = torch.empty(n, h, w, co, device="cuda", dtype=torch.float16)
    part = torch.empty(n * h * w // 64 * 2 * co, device="cuda")
    args = (x.data_ptr(), ci, tx.data_ptr(), wp.data_ptr(), None, y.data_ptr(), co, part.data_ptr(), n, h, w, ci, co, 3, 3, 1, 1,
            h, w, 0, 0, h, w, 1, 1, 0, torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert f(*args) == 0
    torch.cuda.synchronize()
    buf = np.zeros(2048 * 8, dtype=np.uint64)
    assert lb.umi_debug_read_stamps(buf.ctypes.data) == 0
    b = buf.reshape(2048, 8).astype(np.float64)
    b = b[b[:, 5] > 0]
    m = np.median(b[:, :5] / b[:, 5:6], axis=0)
    tiles = b[:, 5] / (ci // 16)
    life = b[:, 6] + b[:, :5].sum(1) + b[:, 7] * tiles
    print(f"   WG life (prologue + loop + epilogues), k cycles: min {life.min() / 1e3:.0f} p10 {np.percentile(life, 10) / 1e3:.0f} "
          f"median {np.median(life) / 1e3:.0f} p90 {np.percentile(life, 90) / 1e3:.0f} max {life.max() / 1e3:.0f}; tiles/WG {tiles.min():.0f}..{tiles.max():.0f}")
    print(f"{name:9s} {ci:4d}->{co:4d}@{h:3d} chunks={ci // 16:2d} | per chunk: load-wait {m[0]:5.0f} stage {m[1]:5.0f} barrier1 {m[2]:4.0f} "
          f"issue+MFMA {m[3]:5.0f} barrier2 {m[4]:4.0f} = {m.sum():5.0f} | per WG: prologue {np.median(b[:, 6]):6.0f} "
          f"loop {np.median(b[:, :5].sum(1)):7.0f} epilogue {np.median(b[:, 7]):6.0f}", flush=True)
