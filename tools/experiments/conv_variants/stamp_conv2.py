"""Diagnostic: in-kernel cycle stamps of the v2 conv3x3 forward (library built by
`python tools/build_variant.py stamp conv_mfma2.hip -DUMI2_STAMP`).  Prints, per layer, median per-wave cycles of prologue /
main loop / epilogue and the share of the loop spent in the staging wait (vmcnt) and in the end-of-chunk barrier."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import numpy as np
import torch
import bench
from umi import lib as L, ops

lib = ctypes.CDLL(os.path.join(REPO, "tools", "_ab", "libunetmi_stamp.so"))
conv = lib.umi_conv_fwd
conv.restype = ctypes.c_int
conv.argtypes = L.SIGNATURES["umi_conv_fwd"][1]
lib.umi_debug_read_stamps2.argtypes = [ctypes.c_void_p]
impl = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lib.umi_tune_conv3x3_impl(impl)
g = torch.Generator(device="cuda").manual_seed(1)
for name, n, h, w, ci, co in bench.double_conv_shapes(1, 64, 512, 512, 16):
    if ci < 16:
        continue
    x = torch.randn(n, h, w, ci, device="cuda", generator=g).half()
    wgt = torch.randn(co, ci, 3, 3, device="cuda", generator=g) * (2.0 / (9 * ci)) ** 0.5
    tx = ops.passthrough_tx(ci, "cuda"); tx[:, 3] = 0.0
    y = torch.empty(n, h, w, co, device="cuda", dtype=torch.float16)
    wp = ops.pack_conv_fwd(wgt, torch.float16, k8=True)
    rows = n * ((w + 31) // 32) * ((h + 15) // 16)
    part = torch.empty(rows * 2 * co, device="cuda")
    args = (x.data_ptr(), ci, tx.data_ptr(), wp.data_ptr(), None, y.data_ptr(), co, part.data_ptr(), n, h, w, ci, co,
            3, 3, 1, 1, h, w, 0, 0, h, w, 1, 1, 0, torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert conv(*args) == 0
    torch.cuda.synchronize()
    buf = np.zeros(2 * 4096 * 8, dtype=np.uint64)
    assert lib.umi_debug_read_stamps2(buf.ctypes.data) == 0
    nblk = min(1024, rows * ((co + 63) // 64))
    b = buf.reshape(-1, 8)[: nblk * 4].astype(np.float64)
    med = np.median(b, axis=0)
    nch = med[5]
    b2 = buf[4096 * 8:].reshape(-1, 8)[: nblk * 4].astype(np.float64)
    m2 = np.median(b2, axis=0) / nch
    print(f"{name:8s} {ci:4d}->{co:4d}@{h:3d} chunks={nch:.0f} prologue={med[0]:7.0f} loop={med[1]:8.0f} ({med[1]/nch:6.0f}/chunk) epilogue={med[2]:7.0f} "
          f"vmwait={med[3]/nch:6.0f}/chunk barrier={med[4]/nch:6.0f}/chunk  total={med[0]+med[1]+med[2]:8.0f} | per chunk: dma={m2[0]:5.0f} dx0={m2[1]:5.0f} stage={m2[4]:5.0f} dx1={m2[2]:5.0f} dx2={m2[3]:5.0f}", flush=True)
