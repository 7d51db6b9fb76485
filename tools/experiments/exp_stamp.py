"""Diagnostic build with in-kernel s_memtime stamps: where does a main-loop iteration of conv3x3_mfma spend its cycles?
(shares, not absolute times: the stamps' fences forbid overlaps the real kernel has)"""
import ctypes, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import numpy as np, torch
from umi import build as B, ops, lib as L
out = os.path.join(REPO, "gpurun_out", "libexp_stamp.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call([B.HIPCC] + B.FLAGS + ["-DUMI_STAMP"] + sys.argv[1:] + B.sources() + ["-o", out])
lb = ctypes.CDLL(out)
f = lb.umi_conv_fwd; f.restype = ctypes.c_int; f.argtypes = L.SIGNATURES["umi_conv_fwd"][1]
for nm, n, h, w, ci, co in [("64->64@512", 16, 512, 512, 64, 64), ("128->128@256", 16, 256, 256, 128, 128),
                            ("512->512@64", 16, 64, 64, 512, 512), ("1024->1024@32", 16, 32, 32, 1024, 1024)]:
    x = torch.randn(n, h, w, ci, device="cuda").half()
    wp = ops.pack_conv_fwd(torch.randn(co, ci, 3, 3, device="cuda") * 0.02, torch.float16, k8=True)
    tx = ops.passthrough_tx(ci, "cuda"); tx[:, 3] = 0
    y = torch.empty(n, h, w, co, device="cuda", dtype=torch.float16)
    part = torch.empty(n * h * w // 64 * 2 * co, device="cuda")
    txp = None if os.environ.get("NOTX") else tx.data_ptr()
    args = (x.data_ptr(), ci, txp, wp.data_ptr(), None, y.data_ptr(), co, part.data_ptr(), n, h, w, ci, co, 3, 3, 1, 1,
            h, w, 0, 0, h, w, 1, 1, 0, torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert f(*args) == 0
    torch.cuda.synchronize()
    buf = np.zeros(2048 * 8, dtype=np.uint64)
    lb.umi_debug_read_stamps.argtypes = [ctypes.c_void_p]
    assert lb.umi_debug_read_stamps(buf.ctypes.data) == 0
    b = buf.reshape(2048, 8).astype(np.float64)
    b = b[b[:, 5] > 0]
    per = b[:, :5] / b[:, 5:6]
    m = np.median(per, axis=0)
    print(f"{nm}: per-chunk cycles (median over waves): load-wait {m[0]:.0f} | transform+ds_write {m[1]:.0f} | barrier1 {m[2]:.0f} | "
          f"issue+MFMA {m[3]:.0f} | barrier2 {m[4]:.0f} | total {m.sum():.0f}  (72 MFMAs = 2304 cyc alone) || per block: prologue "
          f"{np.median(b[:, 6]):.0f}  loop {np.median(b[:, :5].sum(1)):.0f}  epilogue {np.median(b[:, 7]):.0f}", flush=True)
