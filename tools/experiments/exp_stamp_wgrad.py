"""Diagnostic build with in-kernel s_memtime stamps: where does a tile iteration of wgrad3x3_mfma spend its cycles?
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
f = lb.umi_conv_wgrad; f.restype = ctypes.c_int; f.argtypes = L.SIGNATURES["umi_conv_wgrad"][1]
fw = lb.umi_conv_wgrad_ws_bytes; fw.restype = ctypes.c_size_t; fw.argtypes = L.SIGNATURES["umi_conv_wgrad_ws_bytes"][1]
lb.umi_debug_read_stamps_w.argtypes = [ctypes.c_void_p]
for nm, n, h, w, ci, co in [("64->64@512", 16, 512, 512, 64, 64), ("128->128@256", 16, 256, 256, 128, 128),
                            ("512->512@64", 16, 64, 64, 512, 512), ("1024->1024@32", 16, 32, 32, 1024, 1024)]:
    x = torch.randn(n, h, w, ci, device="cuda").half()
    dy = torch.randn(n, h, w, co, device="cuda").half()
    tx = ops.passthrough_tx(ci, "cuda"); tx[:, 3] = 0
    dW = torch.empty(co, ci, 3, 3, device="cuda")
    nb = fw(n, h, w, ci, co, 3, 3, 1, 0)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    txp = None if os.environ.get("NOTX") else tx.data_ptr()
    args = (x.data_ptr(), ci, txp, dy.data_ptr(), co, None, dW.data_ptr(), ci * 9, 9, 1, 1.0, n, h, w, ci, co, 3, 3, 1, 1, h, w, 1, 0,
            ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        assert f(*args) == 0
    e0.record()
    for _ in range(5):
        assert f(*args) == 0
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    buf = np.zeros(2048 * 8, dtype=np.uint64)
    assert lb.umi_debug_read_stamps_w(buf.ctypes.data) == 0
    if not os.environ.get("UMI_WGRAD_CLASSIC"):
        b = buf.reshape(256, 8, 8).astype(np.float64)
        b = b[b[:, 0, 5] > 0]
        if "-DUMI_STAMP_STEPS" in sys.argv:
            print("   consumer tile-start (B + first A fragments landed), cycles per tile:", np.median(b[:, :4, 2] / b[:, :4, 5]))
        per = b[:, :, :2] / b[:, :, 5:6]
        c, pr = np.median(per[:, :4].reshape(-1, 2), axis=0), np.median(per[:, 4:].reshape(-1, 2), axis=0)
        print(f"{nm}: {ms*1e3:.0f} us (stamped build) | per tile: consumer work {c[0]:.0f} barrier-wait {c[1]:.0f} | producer work {pr[0]:.0f} "
              f"barrier-wait {pr[1]:.0f} cycles (72 MFMAs = 2304)", flush=True)
        continue
    b = buf.reshape(2048, 8).astype(np.float64)
    b = b[b[:, 5] > 0]
    per = b[:, :5] / b[:, 5:6]
    m = np.median(per, axis=0)
    print(f"{nm}: {ms*1e3:.0f} us (stamped build) | per-tile cycles: load-wait {m[0]:.0f} | transform+ds_write {m[1]:.0f} | barrier1 {m[2]:.0f} | "
          f"issue+MFMA {m[3]:.0f} | barrier2 {m[4]:.0f} | total {m.sum():.0f} (72 MFMAs = 2304) || per block: tiles {np.median(b[:,5]):.0f} prologue "
          f"{np.median(b[:, 6]):.0f}  loop {np.median(b[:, :5].sum(1)):.0f}  epilogue {np.median(b[:, 7]):.0f}", flush=True)
