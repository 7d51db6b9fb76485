"""Diagnostic: in-kernel cycle stamps of the persistent conv3x3 forward (library built by
`python tools/build_variant.py stamp3 conv_mfma3.hip -DUMI3_STAMP`): per tile, median cycles of the chunk loop, the epilogue
and the plan/barrier tail; per chunk, the staging wait and the barrier."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import numpy as np
import torch
import bench
from umi import lib as L, ops

lib = ctypes.CDLL(os.path.join(REPO, "tools", "_ab", "libunetmi_stamp3.so"))
conv = lib.umi_conv_fwd
conv.restype = ctypes.c_int
conv.argtypes = L.SIGNATURES["umi_conv_fwd"][1]
lib.umi_debug_read_stamps3.argtypes = [ctypes.c_void_p]
lib.umi_tune_conv3x3_impl(5)
g = torch.Generator(device="cuda").manual_seed(1)
for name, n, h, w, ci, co in bench.double_conv_shapes(1, 64, 512, 512, 16):
    if ci < 16:
        continue
    x = torch.randn(n, h, w, ci, device="cuda", generator=g).half()
    wgt = torch.randn(co, ci, 3, 3, device="cuda", generator=g) * (2.0 / (9 * ci)) ** 0.5
    tx = ops.passthrough_tx(ci, "cuda"); tx[:, 3] = 0.0
    y = torch.empty(n, h, w, co, device="cuda", dtype=torch.float16)
    wp = ops.pack_conv_fwd(wgt, torch.float16, k8=True)
    rows = 4 * n * ((w + 31) // 32) * ((h + 15) // 16)
    part = torch.empty(rows * 2 * co, device="cuda")
    args = (x.data_ptr(), ci, tx.data_ptr(), wp.data_ptr(), None, y.data_ptr(), co, part.data_ptr(), n, h, w, ci, co,
            3, 3, 1, 1, h, w, 0, 0, h, w, 1, 1, 0, torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert conv(*args) == 0
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    assert lib.umi_debug_read_stamps3(buf.ctypes.data) == 0
    b = buf.reshape(-1, 8)[: 512 * 4].astype(np.float64)
    b = b[b[:, 4] > 0]
    med = np.median(b, axis=0)
    tiles, nch = med[4], med[7]
    print(f"{name:8s} {ci:4d}->{co:4d}@{h:3d} tiles/WG={tiles:.0f} chunks={nch:.0f} total={med[0]:8.0f} per tile: loop={med[1]/tiles:7.0f} "
          f"({med[1]/tiles/nch:6.0f}/chunk) epilogue={med[2]/tiles:6.0f} plan+bar={med[3]/tiles:6.0f} | per chunk: vmwait={med[6]/tiles/nch:5.0f} "
          f"barrier={med[5]/tiles/nch:5.0f}", flush=True)
