"""Tensor-level wrappers for the TransUNet kernels of libunetmi (see include/unetmi.h, 'TransUNet path')."""
import torch

from . import lib as L
from .ops import _dt, _nhwc, _ptr, _stream, workspace, pack_kn


def _rows(t):
    """[N,H,W,C] view -> (M rows, C, ld)."""
    N, H, W, C, ld = _nhwc(t)
    return N * H * W, C, ld


def wstd_fwd(w, eps=1e-5):
    Co = w.shape[0]
    K = w[0].numel()
    wf = w.detach().float().contiguous()
    ws = torch.empty_like(wf)
    rstd = torch.empty(Co, dtype=torch.float32, device=w.device)
    L.check(L.fn("umi_wstd_fwd")(wf.data_ptr(), ws.data_ptr(), rstd.data_ptr(), Co, K, eps, _stream()), "umi_wstd_fwd")
    return ws, rstd


def wstd_bwd(ws, rstd, g):
    dw = torch.empty_like(ws)
    L.check(L.fn("umi_wstd_bwd")(ws.data_ptr(), rstd.data_ptr(), g.data_ptr(), dw.data_ptr(), ws.shape[0], ws[0].numel(),
                                 _stream()), "umi_wstd_bwd")
    return dw


def gn_fwd(x, gamma, beta, groups, eps, relu, res, y):
    N, H, W, C, ldx = _nhwc(x)
    _, _, _, _, ldy = _nhwc(y)
    mean = torch.empty(N * groups, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    ldr = _nhwc(res)[4] if res is not None else 0
    ws = workspace(L.fn("umi_gn_fwd_ws_bytes")(N, H * W, C), x.device)
    L.check(L.fn("umi_gn_fwd")(x.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(), _ptr(res), ldr, y.data_ptr(), ldy,
                               mean.data_ptr(), rstd.data_ptr(), int(relu), N, H * W, C, groups, eps, _dt(x),
                               ws.data_ptr(), ws.numel(), _stream()),
            "umi_gn_fwd")
    return mean, rstd


def gn_bwd(dy, y, x, mean, rstd, gamma, groups, relu, dx, dres, out_scale, keep_part=False):
    """keep_part: dgamma / dbeta are NOT computed; returns the per-sample rows [N][2][C] for gn_param_grads_group instead."""
    N, H, W, C, ldx = _nhwc(x)
    ws = workspace(L.fn("umi_gn_bwd_ws_bytes")(N, H * W, C, groups), x.device)
    if keep_part:
        part, dg, db = torch.empty(N * 2 * C, dtype=torch.float32, device=x.device), None, None
    else:
        part, dg = None, torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty_like(dg)
    L.check(L.fn("umi_gn_bwd")(dy.data_ptr(), _nhwc(dy)[4], y.data_ptr(), _nhwc(y)[4], x.data_ptr(), ldx, mean.data_ptr(),
                               rstd.data_ptr(), gamma.data_ptr(), int(relu), dx.data_ptr(), _nhwc(dx)[4], _ptr(dres),
                               _nhwc(dres)[4] if dres is not None else 0, _ptr(dg), _ptr(db), out_scale, N, H * W,
                               C, groups, _dt(x), ws.data_ptr(), ws.numel(), _ptr(part), _stream()), "umi_gn_bwd")
    return part if keep_part else (dg, db)


def gn_param_grads_group(parts, N, dgammas, dbetas, out_scale):
    """dgammas[i] / dbetas[i] <- out_scale * sum over the N samples of parts[i] ([N][2][C_i]): one launch per 16 layers."""
    import ctypes
    n = len(parts)
    arr = ctypes.c_void_p * n
    pp, pg, pb = (arr(*[t.data_ptr() for t in ts]) for ts in (parts, dgammas, dbetas))
    cs = (ctypes.c_int * n)(*[t.numel() for t in dgammas])
    L.check(L.fn("umi_gn_param_grads_group")(n, ctypes.cast(pp, ctypes.c_void_p), ctypes.cast(cs, ctypes.c_void_p), N,
                                             ctypes.cast(pg, ctypes.c_void_p), ctypes.cast(pb, ctypes.c_void_p), out_scale,
                                             _stream()), "umi_gn_param_grads_group")


def pool3s2_fwd(x, y, idx=None):
    """idx: optional uint8 tensor [N,Ho,Wo,C] receiving the winning tap of every output element (fp16, C % 8 == 0)."""
    N, H, W, C, ldx = _nhwc(x)
    L.check(L.fn("umi_pool3s2_fwd")(x.data_ptr(), ldx, y.data_ptr(), _nhwc(y)[4], _ptr(idx), N, H, W, C, _dt(x), _stream()),
            "umi_pool3s2_fwd")


def pool3s2_bwd(dy, x, dx, idx=None):
    N, H, W, C, ldx = _nhwc(x)
    L.check(L.fn("umi_pool3s2_bwd")(dy.data_ptr(), _nhwc(dy)[4], x.data_ptr(), ldx, _ptr(idx), dx.data_ptr(), _nhwc(dx)[4],
                                    N, H, W, C, _dt(x), _stream()), "umi_pool3s2_bwd")


def ln_fwd(x, gamma, beta, eps, y):
    M, C, ldx = _rows(x)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    L.check(L.fn("umi_ln_fwd")(x.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), _rows(y)[2],
                               mean.data_ptr(), rstd.data_ptr(), M, C, eps, _dt(x), _stream()), "umi_ln_fwd")
    return mean, rstd


def ln_bwd(dy, x, gamma, mean, rstd, dx, out_scale, keep_part=False):
    """keep_part: dgamma / dbeta are NOT computed; returns (partial rows [rows][2][C], rows) for gn_param_grads_group."""
    import ctypes
    M, C, ldx = _rows(x)
    nb = L.fn("umi_ln_bwd_ws_bytes")(M, C)
    rows = ctypes.c_int(0)
    if keep_part:
        ws, dg, db = torch.empty(nb, dtype=torch.uint8, device=x.device), None, None
    else:
        ws, dg = workspace(nb, x.device), torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty_like(dg)
    L.check(L.fn("umi_ln_bwd")(dy.data_ptr(), _rows(dy)[2], x.data_ptr(), ldx, gamma.data_ptr(), mean.data_ptr(),
                               rstd.data_ptr(), dx.data_ptr(), _rows(dx)[2], _ptr(dg), _ptr(db), out_scale, M, C,
                               _dt(x), ws.data_ptr(), ws.numel(), ctypes.addressof(rows), _stream()), "umi_ln_bwd")
    return (ws.view(torch.float32), rows.value) if keep_part else (dg, db)


def elementwise(mode, x, g, y, bcast_rows=0):
    M, C, ldx = _rows(x)
    ldg = 0
    if g is not None:
        ldg = g.stride(-2) if g.dim() >= 2 else C
    L.check(L.fn("umi_elementwise")(mode, x.data_ptr(), ldx, _ptr(g), ldg, y.data_ptr(), _rows(y)[2], M, C, bcast_rows,
                                    _dt(x), _stream()), "umi_elementwise")


def gelu_fwd(x, y):
    elementwise(0, x, None, y)


def gelu_bwd(pre, g, y):
    elementwise(1, pre, g, y)


def add(a, b, y):
    elementwise(2, a, b, y)


def add_bcast(a, rows_tensor, y, bcast_rows):
    elementwise(3, a, rows_tensor, y, bcast_rows)


def dropout(x, y, mask, backward, p, seed, tx=None, seed_dev=None):
    """seed_dev: optional int32 device scalar added into the seed inside the kernel (HIP-graph replays)."""
    M, C, ldx = _rows(x)
    L.check(L.fn("umi_dropout")(x.data_ptr(), ldx, y.data_ptr(), _rows(y)[2], mask.data_ptr(), int(backward), p,
                                seed & 0xFFFFFFFF, M, C, _dt(x), _ptr(tx), _ptr(seed_dev), _stream()), "umi_dropout")


def dropout_fused(x, y, mask, backward, p, seed, seed_dev=None, aux=None, gelu=False):
    """Forward y = dropout(gelu(x) if gelu else x) + (aux or 0); backward y = dropout'(x) * (gelu'(aux) if gelu else 1).
    False where the fused kernel does not apply (nothing was launched)."""
    M, C, ldx = _rows(x)
    st = L.fn("umi_dropout_fused")(x.data_ptr(), ldx, y.data_ptr(), _rows(y)[2], mask.data_ptr(), int(backward), p,
                                   seed & 0xFFFFFFFF, M, C, _dt(x), _ptr(seed_dev), _ptr(aux),
                                   _rows(aux)[2] if aux is not None else 0, int(gelu), _stream())
    if st == -2:
        return False
    L.check(st, "umi_dropout_fused")
    return True


def linear_fused(x, wp8, bias, y, epi, p, seed, seed_dev, mask, aux=None, y2=None):
    """umi_linear_fused: y = x W + b with the block's elementwise tail in the GEMM epilogue (epi 1: y2 = dropout(gelu(y));
    epi 2: y = dropout(x W + b) + aux).  False where the matrix-core kernel does not apply (nothing was launched)."""
    M, Ci, ldx = _rows(x)
    _, Co, ldy = _rows(y)
    st = L.fn("umi_linear_fused")(x.data_ptr(), ldx, wp8.data_ptr(), _ptr(bias), y.data_ptr(), ldy, M, Ci, Co, int(epi), p,
                                  seed & 0xFFFFFFFF, _ptr(seed_dev), mask.data_ptr(), _ptr(aux),
                                  _rows(aux)[2] if aux is not None else 0, _ptr(y2), _rows(y2)[2] if y2 is not None else 0,
                                  _dt(x), _stream())
    if st == -2:
        return False
    L.check(st, "umi_linear_fused")
    return True


def attn_fwd(q, k, v, o, heads):
    B, _, N, C = q.shape
    D = C // heads
    ld = _nhwc(q)[4]
    assert _nhwc(k)[4] == ld and _nhwc(v)[4] == ld
    lse = torch.empty(B * heads * N, dtype=torch.float32, device=q.device)
    L.check(L.fn("umi_attn_fwd")(q.data_ptr(), k.data_ptr(), v.data_ptr(), ld, o.data_ptr(), _nhwc(o)[4], lse.data_ptr(), B,
                                 N, heads, D, _dt(q), _stream()), "umi_attn_fwd")
    return lse


def attn_bwd(q, k, v, o, dO, lse, dq, dk, dv, heads):
    B, _, N, C = q.shape
    D = C // heads
    ld, ldo, ldd = _nhwc(q)[4], _nhwc(o)[4], _nhwc(dq)[4]
    assert _nhwc(dO)[4] == ldo and _nhwc(dk)[4] == ldd and _nhwc(dv)[4] == ldd
    delta = torch.empty_like(lse)
    L.check(L.fn("umi_attn_bwd")(q.data_ptr(), k.data_ptr(), v.data_ptr(), ld, o.data_ptr(), dO.data_ptr(), ldo,
                                 lse.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), ldd, delta.data_ptr(), B, N,
                                 heads, D, _dt(q), _stream()), "umi_attn_bwd")


def bilinear2x(x, y, backward=False, tx=None):
    if backward:                  # x = dy [N,2H,2W,C], y = dx [N,H,W,C]
        N, H, W, C, ldy = _nhwc(y)
        ldx = _nhwc(x)[4]
    else:
        N, H, W, C, ldx = _nhwc(x)
        ldy = _nhwc(y)[4]
    L.check(L.fn("umi_bilinear2x")(x.data_ptr(), ldx, _ptr(tx), y.data_ptr(), ldy, int(backward), N, H, W, C, _dt(x), _stream()),
            "umi_bilinear2x")


from .ops import pack_conv_dgrad_strided  # noqa: E402,F401  (kept under its old name)
