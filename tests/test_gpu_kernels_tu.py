"""Kernel-level GPU parity for the TransUNet ops of libunetmi vs plain PyTorch fp32 (CPU)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _gpu():
    if not torch.cuda.is_available():
        pytest.fail("needs an MI355X")
    from umi import lib, ops, ops_tu
    return lib, ops, ops_tu


def _close(a, b, tol):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert (a - b).abs().max().item() <= tol * (b.abs().max().item() + 1e-12), ((a - b).abs().max().item(), b.abs().max().item())


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-5), (torch.float16, 4e-3)])
def test_group_norm_fwd_bwd(dt, tol):
    lib, ops, T = _gpu()
    g = torch.Generator().manual_seed(1)
    N, H, W, C, G = 2, 7, 9, 64, 32
    x = (torch.randn(N, H, W, C, generator=g) * 2 + 0.5).to(dt)
    res = torch.randn(N, H, W, C, generator=g).to(dt)
    gamma, beta = 0.5 + torch.rand(C, generator=g), 0.1 * torch.randn(C, generator=g)
    dy = torch.randn(N, H, W, C, generator=g).to(dt)
    xr, rr = x.float().permute(0, 3, 1, 2).requires_grad_(True), res.float().permute(0, 3, 1, 2).requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.relu(F.group_norm(xr, G, gr, br, 1e-6) + rr)
    ref.backward(dy.float().permute(0, 3, 1, 2))
    y = torch.empty(N, H, W, C, device=DEV, dtype=dt)
    xd, rd = x.to(DEV), res.to(DEV)
    mean, rstd = T.gn_fwd(xd, gamma.to(DEV), beta.to(DEV), G, 1e-6, True, rd, y)
    _close(y.permute(0, 3, 1, 2), ref, tol)
    dx, dres = torch.empty_like(y), torch.empty_like(y)
    dg, db = T.gn_bwd(dy.to(DEV), y, xd, mean, rstd, gamma.to(DEV), G, True, dx, dres, 0.5)
    _close(dx.permute(0, 3, 1, 2), xr.grad, 10 * tol)
    _close(dres.permute(0, 3, 1, 2), rr.grad, 10 * tol)
    _close(dg, gr.grad * 0.5, 10 * tol)
    _close(db, br.grad * 0.5, 10 * tol)


def test_layer_norm_gelu_pool_bilinear_fp32():
    lib, ops, T = _gpu()
    g = torch.Generator().manual_seed(2)
    B, N, C = 2, 37, 96
    x = torch.randn(B, 1, N, C, generator=g)
    gamma, beta = 0.5 + torch.rand(C, generator=g), 0.1 * torch.randn(C, generator=g)
    dy = torch.randn(B, 1, N, C, generator=g)
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br, 1e-6)
    ref.backward(dy)
    y = torch.empty_like(x, device=DEV)
    mean, rstd = T.ln_fwd(x.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-6, y)
    _close(y, ref, 2e-5)
    dx = torch.empty_like(y)
    dg, db = T.ln_bwd(dy.to(DEV), x.to(DEV), gamma.to(DEV), mean, rstd, dx, 1.0)
    _close(dx, xr.grad, 2e-4); _close(dg, gr.grad, 2e-4); _close(db, br.grad, 2e-4)
    # GELU
    u = torch.randn(B, 1, N, C, generator=g) * 2
    ur = u.clone().requires_grad_(True)
    F.gelu(ur).backward(dy)
    o, du = torch.empty_like(y), torch.empty_like(y)
    T.gelu_fwd(u.to(DEV), o); T.gelu_bwd(u.to(DEV), dy.to(DEV), du)
    _close(o, F.gelu(u), 2e-6); _close(du, ur.grad, 2e-5)
    # max-pool 3x3 / s2 / p0 on an odd map
    xm = torch.randn(2, 15, 17, 8, generator=g)
    xmr = xm.permute(0, 3, 1, 2).clone().requires_grad_(True)
    pr = F.max_pool2d(xmr, 3, 2, 0)
    gp = torch.randn(pr.shape, generator=g)
    pr.backward(gp)
    yp = torch.empty(2, 7, 8, 8, device=DEV)
    T.pool3s2_fwd(xm.to(DEV), yp)
    _close(yp.permute(0, 3, 1, 2), pr, 0.0)
    dxp = torch.empty(2, 15, 17, 8, device=DEV)
    T.pool3s2_bwd(gp.permute(0, 2, 3, 1).contiguous().to(DEV), xm.to(DEV), dxp)
    _close(dxp.permute(0, 3, 1, 2), xmr.grad, 1e-6)
    # bilinear x2 align_corners=True with a consumer transform on the input, + adjoint
    xb = torch.randn(2, 5, 7, 16, generator=g)
    t = torch.zeros(16, 4); t[:, 1] = 1.5; t[:, 2] = 0.1; t[:, 3] = 0.0
    act = torch.clamp_min(xb * 1.5 + 0.1, 0.0).permute(0, 3, 1, 2).clone().requires_grad_(True)
    ub = F.interpolate(act, scale_factor=2, mode="bilinear", align_corners=True)
    gb = torch.randn(ub.shape, generator=g)
    ub.backward(gb)
    yb = torch.empty(2, 10, 14, 16, device=DEV)
    T.bilinear2x(xb.to(DEV), yb, False, t.to(DEV))
    _close(yb.permute(0, 3, 1, 2), ub, 2e-6)
    dxb = torch.empty(2, 5, 7, 16, device=DEV)
    T.bilinear2x(gb.permute(0, 2, 3, 1).contiguous().to(DEV), dxb, True)
    _close(dxb.permute(0, 3, 1, 2), act.grad, 2e-5)


@pytest.mark.parametrize("dt,tol,N,heads,D", [(torch.float32, 2e-5, 50, 4, 16), (torch.float16, 3e-3, 196, 3, 64),
                                                 (torch.float16, 3e-3, 301, 2, 64), (torch.float16, 3e-3, 17, 1, 64),
                                                 (torch.float16, 3e-3, 40, 2, 32),
                                                 # 1,024 tokens x 12 heads x 64: the ViT-B/16 attention of a 512 x 512 image
                                                 # (BASELINE configs[4], reference vit_seg_modeling.py:82-88)
                                                 (torch.float16, 3e-3, 1024, 12, 64), (torch.float32, 2e-5, 1024, 2, 64)])
def test_attention_fwd_bwd(dt, tol, N, heads, D):
    lib, ops, T = _gpu()
    g = torch.Generator().manual_seed(3)
    B, C = 2, heads * D
    q, k, v = (torch.randn(B, 1, N, C, generator=g).to(dt) for _ in range(3))
    dO = torch.randn(B, 1, N, C, generator=g).to(dt)
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    sp = lambda t: t.view(B, N, heads, D).permute(0, 2, 1, 3)
    p = torch.softmax(sp(qr) @ sp(kr).transpose(-1, -2) / math.sqrt(D), -1)
    ref = (p @ sp(vr)).permute(0, 2, 1, 3).reshape(B, 1, N, C)
    ref.backward(dO.float())
    o = torch.empty(B, 1, N, C, device=DEV, dtype=dt)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    lse = T.attn_fwd(qd, kd, vd, o, heads)
    _close(o, ref, tol)
    dq, dk, dv = torch.empty_like(o), torch.empty_like(o), torch.empty_like(o)
    T.attn_bwd(qd, kd, vd, o, dO.to(DEV), lse, dq, dk, dv, heads)
    _close(dq, qr.grad, 5 * tol); _close(dk, kr.grad, 5 * tol); _close(dv, vr.grad, 5 * tol)


def test_wstd_and_strided_dgrad_fp32():
    lib, ops, T = _gpu()
    g = torch.Generator().manual_seed(4)
    w = torch.randn(24, 16, 3, 3, generator=g)
    wr = w.clone().requires_grad_(True)
    v, m = torch.var_mean(wr, dim=[1, 2, 3], keepdim=True, unbiased=False)
    ws_ref = (wr - m) / torch.sqrt(v + 1e-5)
    gw = torch.randn(w.shape, generator=g)
    ws_ref.backward(gw)
    ws, rstd = T.wstd_fwd(w.to(DEV), 1e-5)
    _close(ws, ws_ref, 2e-6)
    _close(T.wstd_bwd(ws, rstd, gw.to(DEV)), wr.grad, 2e-5)
    # data gradient of a stride-2 3x3 conv (pad 1) and of a stride-2 1x1 conv
    for R, pad in ((3, 1), (1, 0)):
        x = torch.zeros(2, 16, 11, 13, requires_grad=True)
        wc = torch.randn(24, 16, R, R, generator=g) * 0.2
        y = F.conv2d(x, wc, None, 2, pad)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        dx = torch.empty(2, 11, 13, 16, device=DEV)
        from umi.graph_tu import TUTape
        TUTape._strided_dgrad(gy.permute(0, 2, 3, 1).contiguous().to(DEV), T.pack_conv_dgrad_strided(wc.to(DEV), torch.float32),
                              dx, R, R, 2, pad)
        _close(dx.permute(0, 3, 1, 2), x.grad, 2e-5)


@pytest.mark.parametrize("shape", [(2 * 13 * 11, 192, 128), (4704, 768, 768), (300, 256, 128)])
def test_wgrad1x1_mfma_and_colsum(shape):
    lib, ops, T = _gpu()
    M, Ci, Co = shape
    g = torch.Generator().manual_seed(M)
    x = torch.randn(1, 1, M, Ci, generator=g).half()
    dy = (torch.randn(1, 1, M, Co, generator=g) * 0.1).half()
    ref = dy[0, 0].float().t() @ x[0, 0].float()              # [Co, Ci]
    res = {}
    for name, flags in (("mfma", 0), ("generic", lib.CONV_FORCE_GENERIC)):
        gw = torch.empty(Co, Ci, device=DEV)
        ops.conv_wgrad(x.to(DEV), None, dy.to(DEV), None, gw, Ci, 1, 1, 1.0, 1, 1, 1, 0, flags=flags)
        res[name] = gw.cpu()
    scale = ref.abs().max().item()
    assert (res["generic"] - ref).abs().max().item() < 2e-3 * scale
    assert (res["mfma"] - ref).abs().max().item() < 4e-3 * scale
    gb = torch.empty(Co, device=DEV)
    ops.colsum(dy.to(DEV), gb, 0.5)
    _close(gb, dy[0, 0].float().sum(0) * 0.5, 2e-3)
    wide = torch.randn(1, 1, 24, 20000, generator=g)          # position-embedding-like: few rows, very wide
    gp = torch.empty(20000, device=DEV)
    ops.colsum(wide.to(DEV), gp, 1.0)
    _close(gp, wide[0, 0].sum(0), 1e-5)


@pytest.mark.parametrize("R,stride,pad,Ci,Co,H,W", [(3, 2, 1, 64, 128, 11, 13), (1, 2, 0, 128, 64, 11, 13), (3, 2, 1, 128, 128, 55, 55),
                                                   (7, 2, 3, 64, 64, 20, 17), (3, 3, 0, 64, 64, 14, 16)])
def test_strided_conv_mfma_gather_fp16(R, stride, pad, Ci, Co, H, W):
    """Strided convs (resnet_skip.py:52-60) on the tap-gather MFMA kernels: forward, data gradient, weight gradient, with a
    consumer-side transform on the input (zero padding applies AFTER the transform)."""
    lib, ops, T = _gpu()
    from umi.graph_tu import TUTape
    g = torch.Generator().manual_seed(11)
    N = 2
    x = torch.randn(N, Ci, H, W, generator=g).half()
    w = (torch.randn(Co, Ci, R, R, generator=g) * (1.0 / math.sqrt(Ci * R * R)))
    wq = w.half().float()
    scale, shift = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.3
    tx = torch.stack([torch.zeros(Ci), scale, shift, torch.zeros(Ci)], 1).contiguous()          # relu(x*scale + shift)
    xa = torch.relu(x.float() * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).half().float().requires_grad_(True)
    wr = wq.clone().requires_grad_(True)
    y_ref = F.conv2d(xa, wr, None, stride, pad)
    gy = torch.randn(y_ref.shape, generator=g).half()
    y_ref.backward(gy.float())
    Ho, Wo = y_ref.shape[2:]
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = torch.empty(N, Ho, Wo, Co, device=DEV, dtype=torch.float16)
    lay, _ = ops.conv_plan(xd, y, R, R, stride, pad)
    assert lay == 1, "expected the MFMA (k8) path for this shape"
    ops.conv_fwd(xd, tx.to(DEV), lambda l: ops.pack_conv_fwd(w.to(DEV), torch.float16, k8=bool(l)), None, y, R, R, stride, pad)
    _close(y.permute(0, 3, 1, 2), y_ref, 3e-3)
    gyd = gy.permute(0, 2, 3, 1).contiguous().to(DEV)
    dx = torch.empty(N, H, W, Ci, device=DEV, dtype=torch.float16)
    TUTape._strided_dgrad(gyd, lambda l: T.pack_conv_dgrad_strided(w.to(DEV), torch.float16, k8=bool(l)), dx, R, R, stride, pad)
    _close(dx.permute(0, 3, 1, 2), xa.grad, 4e-3)
    dW = torch.empty(Co, Ci, R, R, device=DEV)
    ops.conv_wgrad(xd, tx.to(DEV), gyd, None, dW, Ci * R * R, R * R, 1, 1.0, R, R, stride, pad)
    _close(dW, wr.grad, 4e-3)


def test_root_conv_and_seg_head_narrow_kernels():
    """ResNetV2 root (3 -> 64, 7x7 / s2 / p3, resnet_skip.py:120) and SegmentationHead (16 -> n_classes, 3x3 / p1, fp32 logits
    + bias, vit_seg_modeling.py:317-323): forward and weight gradient, fast kernels vs torch and vs the generic kernels."""
    lib, ops, T = _gpu()
    g = torch.Generator().manual_seed(21)
    # root
    N, H, W, Co = 2, 37, 41, 64
    x = torch.randn(N, 3, H, W, generator=g).half()
    w = torch.randn(Co, 3, 7, 7, generator=g) * 0.1
    wr = w.half().float().clone().requires_grad_(True)
    y_ref = F.conv2d(x.float(), wr, None, 2, 3)
    gy = torch.randn(y_ref.shape, generator=g).half()
    y_ref.backward(gy.float())
    Ho, Wo = y_ref.shape[2:]
    xd = torch.empty(N, H, W, 3, dtype=torch.float16, device=DEV)
    xd.copy_(x.permute(0, 2, 3, 1))
    gyd = gy.permute(0, 2, 3, 1).contiguous().to(DEV)
    for flags in (0, lib.CONV_FORCE_GENERIC):
        y = torch.empty(N, Ho, Wo, Co, device=DEV, dtype=torch.float16)
        ops.conv_fwd(xd, None, lambda l: ops.pack_conv_fwd(w.to(DEV), torch.float16, k8=bool(l)), None, y, 7, 7, 2, 3, flags=flags)
        _close(y.permute(0, 3, 1, 2), y_ref, 3e-3)
        dW = torch.empty(Co, 3, 7, 7, device=DEV)
        ops.conv_wgrad(xd, None, gyd, None, dW, 3 * 49, 49, 1, 1.0, 7, 7, 2, 3, flags=flags)
        _close(dW, wr.grad, 3e-3)
    # segmentation head
    N, H, W, Ci = 2, 21, 19, 16
    for ncls in (2, 4):
        x = torch.randn(N, Ci, H, W, generator=g).half()
        scale, shift = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.3
        tx = torch.stack([torch.zeros(Ci), scale, shift, torch.zeros(Ci)], 1).contiguous()
        xa = torch.relu(x.float() * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
        w = torch.randn(ncls, Ci, 3, 3, generator=g) * 0.2
        b = torch.randn(ncls, generator=g)
        wr = w.half().float().clone().requires_grad_(True)
        y_ref = F.conv2d(xa, wr, b, 1, 1)
        gl = torch.randn(y_ref.shape, generator=g).half()
        y_ref.backward(gl.float())
        xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
        gld = torch.empty(N, H, W, ncls, dtype=torch.float16, device=DEV)
        gld.copy_(gl.permute(0, 2, 3, 1))
        for flags in (0, lib.CONV_FORCE_GENERIC):
            y = torch.empty(N, H, W, ncls, device=DEV, dtype=torch.float32)
            ops.conv_fwd(xd, tx.to(DEV), lambda l: ops.pack_conv_fwd(w.to(DEV), torch.float16, k8=bool(l)), b.to(DEV), y, 3, 3, 1, 1,
                         flags=flags)
            _close(y.permute(0, 3, 1, 2), y_ref, 2e-3)
            dW = torch.empty(ncls, Ci, 3, 3, device=DEV)
            ops.conv_wgrad(xd, tx.to(DEV), gld, None, dW, Ci * 9, 9, 1, 1.0, 3, 3, 1, 1, flags=flags)
            _close(dW, wr.grad, 3e-3)


def test_bilinear_pool_gn_fp16_vector_paths():
    """fp16 (16-B vectorised) variants of bilinear x2 (+ consumer transform) and its adjoint, and GroupNorm with fewer than 8
    channels per group (ResNetV2 root / first block: 64 channels, 32 groups)."""
    lib, ops, T = _gpu()
    g = torch.Generator().manual_seed(31)
    xb = torch.randn(2, 5, 7, 24, generator=g).half()
    t = torch.zeros(24, 4); t[:, 1] = 1.5; t[:, 2] = 0.1
    act = torch.clamp_min(xb.float() * 1.5 + 0.1, 0.0).permute(0, 3, 1, 2).clone().requires_grad_(True)
    ub = F.interpolate(act, scale_factor=2, mode="bilinear", align_corners=True)
    gb = torch.randn(ub.shape, generator=g).half()
    ub.backward(gb.float())
    yb = torch.empty(2, 10, 14, 24, device=DEV, dtype=torch.float16)
    T.bilinear2x(xb.to(DEV), yb, False, t.to(DEV))
    _close(yb.permute(0, 3, 1, 2), ub, 2e-3)
    dxb = torch.empty(2, 5, 7, 24, device=DEV, dtype=torch.float16)
    T.bilinear2x(gb.permute(0, 2, 3, 1).contiguous().to(DEV), dxb, True)
    _close(dxb.permute(0, 3, 1, 2), act.grad, 3e-3)
    # GroupNorm, 2 channels per group
    N, H, W, C, G = 2, 9, 11, 64, 32
    x = (torch.randn(N, H, W, C, generator=g) * 2 + 0.5).half()
    gamma, beta = 0.5 + torch.rand(C, generator=g), 0.1 * torch.randn(C, generator=g)
    dy = torch.randn(N, H, W, C, generator=g).half()
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.relu(F.group_norm(xr, G, gr, br, 1e-6))
    ref.backward(dy.float().permute(0, 3, 1, 2))
    y = torch.empty(N, H, W, C, device=DEV, dtype=torch.float16)
    xd = x.to(DEV)
    mean, rstd = T.gn_fwd(xd, gamma.to(DEV), beta.to(DEV), G, 1e-6, True, None, y)
    _close(y.permute(0, 3, 1, 2), ref, 4e-3)
    dx = torch.empty_like(y)
    dg, db = T.gn_bwd(dy.to(DEV), y, xd, mean, rstd, gamma.to(DEV), G, True, dx, None, 1.0)
    _close(dx.permute(0, 3, 1, 2), xr.grad, 4e-2)
    _close(dg, gr.grad, 2e-2)
    _close(db, br.grad, 2e-2)


@pytest.mark.parametrize("case", [(2, 9, 13, 128, 32, True), (1, 16, 16, 32, 128, False), (2, 7, 5, 96, 40, True),
                                  (1, 20, 20, 64, 64, True), (1, 8, 8, 16, 16, False)])
def test_pointwise_mfma_partial_tiles_and_bn_stats_pass(case):
    """Plain 1x1 conv on the MFMA kernel with channel counts that are only multiples of 8 (partial last K chunk / partial
    output tile: the attention gates' 32-channel branches), against torch and the generic kernel; and umi_bn_stats (the
    separate BatchNorm statistics pass used when the producer has no statistics epilogue) against torch sums."""
    lib, ops, T = _gpu()
    N, H, W, Ci, Co, use_tx = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = torch.randn(N, H, W, Ci, generator=g).half()
    w = torch.randn(Co, Ci, 1, 1, generator=g) * (2.0 / Ci) ** 0.5
    t = None
    if use_tx:
        t = torch.empty(Ci, 4)
        t[:, 0] = 0.1 * torch.randn(Ci, generator=g)
        t[:, 1] = 0.5 + torch.rand(Ci, generator=g)
        t[:, 2] = 0.2 * torch.randn(Ci, generator=g)
        t[:, 3] = 0.0
    a = x.float() if t is None else torch.maximum(x.float() * t[:, 1] + t[:, 2], t[:, 3])
    ref = torch.nn.functional.conv2d(a.permute(0, 3, 1, 2), w.half().float()).permute(0, 2, 3, 1)
    xd, wd = x.to(DEV), w.to(DEV)
    td = t.to(DEV).contiguous() if t is not None else None
    outs = {}
    for name, flags in (("mfma", 0), ("generic", lib.CONV_FORCE_GENERIC)):
        y = torch.full((N, H, W, Co + 8), 7.0, device=DEV, dtype=torch.float16)[..., :Co]      # channel slice: ld > Co
        lay, _ = ops.conv_plan(xd, y, 1, 1, 1, 0, flags)
        assert lay == (1 if name == "mfma" else 0)
        ops.conv_fwd(xd, td, lambda l: ops.pack_conv_fwd(wd, torch.float16, k8=bool(l)), None, y, 1, 1, 1, 0, flags=flags)
        outs[name] = y.float().cpu()
        assert float(y._base[..., Co:].min()) == 7.0 if y._base is not None else True           # nothing written past Co
    scale = ref.abs().max().item()
    assert (outs["mfma"] - ref).abs().max().item() < 4e-3 * scale
    assert (outs["generic"] - ref).abs().max().item() < 4e-3 * scale
    yd = outs["mfma"].half().to(DEV)
    part = ops.bn_stats(yd)
    assert part is not None
    sums = part.view(-1, 2, Co).sum(0).cpu()
    yf = outs["mfma"].half().float().reshape(-1, Co)
    torch.testing.assert_close(sums[0], yf.sum(0), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(sums[1], (yf * yf).sum(0), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("case", [(2, 9, 13, 32, 1), (1, 40, 40, 256, 1), (2, 7, 5, 64, 3), (3, 33, 31, 128, 8)])
def test_narrow_pointwise_fp16_out_with_stat_epilogue(case):
    """1x1 conv with <= 8 output channels and an fp16 output that feeds a BatchNorm (the attention gate's psi branch, reference
    Model.py:283-287): the narrow-output kernel with its statistics epilogue against torch and against the generic kernel."""
    lib, ops, T = _gpu()
    N, H, W, Ci, Co = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, H, W, Ci, generator=g).half()
    w = torch.randn(Co, Ci, 1, 1, generator=g) * (2.0 / Ci) ** 0.5
    t = torch.empty(Ci, 4)
    t[:, 0] = 0.0
    t[:, 1] = 0.5 + torch.rand(Ci, generator=g)
    t[:, 2] = 0.2 * torch.randn(Ci, generator=g)
    t[:, 3] = 0.0
    a = torch.maximum(x.float() * t[:, 1] + t[:, 2], t[:, 3])
    ref = torch.nn.functional.conv2d(a.permute(0, 3, 1, 2), w.half().float()).permute(0, 2, 3, 1)
    xd, wd, td = x.to(DEV), w.to(DEV), t.to(DEV).contiguous()
    res = {}
    for name, flags in (("narrow", 0), ("generic", lib.CONV_FORCE_GENERIC)):
        y = torch.empty(N, H, W, Co, device=DEV, dtype=torch.float16)
        lay, rows = ops.conv_plan(xd, y, 1, 1, 1, 0, flags)
        assert lay == 0
        part = ops.conv_fwd(xd, td, lambda l: ops.pack_conv_fwd(wd, torch.float16, k8=bool(l)), None, y, 1, 1, 1, 0,
                            want_stats=True, flags=flags)
        assert part.numel() == rows * 2 * Co
        res[name] = (y.float().cpu(), part.view(-1, 2, Co).sum(0).cpu())
    assert res["narrow"][1].shape == res["generic"][1].shape
    scale = ref.abs().max().item()
    for name in res:
        y, sums = res[name]
        assert (y - ref).abs().max().item() < 2e-3 * scale, name
        yf = y.reshape(-1, Co)                                  # statistics are those of the ROUNDED stored values
        torch.testing.assert_close(sums[0], yf.sum(0), rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(sums[1], (yf * yf).sum(0), rtol=1e-4, atol=1e-3)
    # without statistics the same kernel runs (fp16 out), and the fp32-out instance still serves the logits
    y0 = torch.empty(N, H, W, Co, device=DEV, dtype=torch.float16)
    ops.conv_fwd(xd, td, lambda l: ops.pack_conv_fwd(wd, torch.float16, k8=bool(l)), None, y0, 1, 1, 1, 0)
    assert torch.equal(y0.float().cpu(), res["narrow"][0])
    y32 = torch.empty(N, H, W, Co, device=DEV, dtype=torch.float32)
    ops.conv_fwd(xd, td, lambda l: ops.pack_conv_fwd(wd, torch.float16, k8=bool(l)), None, y32, 1, 1, 1, 0)
    assert (y32.cpu() - ref).abs().max().item() < 2e-4 * scale


@pytest.mark.parametrize("C,dt", [(32, torch.float16), (256, torch.float16), (24, torch.float16), (12, torch.float16),
                                  (20, torch.float32)])
def test_add2_relu_fwd_bwd_vector_and_scalar_paths(C, dt):
    """E = relu(tx_a(a) + tx_b(b)) of the attention gate (reference Model.py:302) and its backward, against torch: 16-B lanes
    when C/8 is a power of two (32, 256), the scalar kernels otherwise (24, 12, fp32); strided (channel-slice) operands."""
    lib, ops, T = _gpu()
    N, H, W = 2, 7, 9
    g = torch.Generator().manual_seed(C)
    wide = lambda: torch.randn(N, H, W, C + 8, generator=g).to(dt).to(DEV)
    A, B = wide(), wide()
    a, b = A[..., :C], B[..., 8:]
    def mk():
        t = torch.empty(C, 4)
        t[:, 0] = 0.0
        t[:, 1] = 0.5 + torch.rand(C, generator=g)
        t[:, 2] = 0.3 * torch.randn(C, generator=g)
        t[:, 3] = float("-inf")
        return t
    ta, tb = mk(), mk()
    y = torch.empty(N, H, W, C, device=DEV, dtype=dt)
    ops.add2_relu(a, ta.to(DEV), b, tb.to(DEV), y)
    ref = torch.relu(a.float().cpu() * ta[:, 1] + ta[:, 2] + b.float().cpu() * tb[:, 1] + tb[:, 2])
    tol = 2e-3 if dt == torch.float16 else 1e-6
    assert (y.float().cpu() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    y2 = torch.empty_like(y)
    ops.add2_relu(a, None, b, tb.to(DEV), y2)                       # one side consumed as stored
    ref2 = torch.relu(a.float().cpu() + b.float().cpu() * tb[:, 1] + tb[:, 2])
    assert (y2.float().cpu() - ref2).abs().max().item() <= tol * max(1.0, ref2.abs().max().item())
    dy = torch.randn(N, H, W, C, generator=g).to(dt).to(DEV)
    DA = torch.full((N, H, W, C + 8), 3.0, device=DEV, dtype=dt)
    da, db = DA[..., :C], torch.empty(N, H, W, C, device=DEV, dtype=dt)
    ops.add2_relu_bwd(dy, y, da, db)
    want = torch.where(y > 0, dy, torch.zeros_like(dy))
    assert torch.equal(da, want) and torch.equal(db, want)
    assert float(DA[..., C:].min()) == 3.0                            # nothing written past the slice


def test_conv_accumulate_flag_adds_into_the_output():
    """UMI_CONV_ACCUMULATE (second gradient contribution of a tensor with two consumers): the pointwise and the tap-gather
    (ConvTranspose data gradient) MFMA kernels add their result to the stored fp16 tensor, bit-identically to computing into a
    fresh tensor and adding; other problems are refused by the plan, and nothing is written."""
    lib, ops, T = _gpu()
    g = torch.Generator().manual_seed(77)
    # (a) plain 1x1 (the gate's W_x data gradient), output is a channel slice of a wider tensor
    N, H, W, Ci, Co = 2, 9, 13, 32, 64
    x = torch.randn(N, H, W, Ci, generator=g).half().to(DEV)
    w = (torch.randn(Co, Ci, 1, 1, generator=g) * 0.2).to(DEV)
    pk = lambda l: ops.pack_conv_fwd(w, torch.float16, k8=bool(l))
    fresh = torch.empty(N, H, W, Co, device=DEV, dtype=torch.float16)
    ops.conv_fwd(x, None, pk, None, fresh, 1, 1, 1, 0)
    wide = torch.randn(N, H, W, Co + 16, generator=g).half().to(DEV)
    tgt = wide[..., 8:8 + Co]
    want = tgt + fresh
    keep = wide.clone()
    assert ops.conv_accumulate_ok(x, tgt, 1, 1, 1, 0)
    ops.conv_fwd(x, None, pk, None, tgt, 1, 1, 1, 0, flags=lib.CONV_ACCUMULATE)
    assert torch.equal(tgt, want)
    assert torch.equal(wide[..., :8], keep[..., :8]) and torch.equal(wide[..., 8 + Co:], keep[..., 8 + Co:])
    # (b) ConvTranspose2d(2,2) data gradient = stride-2 2x2 tap gather
    Cin, Cout, h, wd = 128, 64, 6, 5
    gy = torch.randn(N, 2 * h, 2 * wd, Cout, generator=g).half().to(DEV)
    wt = (torch.randn(Cin, Cout, 2, 2, generator=g) * 0.1).to(DEV)
    pkt = lambda l: ops.pack_convT_dgrad(wt, torch.float16, k8=bool(l))
    fresh = torch.empty(N, h, wd, Cin, device=DEV, dtype=torch.float16)
    ops.conv_fwd(gy, None, pkt, None, fresh, 2, 2, 2, 0)
    ref = torch.nn.functional.conv2d(gy.float().permute(0, 3, 1, 2), wt.half().float(), stride=2).permute(0, 2, 3, 1)
    assert (fresh.float() - ref).abs().max().item() < 4e-3 * ref.abs().max().item()
    tgt = torch.randn(N, h, wd, Cin, generator=g).half().to(DEV)
    want = tgt + fresh
    assert ops.conv_accumulate_ok(gy, tgt, 2, 2, 2, 0)
    ops.conv_fwd(gy, None, pkt, None, tgt, 2, 2, 2, 0, flags=lib.CONV_ACCUMULATE)
    assert torch.equal(tgt, want)
    # (c) refused elsewhere: 3x3, fp32, forced generic
    x3 = torch.randn(1, 8, 8, 64, generator=g).half().to(DEV)
    y3 = torch.zeros(1, 8, 8, 64, device=DEV, dtype=torch.float16)
    assert not ops.conv_accumulate_ok(x3, y3, 3, 3, 1, 1)
    assert not ops.conv_accumulate_ok(x3, y3, 1, 1, 1, 0, flags=lib.CONV_FORCE_GENERIC)
    assert not ops.conv_accumulate_ok(x3.float(), y3.float(), 1, 1, 1, 0)
    w3 = (torch.randn(64, 64, 3, 3, generator=g) * 0.1).to(DEV)
    with pytest.raises(RuntimeError):
        ops.conv_fwd(x3, None, lambda l: ops.pack_conv_fwd(w3, torch.float16, k8=bool(l)), None, y3, 3, 3, 1, 1,
                     flags=lib.CONV_ACCUMULATE)
    assert float(y3.abs().max()) == 0.0


@pytest.mark.parametrize("shape", [(2 * 9 * 13, 128, 32), (400, 32, 128), (70, 96, 40), (256, 16, 16)])
def test_wgrad1x1_mfma_partial_channel_tiles(shape):
    """Weight gradient of a plain 1x1 conv on the MFMA kernel with channel counts that are only multiples of 8 (attention
    gates: 32 hidden channels): against torch and the generic kernel; the elements next to the tensors must not matter."""
    lib, ops, T = _gpu()
    M, Ci, Co = shape
    g = torch.Generator().manual_seed(M + Ci)
    xb = torch.randn(1, 1, M + 3, Ci, generator=g).half()          # extra rows: what a partial tile reads past the channels
    dyb = (torch.randn(1, 1, M + 3, Co, generator=g) * 0.1).half()
    xb[0, 0, M:] = float("nan")
    dyb[0, 0, M:] = float("nan")
    x, dy = xb[:, :, :M], dyb[:, :, :M]
    ref = dy[0, 0].float().t() @ x[0, 0].float()
    res = {}
    for name, flags in (("mfma", 0), ("generic", lib.CONV_FORCE_GENERIC)):
        gw = torch.full((Co, Ci), 7.0, device=DEV)
        ops.conv_wgrad(xb.to(DEV)[:, :, :M], None, dyb.to(DEV)[:, :, :M], None, gw, Ci, 1, 1, 1.0, 1, 1, 1, 0, flags=flags)
        res[name] = gw.cpu()
    scale = ref.abs().max().item()
    assert torch.isfinite(res["mfma"]).all()
    assert (res["generic"] - ref).abs().max().item() < 2e-3 * scale
    assert (res["mfma"] - ref).abs().max().item() < 4e-3 * scale


@pytest.mark.parametrize("n,M,Ci,Co,strided", [(12, 4704, 768, 768, False), (3, 301, 256, 384, False), (18, 150, 72, 136, False),
                                                (4, 640, 128, 256, True)])
def test_grouped_pointwise_wgrad_matches_per_layer_launches(n, M, Ci, Co, strided):
    """umi_conv_wgrad_group (the weight gradients of n same-shaped linears in one launch, no split-K) against n
    umi_conv_wgrad calls and an fp32 matmul: more than 16 problems (two launches), ragged row / channel tiles, and operands
    that are column slices of wider tensors (the fused Q/K/V gradient)."""
    lib, ops, T = _gpu()
    g = torch.Generator().manual_seed(n * M)
    pad = 64 if strided else 0
    xs = [torch.randn(1, 1, M, Ci + pad, generator=g).half().to(DEV)[..., pad // 2:pad // 2 + Ci] for _ in range(n)]
    dys = [(torch.randn(1, 1, M, Co + pad, generator=g) * 0.1).half().to(DEV)[..., pad // 2:pad // 2 + Co] for _ in range(n)]
    gws = [torch.full((Co, Ci), float("nan"), device=DEV) for _ in range(n)]
    assert ops.conv_wgrad_group(xs, dys, gws, Ci, 1, 0.25)
    for x, dy, gw in zip(xs, dys, gws):
        ref = 0.25 * (dy[0, 0].float().t() @ x[0, 0].float())
        one = torch.empty(Co, Ci, device=DEV)
        ops.conv_wgrad(x, None, dy, None, one, Ci, 1, 1, 0.25, 1, 1, 1, 0)
        scale = ref.abs().max().item()
        assert (gw - ref).abs().max().item() < 2e-3 * scale          # fp16 operands, fp32 accumulation
        assert (gw - one).abs().max().item() < 1e-4 * scale          # same products, another summation order


@pytest.mark.parametrize("n,M,C,strided", [(12, 4704, 3072, False), (18, 333, 72, False), (3, 1000, 768, True)])
def test_grouped_colsum_matches_per_tensor_launches(n, M, C, strided):
    """umi_colsum_group (bias gradients of n same-shaped layers, two launches per 16) against umi_colsum per tensor: the same
    partial-row kernel body, fp64 second stages that differ only in summation order."""
    lib, ops, T = _gpu()
    g = torch.Generator().manual_seed(n + M + C)
    pad = 64 if strided else 0
    xs = [torch.randn(1, 1, M, C + pad, generator=g).half().to(DEV)[..., pad // 2:pad // 2 + C] for _ in range(n)]
    outs = [torch.full((C,), float("nan"), device=DEV) for _ in range(n)]
    assert ops.colsum_group(xs, outs, 0.5)
    for x, o in zip(xs, outs):
        one = torch.empty(C, device=DEV)
        ops.colsum(x, one, 0.5)
        _close(o, one, 1e-6)
        _close(o, x[0, 0].float().sum(0) * 0.5, 2e-3)


@pytest.mark.parametrize("shape", [(2, 15, 17, 64), (1, 112, 112, 64), (3, 8, 9, 24)])
def test_pool3s2_fp16_with_recorded_taps(shape):
    """MaxPool2d(3, 2) fp16 vector kernels: forward against torch (ties included: values drawn from 8 levels), backward with the
    recorded winning taps against the backward that re-derives them from the input -- identical, and both equal to torch's
    max_pool2d gradient (first maximum in scan order)."""
    lib, ops, T = _gpu()
    N, H, W, C = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randint(0, 8, (N, H, W, C), generator=g).float() * 0.25 - 1.0).half()
    Ho, Wo = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    dy = torch.randn(N, Ho, Wo, C, generator=g).half()
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.max_pool2d(xr, 3, 2, 0)
    ref.backward(dy.float().permute(0, 3, 1, 2))
    xd, dyd = x.to(DEV), dy.to(DEV)
    y = torch.empty(N, Ho, Wo, C, device=DEV, dtype=torch.float16)
    idx = torch.full((N, Ho, Wo, C), 255, dtype=torch.uint8, device=DEV)
    T.pool3s2_fwd(xd, y, idx)
    assert torch.equal(y.cpu().float().permute(0, 3, 1, 2), ref.detach())
    assert int(idx.max()) <= 8
    dx_a = torch.empty(N, H, W, C, device=DEV, dtype=torch.float16)
    dx_b = torch.empty_like(dx_a)
    T.pool3s2_bwd(dyd, xd, dx_a, idx)
    T.pool3s2_bwd(dyd, xd, dx_b, None)
    assert torch.equal(dx_a, dx_b)
    _close(dx_a.permute(0, 3, 1, 2), xr.grad, 2e-3)


def test_group_norm_per_channel_groups_and_deferred_param_grads():
    """GroupNorm(C, C) (the projection shortcuts, resnet_skip.py:62-63: one channel per group, instance-norm-like) on channels
    whose mean is 50x their spread -- the case the shifted statistics exist for -- and dgamma / dbeta formed later from the
    per-sample rows (umi_gn_param_grads_group) against the immediate ones."""
    lib, ops, T = _gpu()
    N, H, W, C, G = 3, 14, 14, 512, 512
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(N, H, W, C, generator=g) * 0.1 + 5.0 * torch.randn(1, 1, 1, C, generator=g)).half()
    gamma, beta = 0.5 + torch.rand(C, generator=g), 0.1 * torch.randn(C, generator=g)
    dy = torch.randn(N, H, W, C, generator=g).half()
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.group_norm(xr, G, gr, br, 1e-5)
    ref.backward(dy.float().permute(0, 3, 1, 2))
    xd, y = x.to(DEV), torch.empty(N, H, W, C, device=DEV, dtype=torch.float16)
    mean, rstd = T.gn_fwd(xd, gamma.to(DEV), beta.to(DEV), G, 1e-5, False, None, y)
    want_rstd = 1.0 / torch.sqrt(x.float().var(dim=(1, 2), unbiased=False) + 1e-5)
    assert ((rstd.cpu().view(N, C) - want_rstd).abs() / want_rstd).max().item() < 1e-4
    _close(y.permute(0, 3, 1, 2), ref, 4e-3)
    dx = torch.empty_like(y)
    dg, db = T.gn_bwd(dy.to(DEV), y, xd, mean, rstd, gamma.to(DEV), G, False, dx, None, 0.5)
    dx2 = torch.empty_like(y)
    part = T.gn_bwd(dy.to(DEV), y, xd, mean, rstd, gamma.to(DEV), G, False, dx2, None, 0.5, keep_part=True)
    dg2, db2 = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    T.gn_param_grads_group([part], N, [dg2], [db2], 0.5)
    assert torch.equal(dx, dx2)
    _close(dg2, dg, 1e-6)
    _close(db2, db, 1e-6)
    _close(db, br.grad * 0.5, 2e-3)
    _close(dg, gr.grad * 0.5, 2e-2)


@pytest.mark.parametrize("gelu,with_add", [(True, False), (False, True), (True, True)])
def test_fused_dropout_matches_the_separate_kernels(gelu, with_add):
    """umi_dropout_fused (GELU -> dropout -> + residual in one kernel, and its backward) against GELU, dropout and add run one
    after the other with the same seed: the same mask bytes, values equal up to the roundings the fusion removes."""
    lib, ops, T = _gpu()
    M, C, p, seed = 300, 768, 0.1, 12345
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 1, M, C, generator=g).half().to(DEV)
    h = torch.randn(1, 1, M, C, generator=g).half().to(DEV)
    dy = torch.randn(1, 1, M, C, generator=g).half().to(DEV)
    y, mask = torch.empty_like(x), torch.empty(M * C, dtype=torch.uint8, device=DEV)
    assert T.dropout_fused(x, y, mask, False, p, seed, None, h if with_add else None, gelu)
    a = x
    if gelu:
        a = torch.empty_like(x)
        T.gelu_fwd(x, a)
    y_ref, mask_ref = torch.empty_like(x), torch.empty_like(mask)
    T.dropout(a, y_ref, mask_ref, False, p, seed)
    if with_add:
        T.add(y_ref, h, y_ref)
    assert torch.equal(mask, mask_ref)
    assert 0.85 < mask.float().mean().item() < 0.95
    _close(y, y_ref, 2e-3)
    dx, dx_ref = torch.empty_like(x), torch.empty_like(x)
    if gelu:
        assert T.dropout_fused(dy, dx, mask, True, p, 0, None, x, True)
        tmp = torch.empty_like(x)
        T.dropout(dy, tmp, mask, True, p, 0)
        T.gelu_bwd(x, tmp, dx_ref)
        _close(dx, dx_ref, 2e-3)


def test_pack_cache_wstd_and_side_by_side_operands_match_the_single_launch_paths():
    """PackCache.refresh brings the standardised StdConv2d weights (one umi_wstd_fwd_multi launch), their kernel layouts and the
    side-by-side Q/K/V operands (umi_pack_desc.ldn) up to date after a parameter update: identical to umi_wstd_fwd per conv, to
    packing a torch.cat of the three weights, and the grouped standardisation backward to umi_wstd_bwd per conv."""
    lib, ops, T = _gpu()
    g = torch.Generator().manual_seed(11)
    convs = [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in [(64, 3, 7, 7), (128, 64, 1, 1), (64, 64, 3, 3),
                                                                                (256, 128, 1, 1)]]
    qkv = [torch.nn.Parameter(torch.randn(96, 72, generator=g).to(DEV)) for _ in range(3)]
    bias = [torch.nn.Parameter(torch.randn(96, generator=g).to(DEV)) for _ in range(3)]
    c = ops.PackCache()

    def check():
        ents = [c.wstd(w, 1e-5) for w in convs]
        for w, e in zip(convs, ents):
            ws, rstd = T.wstd_fwd(w, 1e-5)
            assert torch.equal(e.ws, ws) and torch.equal(e.rstd, rstd)
            k8 = w.shape[1] % 8 == 0                    # (the 3-channel root conv runs on its own kernel, plain layout)
            for kind in ("conv_fwd", "conv_dgrad", "conv_dgrad_strided"):
                assert torch.equal(c.get(kind, e.ws, torch.float16, k8), ops.PACKERS[kind](ws, torch.float16, k8=k8))
        wcat = torch.cat([w.detach() for w in qkv], 0).reshape(288, 72, 1, 1)
        for kind in ("conv_fwd", "conv_dgrad"):
            for k8 in (True, False):
                assert torch.equal(c.get_cat(kind, qkv, torch.float16, k8), ops.PACKERS[kind](wcat, torch.float16, k8=k8)), (kind, k8)
        assert torch.equal(c.get_cat("bias", bias, torch.float32, False), torch.cat([b.detach() for b in bias]))
        return ents

    check()                                             # first use: per-entry launches
    with torch.no_grad():                               # an optimizer step: everything is stale
        for p in convs + qkv + bias:
            p.mul_(0.9).add_(0.01)
    c.refresh()                                         # one wstd launch + one pack launch per storage dtype
    stale = [e for e in c.ents.values() if e.ver != c._ver(e.w())]
    assert not stale
    ents = check()
    gflat = torch.randn(c.wstd_total, generator=g).to(DEV)
    dflat = torch.empty_like(gflat)
    c.wstd_bwd(ents, gflat, dflat)
    for w, e in zip(convs, ents):
        gw = gflat[e.off:e.off + w.numel()].view(w.shape)
        assert torch.equal(dflat[e.off:e.off + w.numel()].view(w.shape), T.wstd_bwd(e.ws, e.rstd, gw.contiguous()))


@pytest.mark.parametrize("M,Ci,Co", [(4704, 768, 3072), (300, 256, 128), (77, 64, 192)])
def test_linear_fused_epilogues_match_the_two_kernel_sequence(M, Ci, Co):
    """umi_linear_fused (GEMM epilogue = bias + GELU + dropout -> second output + mask bytes; or bias + dropout + residual add)
    against umi_conv_fwd followed by umi_dropout_fused with the same seed: identical outputs and masks, bit for bit
    (reference vit_seg_modeling.py:113-119 Mlp.forward, :177-187 Block.forward)."""
    if not torch.cuda.is_available():
        pytest.fail("needs an MI355X")
    from umi import ops, ops_tu
    g = torch.Generator(device=DEV).manual_seed(M + Ci)
    x = torch.randn(1, 1, M, Ci, device=DEV, generator=g).half()
    w = torch.randn(Co, Ci, 1, 1, device=DEV, generator=g) * Ci ** -0.5
    b = torch.randn(Co, device=DEV, generator=g) * 0.1
    res = torch.randn(1, 1, M, Co, device=DEV, generator=g).half()
    seed_dev = torch.tensor([5], dtype=torch.int32, device=DEV)
    p, seed = 0.1, 1234567
    y = torch.empty(1, 1, M, Co, device=DEV, dtype=torch.float16)
    lay, _ = ops.conv_plan(x, y, 1, 1, 1, 0, has_bias=True)
    assert lay == 1
    wp = ops.pack_conv_fwd(w, torch.float16, k8=True)
    ops.conv_fwd(x, None, lambda l: wp, b, y, 1, 1, 1, 0)
    for epi in (1, 2):
        ref_out, ref_mask = torch.empty_like(y), torch.empty(M * Co, dtype=torch.uint8, device=DEV)
        assert ops_tu.dropout_fused(y, ref_out, ref_mask, False, p, seed, seed_dev, None if epi == 1 else res, epi == 1)
        yf = torch.full_like(y, 7.0)
        y2 = torch.full_like(y, 7.0) if epi == 1 else None
        mask = torch.zeros(M * Co, dtype=torch.uint8, device=DEV)
        assert ops_tu.linear_fused(x, wp, b, yf, epi, p, seed, seed_dev, mask, aux=res if epi == 2 else None, y2=y2)
        torch.cuda.synchronize()
        assert torch.equal(mask, ref_mask)
        assert 0.05 < 1.0 - mask.float().mean().item() < 0.15          # about p of the elements dropped
        if epi == 1:
            assert torch.equal(yf, y)                                   # the pre-activation, kept for the GELU backward
            assert torch.equal(y2, ref_out)
        else:
            assert torch.equal(yf, ref_out)
