"""Kernel-level GPU parity: every libunetmi op against plain PyTorch fp32 (CPU) on odd shapes, and the
MFMA fast paths against the generic kernels (UMI_CONV_FORCE_GENERIC) on the same inputs."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _gpu():
    if not torch.cuda.is_available():
        pytest.fail("needs an MI355X")
    from umi import lib, ops
    return lib, ops


def _tx(C, gen, relu=True):
    t = torch.empty(C, 4)
    t[:, 0] = 0.3 * torch.randn(C, generator=gen)
    t[:, 1] = (0.5 + torch.rand(C, generator=gen)) * torch.where(torch.rand(C, generator=gen) < 0.15, -1.0, 1.0)
    t[:, 2] = 0.2 * torch.randn(C, generator=gen)
    t[:, 3] = 0.0 if relu else float("-inf")
    return t


def _apply_tx(x_nhwc, t):
    return torch.maximum(x_nhwc * t[:, 1] + t[:, 2], t[:, 3])      # t[:, 0] (mean) is only used by BN backward


def _ref_conv(x_nhwc, t, w, stride=1, pad=1):
    a = _apply_tx(x_nhwc.float(), t) if t is not None else x_nhwc.float()
    return F.conv2d(a.permute(0, 3, 1, 2), w, None, stride, pad).permute(0, 2, 3, 1).contiguous()


CASES = [  # N, H, W, Ci, Co, ldx_extra, ldy_extra, use_tx
    (2, 20, 45, 32, 64, 0, 0, True),       # odd spatial size, BN=64 tile config
    (1, 16, 64, 64, 128, 0, 0, True),      # BN=128 tile config
    (2, 9, 33, 16, 192, 16, 64, True),     # channel-slice views (ld > C), 3 co-blocks of 64
    (1, 40, 32, 128, 256, 0, 128, False),  # no transform (dgrad-style), output into a concat half
    (3, 7, 5, 48, 64, 16, 0, True),        # image smaller than one tile
    (2, 20, 45, 64, 16, 0, 0, True),       # Co = 16: partial output-channel tile (TransUNet decoder tail)
    (2, 19, 33, 16, 16, 0, 16, True),      # Ci = Co = 16 into a channel slice
    (1, 16, 40, 32, 200, 0, 0, False),     # 128 + 72 output channels
    (2, 20, 45, 96, 128, 0, 0, True),      # 16x16x32 form (Co % 128 == 0, Ci % 32 == 0): 3 super-chunks, ragged tiles
    (1, 9, 33, 64, 256, 32, 64, False),    # same form, no transform, channel-slice views on both sides
    (1, 8, 32, 32, 128, 0, 0, True),       # same form, ONE super-chunk (the prologue's two weight stages + one more)
]


@pytest.mark.parametrize("case", CASES)
def test_conv3x3_mfma_vs_reference_and_generic(case):
    lib, ops = _gpu()
    N, H, W, Ci, Co, ex, ey, use_tx = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    xb = torch.randn(N, H, W, Ci + ex, generator=g).half()
    x = xb[..., ex // 2: ex // 2 + Ci] if ex else xb
    w = torch.randn(Co, Ci, 3, 3, generator=g) * (2.0 / (9 * Ci)) ** 0.5
    t = _tx(Ci, g) if use_tx else None
    ref = _ref_conv(x, t, w.half().float())
    xd_b = xb.to(DEV)
    xd = xd_b[..., ex // 2: ex // 2 + Ci] if ex else xd_b
    td = t.to(DEV).contiguous() if use_tx else None
    wd = w.to(DEV)
    outs = {}
    for name, flags in (("mfma", 0), ("generic", lib.CONV_FORCE_GENERIC)):
        yb = torch.full((N, H, W, Co + ey), 7.0, device=DEV, dtype=torch.float16)
        y = yb[..., ey // 2: ey // 2 + Co] if ey else yb
        lay, rows = ops.conv_plan(xd, y, 3, 3, 1, 1, flags)
        assert lay == (1 if name == "mfma" else 0)
        part = ops.conv_fwd(xd, td, lambda l: ops.pack_conv_fwd(wd, torch.float16, k8=bool(l)), None, y, 3, 3, 1, 1,
                            want_stats=True, flags=flags)
        torch.cuda.synchronize()
        if ey:      # bytes outside the channel slice are untouched
            assert (yb[..., : ey // 2] == 7.0).all() and (yb[..., ey // 2 + Co:] == 7.0).all()
        outs[name] = (y.float().cpu(), part.view(-1, 2, Co).sum(0).cpu())
        assert part.numel() == rows * 2 * Co
    ym, sm = outs["mfma"]
    yg, sg = outs["generic"]
    scale = ref.abs().max().item()
    # the MFMA path rounds the activated input to fp16 before the matrix core; the generic one keeps fp32
    assert (ym - ref).abs().max().item() < 4e-3 * scale
    assert (yg - ref).abs().max().item() < 2e-3 * scale
    assert (ym - yg).abs().max().item() < 4e-3 * scale
    # BatchNorm partial sums are taken over the stored (rounded) outputs
    for (y_, s_) in ((ym, sm), (yg, sg)):
        torch.testing.assert_close(s_[0], y_.sum((0, 1, 2)), rtol=1e-3, atol=1e-2 * scale)
        torch.testing.assert_close(s_[1], (y_ * y_).sum((0, 1, 2)), rtol=1e-3, atol=1e-2 * scale * scale)


def test_conv3x3_mfma_dgrad_matches_autograd():
    """dgrad = the same kernel on rotated/transposed weights: compare with torch autograd of conv2d."""
    lib, ops = _gpu()
    N, H, W, Ci, Co = 2, 24, 40, 64, 128
    g = torch.Generator().manual_seed(5)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.05).half().float()
    dy = torch.randn(N, H, W, Co, generator=g).half()
    xr = torch.zeros(N, Ci, H, W, requires_grad=True)
    F.conv2d(xr, w, None, 1, 1).backward(dy.float().permute(0, 3, 1, 2))
    ref = xr.grad.permute(0, 2, 3, 1)
    dx = torch.empty(N, H, W, Ci, device=DEV, dtype=torch.float16)
    wd = w.to(DEV)
    ops.conv_fwd(dy.to(DEV), None, lambda l: ops.pack_conv_dgrad(wd, torch.float16, k8=bool(l)), None, dx, 3, 3, 1, 1)
    assert ops.conv_plan(dy.to(DEV), dx, 3, 3, 1, 1)[0] == 1
    assert (dx.float().cpu() - ref).abs().max().item() < 3e-3 * ref.abs().max().item()


@pytest.mark.parametrize("shape", [(2, 20, 45, 32, 64), (1, 16, 64, 64, 128), (2, 9, 33, 16, 192), (1, 12, 12, 128, 64),
                                   (2, 20, 45, 64, 16), (2, 19, 33, 16, 16), (1, 10, 40, 72, 24)])
def test_wgrad_fast_path_vs_reference_and_generic(shape):
    lib, ops = _gpu()
    N, H, W, Ci, Co = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(N, H, W, Ci, generator=g).half()
    dy = (torch.randn(N, H, W, Co, generator=g) * 0.1).half()
    t = _tx(Ci, g)
    a = _apply_tx(x.float(), t).permute(0, 3, 1, 2).requires_grad_(False)
    wr = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    F.conv2d(a, wr, None, 1, 1).backward(dy.float().permute(0, 3, 1, 2))
    ref = wr.grad * 0.5
    res = {}
    for name, flags in (("fast", 0), ("generic", lib.CONV_FORCE_GENERIC)):
        gw = torch.empty(Co, Ci, 3, 3, device=DEV)
        ops.conv_wgrad(x.to(DEV), t.to(DEV), dy.to(DEV), None, gw, Ci * 9, 9, 1, 0.5, 3, 3, 1, 1, flags=flags)
        res[name] = gw.cpu()
    scale = ref.abs().max().item()
    assert (res["generic"] - ref).abs().max().item() < 2e-3 * scale
    assert (res["fast"] - ref).abs().max().item() < 6e-3 * scale


@pytest.mark.parametrize("shape", [(2, 5, 6, 128, 64), (1, 16, 16, 64, 128), (2, 9, 7, 256, 128),
                                   (1, 80, 81, 128, 256)])       # last: a 6,480-pixel map, 1,024 output columns
def test_conv_transpose_mfma_fwd_and_dgrad(shape):
    """ConvTranspose2d(k2,s2) forward (scatter epilogue) and its data gradient (space-to-depth gather) on the
    pointwise MFMA kernel vs torch and vs the generic kernels; destination is a padded channel slice."""
    lib, ops = _gpu()
    N, h, w, Cin, Cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(N, h, w, Cin, generator=g).half()
    wt = (torch.randn(Cin, Cout, 2, 2, generator=g) * (1.0 / Cin) ** 0.5).half().float()
    b = 0.1 * torch.randn(Cout, generator=g)
    t = _tx(Cin, g)
    a = _apply_tx(x.float(), t).permute(0, 3, 1, 2)
    ref = F.conv_transpose2d(a, wt, b, stride=2).permute(0, 2, 3, 1)
    Hd, Wd, oy, ox = 2 * h + 1, 2 * w + 2, 0, 1
    wd = wt.to(DEV)
    res = {}
    for name, flags in (("mfma", lib.CONV_UPSAMPLE2), ("generic", lib.CONV_UPSAMPLE2 | lib.CONV_FORCE_GENERIC)):
        buf = torch.zeros(N, Hd, Wd, 2 * Cout, device=DEV, dtype=torch.float16)
        dest = buf[..., Cout:]
        ops.conv_fwd(x.to(DEV), t.to(DEV), lambda l: ops.pack_convT_fwd(wd, torch.float16, k8=bool(l)), b.to(DEV), dest,
                     2, 2, 2, 0, flags=flags, up_offset=(oy, ox))
        assert ops.conv_plan(x.to(DEV), dest, 2, 2, 2, 0, flags)[0] == (1 if name == "mfma" else 0)
        got = buf.float().cpu()
        assert (got[..., :Cout] == 0).all()
        inner = got[:, oy:oy + 2 * h, ox:ox + 2 * w, Cout:]
        assert (inner - ref).abs().max().item() < 4e-3 * ref.abs().max().item(), name
        mask = torch.ones(Hd, Wd, dtype=torch.bool)
        mask[oy:oy + 2 * h, ox:ox + 2 * w] = False
        assert (got[:, mask][..., Cout:] == 0).all()
        res[name] = inner
    # data gradient: stride-2 2x2 conv over d(up)
    dup = (torch.randn(N, 2 * h, 2 * w, Cout, generator=g) * 0.1).half()
    xr = torch.zeros(N, Cin, h, w, requires_grad=True)
    F.conv_transpose2d(xr, wt, None, stride=2).backward(dup.float().permute(0, 3, 1, 2))
    refd = xr.grad.permute(0, 2, 3, 1)
    for name, flags in (("mfma", 0), ("generic", lib.CONV_FORCE_GENERIC)):
        dx = torch.empty(N, h, w, Cin, device=DEV, dtype=torch.float16)
        ops.conv_fwd(dup.to(DEV), None, lambda l: ops.pack_convT_dgrad(wd, torch.float16, k8=bool(l)), None, dx, 2, 2, 2, 0,
                     flags=flags)
        assert ops.conv_plan(dup.to(DEV), dx, 2, 2, 2, 0, flags)[0] == (1 if name == "mfma" else 0)
        assert (dx.float().cpu() - refd).abs().max().item() < 4e-3 * refd.abs().max().item(), name


def test_conv1x1_mfma_plain():
    lib, ops = _gpu()
    N, H, W, Ci, Co = 2, 13, 11, 192, 64
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, H, W, Ci, generator=g).half()
    wt = (torch.randn(Co, Ci, 1, 1, generator=g) * (1.0 / Ci) ** 0.5).half().float()
    b = 0.1 * torch.randn(Co, generator=g)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wt, b).permute(0, 2, 3, 1)
    y = torch.empty(N, H, W, Co, device=DEV, dtype=torch.float16)
    wd = wt.to(DEV)
    ops.conv_fwd(x.to(DEV), None, lambda l: ops.pack_conv_fwd(wd, torch.float16, k8=bool(l)), b.to(DEV), y, 1, 1, 1, 0)
    assert ops.conv_plan(x.to(DEV), y, 1, 1, 1, 0, 0, True)[0] == 1
    assert (y.float().cpu() - ref).abs().max().item() < 3e-3 * ref.abs().max().item()


@pytest.mark.parametrize("case", [(1, 25, 40, 128, 64, True, True),      # M = 1000 (ragged last tile), 64-column tiles
                                  (2, 65, 100, 256, 512, True, False),   # M = 13000: 128-column tiles, transform, no bias
                                  (1, 48, 98, 768, 768, False, True),    # a ViT linear's shape (4704 tokens)
                                  (1, 7, 9, 384, 192, False, False)])    # one partial tile
def test_conv1x1_mfma_shapes_and_tile_override(case):
    """The pointwise MFMA kernel on GEMM-like shapes (ragged pixel count, transform, bias, channel-slice views on both sides) vs torch,
    and its default tile pick vs a forced tile (two summation-order-identical runs of the same kernel family)."""
    import os
    lib, ops = _gpu()
    N, H, W, Ci, Co, use_tx, use_b = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    xb = torch.randn(N, H, W, Ci + 16, generator=g).half()
    x = xb[..., 8:8 + Ci]
    wt = (torch.randn(Co, Ci, 1, 1, generator=g) * (1.0 / Ci) ** 0.5).half().float()
    b = 0.1 * torch.randn(Co, generator=g) if use_b else None
    t = _tx(Ci, g) if use_tx else None
    a = _apply_tx(x.float(), t).half().float() if use_tx else x.float()
    ref = F.conv2d(a.permute(0, 3, 1, 2), wt, b).permute(0, 2, 3, 1)
    xd = xb.to(DEV)[..., 8:8 + Ci]
    wd = wt.to(DEV)
    outs = []
    for tile in (None, "256x64"):
        if tile:
            os.environ["UMI_C1_TILE"] = tile
        try:
            yb = torch.full((N, H, W, Co + 32), 3.0, device=DEV, dtype=torch.float16)
            y = yb[..., 16:16 + Co]
            ops.conv_fwd(xd, t.to(DEV) if use_tx else None, lambda l: ops.pack_conv_fwd(wd, torch.float16, k8=bool(l)),
                         b.to(DEV) if use_b else None, y, 1, 1, 1, 0)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("UMI_C1_TILE", None)
        assert (yb[..., :16] == 3.0).all() and (yb[..., 16 + Co:] == 3.0).all()
        outs.append(y.float().cpu())
    scale = ref.abs().max().item()
    assert (outs[0] - ref).abs().max().item() < 3e-3 * scale
    assert (outs[0] - outs[1]).abs().max().item() < 2e-3 * scale      # two summation orders of the same fp16 operands


@pytest.mark.parametrize("cin", [1, 3])
def test_stem_and_head_fast_paths(cin):
    """First conv (Ci <= 4) forward + stats + wgrad and the OutConv head (fwd fp32, dgrad, wgrad) vs the generic kernels."""
    lib, ops = _gpu()
    N, H, W, C, ncls = 2, 37, 29, 64, 2 if cin == 1 else 4
    g = torch.Generator().manual_seed(cin)
    x = torch.randn(N, H, W, cin, generator=g).half().to(DEV)
    w = (torch.randn(C, cin, 3, 3, generator=g) * 0.3).to(DEV)
    outs = {}
    for name, flags in (("fast", 0), ("generic", lib.CONV_FORCE_GENERIC)):
        y = torch.empty(N, H, W, C, device=DEV, dtype=torch.float16)
        part = ops.conv_fwd(x, None, lambda l: ops.pack_conv_fwd(w, torch.float16, k8=bool(l)), None, y, 3, 3, 1, 1,
                            want_stats=True, flags=flags)
        gw = torch.empty(C, cin, 3, 3, device=DEV)
        dy = (torch.randn(N, H, W, C, generator=g) * 0.1).half().to(DEV)
        ops.conv_wgrad(x, None, dy, None, gw, cin * 9, 9, 1, 0.25, 3, 3, 1, 1, flags=flags)
        outs[name] = (y.float().cpu(), part.view(-1, 2, C).sum(0).cpu(), gw.cpu())
        g = torch.Generator().manual_seed(cin)          # same dy for both arms
        _ = torch.randn(N, H, W, cin, generator=g); _ = torch.randn(C, cin, 3, 3, generator=g)
    (yf, sf, gf), (yg, sg, gg) = outs["fast"], outs["generic"]
    assert (yf - yg).abs().max().item() <= 2e-3 * yg.abs().max().item()
    torch.testing.assert_close(sf, sg, rtol=2e-3, atol=1e-2)
    assert (gf - gg).abs().max().item() < 2e-3 * gg.abs().max().item()
    ref = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.half().float().cpu(), None, 1, 1).permute(0, 2, 3, 1)
    assert (yf - ref).abs().max().item() < 2e-3 * ref.abs().max().item()

    # head
    a = torch.randn(N, H, W, C, generator=g).half().to(DEV)
    t = _tx(C, g).to(DEV)
    wo = (torch.randn(ncls, C, 1, 1, generator=g) * 0.2).to(DEV)
    b = (0.1 * torch.randn(ncls, generator=g)).to(DEV)
    dl = (torch.randn(N, H, W, ncls, generator=g) * 0.05).half().to(DEV)
    res = {}
    for name, flags in (("fast", 0), ("generic", lib.CONV_FORCE_GENERIC)):
        logits = torch.empty(N, H, W, ncls, device=DEV, dtype=torch.float32)
        ops.conv_fwd(a, t, lambda l: ops.pack_conv_fwd(wo, torch.float16, k8=bool(l)), b, logits, 1, 1, 1, 0, flags=flags)
        da = torch.empty(N, H, W, C, device=DEV, dtype=torch.float16)
        ops.conv_fwd(dl, None, lambda l: ops.pack_conv_dgrad(wo, torch.float16, k8=bool(l)), None, da, 1, 1, 1, 0, flags=flags)
        gwo = torch.empty(ncls, C, 1, 1, device=DEV)
        ops.conv_wgrad(a, t, dl, None, gwo, C, 1, 1, 2.0, 1, 1, 1, 0, flags=flags)
        res[name] = (logits.cpu(), da.float().cpu(), gwo.cpu())
    for f, gnr in zip(res["fast"], res["generic"]):
        assert (f - gnr).abs().max().item() < 2e-3 * gnr.abs().max().item()
    act = _apply_tx(a.float().cpu(), t.cpu())
    ref = F.conv2d(act.permute(0, 3, 1, 2), wo.half().float().cpu(), b.cpu()).permute(0, 2, 3, 1)
    assert (res["fast"][0] - ref).abs().max().item() < 1e-3 * ref.abs().max().item()


@pytest.mark.parametrize("shape", [(2, 5, 6, 128, 64), (1, 16, 16, 64, 128), (2, 9, 37, 256, 128)])
def test_conv_transpose_wgrad_mfma(shape):
    """dW of ConvTranspose2d(k2,s2): hi-res gradient x low-res (BN+ReLU-on-load) input, MFMA vs generic vs autograd."""
    lib, ops = _gpu()
    N, h, w, Cin, Cout = shape
    g = torch.Generator().manual_seed(sum(shape) + 1)
    xin = torch.randn(N, h, w, Cin, generator=g).half()
    t = _tx(Cin, g)
    dup = (torch.randn(N, 2 * h, 2 * w, Cout, generator=g) * 0.1).half()
    a = _apply_tx(xin.float(), t).permute(0, 3, 1, 2)
    wr = torch.zeros(Cin, Cout, 2, 2, requires_grad=True)
    F.conv_transpose2d(a, wr, None, stride=2).backward(dup.float().permute(0, 3, 1, 2))
    ref = wr.grad
    res = {}
    for name, flags in (("mfma", 0), ("generic", lib.CONV_FORCE_GENERIC)):
        gw = torch.empty(Cin, Cout, 2, 2, device=DEV)
        ops.conv_wgrad(dup.to(DEV), None, xin.to(DEV), t.to(DEV), gw, Cout * 4, 4, 1, 1.0, 2, 2, 2, 0, flags=flags)
        res[name] = gw.cpu()
    scale = ref.abs().max().item()
    assert (res["generic"] - ref).abs().max().item() < 2e-3 * scale
    assert (res["mfma"] - ref).abs().max().item() < 6e-3 * scale


@pytest.mark.parametrize("C,tdt", [(2, torch.int64), (4, torch.float32), (3, torch.uint8), (8, torch.int32)])
def test_fused_dice_ce_loss_matches_composite(C, tdt):
    """loss.calc_loss('dice_bce_mc') on device logits = two fused kernels per direction; must equal the composite
    torch formulation of the reference (loss.py:488-500) in value and gradient."""
    lib, ops = _gpu()
    import loss as LS
    g = torch.Generator().manual_seed(C)
    N, H, W = 3, 37, 29
    logits = (torch.randn(N, C, H, W, generator=g) * 3).to(DEV).requires_grad_(True)
    target = torch.randint(0, C, (N, H, W), generator=g).to(tdt).to(DEV)
    LS.CLASS_NUMBER = C
    assert LS._fused_ok(logits, target)
    fused = LS.calc_loss(logits, target, loss_type="dice_bce_mc")
    (fused * 1.7).backward()
    gf = logits.grad.clone()
    ref_in = logits.detach().double().cpu().requires_grad_(True)
    tl = target.cpu().long()
    ce = F.cross_entropy(ref_in, tl)
    p = torch.softmax(ref_in, 1)
    dice = 0.0
    for c in range(C):
        t = (tl == c).double()
        dice = dice + (1 - (2 * (p[:, c] * t).sum() + 1e-5) / ((p[:, c] ** 2).sum() + (t * t).sum() + 1e-5))
    ref = 0.5 * ce + 0.5 * dice / C
    (ref * 1.7).backward()
    assert abs(fused.item() - ref.item()) < 2e-6 * max(1.0, abs(ref.item()))
    assert (gf.cpu().double() - ref_in.grad).abs().max().item() < 2e-5 * ref_in.grad.abs().max().item()


@pytest.mark.parametrize("M,C,pad", [(4 * 37 * 29, 2, 0), (2 * 16 * 16, 4, 0), (1001, 2, 0), (4 * 37 * 29, 2, 6), (5000, 64, 0), (777, 768, 0),
                                     (300, 3072, 0), (100, 72, 8)])
def test_colsum_paths(M, C, pad):
    """Bias-gradient column sums: vectorised (C % 8 == 0), folded small-C (dense C in {1,2,4,8}) and scalar fallbacks."""
    lib, ops = _gpu()
    g = torch.Generator().manual_seed(M + C)
    buf = torch.randn(1, 1, M, C + pad, generator=g).half()
    x = buf[..., :C]
    ref = x.double().sum((0, 1, 2)) * 0.25
    out = torch.empty(C, device=DEV)
    xd = buf.to(DEV)[..., :C]
    ops.colsum(xd, out, 0.25)
    assert (out.cpu().double() - ref).abs().max().item() < 1e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("rows,C", [(8192, 64), (4096, 128), (1000, 40), (513, 200), (300, 64), (64, 1024), (20000, 64)])
def test_bn_finalize_row_reduction_forms(rows, C):
    """BatchNorm statistics finalize over partial rows: the one-workgroup-per-channel form (< 512 rows), the 2-D form whose last
    workgroup per channel block finishes (ragged last slice / last channel block) and the size fallback, against a float64
    reduction; eleven launches (more than the scratch regions) are bit-identical, i.e. the counters reset themselves and the
    summation order does not depend on which workgroup finishes last."""
    lib, ops = _gpu()
    g = torch.Generator().manual_seed(rows + C)
    part = torch.randn(rows, 2, C, generator=g)
    part[:, 1] = part[:, 1].abs() * 3 + part[:, 0] ** 2
    count = float(rows * 7)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    mean = part[:, 0].double().sum(0) / count
    var = (part[:, 1].double().sum(0) / count - mean * mean).clamp_min(0)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    pd, gd, bd = part.to(DEV).contiguous().view(-1), gamma.to(DEV), beta.to(DEV)
    outs = []
    for _ in range(11):
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        tx, rs = ops.bn_finalize(pd, C, count, gd, bd, 1e-5, 0.1, rm, rv)
        outs.append((tx.cpu(), rs.cpu(), rm.cpu(), rv.cpu()))
    tx, rs, rm, rv = outs[0]
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(outs[0], o))
    scale = gamma.double() * rstd
    assert (tx[:, 0].double() - mean).abs().max().item() < 1e-6 * max(1.0, mean.abs().max().item())
    assert (tx[:, 1].double() - scale).abs().max().item() < 2e-6 * scale.abs().max().item()
    assert (tx[:, 2].double() - (beta.double() - mean * scale)).abs().max().item() < 1e-5
    assert (rs.double() - rstd).abs().max().item() < 2e-6 * rstd.abs().max().item()
    assert (rm.double() - 0.1 * mean).abs().max().item() < 1e-6
    assert (rv.double() - (0.9 + 0.1 * var * count / (count - 1))).abs().max().item() < 1e-5 * max(1.0, var.max().item())


@pytest.mark.parametrize("M,C", [(16 * 256 * 256, 64), (70000, 24)])
def test_bn_backward_reduce_many_rows(M, C):
    """Both stages of the BatchNorm backward reduction with many partial rows against float64."""
    lib, ops = _gpu()
    g = torch.Generator().manual_seed(C)
    y = torch.randn(1, 1, M, C, generator=g).half()
    da = torch.randn(1, 1, M, C, generator=g).half()
    t = _tx(C, g)
    rstd = (0.5 + torch.rand(C, generator=g))
    z = y.double() * t[:, 1].double() + t[:, 2].double()
    dz = torch.where(z > 0, da.double(), torch.zeros((), dtype=torch.float64))
    xhat = (y.double() - t[:, 0].double()) * rstd.double()
    ref0, ref1 = dz.sum((0, 1, 2)), (dz * xhat).sum((0, 1, 2))
    s0, s1 = ops.bn_bwd(da.to(DEV), y.to(DEV), t.to(DEV), rstd.to(DEV), apply=False)
    assert (s0.cpu().double() - ref0).abs().max().item() < 1e-4 * max(1.0, ref0.abs().max().item())
    assert (s1.cpu().double() - ref1).abs().max().item() < 1e-4 * max(1.0, ref1.abs().max().item())


# ---- multi-tensor optimizer step and weight re-pack ---------------------------------------------
def _param_set(gen):
    """Tensors of very unequal sizes, incl. odd lengths and 4-byte-aligned views (flat-bucket slices)."""
    shapes = [(64, 32, 3, 3), (64,), (2,), (2, 64, 1, 1), (128, 64, 3, 3), (5, 7), (1,), (4099,), (300, 41)]
    ps = [torch.randn(s, generator=gen).to(DEV) for s in shapes]
    flat = torch.randn(3 + sum(p.numel() for p in ps), generator=gen).to(DEV)
    return ps, flat


def _run_optim(make, steps, flat_grads):
    gen = torch.Generator().manual_seed(5)
    ps, flat = _param_set(gen)
    params = [torch.nn.Parameter(p.clone()) for p in ps]
    opt = make(params)
    for s in range(steps):
        off = 3                                             # misaligned start: views are only 4-byte aligned
        for p in params:
            g = torch.randn(p.shape, generator=gen).to(DEV) * (0.5 + s)
            if flat_grads:
                v = flat[off:off + p.numel()].view(p.shape)
                v.copy_(g)
                p.grad = v
                off += p.numel()
            else:
                p.grad = g
        if s == 2:
            for group in opt.param_groups:                  # poly-LR style rewrite (reference Trainer.py:722-725)
                group["lr"] *= 0.7
        opt.step()
    torch.cuda.synchronize()
    return [p.detach().cpu() for p in params], opt


@pytest.mark.parametrize("kw", [dict(lr=0.01, momentum=0.9, weight_decay=1e-4),
                                dict(lr=0.05),
                                dict(lr=0.02, momentum=0.8, nesterov=True, weight_decay=1e-3),
                                dict(lr=0.02, momentum=0.9, dampening=0.1)])
@pytest.mark.parametrize("flat_grads", [False, True])
def test_optim_sgd_matches_torch(kw, flat_grads):
    _gpu()
    from umi import optim as uo
    mine, o1 = _run_optim(lambda ps: uo.SGD(ps, **kw), 5, flat_grads)
    ref, o2 = _run_optim(lambda ps: torch.optim.SGD(ps, foreach=True, **kw), 5, flat_grads)
    for a, b in zip(mine, ref):
        torch.testing.assert_close(a, b, rtol=2e-6, atol=1e-7)       # same op order; fma contraction may differ by an ulp
    s1, s2 = o1.state_dict(), o2.state_dict()
    assert s1["state"].keys() == s2["state"].keys()
    for k in s1["state"]:
        assert s1["state"][k].keys() == s2["state"][k].keys()
        if "momentum_buffer" in s1["state"][k] and s1["state"][k]["momentum_buffer"] is not None:
            torch.testing.assert_close(s1["state"][k]["momentum_buffer"].cpu(), s2["state"][k]["momentum_buffer"].cpu(),
                                       rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("kw", [dict(lr=5e-4, weight_decay=1e-4), dict(lr=1e-3, betas=(0.8, 0.95), eps=1e-6)])
def test_optim_adam_matches_torch(kw):
    _gpu()
    from umi import optim as uo
    mine, o1 = _run_optim(lambda ps: uo.Adam(ps, **kw), 6, True)
    ref, o2 = _run_optim(lambda ps: torch.optim.Adam(ps, foreach=True, **kw), 6, True)
    for a, b in zip(mine, ref):
        torch.testing.assert_close(a, b, rtol=5e-6, atol=2e-7)
    s1, s2 = o1.state_dict()["state"], o2.state_dict()["state"]
    for k in s1:
        assert s1[k].keys() == s2[k].keys() and float(s1[k]["step"]) == float(s2[k]["step"]) == 6.0
        for name in ("exp_avg", "exp_avg_sq"):
            # one-ulp differences of O(1) intermediates (fma contraction) show as absolute error where exp_avg cancels
            torch.testing.assert_close(s1[k][name].cpu(), s2[k][name].cpu(), rtol=5e-6, atol=1e-6)


def test_optim_refuses_cpu_parameters():
    _gpu()
    from umi import optim as uo
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="MI355X"):
        uo.SGD([p], lr=0.1).step()


def test_pack_cache_multi_repack_is_bit_identical_to_single_packs():
    _, ops = _gpu()
    gen = torch.Generator().manual_seed(9)
    ws = {"conv_fwd": torch.randn(64, 32, 3, 3, generator=gen), "conv_dgrad": torch.randn(128, 64, 3, 3, generator=gen),
          "convT_fwd": torch.randn(64, 32, 2, 2, generator=gen), "convT_dgrad": torch.randn(32, 16, 2, 2, generator=gen)}
    params = {k: torch.nn.Parameter(v.to(DEV)) for k, v in ws.items()}
    params["small"] = torch.nn.Parameter(torch.randn(2, 64, 1, 1, generator=gen).to(DEV))
    cache = ops.PackCache()
    combos = [(k, params[k], torch.float16, True) for k in ws] + [(k, params[k], torch.float16, False) for k in ws] + \
             [("conv_fwd", params["small"], torch.float16, False), ("conv_fwd", params["conv_fwd"], torch.float32, False)]

    def check():
        for kind, w, dt, k8 in combos:
            got = cache.get(kind, w, dt, k8)
            want = ops.PACKERS[kind](w.detach(), dt, k8=k8)
            assert torch.equal(got.cpu(), want.cpu()), (kind, dt, k8)
    check()                                                  # first use: packed one by one
    ptrs = [cache.get(*c).data_ptr() for c in combos]
    with torch.no_grad():
        for w in params.values():
            w.mul_(1.5).add_(0.25)                           # in-place update bumps ._version -> every entry stale
    cache.refresh()                                          # one launch per storage dtype
    assert all(e.ver == cache._ver(e.w()) for e in cache.ents.values())
    check()
    assert ptrs == [cache.get(*c).data_ptr() for c in combos]   # persistent buffers (no per-step allocation)
    with torch.no_grad():
        params["conv_dgrad"].zero_()
    check()                                                  # a stale entry met without refresh() is re-packed on use


# ---- inference pre-/post-processing --------------------------------------------------------------
def _ref_preprocess(img):
    """Restatement of the reference's `preprocess` arithmetic (test_mc3serousv5.py:115-127), no resize."""
    import numpy as np
    mean3d = np.mean(img, axis=(0, 1))
    std3d = np.std(img, axis=(0, 1))
    x = (img - mean3d) / std3d
    if img.ndim == 2:
        return x.astype(np.float32)[None, None]
    return np.ascontiguousarray(x.transpose((2, 0, 1))[::-1]).astype(np.float32)[None]


@pytest.mark.parametrize("shape,dtype", [((37, 53), "uint8"), ((64, 48, 3), "uint8"), ((33, 31, 3), "float32"),
                                         ((512, 512), "uint8"), ((20, 24, 4), "uint8"), ((18, 22, 2), "float32"),
                                         ((16, 16, 1), "uint8")])      # every HWC input is channel-reversed, as in the reference
def test_znorm_preprocess_matches_numpy(shape, dtype):
    _gpu()
    import numpy as np
    from umi import infer
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=shape).astype(np.uint8) if dtype == "uint8" else \
        (rng.standard_normal(shape) * 40 + 100).astype(np.float32)
    # uint8 (what cv2.imread hands the reference) goes through fp64 in numpy; float32 images are this build's extension
    # and use the same fp64 statistics (numpy would reduce them in fp32)
    want = _ref_preprocess(img if dtype == "uint8" else img.astype(np.float64))
    got = infer.preprocess(img).cpu().numpy()
    assert got.shape == want.shape and got.dtype == np.float32
    # fp64 statistics on both sides, one rounding to fp32: at most the last bit differs, and only rarely
    np.testing.assert_allclose(got, want, rtol=0, atol=2.5e-7 * max(1.0, float(np.abs(want).max())))
    assert (got == want).mean() > 0.995


@pytest.mark.parametrize("N,C,H,W", [(2, 2, 64, 64), (1, 4, 33, 47), (3, 3, 20, 20), (1, 7, 5, 9), (2, 2, 512, 512)])
def test_argmax_mask_matches_softmax_argmax(N, C, H, W):
    _gpu()
    from umi import infer
    gen = torch.Generator().manual_seed(N * 100 + C)
    logits = torch.randn(N, C, H, W, generator=gen) * 3
    logits[0, :, 0, 0] = 1.5                                  # an exact tie: the first class wins
    want = torch.argmax(F.softmax(logits, dim=1), dim=1).to(torch.uint8)
    got = infer.argmax_mask(logits.to(DEV)).cpu()
    top2 = torch.topk(logits, 2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-5                  # softmax can merge near-ties that the logits still separate
    assert got.dtype == torch.uint8 and got.shape == (N, H, W)
    assert torch.equal(got[safe], want[safe])
    assert int(got[0, 0, 0]) == 0
    assert torch.equal(got, torch.argmax(logits, dim=1).to(torch.uint8))


@pytest.mark.parametrize("shape,acc", [((2, 16, 24, 64), True), ((1, 8, 8, 8), False), ((3, 32, 16, 128), True)])
def test_pool2_bwd_with_bn_reduction_matches_separate_kernels(shape, acc):
    """umi_pool2_bwd_bnred (max-pool backward + stage 1 of the pooled layer's BatchNorm backward) against umi_pool2_bwd
    followed by umi_bn_bwd_reduce on the same inputs: identical routed gradient, sums equal up to fp32 summation order."""
    lib, ops = _gpu()
    N, H, W, C = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(N, H, W, C, generator=g).half().to(DEV)
    t = _tx(C, g)
    t[:, 0] = 0.1 * torch.randn(C, generator=g)
    td = t.to(DEV).contiguous()
    rstd = (0.5 + torch.rand(C, generator=g)).to(DEV)
    dp = (torch.randn(N, H // 2, W // 2, C, generator=g) * 0.1).half().to(DEV)
    da0 = (torch.randn(N, H, W, C, generator=g) * 0.1).half().to(DEV)
    da_a, da_b = da0.clone(), da0.clone()
    ops.pool2_bwd(dp, x, td, da_a, acc)
    part = ops.pool2_bwd_bnred(dp, x, td, rstd, da_b, acc)
    assert part is not None
    assert torch.equal(da_a, da_b)
    sums = torch.empty(2, C, dtype=torch.float32, device=DEV)
    lib.check(lib.fn("umi_bn_bwd_from_partials")(part.data_ptr(), part.numel() // (2 * C), C, sums[0].data_ptr(),
                                                 sums[1].data_ptr(), ops._stream()), "from_partials")
    ref = torch.empty(2, C, dtype=torch.float32, device=DEV)
    M = N * H * W
    nb = lib.fn("umi_bn_bwd_ws_bytes")(M, C)
    ws = ops.workspace(nb, x.device)
    lib.check(lib.fn("umi_bn_bwd_reduce")(da_a.data_ptr(), C, x.data_ptr(), C, td.data_ptr(), rstd.data_ptr(), ref[0].data_ptr(),
                                          ref[1].data_ptr(), M, C, lib.UMI_F16, ws.data_ptr(), ws.numel(), ops._stream()),
              "reduce")
    torch.testing.assert_close(sums.cpu(), ref.cpu(), rtol=1e-4, atol=1e-4 * float(ref.abs().max()))


@pytest.mark.parametrize("shape", [(2, 5, 6, 128, 64), (1, 16, 16, 64, 128), (2, 9, 37, 256, 128), (1, 40, 40, 128, 256)])
def test_convT_wgrad_with_bias_gradient_matches_separate_kernels(shape):
    """umi_conv_wgrad_bias (ConvTranspose2d(2,2) weight gradient that also sums the upsampled gradient's columns = the bias gradient)
    against umi_conv_wgrad + umi_colsum: the same weight gradient bit for bit, bias gradient equal up to fp32 summation order."""
    lib, ops = _gpu()
    N, h, w, Cin, Cout = shape
    g = torch.Generator().manual_seed(sum(shape) + 9)
    xin = torch.randn(N, h, w, Cin, generator=g).half().to(DEV)
    t = _tx(Cin, g).to(DEV)
    dupb = (torch.randn(N, 2 * h, 2 * w, 2 * Cout, generator=g) * 0.1).half().to(DEV)
    dup = dupb[..., Cout:]                                   # a channel slice, as in the concat buffers
    gw_a = torch.empty(Cin, Cout, 2, 2, device=DEV)
    gw_b = torch.empty_like(gw_a)
    gb_a = torch.empty(Cout, device=DEV)
    gb_b = torch.empty_like(gb_a)
    ops.conv_wgrad(dup, None, xin, t, gw_a, Cout * 4, 4, 1, 0.5, 2, 2, 2, 0)
    ops.colsum(dup, gb_a, 0.5)
    assert ops.convT_wgrad_bias(dup, xin, t, gw_b, gb_b, 0.5)
    assert torch.equal(gw_a, gw_b)
    torch.testing.assert_close(gb_b.cpu(), gb_a.cpu(), rtol=1e-4, atol=1e-4 * float(gb_a.abs().max()))
    defer = []
    gw_c, gb_c = torch.empty_like(gw_a), torch.empty_like(gb_a)
    assert ops.convT_wgrad_bias(dup, xin, t, gw_c, gb_c, 0.5, defer=defer)
    ops.wgrad_reduce_flush(defer)
    assert torch.equal(gw_a, gw_c) and torch.equal(gb_b, gb_c)


@pytest.mark.parametrize("shape", [(2, 5, 6, 128, 64), (1, 16, 16, 64, 128), (2, 9, 37, 256, 128), (1, 40, 40, 128, 256)])
def test_convT_dgrad_with_bn_reduction_matches_separate_kernels(shape):
    """umi_conv_gather_bnred (ConvTranspose2d(2,2) data gradient on the tap-gather matrix-core kernel + stage 1 of the BatchNorm
    backward of the layer below) against umi_conv_fwd followed by umi_bn_bwd_reduce: identical gradient tensor, sums equal up to
    fp32 summation order.  The gradient arrives as a channel slice of a wider tensor, as in the U-Net's concat buffers."""
    lib, ops = _gpu()
    N, h, w, Cin, Cout = shape
    g = torch.Generator().manual_seed(sum(shape) + 5)
    wt = (torch.randn(Cin, Cout, 2, 2, generator=g) * (1.0 / Cin) ** 0.5).to(DEV)
    dupb = (torch.randn(N, 2 * h, 2 * w, 2 * Cout, generator=g) * 0.1).half().to(DEV)
    dup = dupb[..., Cout:]
    y = torch.randn(N, h, w, Cin, generator=g).half().to(DEV)          # raw output of the BatchNorm layer below the ConvT
    t = _tx(Cin, g)
    t[:, 0] = 0.1 * torch.randn(Cin, generator=g)
    td = t.to(DEV).contiguous()
    rstd = (0.5 + torch.rand(Cin, generator=g)).to(DEV)
    wp = ops.pack_convT_dgrad(wt, torch.float16, k8=True)
    dx_a = torch.empty(N, h, w, Cin, device=DEV, dtype=torch.float16)
    dx_b = torch.empty_like(dx_a)
    ops.conv_fwd(dup, None, lambda lay: wp, None, dx_a, 2, 2, 2, 0)
    part = ops.conv_gather_bnred(dup, wp, dx_b, y, td, rstd, 2, 2, 2, 0)
    assert part is not None
    assert torch.equal(dx_a, dx_b)
    sums = torch.empty(2, Cin, dtype=torch.float32, device=DEV)
    lib.check(lib.fn("umi_bn_bwd_from_partials")(part.data_ptr(), part.numel() // (2 * Cin), Cin, sums[0].data_ptr(),
                                                 sums[1].data_ptr(), ops._stream()), "from_partials")
    ref = torch.empty(2, Cin, dtype=torch.float32, device=DEV)
    M = N * h * w
    nb = lib.fn("umi_bn_bwd_ws_bytes")(M, Cin)
    ws = ops.workspace(nb, y.device)
    lib.check(lib.fn("umi_bn_bwd_reduce")(dx_a.data_ptr(), Cin, y.data_ptr(), Cin, td.data_ptr(), rstd.data_ptr(), ref[0].data_ptr(),
                                          ref[1].data_ptr(), M, Cin, lib.UMI_F16, ws.data_ptr(), ws.numel(), ops._stream()),
              "reduce")
    torch.testing.assert_close(sums.cpu(), ref.cpu(), rtol=1e-4, atol=1e-4 * float(ref.abs().max()))


@pytest.mark.parametrize("shape", [(2, 16, 24, 64, 2), (1, 9, 7, 64, 4), (3, 32, 16, 128, 1), (1, 8, 8, 16, 8)])
def test_head_dgrad_with_bn_reduction_matches_separate_kernels(shape):
    """umi_head_dgrad_bnred (OutConv's data gradient + stage 1 of the last DoubleConv's BatchNorm backward) against the narrow
    pointwise data gradient followed by umi_bn_bwd_reduce: identical gradient tensor, sums equal up to fp32 summation order."""
    lib, ops = _gpu()
    N, H, W, C, ncls = shape
    g = torch.Generator().manual_seed(sum(shape))
    y = torch.randn(N, H, W, C, generator=g).half().to(DEV)            # the BatchNorm layer's raw output
    t = _tx(C, g)
    t[:, 0] = 0.1 * torch.randn(C, generator=g)
    td = t.to(DEV).contiguous()
    rstd = (0.5 + torch.rand(C, generator=g)).to(DEV)
    dl = (torch.randn(N, H, W, ncls, generator=g) * 0.1).half().to(DEV)
    wo = (torch.randn(ncls, C, 1, 1, generator=g) * 0.2).to(DEV)
    wp = ops.pack_conv_dgrad(wo, torch.float16, k8=False)
    da_a = torch.empty(N, H, W, C, device=DEV, dtype=torch.float16)
    da_b = torch.empty_like(da_a)
    ops.conv_fwd(dl, None, lambda lay: ops.pack_conv_dgrad(wo, torch.float16, k8=bool(lay)), None, da_a, 1, 1, 1, 0)
    part = ops.head_dgrad_bnred(dl, wp, da_b, y, td, rstd)
    assert part is not None
    assert torch.equal(da_a, da_b)
    sums = torch.empty(2, C, dtype=torch.float32, device=DEV)
    lib.check(lib.fn("umi_bn_bwd_from_partials")(part.data_ptr(), part.numel() // (2 * C), C, sums[0].data_ptr(),
                                                 sums[1].data_ptr(), ops._stream()), "from_partials")
    ref = torch.empty(2, C, dtype=torch.float32, device=DEV)
    M = N * H * W
    nb = lib.fn("umi_bn_bwd_ws_bytes")(M, C)
    ws = ops.workspace(nb, y.device)
    lib.check(lib.fn("umi_bn_bwd_reduce")(da_a.data_ptr(), C, y.data_ptr(), C, td.data_ptr(), rstd.data_ptr(), ref[0].data_ptr(),
                                          ref[1].data_ptr(), M, C, lib.UMI_F16, ws.data_ptr(), ws.numel(), ops._stream()),
              "reduce")
    torch.testing.assert_close(sums.cpu(), ref.cpu(), rtol=1e-4, atol=1e-4 * float(ref.abs().max()))
    # ... and with the head's weight gradient from the same pass (umi_head_bwd_fused) against umi_conv_wgrad
    gw_ref = torch.empty(ncls, C, 1, 1, device=DEV)
    ops.conv_wgrad(y, td, dl, None, gw_ref, C, 1, 1, 0.5, 1, 1, 1, 0)
    gw = torch.empty_like(gw_ref)
    da_c = torch.empty_like(da_a)
    part2 = ops.head_dgrad_bnred(dl, wp, da_c, y, td, rstd, dW=gw, out_scale=0.5)
    assert part2 is not None and torch.equal(da_c, da_a) and torch.equal(part2, part)
    torch.testing.assert_close(gw.cpu(), gw_ref.cpu(), rtol=2e-4, atol=2e-4 * float(gw_ref.abs().max()))


@pytest.mark.parametrize("shape", [(2, 37, 29, 1, 64), (1, 16, 48, 3, 64), (3, 9, 70, 1, 32)])
def test_stem_wgrad_with_bn_backward_apply_matches_separate_kernels(shape):
    """umi_conv_wgrad_bnapply with dz = NULL on the network's first conv (<= 4 input channels, input takes no gradient): the
    narrow-input weight-gradient kernel forms stage 3 of the BatchNorm backward on the fly; same weight gradient as
    umi_bn_bwd_apply followed by umi_conv_wgrad, bit for bit (same expression, same rounding, same summation order)."""
    lib, ops = _gpu()
    N, H, W, Ci, Co = shape
    g = torch.Generator().manual_seed(sum(shape) + 3)
    x = torch.randn(N, H, W, Ci, generator=g).half().to(DEV)
    y = torch.randn(N, H, W, Co, generator=g).half().to(DEV)
    t = _tx(Co, g)
    t[:, 0] = 0.1 * torch.randn(Co, generator=g)
    td = t.to(DEV).contiguous()
    rstd = (0.5 + torch.rand(Co, generator=g)).to(DEV)
    da = (torch.randn(N, H, W, Co, generator=g) * 0.1).half().to(DEV)
    sums = (torch.randn(2, Co, generator=g) * 0.05 * N * H * W).to(DEV)
    gw_a = torch.empty(Co, Ci, 3, 3, device=DEV)
    gw_b = torch.empty_like(gw_a)
    assert ops.conv_wgrad_bnapply(x, None, da, y, td, rstd, sums[0], sums[1], None, gw_b, Ci * 9, 9, 1, 0.25, 3, 3, 1, 1)
    dz = da.clone()
    ops.bn_bwd_apply(dz, y, td, rstd, sums[0], sums[1])
    ops.conv_wgrad(x, None, dz, None, gw_a, Ci * 9, 9, 1, 0.25, 3, 3, 1, 1)
    assert torch.equal(gw_a, gw_b)


@pytest.mark.parametrize("shape", [(2, 16, 64, 64, 64), (1, 13, 37, 72, 136), (3, 8, 32, 128, 64), (1, 5, 70, 16, 24),
                                   (2, 64, 64, 64, 128)])
@pytest.mark.parametrize("strided", [False, True])
def test_wgrad_with_bn_backward_apply_matches_separate_kernels(shape, strided):
    """umi_conv_wgrad_bnapply (3x3 weight gradient whose producer waves also do stage 3 of the following BatchNorm's backward)
    against umi_bn_bwd_apply + umi_conv_wgrad on the same inputs: the gradient of the raw conv output and the weight gradient
    are both identical bit for bit (same expression, same rounding, same split-K order).  Ragged tiles (H % 4, W % 32, channel
    counts that are not multiples of 64) and, with strided=True, operands that are channel slices of wider tensors."""
    lib, ops = _gpu()
    N, H, W, Ci, Co = shape
    g = torch.Generator().manual_seed(sum(shape))

    def slab(c):                                       # [N,H,W,c] view, optionally a channel slice of a wider tensor
        full = torch.randn(N, H, W, c + (16 if strided else 0), generator=g).half().to(DEV)
        return full[..., 8:8 + c] if strided else full

    x, y, da = slab(Ci), slab(Co), slab(Co)
    da.mul_(0.1)
    txa = _tx(Ci, g).to(DEV).contiguous()
    tb = _tx(Co, g)
    tb[:, 0] = 0.1 * torch.randn(Co, generator=g)
    tb = tb.to(DEV).contiguous()
    rstd = (0.5 + torch.rand(Co, generator=g)).to(DEV)
    da_ref = da.clone()
    s1, s2 = ops.bn_bwd(da_ref, y, tb, rstd)                       # in place: da_ref <- dz
    gw_ref = torch.empty(Co, Ci, 3, 3, device=DEV)
    ops.conv_wgrad(x, txa, da_ref, None, gw_ref, Ci * 9, 9, 1, 0.5, 3, 3, 1, 1)
    keep = da.clone()
    t1, t2 = ops.bn_bwd(da, y, tb, rstd, apply=False)
    assert torch.equal(da, keep) and torch.equal(s1, t1) and torch.equal(s2, t2)
    dz = torch.full((N, H, W, Co), float("nan"), device=DEV, dtype=torch.float16)
    gw = torch.empty(Co, Ci, 3, 3, device=DEV)
    assert ops.conv_wgrad_bnapply(x, txa, da, y, tb, rstd, t1, t2, dz, gw, Ci * 9, 9, 1, 0.5, 3, 3, 1, 1)
    torch.cuda.synchronize()
    assert torch.equal(da, keep)                                   # the fused kernel only reads dA
    assert torch.equal(dz, da_ref.contiguous())
    assert torch.equal(gw, gw_ref)
