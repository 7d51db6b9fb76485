"""Exact-arithmetic parity of the matrix-core kernels at the benchmark's channel widths.

The benchmarked path stores fp16 and accumulates in fp32, so against the fp32 oracle it can only be compared within a rounding
tolerance (tests/test_gpu_kernels.py, tests/test_gpu_unet.py).  Here the operands are small integers: every product and every
partial sum is exactly representable in fp16 / fp32, whatever the summation order, so the MFMA kernels -- the very binaries
the bench runs, at 64 ... 1024 channels -- must reproduce the reference convolution BIT FOR BIT.  Indexing, tiling, halo,
tap order, channel-block order, split-K partial sums and the epilogue statistics are checked with zero tolerance; what the
tolerance tests allow for is then rounding alone.  Reference: torch fp32 conv2d / conv_transpose2d on the CPU (exact on this
data too), i.e. /root/reference/Model.py:15-22 (DoubleConv convs), :56-57 (ConvTranspose2d), :126-135 (OutConv)."""
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


def _ints(shape, lo, hi, gen):
    return torch.randint(lo, hi + 1, shape, generator=gen).float()


def _int_tx(C, gen):
    """Consumer-side transform rows with integer effect: scale in {1, -1, 2}, integer shift, ReLU."""
    t = torch.zeros(C, 4)
    t[:, 1] = torch.tensor([1.0, -1.0, 2.0])[torch.randint(0, 3, (C,), generator=gen)]
    t[:, 2] = _ints((C,), -1, 1, gen)
    return t


def _apply(x, t):
    return torch.clamp_min(x * t[:, 1] + t[:, 2], 0.0)


# N, H, W, Ci, Co, transform on load          (Co % 128 == 0 and Ci % 32 == 0: the 16x16x32 form; else the 32x32x16 form)
CONV_CASES = [
    (1, 16, 64, 64, 64, True),        # enc0.c2 / dec3.c2 widths (32-channel halo tiles, 64-channel tile)
    (1, 16, 64, 128, 64, True),       # dec3.c1
    (2, 16, 32, 64, 128, True),       # enc1.c1: two super-chunks
    (1, 16, 32, 128, 128, False),     # a data gradient's shape (no transform)
    (1, 8, 32, 512, 512, True),       # enc3.c2 / dec0.c2 widths, four channel blocks
    (1, 8, 32, 1024, 512, True),      # dec0.c1: 32 super-chunks
    (1, 8, 32, 1024, 1024, False),    # enc4.c2 as a data gradient
    (2, 11, 37, 96, 128, True),       # ragged tiles, 3 super-chunks
    (1, 20, 45, 48, 72, True),        # 32x32x16 form, partial channel tile, odd sizes
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3_forward_is_exact_on_integer_data(case):
    lib, ops = _gpu()
    N, H, W, Ci, Co, use_tx = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = _ints((N, H, W, Ci), -2, 2, g)
    w = _ints((Co, Ci, 3, 3), -1, 1, g)
    t = _int_tx(Ci, g) if use_tx else None
    a = _apply(x, t) if use_tx else x
    ref = F.conv2d(a.permute(0, 3, 1, 2), w, None, 1, 1).permute(0, 2, 3, 1).contiguous()
    assert ref.abs().max().item() < 2048          # every output is an fp16 integer
    xd, wd = x.half().to(DEV), w.to(DEV)
    y = torch.empty(N, H, W, Co, device=DEV, dtype=torch.float16)
    assert ops.conv_plan(xd, y, 3, 3, 1, 1, 0)[0] == 1
    part = ops.conv_fwd(xd, t.to(DEV) if use_tx else None, lambda l: ops.pack_conv_fwd(wd, torch.float16, k8=bool(l)), None, y,
                        3, 3, 1, 1, want_stats=True)
    got = y.float().cpu()
    assert torch.equal(got, ref)
    # the BatchNorm-statistics epilogue: integer sums below 2^24, exact in fp32 in any order
    s = part.view(-1, 2, Co).sum(0).cpu()
    assert (ref * ref).sum((0, 1, 2)).max().item() < 2 ** 24
    assert torch.equal(s[0], ref.sum((0, 1, 2))) and torch.equal(s[1], (ref * ref).sum((0, 1, 2)))


@pytest.mark.parametrize("case", [(1, 16, 64, 64, 64), (2, 16, 32, 128, 64), (1, 8, 32, 512, 1024), (1, 8, 32, 256, 128),
                                  (2, 11, 37, 128, 96)])
def test_conv3x3_data_gradient_is_exact_on_integer_data(case):
    """dgrad = the same kernel on the flipped / transposed weight panel, against autograd of conv2d."""
    lib, ops = _gpu()
    N, H, W, Ci, Co = case                                     # forward widths: the gradient maps Co -> Ci channels
    g = torch.Generator().manual_seed(sum(case))
    w = _ints((Co, Ci, 3, 3), -1, 1, g)
    dy = _ints((N, H, W, Co), -1, 1, g)
    xr = torch.zeros(N, Ci, H, W, requires_grad=True)
    F.conv2d(xr, w, None, 1, 1).backward(dy.permute(0, 3, 1, 2))
    ref = xr.grad.permute(0, 2, 3, 1).contiguous()
    assert ref.abs().max().item() < 2048
    dx = torch.empty(N, H, W, Ci, device=DEV, dtype=torch.float16)
    wd, dyd = w.to(DEV), dy.half().to(DEV)
    assert ops.conv_plan(dyd, dx, 3, 3, 1, 1)[0] == 1
    ops.conv_fwd(dyd, None, lambda l: ops.pack_conv_dgrad(wd, torch.float16, k8=bool(l)), None, dx, 3, 3, 1, 1)
    assert torch.equal(dx.float().cpu(), ref)


@pytest.mark.parametrize("case", [(2, 32, 64, 64, 64), (1, 32, 32, 128, 64), (2, 16, 32, 256, 256), (1, 16, 16, 1024, 512),
                                  (2, 16, 16, 512, 1024), (2, 20, 45, 32, 64)])
def test_conv3x3_weight_gradient_is_exact_on_integer_data(case):
    """Warp-specialised weight-gradient kernel incl. its split-K partial slabs and their reduction, against autograd."""
    lib, ops = _gpu()
    N, H, W, Ci, Co = case
    g = torch.Generator().manual_seed(sum(case))
    x = _ints((N, H, W, Ci), -2, 2, g)
    dy = _ints((N, H, W, Co), -1, 1, g)
    t = _int_tx(Ci, g)
    a = _apply(x, t).permute(0, 3, 1, 2)
    wr = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    F.conv2d(a, wr, None, 1, 1).backward(dy.permute(0, 3, 1, 2))
    ref = wr.grad * 0.5                                        # out_scale = 1 / loss scale: a power of two
    gw = torch.empty(Co, Ci, 3, 3, device=DEV)
    ops.conv_wgrad(x.half().to(DEV), t.to(DEV), dy.half().to(DEV), None, gw, Ci * 9, 9, 1, 0.5, 3, 3, 1, 1)
    assert torch.equal(gw.cpu(), ref)


@pytest.mark.parametrize("case", [(1, 16, 16, 128, 64), (1, 8, 16, 1024, 512), (2, 9, 7, 256, 128)])
def test_conv_transpose_trio_is_exact_on_integer_data(case):
    """ConvTranspose2d(2, 2) forward (scatter into a concat slice), data gradient and weight + bias gradient."""
    lib, ops = _gpu()
    N, h, w, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    x = _ints((N, h, w, Cin), -2, 2, g)
    wt = _ints((Cin, Cout, 2, 2), -1, 1, g)
    b = _ints((Cout,), -3, 3, g)
    t = _int_tx(Cin, g)
    a = _apply(x, t).permute(0, 3, 1, 2)
    ref = F.conv_transpose2d(a, wt, b, stride=2).permute(0, 2, 3, 1).contiguous()
    assert ref.abs().max().item() < 2048
    wd, xd, td = wt.to(DEV), x.half().to(DEV), t.to(DEV)
    buf = torch.zeros(N, 2 * h, 2 * w, 2 * Cout, device=DEV, dtype=torch.float16)
    dest = buf[..., Cout:]
    assert ops.conv_plan(xd, dest, 2, 2, 2, 0, lib.CONV_UPSAMPLE2)[0] == 1
    ops.conv_fwd(xd, td, lambda l: ops.pack_convT_fwd(wd, torch.float16, k8=bool(l)), b.to(DEV), dest, 2, 2, 2, 0,
                 flags=lib.CONV_UPSAMPLE2, up_offset=(0, 0))
    got = buf.float().cpu()
    assert torch.equal(got[..., Cout:], ref) and (got[..., :Cout] == 0).all()
    # data gradient
    dup = _ints((N, 2 * h, 2 * w, Cout), -1, 1, g)
    xr = torch.zeros(N, Cin, h, w, requires_grad=True)
    wr = wt.clone().requires_grad_(True)
    br = torch.zeros(Cout, requires_grad=True)
    F.conv_transpose2d(xr, wt, None, stride=2).backward(dup.permute(0, 3, 1, 2))
    F.conv_transpose2d(a, wr, br, stride=2).backward(dup.permute(0, 3, 1, 2))
    dx = torch.empty(N, h, w, Cin, device=DEV, dtype=torch.float16)
    dupd = dup.half().to(DEV)
    assert ops.conv_plan(dupd, dx, 2, 2, 2, 0, 0)[0] == 1
    ops.conv_fwd(dupd, None, lambda l: ops.pack_convT_dgrad(wd, torch.float16, k8=bool(l)), None, dx, 2, 2, 2, 0)
    assert xr.grad.abs().max().item() < 2048
    assert torch.equal(dx.float().cpu(), xr.grad.permute(0, 2, 3, 1))
    # weight gradient with the bias gradient riding on it
    gw = torch.empty(Cin, Cout, 2, 2, device=DEV)
    gb = torch.empty(Cout, device=DEV)
    ops.convT_wgrad_bias(dupd, xd, td, gw, gb, 1.0)
    assert torch.equal(gw.cpu(), wr.grad) and torch.equal(gb.cpu(), br.grad)


@pytest.mark.parametrize("case", [(1, 48, 98, 768, 768), (1, 48, 98, 768, 3072), (1, 25, 40, 3072, 768), (2, 13, 11, 192, 64)])
def test_pointwise_gemm_is_exact_on_integer_data(case):
    """The pointwise MFMA kernel on the ViT linears' shapes (4704 tokens) with an integer bias."""
    lib, ops = _gpu()
    N, H, W, Ci, Co = case
    g = torch.Generator().manual_seed(sum(case))
    x = _ints((N, H, W, Ci), -1, 1, g)
    wt = _ints((Co, Ci, 1, 1), -1, 1, g)
    b = _ints((Co,), -4, 4, g)
    ref = F.conv2d(x.permute(0, 3, 1, 2), wt, b).permute(0, 2, 3, 1).contiguous()
    assert ref.abs().max().item() < 2048
    y = torch.empty(N, H, W, Co, device=DEV, dtype=torch.float16)
    wd, xd = wt.to(DEV), x.half().to(DEV)
    assert ops.conv_plan(xd, y, 1, 1, 1, 0, 0, True)[0] == 1
    ops.conv_fwd(xd, None, lambda l: ops.pack_conv_fwd(wd, torch.float16, k8=bool(l)), b.to(DEV), y, 1, 1, 1, 0)
    assert torch.equal(y.float().cpu(), ref)


@pytest.mark.parametrize("cin", [1, 3])
def test_stem_and_head_are_exact_on_integer_data(cin):
    """First conv (Ci <= 4: forward, statistics, weight gradient) and the OutConv head (fp32 logits with bias, data and weight
    gradient) -- the narrow kernels around the MFMA trunk (csrc/stem_head.hip)."""
    lib, ops = _gpu()
    N, H, W, C, ncls = 2, 37, 29, 64, 2 if cin == 1 else 4
    g = torch.Generator().manual_seed(10 + cin)
    x = _ints((N, H, W, cin), -3, 3, g)
    w = _ints((C, cin, 3, 3), -2, 2, g)
    dy = _ints((N, H, W, C), -1, 1, g)
    wr = w.clone().requires_grad_(True)
    ref = F.conv2d(x.permute(0, 3, 1, 2), wr, None, 1, 1)
    ref.backward(dy.permute(0, 3, 1, 2))
    ref = ref.detach().permute(0, 2, 3, 1).contiguous()
    xd, wd = x.half().to(DEV), w.to(DEV)
    y = torch.empty(N, H, W, C, device=DEV, dtype=torch.float16)
    part = ops.conv_fwd(xd, None, lambda l: ops.pack_conv_fwd(wd, torch.float16, k8=bool(l)), None, y, 3, 3, 1, 1, want_stats=True)
    gw = torch.empty(C, cin, 3, 3, device=DEV)
    ops.conv_wgrad(xd, None, dy.half().to(DEV), None, gw, cin * 9, 9, 1, 0.25, 3, 3, 1, 1)
    assert torch.equal(y.float().cpu(), ref)
    s = part.view(-1, 2, C).sum(0).cpu()
    assert torch.equal(s[0], ref.sum((0, 1, 2))) and torch.equal(s[1], (ref * ref).sum((0, 1, 2)))
    assert torch.equal(gw.cpu(), wr.grad * 0.25)

    a = _ints((N, H, W, C), -2, 2, g)
    t = _int_tx(C, g)
    wo = _ints((ncls, C, 1, 1), -2, 2, g)
    b = _ints((ncls,), -3, 3, g)
    dl = _ints((N, H, W, ncls), -2, 2, g)
    act = _apply(a, t).permute(0, 3, 1, 2).requires_grad_(True)
    wor = wo.clone().requires_grad_(True)
    out = F.conv2d(act, wor, b)
    out.backward(dl.permute(0, 3, 1, 2))
    ad, td, wod, dld = a.half().to(DEV), t.to(DEV), wo.to(DEV), dl.half().to(DEV)
    logits = torch.empty(N, H, W, ncls, device=DEV, dtype=torch.float32)
    ops.conv_fwd(ad, td, lambda l: ops.pack_conv_fwd(wod, torch.float16, k8=bool(l)), b.to(DEV), logits, 1, 1, 1, 0)
    da = torch.empty(N, H, W, C, device=DEV, dtype=torch.float16)
    ops.conv_fwd(dld, None, lambda l: ops.pack_conv_dgrad(wod, torch.float16, k8=bool(l)), None, da, 1, 1, 1, 0)
    gwo = torch.empty(ncls, C, 1, 1, device=DEV)
    ops.conv_wgrad(ad, td, dld, None, gwo, C, 1, 1, 2.0, 1, 1, 1, 0)
    assert torch.equal(logits.cpu(), out.detach().permute(0, 2, 3, 1))
    assert torch.equal(da.float().cpu(), act.grad.permute(0, 2, 3, 1))
    assert torch.equal(gwo.cpu(), wor.grad * 2.0)


def test_maxpool_with_transform_is_exact_on_integer_data():
    """MaxPool2d(2) of the lazily-activated tensor (transform on load) and its backward scatter (ties: torch's first-max rule
    is not assumed -- the data has no ties inside a window)."""
    lib, ops = _gpu()
    N, H, W, C = 2, 12, 20, 64
    g = torch.Generator().manual_seed(3)
    # distinct values inside every 2 x 2 window: a permutation of 0..3 per window, times a positive per-channel step
    base = torch.stack([torch.randperm(4, generator=g) for _ in range(N * (H // 2) * (W // 2) * C)]).float()
    base = base.view(N, H // 2, W // 2, C, 2, 2).permute(0, 1, 4, 2, 5, 3).reshape(N, H, W, C)
    x = base + 1.0
    t = torch.zeros(C, 4)
    t[:, 1] = torch.tensor([1.0, 2.0])[torch.randint(0, 2, (C,), generator=g)]
    t[:, 2] = _ints((C,), -2, 0, g)
    act = _apply(x, t).permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.max_pool2d(act, 2)
    dp = _ints((N, H // 2, W // 2, C), -3, 3, g)
    ref.backward(dp.permute(0, 3, 1, 2))
    xd, td = x.half().to(DEV), t.to(DEV)
    y = torch.empty(N, H // 2, W // 2, C, device=DEV, dtype=torch.float16)
    ops.pool2_fwd(xd, td, y)
    assert torch.equal(y.float().cpu(), ref.detach().permute(0, 2, 3, 1))
    da = torch.empty(N, H, W, C, device=DEV, dtype=torch.float16)
    ops.pool2_bwd(dp.half().to(DEV), xd, td, da, False)
    # windows whose activated values tie at 0 (ReLU clipped) route the gradient by convention: compare where the max is unique
    a4 = act.detach().view(N, C, H // 2, 2, W // 2, 2)
    mx = a4.amax((3, 5), keepdim=True)
    unique = ((a4 == mx).sum((3, 5), keepdim=True) == 1).expand_as(a4).reshape(N, C, H, W).permute(0, 2, 3, 1)
    assert unique.float().mean().item() > 0.5
    assert torch.equal(da.float().cpu()[unique], act.grad.permute(0, 2, 3, 1)[unique])


@pytest.mark.parametrize("R,stride,pad,Ci,Co,H,W", [(3, 2, 1, 64, 128, 11, 13), (1, 2, 0, 128, 64, 11, 13), (3, 2, 1, 256, 256, 28, 28),
                                                   (1, 1, 0, 1024, 256, 14, 14), (7, 2, 3, 64, 64, 20, 17)])
def test_strided_convs_are_exact_on_integer_data(R, stride, pad, Ci, Co, H, W):
    """TransUNet's strided / 1x1 bottleneck convs on the tap-gather MFMA kernels (reference
    TransUnet/vit_seg_modeling_resnet_skip.py:52-60): forward with a transform on load (zero padding applied AFTER it), the
    fractionally-strided data gradient and the weight gradient."""
    lib, ops = _gpu()
    from umi.graph_tu import TUTape
    g = torch.Generator().manual_seed(R * 100 + Ci)
    N = 2
    x = _ints((N, H, W, Ci), -2, 2, g)
    w = _ints((Co, Ci, R, R), -1, 1, g)
    t = _int_tx(Ci, g)
    xa = _apply(x, t).permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(xa, wr, None, stride, pad)
    assert y_ref.abs().max().item() < 2048
    gy = _ints(tuple(y_ref.shape), -1, 1, g)
    y_ref.backward(gy)
    assert xa.grad.abs().max().item() < 2048
    Ho, Wo = y_ref.shape[2:]
    xd, td, wd = x.half().to(DEV), t.to(DEV), w.to(DEV)
    y = torch.empty(N, Ho, Wo, Co, device=DEV, dtype=torch.float16)
    assert ops.conv_plan(xd, y, R, R, stride, pad)[0] == 1
    ops.conv_fwd(xd, td, lambda l: ops.pack_conv_fwd(wd, torch.float16, k8=bool(l)), None, y, R, R, stride, pad)
    assert torch.equal(y.float().cpu(), y_ref.detach().permute(0, 2, 3, 1))
    gyd = gy.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    dx = torch.empty(N, H, W, Ci, device=DEV, dtype=torch.float16)
    if stride > 1 or R > 1:
        TUTape._strided_dgrad(gyd, lambda l: ops.pack_conv_dgrad_strided(wd, torch.float16, k8=bool(l)), dx, R, R, stride, pad)
    else:
        ops.conv_fwd(gyd, None, lambda l: ops.pack_conv_dgrad(wd, torch.float16, k8=bool(l)), None, dx, 1, 1, 1, 0)
    assert torch.equal(dx.float().cpu(), xa.grad.permute(0, 2, 3, 1))
    dW = torch.empty(Co, Ci, R, R, device=DEV)
    ops.conv_wgrad(xd, td, gyd, None, dW, Ci * R * R, R * R, 1, 1.0, R, R, stride, pad)
    assert torch.equal(dW.cpu(), wr.grad)
