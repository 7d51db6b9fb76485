"""GPU parity tests (run on the MI355X with -m gpu): HIP path vs the golden fixtures (outputs of the
reference itself) and vs the CPU oracle on the same seeded inputs.

Tolerances: fp32 mode -- logits within rtol 1e-4 (north_star), gradients 2e-3 of the tensor's scale
(different summation order through 23 layers); fp16 mode -- stated per test.
"""
import os

import numpy as np
import pytest
import torch

from oracle import recipe, ref_unet

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X: torch.cuda.is_available() is False")


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def max_err_scaled(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def test_library_loaded_in_process():
    _need_gpu()
    from umi import lib
    assert lib.fn("umi_arch")() == b"gfx950"
    maps = open("/proc/self/maps").read()
    assert "libunetmi.so" in maps


# ---- blocks vs reference outputs -----------------------------------------------------------
def _load_block(g, prefix, mod, seed):
    mod.load_state_dict(recipe.fill_state_dict(mod.state_dict(), seed=seed))
    return mod.to(DEV)


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-5), ("fp16", 6e-3)])
def test_double_conv_block(golden_dir, dtype, tol):
    _need_gpu()
    import Model
    g = np.load(os.path.join(golden_dir, "unet_blocks.npz"))
    dc = _load_block(g, "dc", Model.DoubleConv(3, 8, compute_dtype=dtype), 1)
    x = torch.from_numpy(g["dc_x"]).to(DEV).requires_grad_(True)
    dc.train()
    y = dc(x)
    y.backward(torch.from_numpy(g["dc_gy"]).to(DEV))
    assert max_err_scaled(y, torch.from_numpy(g["dc_train_y"])) < tol
    assert rel_err(x.grad, torch.from_numpy(g["dc_train_gx"])) < 30 * tol
    for k, p in dc.named_parameters():
        assert rel_err(p.grad, torch.from_numpy(g["dc_grad." + k])) < 30 * tol, k
    for k, b in dc.named_buffers():
        ref = torch.from_numpy(g["dc_buf." + k])
        if "num_batches" in k:
            assert int(b) == int(ref)
        else:
            assert max_err_scaled(b, ref) < max(tol, 1e-5), k
    dc.eval()
    with torch.no_grad():
        ye = dc(x.detach())
    assert max_err_scaled(ye, torch.from_numpy(g["dc_eval_y"])) < tol


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-5), ("fp16", 6e-3)])
def test_down_up_out_blocks(golden_dir, dtype, tol):
    _need_gpu()
    import Model
    g = np.load(os.path.join(golden_dir, "unet_blocks.npz"))
    dn = _load_block(g, "down", Model.Down(8, 16, compute_dtype=dtype), 2).train()
    x = torch.from_numpy(g["down_x"]).to(DEV).requires_grad_(True)
    y = dn(x)
    y.backward(torch.from_numpy(g["down_gy"]).to(DEV))
    assert max_err_scaled(y, torch.from_numpy(g["down_y"])) < tol
    assert rel_err(x.grad, torch.from_numpy(g["down_gx"])) < 30 * tol
    for k, p in dn.named_parameters():
        assert rel_err(p.grad, torch.from_numpy(g["down_grad." + k])) < 30 * tol, k

    up = _load_block(g, "up", Model.Up(16, 8, compute_dtype=dtype), 3).train()
    x1 = torch.from_numpy(g["up_x1"]).to(DEV).requires_grad_(True)
    x2 = torch.from_numpy(g["up_x2"]).to(DEV).requires_grad_(True)
    y = up(x1, x2)                                   # odd skip size: exercises the F.pad path
    y.backward(torch.from_numpy(g["up_gy"]).to(DEV))
    assert max_err_scaled(y, torch.from_numpy(g["up_y"])) < tol
    assert rel_err(x1.grad, torch.from_numpy(g["up_gx1"])) < 30 * tol
    assert rel_err(x2.grad, torch.from_numpy(g["up_gx2"])) < 30 * tol
    for k, p in up.named_parameters():
        assert rel_err(p.grad, torch.from_numpy(g["up_grad." + k])) < 30 * tol, k

    oc = _load_block(g, "outc", Model.OutConv(8, 3, compute_dtype=dtype), 4)
    y = oc(torch.from_numpy(g["outc_x"]).to(DEV))
    assert max_err_scaled(y, torch.from_numpy(g["outc_y"])) < tol


# ---- whole network -------------------------------------------------------------------------------
def _oracle_run(g, steps):
    cin, ncls, feat = int(g["cin"]), int(g["ncls"]), int(g["feat"])
    B, H, W, seed = int(g["B"]), int(g["H"]), int(g["W"]), int(g["seed"])
    m = ref_unet.RefUNet(cin, ncls, feat, False)
    m.load_state_dict(recipe.fill_state_dict(m.state_dict(), seed=seed))
    x, lab = recipe.synthetic_batch(B, cin, H, W, ncls, seed=seed)
    return m, x, lab


@pytest.mark.parametrize("name", ["unet_1_2_8", "unet_3_4_8"])
def test_unet_fp32_parity(golden_dir, name):
    """fp32 HIP path: logits rtol 1e-4 vs the REFERENCE's logits, argmax identical off near-ties,
    loss/grads/3 SGD steps/eval-mode logits vs the oracle."""
    _need_gpu()
    import Model
    import loss as L
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    ref, x, lab = _oracle_run(g, 3)
    ncls = int(g["ncls"])
    L.CLASS_NUMBER = ncls
    m = Model.UNet(int(g["cin"]), ncls, int(g["feat"]), False, compute_dtype="fp32")
    m.load_state_dict(ref.state_dict())
    m.to(DEV).train()
    ref.train()
    xd, labd = x.to(DEV), lab.to(DEV)
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    ropt = torch.optim.SGD(ref.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    for step in range(3):
        logits = m(xd)
        loss = L.calc_loss(logits, labd, loss_type="dice_bce_mc")
        opt.zero_grad()
        loss.backward()
        rlogits = ref(x)
        rloss = ref_unet.dice_bce_mc(rlogits, lab, ncls)
        ropt.zero_grad()
        rloss.backward()
        if step == 0:
            gl = torch.from_numpy(g["logits"])
            np.testing.assert_allclose(logits.detach().cpu().numpy(), g["logits"], rtol=1e-4, atol=1e-4 * gl.abs().max().item())
            top2 = torch.topk(gl, 2, dim=1).values
            safe = (top2[:, 0] - top2[:, 1]) > 1e-4 * gl.abs().max()
            am = logits.argmax(1).cpu()
            assert (am == torch.from_numpy(g["argmax"]).long())[safe].all()
            assert abs(loss.item() - float(g["loss0"])) < 1e-5
        assert abs(loss.item() - float(g[f"loss{step}"])) < 5e-5, step
        for (k, p), (_, rp) in zip(m.named_parameters(), ref.named_parameters()):
            # step 0 is the parity statement; after an SGD step the two runs start from weights that differ by fp32
            # summation order, a few ReLU / max-pool masks flip, and a gradient moves by ~sqrt(flipped fraction)
            # (measured up to 4.6 % on the first layer at step 2 with a different split-K order): loose bound there,
            # tight bound on the loss trajectory above
            assert rel_err(p.grad, rp.grad) < (2e-3 if step == 0 else 8e-2), (step, k)
        opt.step()
        ropt.step()
    for k, v in m.state_dict().items():
        rv = ref.state_dict()[k]
        if "num_batches" in k:
            assert int(v) == int(rv) == 3
        else:
            assert rel_err(v.float(), rv.float()) < 1e-3, k
    m.eval()
    with torch.no_grad():
        ev = m(xd)
    np.testing.assert_allclose(ev.cpu().numpy(), g["eval_logits"], rtol=2e-3,
                               atol=2e-3 * float(np.abs(g["eval_logits"]).max()))


def _cos(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-30)).item()


@pytest.mark.parametrize("name", ["unet_1_2_8", "unet_3_4_8"])
def test_unet_fp16_close_to_oracle(golden_dir, name):
    """fp16-storage path (the benchmarked one).
    vs the fp32 reference: logits within 2e-2 of the logit scale, loss within 2e-3, every parameter
    gradient has cosine similarity > 0.9 (ReLU / max-pool masks flip under fp16 rounding of the stored
    activations, which moves gradients by ~sqrt(flipped fraction): inherent to fp16 storage).
    vs the oracle with the same fp16 rounding points (RefUNet(quant='fp16')), where the masks coincide:
    logits within 2e-3 of scale, gradients within 3e-2 relative L2."""
    _need_gpu()
    import Model
    import loss as L
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    ref, x, lab = _oracle_run(g, 1)
    ncls = int(g["ncls"])
    L.CLASS_NUMBER = ncls
    m = Model.UNet(int(g["cin"]), ncls, int(g["feat"]), False, compute_dtype="fp16")
    m.load_state_dict(ref.state_dict())
    m.to(DEV).train()
    logits = m(x.to(DEV))
    loss = L.calc_loss(logits, lab.to(DEV), loss_type="dice_bce_mc")
    loss.backward()
    assert logits.dtype == torch.float32
    assert max_err_scaled(logits, torch.from_numpy(g["logits"])) < 2e-2
    assert abs(loss.item() - float(g["loss0"])) < 2e-3
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    ref.train()
    ref_unet.dice_bce_mc(ref(x), lab, ncls).backward()
    cos = {k: _cos(p.grad, rp.grad) for (k, p), (_, rp) in zip(m.named_parameters(), ref.named_parameters())}
    assert min(cos.values()) > 0.9, min(cos.items(), key=lambda kv: kv[1])

    q = ref_unet.RefUNet(int(g["cin"]), ncls, int(g["feat"]), False, quant="fp16")
    q.load_state_dict(ref.state_dict())
    q.train()
    ql = q(x)
    ref_unet.dice_bce_mc(ql, lab, ncls).backward()
    assert max_err_scaled(logits, ql) < 1e-2
    errs = {k: rel_err(p.grad, qp.grad) for (k, p), (_, qp) in zip(m.named_parameters(), q.named_parameters())}
    # Rounding chaos floor: the quantised oracle against ITSELF with 1e-7 relative noise (summation-order
    # size) injected before each fp16 rounding.  The kernel path must not be further away than that.
    qn = ref_unet.RefUNet(int(g["cin"]), ncls, int(g["feat"]), False, quant="fp16")
    qn.load_state_dict(ref.state_dict())
    qn.train()
    qn.noise = 1e-7
    ref_unet.dice_bce_mc(qn(x), lab, ncls).backward()
    floor = {k: rel_err(p.grad, qp.grad) for (k, p), (_, qp) in zip(qn.named_parameters(), q.named_parameters())}
    med, med_floor = float(np.median(list(errs.values()))), float(np.median(list(floor.values())))
    print(f"fp16 grads vs quantised oracle: median rel-L2 {med:.3f} worst {max(errs.values()):.3f} | oracle self-noise "
          f"floor: median {med_floor:.3f} worst {max(floor.values()):.3f} | worst cosine vs fp32 {min(cos.values()):.3f}")
    # (the worst single tensor of one chaotic sample scatters more than the median: 2x its floor; with the 8..64-channel
    #  layers of these small fixtures on the matrix-core kernels the measured worst case is 1.7x)
    assert med < 1.5 * med_floor + 0.01 and max(errs.values()) < 2.0 * max(floor.values()) + 0.02


@pytest.mark.parametrize("cin,ncls", [(1, 2), (3, 4)])
def test_unet_fp16_feat64_benchmark_widths(cin, ncls):
    """The benchmarked path at the benchmark's channel widths: UNet(1,2,64) / UNet(3,4,64) in fp16 storage, 2 x C x 64 x 64
    (64 .. 1024 channels: every tile configuration of the matrix-core kernels, no layer on the generic VALU kernels --
    asserted through UMI_TRACE_GENERIC in a child process, tools/check_fp16_feat64.py) against the CPU oracle in fp32 and
    with the same fp16 rounding points:
      * logits within 2e-2 of the logit scale (fp32 oracle) and 1e-2 (fp16 oracle); loss within 5e-3;
      * argmax masks identical on every pixel whose top-2 margin exceeds 4x the worst logit error (near-ties may flip);
      * gradients: finite; cosine vs the fp32 oracle > 0.9 for every tensor; relative L2 vs the fp16 oracle bounded by
        the oracle's own rounding-chaos floor (median < 1.5x floor + 0.01, worst < 2x worst floor + 0.02) and by an
        absolute 0.5 for the worst tensor / 0.25 for the median (measured: median 0.16 against a floor of 0.12).
    The measured values of one such run are committed as profiles/r02_fp16_feat64_parity.json."""
    _need_gpu()
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UMI_TRACE_GENERIC="1")
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "check_fp16_feat64.py"), str(cin), str(ncls)],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "[umi generic" not in r.stderr, [l for l in r.stderr.splitlines() if "[umi generic" in l][:8]
    m = json.loads([l for l in r.stdout.splitlines() if l.startswith("FP16_FEAT64 ")][-1][len("FP16_FEAT64 "):])
    print(m)
    assert m["grads_finite"]
    assert m["logits_max_err_over_scale_vs_fp32_oracle"] < 2e-2 and m["logits_max_err_over_scale_vs_fp16_oracle"] < 1e-2
    assert abs(m["loss"] - m["loss_fp32_oracle"]) < 5e-3 and abs(m["loss"] - m["loss_fp16_oracle"]) < 5e-3
    assert m["argmax_mismatch_clear"] == 0 and m["pixels_clear_of_near_ties"] > 0.5 * m["pixels"]
    gq, fl = m["grad_rel_l2_vs_fp16_oracle"], m["grad_rel_l2_fp16_oracle_self_noise_floor"]
    assert gq["median"] < 1.5 * fl["median"] + 0.01 and gq["worst"] < 2.0 * fl["worst"] + 0.02, (gq, fl)
    assert gq["median"] < 0.25 and gq["worst"] < 0.5, gq
    assert m["grad_cosine_vs_fp32_oracle"]["worst"] > 0.9, m["grad_cosine_vs_fp32_oracle"]


def test_unet_config1_scale_fp32(golden_dir):
    """Config 1 shape (UNet(1,2,64), B=2, 256x256): logits signature + argmax mask vs the reference."""
    _need_gpu()
    import Model
    import loss as L
    from tests.test_oracle_golden import sig
    g = np.load(os.path.join(golden_dir, "unet_c1.npz"))
    L.CLASS_NUMBER = 2
    m = Model.UNet(1, 2, 64, False, compute_dtype="fp32")
    m.load_state_dict(recipe.fill_state_dict(m.state_dict(), seed=int(g["seed"])))
    m.to(DEV).train()
    x, lab = recipe.synthetic_batch(2, 1, 256, 256, 2, seed=int(g["seed"]))
    logits = m(x.to(DEV))
    loss = L.calc_loss(logits, lab.to(DEV), loss_type="dice_bce_mc")
    loss.backward()
    s = sig(logits.cpu())
    np.testing.assert_allclose(s[[0, 2]], g["logits_sig"][[0, 2]], rtol=1e-4)
    np.testing.assert_allclose(s[3:], g["logits_sig"][3:], rtol=1e-4, atol=1e-4 * s[0] / 300)
    assert abs(loss.item() - float(g["loss0"])) < 1e-5
    am = logits.argmax(1).cpu().numpy().astype(np.uint8)
    mism = int((am != g["argmax"]).sum())
    # near-tie pixels (|top-2 margin| below the 1e-4 tolerance band) may flip; count and bound them
    top2 = torch.topk(logits.detach(), 2, dim=1).values
    near = int(((top2[:, 0] - top2[:, 1]) < 1e-4 * logits.abs().max()).sum())
    assert mism <= near, (mism, near)
    for k, p in m.named_parameters():
        np.testing.assert_allclose(p.grad.double().norm().item(), g["grad_sig." + k][0], rtol=5e-3)


def test_forward_is_deterministic():
    _need_gpu()
    import Model
    torch.manual_seed(1)
    m = Model.UNet(1, 2, 8, compute_dtype="fp16").to(DEV).train()
    x = torch.randn(2, 1, 64, 64, device=DEV)
    outs = []
    for _ in range(2):
        m.zero_grad()
        y = m(x)
        y.square().mean().backward()
        outs.append((y.detach().clone(), [p.grad.clone() for p in m.parameters()]))
    assert torch.equal(outs[0][0], outs[1][0])
    assert all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))


def test_grad_reducer_sink_path_matches_plain_backward():
    """Data-parallel sink (flat buckets written directly by the wgrad kernels, umi/ddp.py) with world size 1:
    gradients must equal the plain tape's bit for bit, and land in the reducer's buckets."""
    _need_gpu()
    import Model
    from umi import ddp
    torch.manual_seed(3)
    m = Model.UNet(1, 2, 8, compute_dtype="fp16").to(DEV).train()
    x = torch.randn(2, 1, 32, 32, device=DEV)
    m(x).square().mean().backward()
    plain = [p.grad.clone() for p in m.parameters()]
    m.zero_grad()
    red = ddp.GradReducer(m, world_size=1, bucket_mb=0.5)
    assert len(red.buckets) > 1
    m(x).square().mean().backward()
    red.sync()
    for p, g in zip(m.parameters(), plain):
        assert torch.equal(p.grad, g)
        assert torch.equal(red.buffer_for(p), g)


def test_grad_reducer_sink_path_matches_plain_backward_transunet():
    """The same for the TransUNet tape: its grouped end-of-backward launches (token-linear weight
    gradients per shape, bias column sums, GroupNorm / LayerNorm parameter gradients, the StdConv2d standardisation backward)
    stay enabled under a sink and write straight into the bucket slots; .grad IS the slot (no autograd copy).  Launch counts
    of the full R50-ViT-B/16 with and without the sink: tools/check_tu_sink.py (770 vs 795 per step)."""
    _need_gpu()
    from oracle import ref_transunet
    from tests.test_gpu_transunet import product_config
    from TransUnet.vit_seg_modeling import VisionTransformer
    from umi import ddp
    torch.manual_seed(3)
    cfg = ref_transunet.small_config(2)
    m = VisionTransformer(product_config(cfg, 64), img_size=64, num_classes=2, compute_dtype="fp16").to(DEV).train()
    x = torch.randn(2, 1, 64, 64, device=DEV)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m(x).square().mean().backward()
    plain = [p.grad.clone() for p in m.parameters()]
    m.zero_grad(set_to_none=True)
    m.load_state_dict(sd)
    red = ddp.GradReducer(m, world_size=1, bucket_mb=0.5)
    assert len(red.buckets) > 1
    m(x).square().mean().backward()
    red.sync()
    for (k, p), g in zip(m.named_parameters(), plain):
        assert p.grad.data_ptr() == red.buffer_for(p).data_ptr(), k
        assert torch.equal(p.grad, g), k


def test_fused_dgrad_bn_reduction_matches_separate_kernels(monkeypatch):
    """DoubleConv's second conv: its data-gradient kernel also produces stage 1 of the first layer's BatchNorm backward
    reduction (umi_conv_dgrad_bnred).  Gradients must agree with the separate-kernel path (same values, other sum order)."""
    _need_gpu()
    import Model
    torch.manual_seed(4)
    m = Model.UNet(1, 2, 64, compute_dtype="fp16").to(DEV).train()
    x = torch.randn(2, 1, 96, 64, device=DEV)
    grads = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("UMI_NO_BNRED_FUSION", mode)
        m.zero_grad()
        m(x).square().mean().backward()
        grads[mode] = [p.grad.clone() for p in m.parameters()]
    # the two paths add the same fp32 terms in a different order; a last-bit change of a channel sum flips fp16 roundings
    # of that layer's dz, which later layers see as ~1e-3 relative noise (measured 1.6e-3 worst)
    worst = max(rel_err(a, b) for a, b in zip(grads["0"], grads["1"]))
    assert worst < 5e-3, worst
    # kernel level, odd sizes (partial tiles): the fused epilogue's sums == the stand-alone reduction's, to fp32 accuracy
    from umi import ops
    g = torch.Generator().manual_seed(9)
    N, H, W, Ci, Co = 2, 21, 37, 64, 128                 # data gradient: dy has Ci channels, da has Co
    dy = torch.randn(N, H, W, Ci, generator=g).half().to(DEV)
    ybn = torch.randn(N, H, W, Co, generator=g).half().to(DEV)
    wt = (torch.randn(Ci, Co, 3, 3, generator=g) * 0.05).to(DEV)     # forward conv Co -> Ci, so its dgrad maps Ci -> Co
    tx = torch.stack([torch.randn(Co, generator=g) * 0.1, torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.2,
                      torch.zeros(Co)], 1).contiguous().to(DEV)
    rstd = (torch.rand(Co, generator=g) + 0.5).to(DEV)
    da1, da2 = (torch.empty(N, H, W, Co, device=DEV, dtype=torch.float16) for _ in range(2))
    wp = ops.pack_conv_dgrad(wt, torch.float16, k8=True)
    part = ops.conv_dgrad_bnred(dy, wp, da1, ybn, tx, rstd)
    assert part is not None
    ops.conv_fwd(dy, None, lambda l: wp, None, da2, 3, 3, 1, 1)
    assert torch.equal(da1, da2)
    s_f = ops.bn_bwd(da1, ybn, tx, rstd, partials=part)
    s_r = ops.bn_bwd(da2, ybn, tx, rstd)
    for a, b in zip(s_f, s_r):
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()


def test_graphed_step_matches_eager_trajectory():
    """umi.graphs.GraphedStep: a training step captured in a HIP graph and replayed must follow the eager trajectory
    (same kernels, same order: bit-identical losses and weights).  Runs in a child process: stream capture is sensitive to
    autograd state left by earlier eager steps of the same process, and a runtime crash must not take the suite down."""
    _need_gpu()
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_graphed_step.py")
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "GRAPHED_STEP_OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])


def test_dropout_kernel_with_consumer_transform_and_unet_dropout_mode():
    """U-Net `dropout=True` (reference Model.py:34-41,59-61,79-83): the dropout kernel applies the producer's lazy BN+ReLU,
    scales the kept values by 1/(1-p) and its backward reuses the mask; the network trains with it and ignores it in eval."""
    _need_gpu()
    from umi import ops_tu
    g = torch.Generator().manual_seed(5)
    N, H, W, C, p = 2, 9, 7, 16, 0.3
    for dt in (torch.float16, torch.float32):
        x = torch.randn(N, H, W, C, generator=g).to(dt).to(DEV)
        tx = torch.stack([torch.zeros(C), torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2, torch.zeros(C)], 1)
        y = torch.empty_like(x)
        mask = torch.empty(x.numel(), dtype=torch.uint8, device=DEV)
        ops_tu.dropout(x, y, mask, False, p, 1234, tx.to(DEV).contiguous())
        act = torch.relu(x.float().cpu() * tx[:, 1] + tx[:, 2])
        keep = mask.cpu().view(N, H, W, C).bool()
        exp = torch.where(keep, act / (1 - p), torch.zeros(()))
        assert (y.float().cpu() - exp).abs().max().item() <= 2e-3 * exp.abs().max().item()
        assert 0.55 < keep.float().mean().item() < 0.85
        gy = torch.randn(N, H, W, C, generator=g).to(dt).to(DEV)
        dx = torch.empty_like(gy)
        ops_tu.dropout(gy, dx, mask, True, p, 0)
        exp_dx = torch.where(keep, gy.float().cpu() / (1 - p), torch.zeros(()))
        assert (dx.float().cpu() - exp_dx).abs().max().item() <= 2e-3 * exp_dx.abs().max().item()
    import Model
    torch.manual_seed(11)
    m = Model.UNet(1, 2, 8, True, True, 0.25, compute_dtype="fp16").to(DEV)
    assert "down1.maxpool_conv.2.double_conv.0.weight" in m.state_dict()        # reference key layout with dropout=True
    xin = torch.randn(2, 1, 32, 32, device=DEV)
    m.train()
    a = m(xin)
    a.square().mean().backward()
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in m.parameters())
    # The mask stream = a per-model base seed drawn once from torch's generator (first training forward) + a device-side step
    # counter: a RUN is reproducible under torch.manual_seed, every step draws fresh masks (also when replayed from a HIP
    # graph), and different seeds give different streams.
    sd = m.state_dict()

    def run(seed):
        torch.manual_seed(seed)
        mm = Model.UNet(1, 2, 8, True, True, 0.25, compute_dtype="fp16").to(DEV)
        mm.load_state_dict(sd)
        mm.train()
        with torch.no_grad():
            return mm(xin), mm(xin)
    a1, a2 = run(1)
    b1, b2 = run(1)
    c1, _ = run(2)
    assert torch.equal(a1, b1) and torch.equal(a2, b2) and not torch.equal(a1, a2) and not torch.equal(a1, c1)
    m.eval()
    with torch.no_grad():
        e1, e2 = m(xin), m(xin)
    assert torch.equal(e1, e2)


def test_fused_optimizer_and_pack_cache_leave_the_training_trajectory_unchanged(monkeypatch):
    """umi.optim.SGD + the model's PackCache (one update launch + one re-pack launch per step) against
    torch.optim.SGD + per-use packing, same seeds: same losses and weights up to fp32 rounding of the update."""
    import Model
    import loss as L
    from umi import optim as uo
    L.CLASS_NUMBER = 2

    def run(fused):
        monkeypatch.setenv("UMI_NO_PACK_CACHE", "0" if fused else "1")
        torch.manual_seed(3)
        m = Model.UNet(1, 2, 16, compute_dtype=torch.float16).to("cuda").train()
        opt = (uo.SGD if fused else torch.optim.SGD)(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        g = torch.Generator().manual_seed(11)
        losses = []
        for _ in range(4):
            x = torch.randn(2, 1, 64, 64, generator=g).cuda()
            y = torch.randint(0, 2, (2, 64, 64), generator=g).float().cuda()
            opt.zero_grad(set_to_none=True)
            l = L.calc_loss(m(x), y, loss_type="dice_bce_mc")
            l.backward()
            opt.step()
            losses.append(float(l.detach()))
        if fused:
            assert len(m._umi_pack_cache.ents) >= 40      # every conv weight in both layouts
        return losses, [p.detach().float().cpu() for p in m.parameters()]

    l1, p1 = run(True)
    l0, p0 = run(False)
    assert max(abs(a - b) for a, b in zip(l1, l0)) < 2e-5, (l1, l0)
    for a, b in zip(p1, p0):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("name,pcls,rcls", [("unet_multitask_1_2_8_s10", "UNet_multitask", "RefUNetMultitask"),
                                            ("unet_attention_1_2_8_s16", "UNet_attention", "RefUNetAttention")])
def test_variant_unpicked_seeds(golden_dir, name, pcls, rcls):
    """The variant fixtures of the two tests below were generated on seeds screened to be free of ReLU near-ties on both
    sides, which is what lets them carry a 2e-3 per-tensor gradient bar.  These two seeds were NOT screened (seed 10 is one
    where the reference's own fp32 and fp64 runs disagree on a mask; 16 is one where this build's summation order does).
    Stated bound for such seeds, fp32 path: logits still within 1e-4 of the reference's (fixture), loss within 1e-4; gradients
    within 3e-2 relative L2 for EVERY tensor and 2e-3 for the median tensor (one flipped ReLU element moves the tensors
    downstream of it by 1e-3 .. 1e-2)."""
    _need_gpu()
    import Model
    import loss as L
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cin, ncls, feat = int(g["cin"]), int(g["ncls"]), int(g["feat"])
    B, H, W, seed = int(g["B"]), int(g["H"]), int(g["W"]), int(g["seed"])
    ref = getattr(ref_unet, rcls)(cin, ncls, feat, False)
    ref.load_state_dict(recipe.fill_state_dict(ref.state_dict(), seed=seed))
    x, lab1 = recipe.synthetic_batch(B, cin, H, W, ncls, seed=seed)
    _, lab2 = recipe.synthetic_batch(B, cin, H, W, ncls, seed=seed + 100)
    L.CLASS_NUMBER = ncls
    m = getattr(Model, pcls)(cin, ncls, feat, False, compute_dtype="fp32")
    m.load_state_dict(ref.state_dict())
    m.to(DEV).train()
    ref.train()
    out, rout = m(x.to(DEV)), ref(x)
    if isinstance(out, tuple):
        loss = L.calc_loss(out[0], lab1.to(DEV), loss_type="dice_bce_mc") + L.calc_loss(out[1], lab2.to(DEV), loss_type="dice_bce_mc")
        rloss = ref_unet.dice_bce_mc(rout[0], lab1, ncls) + ref_unet.dice_bce_mc(rout[1], lab2, ncls)
        pairs = [(out[0], g["logits1"]), (out[1], g["logits2"])]
    else:
        loss = L.calc_loss(out, lab1.to(DEV), loss_type="dice_bce_mc")
        rloss = ref_unet.dice_bce_mc(rout, lab1, ncls)
        pairs = [(out, g["logits"])]
    loss.backward()
    rloss.backward()
    for o, gl in pairs:
        np.testing.assert_allclose(o.detach().cpu().numpy(), gl, rtol=1e-4, atol=1e-4 * float(np.abs(gl).max()))
    assert abs(loss.item() - float(g["loss0"])) < 1e-4
    errs = {k: rel_err(p.grad, rp.grad) for (k, p), (_, rp) in zip(m.named_parameters(), ref.named_parameters())
            if rp.grad.abs().max() > 1e-6}
    print(name, "grad rel-L2: median", float(np.median(list(errs.values()))), "worst", max(errs.items(), key=lambda kv: kv[1]))
    assert max(errs.values()) < 3e-2, max(errs.items(), key=lambda kv: kv[1])
    assert float(np.median(list(errs.values()))) < 2e-3


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_unet_multitask_parity(golden_dir, dtype):
    """`Model.UNet_multitask` (two decoders on one tape, encoder gradients summed) against the
    REFERENCE's logits (fixture) and the oracle's loss / gradients / 3 SGD steps.  fp32: the 1e-4 logits bar;
    fp16: the storage-rounding bar of test_unet_fp16_close_to_oracle."""
    _need_gpu()
    import Model
    import loss as L
    g = np.load(os.path.join(golden_dir, "unet_multitask_1_2_8.npz"))
    cin, ncls, feat = int(g["cin"]), int(g["ncls"]), int(g["feat"])
    B, H, W, seed = int(g["B"]), int(g["H"]), int(g["W"]), int(g["seed"])
    ref = ref_unet.RefUNetMultitask(cin, ncls, feat, False)
    ref.load_state_dict(recipe.fill_state_dict(ref.state_dict(), seed=seed))
    x, lab1 = recipe.synthetic_batch(B, cin, H, W, ncls, seed=seed)
    _, lab2 = recipe.synthetic_batch(B, cin, H, W, ncls, seed=seed + 100)
    L.CLASS_NUMBER = ncls
    m = Model.UNet_multitask(cin, ncls, feat, False, compute_dtype=dtype)
    assert list(m.state_dict().keys()) == list(ref.state_dict().keys())
    m.load_state_dict(ref.state_dict())
    m.to(DEV).train()
    ref.train()
    xd, l1d, l2d = x.to(DEV), lab1.to(DEV), lab2.to(DEV)
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    ropt = torch.optim.SGD(ref.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    tol = 1e-4 if dtype == "fp32" else 2e-2
    for step in range(3):
        o1, o2 = m(xd)
        loss = L.calc_loss(o1, l1d, loss_type="dice_bce_mc") + L.calc_loss(o2, l2d, loss_type="dice_bce_mc")
        opt.zero_grad()
        loss.backward()
        r1, r2 = ref(x)
        rloss = ref_unet.dice_bce_mc(r1, lab1, ncls) + ref_unet.dice_bce_mc(r2, lab2, ncls)
        ropt.zero_grad()
        rloss.backward()
        if step == 0:
            for o, key in ((o1, "logits1"), (o2, "logits2")):
                np.testing.assert_allclose(o.detach().cpu().numpy(), g[key], rtol=tol, atol=tol * float(np.abs(g[key]).max()))
        assert abs(loss.item() - float(g[f"loss{step}"])) < (1e-4 if dtype == "fp32" else 2e-2), step
        for (k, p), (_, rp) in zip(m.named_parameters(), ref.named_parameters()):
            if dtype == "fp32":
                assert rel_err(p.grad, rp.grad) < (2e-3 if step == 0 else 8e-2), (step, k)
            elif step == 0:
                # fp16 storage flips ReLU / pool masks: same per-tensor bar as test_unet_fp16_close_to_oracle
                assert _cos(p.grad, rp.grad) > 0.9, (k, _cos(p.grad, rp.grad))
        opt.step()
        ropt.step()
    if dtype == "fp32":
        m.eval()
        with torch.no_grad():
            e1, e2 = m(xd)
        for e, key in ((e1, "eval_logits1"), (e2, "eval_logits2")):
            np.testing.assert_allclose(e.cpu().numpy(), g[key], rtol=2e-3, atol=2e-3 * float(np.abs(g[key]).max()))


def test_predict_mask_matches_reference_eval_argmax(golden_dir):
    """eval-mode forward + argmax kernel against the argmax of the oracle's eval-mode logits on the
    same weights (the oracle's eval logits are pinned to the reference's by test_oracle_golden), identical off near-ties;
    the weight packings are cached across calls (no re-pack when nothing changed)."""
    _need_gpu()
    import Model
    from umi import infer
    g = np.load(os.path.join(golden_dir, "unet_3_4_8.npz"))
    ref, x, lab = _oracle_run(g, 0)
    m = Model.UNet(int(g["cin"]), int(g["ncls"]), int(g["feat"]), False, compute_dtype="fp32")
    m.load_state_dict(ref.state_dict())
    m.to(DEV).train()
    ref.eval()
    with torch.no_grad():
        want_logits = ref(x)
    mask = infer.predict_mask(m, x)
    assert m.training                            # mode restored
    assert mask.dtype == torch.uint8 and tuple(mask.shape) == (x.shape[0], x.shape[2], x.shape[3])
    top2 = torch.topk(want_logits, 2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-3 * want_logits.abs().max()
    assert safe.float().mean() > 0.9
    assert torch.equal(mask.cpu()[safe], want_logits.argmax(1).to(torch.uint8)[safe])
    vers = {k: e.ver for k, e in m._umi_pack_cache.ents.items()}
    mask2 = infer.predict_mask(m, x)
    assert torch.equal(mask, mask2) and vers == {k: e.ver for k, e in m._umi_pack_cache.ents.items()}


@pytest.mark.parametrize("cls,cfg", [("UNet", (1, 2, 32)), ("UNet", (3, 4, 64)), ("UNet_attention", (1, 2, 32))])
def test_eval_forward_with_bn_relu_on_store(monkeypatch, cls, cfg):
    """SURVEY 8(f) rank 4 (reference evaluation loop test_mc3serousv5.py:877-887 over Model.py:7-26): in eval mode the fp16
    path runs every matrix-core 3x3 convolution with its own BatchNorm (running statistics) + ReLU applied in the epilogue
    (`umi_conv3x3_fwd_act`): tensors are stored activated, no statistics, no transform in the consumers.  Checked against the
    same network with the epilogue switched off (UMI_NO_EVAL_FOLD=1: lazy consumer-side transform, the round-1 eval path)
    and against the fp32 oracle: logits within 1e-2 of scale of both, argmax identical off near-ties."""
    _need_gpu()
    import Model
    cin, ncls, feat = cfg
    torch.manual_seed(5)
    ref = getattr(ref_unet, {"UNet": "RefUNet", "UNet_attention": "RefUNetAttention"}[cls])(cin, ncls, feat, False)
    ref.load_state_dict(recipe.fill_state_dict(ref.state_dict(), seed=50 + feat))
    # non-trivial running statistics: a few training steps of the oracle
    x, _ = recipe.synthetic_batch(2, cin, 64, 80, ncls, seed=50 + feat)
    ref.train()
    with torch.no_grad():
        for _ in range(3):
            ref(x + 0.1 * torch.randn_like(x))
    ref.eval()
    with torch.no_grad():
        want = ref(x)
    m = getattr(Model, cls)(cin, ncls, feat, False, compute_dtype="fp16")
    m.load_state_dict(ref.state_dict())
    m.to(DEV).eval()
    with torch.no_grad():
        monkeypatch.setenv("UMI_NO_EVAL_FOLD", "1")
        lazy = m(x.to(DEV)).float().cpu()
        monkeypatch.setenv("UMI_NO_EVAL_FOLD", "0")
        fold = m(x.to(DEV)).float().cpu()
    scale = want.abs().max().item()
    assert (fold - lazy).abs().max().item() < 1e-2 * scale
    assert (fold - want).abs().max().item() < 1e-2 * scale and (lazy - want).abs().max().item() < 1e-2 * scale
    assert not torch.equal(fold, lazy) or feat < 16           # (the two paths round at different points)
    top2 = torch.topk(want, 2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 4e-2 * scale
    assert safe.float().mean() > 0.5 and torch.equal(fold.argmax(1)[safe], want.argmax(1)[safe])


def test_attention_dgrad_accumulate_is_bit_identical(monkeypatch):
    """Tensors with two consumers (attention gate: g and x) get their second gradient contribution added by the data-gradient
    kernel itself (UMI_CONV_ACCUMULATE); with the knob off the tape computes into a fresh tensor and adds.  Same bits."""
    _need_gpu()
    import Model
    import loss as L
    from umi import ops
    L.CLASS_NUMBER = 2
    torch.manual_seed(5)
    m = Model.UNet_attention(1, 2, 64, False, compute_dtype="fp16").to(DEV).train()
    x = torch.randn(2, 1, 64, 64, device=DEV)
    lab = torch.randint(0, 2, (2, 64, 64), device=DEV).float()
    calls = []
    real = ops.conv_fwd

    def spy(*a, **k):
        calls.append(k.get("flags", 0))
        return real(*a, **k)
    monkeypatch.setattr(ops, "conv_fwd", spy)
    grads = {}
    for knob in ("0", "1"):
        monkeypatch.setenv("UMI_NO_DGRAD_ACCUMULATE", knob)
        calls.clear()
        m.zero_grad(set_to_none=True)
        L.calc_loss(m(x), lab, loss_type="dice_bce_mc").backward()
        n_acc = sum(1 for f in calls if f & 8)
        assert n_acc == (8 if knob == "0" else 0), (knob, n_acc)      # 4 gates x (ConvT data gradient, W_x data gradient)
        grads[knob] = {k: p.grad.clone() for k, p in m.named_parameters()}
    for k in grads["0"]:
        assert torch.equal(grads["0"][k], grads["1"][k]), k


def _is_dead_bias(k):
    """Conv biases directly in front of a BatchNorm (attention gate W_q / W_x / psi): their gradient vanishes identically;
    the reference leaves ~1e-10 of rounding noise there, the HIP path writes exact zeros."""
    # (the gate's ConvTranspose2d bias is a per-channel constant in front of W_q's BatchNorm: cancelled the same way)
    return k.endswith((".W_q.0.bias", ".W_x.0.bias", ".psi.0.bias")) or ("attenion" in k and k.endswith(".up.bias"))


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_unet_attention_parity(golden_dir, dtype):
    """`Model.UNet_attention` (attention gates: ConvT, 1x1 conv + BatchNorm branches, fused add-ReLU and
    sigmoid-gate kernels) against the REFERENCE's logits (fixture) and the oracle's loss / gradients / running statistics /
    3 SGD steps / eval-mode logits."""
    _need_gpu()
    import Model
    import loss as L
    g = np.load(os.path.join(golden_dir, "unet_attention_1_2_8.npz"))
    cin, ncls, feat = int(g["cin"]), int(g["ncls"]), int(g["feat"])
    B, H, W, seed = int(g["B"]), int(g["H"]), int(g["W"]), int(g["seed"])
    ref = ref_unet.RefUNetAttention(cin, ncls, feat, False)
    ref.load_state_dict(recipe.fill_state_dict(ref.state_dict(), seed=seed))
    x, lab = recipe.synthetic_batch(B, cin, H, W, ncls, seed=seed)
    L.CLASS_NUMBER = ncls
    m = Model.UNet_attention(cin, ncls, feat, False, compute_dtype=dtype)
    assert list(m.state_dict().keys()) == list(ref.state_dict().keys())
    m.load_state_dict(ref.state_dict())
    m.to(DEV).train()
    ref.train()
    xd, labd = x.to(DEV), lab.to(DEV)
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    ropt = torch.optim.SGD(ref.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    tol = 1e-4 if dtype == "fp32" else 2e-2
    for step in range(3):
        logits = m(xd)
        loss = L.calc_loss(logits, labd, loss_type="dice_bce_mc")
        opt.zero_grad()
        loss.backward()
        rloss = ref_unet.dice_bce_mc(ref(x), lab, ncls)
        ropt.zero_grad()
        rloss.backward()
        if step == 0:
            np.testing.assert_allclose(logits.detach().cpu().numpy(), g["logits"], rtol=tol,
                                       atol=tol * float(np.abs(g["logits"]).max()))
        assert abs(loss.item() - float(g[f"loss{step}"])) < (1e-4 if dtype == "fp32" else 2e-2), step
        for (k, p), (_, rp) in zip(m.named_parameters(), ref.named_parameters()):
            if _is_dead_bias(k):
                # (the ConvTranspose2d bias gradient is a real column sum of fp16 values that cancels: ~1e-6 left in fp16)
                assert float(p.grad.abs().max()) < (1e-6 if dtype == "fp32" else 1e-4), k
                assert float(rp.grad.abs().max()) < 1e-6, k
            elif dtype == "fp32":
                assert rel_err(p.grad, rp.grad) < (2e-3 if step == 0 else 8e-2), (step, k)
        if dtype == "fp16" and step == 0:
            # fp16 storage against the fp32 oracle (no quantised restatement of the gates exists): masks flip and the gates'
            # tiny BatchNorm gradients (1 .. 8 numbers, sums with heavy cancellation) scatter, so the bar is the direction of
            # the whole gradient plus a per-tensor bar on the large majority (measured: global 0.96, 90 % of tensors > 0.9)
            live = [(p.grad.detach().float().cpu().flatten(), rp.grad.float().flatten())
                    for (k, p), (_, rp) in zip(m.named_parameters(), ref.named_parameters()) if not _is_dead_bias(k)]
            assert _cos(torch.cat([a for a, _ in live]), torch.cat([b for _, b in live])) > 0.93
            assert np.mean([_cos(a, b) > 0.9 for a, b in live]) > 0.8
        opt.step()
        ropt.step()
    if dtype == "fp32":
        for k, v in m.state_dict().items():
            rv = ref.state_dict()[k]
            if "num_batches" in k:
                assert int(v) == int(rv) == 3
            elif not _is_dead_bias(k):
                assert rel_err(v.float(), rv.float()) < 1e-3, k       # incl. running_mean of the biased 1x1 convs
    m.eval()
    ref.eval()
    with torch.no_grad():
        ev, rev = m(xd), ref(x)
    etol = 2e-3 if dtype == "fp32" else 5e-2
    np.testing.assert_allclose(ev.cpu().numpy(), rev.numpy(), rtol=etol, atol=etol * float(rev.abs().max()))
    if dtype == "fp32":
        np.testing.assert_allclose(ev.cpu().numpy(), g["eval_logits"], rtol=2e-3, atol=2e-3 * float(np.abs(g["eval_logits"]).max()))


def test_attention_block_odd_shapes_and_input_gradients():
    """Attention_block as a standalone module (reference Model.py:265-305) on a non-square map, gradients w.r.t. both
    inputs and all parameters against the oracle's functional restatement (fp32)."""
    _need_gpu()
    import Model
    ref = ref_unet.RefUNetAttention(1, 2, 8, False)
    ref.load_state_dict(recipe.fill_state_dict(ref.state_dict(), seed=4))
    node = ref.attenion3                                      # C_q = 64, C_x = 32, hidden 16
    blk = Model.Attention_block(64, 32, 16, compute_dtype="fp32")
    blk.load_state_dict(node.state_dict())
    blk.to(DEV).train()
    ref.train()
    gen = torch.Generator().manual_seed(12)
    q = torch.randn(2, 64, 5, 7, generator=gen, requires_grad=True)
    x = torch.randn(2, 32, 10, 14, generator=gen).relu_().requires_grad_(True)
    w = torch.randn(2, 32, 10, 14, generator=gen)
    qd, xd = q.detach().to(DEV).requires_grad_(True), x.detach().to(DEV).requires_grad_(True)
    out = blk(qd, xd)
    rout = ref._gate(q, x, node)
    np.testing.assert_allclose(out.detach().cpu().numpy(), rout.detach().numpy(), rtol=1e-4, atol=1e-5)
    (out * w.to(DEV)).sum().backward()
    (rout * w).sum().backward()
    assert rel_err(qd.grad, q.grad) < 1e-3 and rel_err(xd.grad, x.grad) < 1e-3
    for (k, p), (_, rp) in zip(blk.named_parameters(), node.named_parameters()):
        if _is_dead_bias("." + k) or k == "up.bias":
            assert float(p.grad.abs().max()) < 1e-4, k
        else:
            assert rel_err(p.grad, rp.grad) < 1e-3, k


def test_trainer_multitask_on_hip_model_follows_reference_run(golden_dir, tmp_path):
    """Product Trainer.multi_task_train (reference Trainer.py:831-992) driving the HIP `Model.UNet_multitask(1, 1, 8)` (one
    regression map per head, ReLU + 'mse' in the trainer) against the reference's own run of that loop (fixture)."""
    _need_gpu()
    import Model
    from torch.utils.data import DataLoader
    from Trainer import Trainer
    from tools.gen_golden import PairLabels, multitask_trainer_data
    g = np.load(os.path.join(golden_dir, "trainer_multitask.npz"))
    m = Model.UNet_multitask(1, 1, 8, False, compute_dtype="fp32")
    m.load_state_dict(recipe.fill_state_dict(m.state_dict(), seed=22))
    m.to(DEV)
    xs, l1, l2 = multitask_trainer_data()
    loaders = {"train": DataLoader(PairLabels(xs[:4], l1[:4], l2[:4]), batch_size=2, shuffle=False),
               "val": DataLoader(PairLabels(xs[4:], l1[4:], l2[4:]), batch_size=1)}
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    tr = Trainer(m, "multi_task", torch.cuda.FloatTensor, DEV, str(tmp_path), loaders, 2, opt, 25, 2, "mse", "mse",
                 lr_scheduler=True)
    tr.train()
    for mine, key in ((tr.train_loss_list, "train_loss"), (tr.val_loss_list, "val_loss"),
                      (tr.train_loss_list_1, "train_loss_1"), (tr.val_loss_list_2, "val_loss_2")):
        np.testing.assert_allclose(mine, g[key], rtol=2e-4, atol=2e-5)
    assert sorted(os.listdir(tmp_path / "models")) == list(g["files"])


@pytest.mark.parametrize("shape", [(1, 1, 70, 90), (3, 3, 33, 47), (2, 1, 16, 16)])
def test_unet_ragged_sizes_match_oracle(shape):
    """Spatial sizes that are not multiples of 16 (every level floors in the encoder and pads in the decoder, reference
    Model.py:69-73), batch 1 and 3, and the smallest map the 4-level network accepts (1x1 at the bottom; batch 2 because
    BatchNorm refuses a single value per channel in training mode, in the reference too): fp32 logits and all gradients vs
    the oracle."""
    _need_gpu()
    import Model
    import loss as L
    B, cin, H, W = shape
    ncls = 3
    L.CLASS_NUMBER = ncls
    ref = ref_unet.RefUNet(cin, ncls, 8, False)
    ref.load_state_dict(recipe.fill_state_dict(ref.state_dict(), seed=40 + H))
    x, lab = recipe.synthetic_batch(B, cin, H, W, ncls, seed=40 + H)
    m = Model.UNet(cin, ncls, 8, False, compute_dtype="fp32")
    m.load_state_dict(ref.state_dict())
    m.to(DEV).train()
    ref.train()
    logits = m(x.to(DEV))
    rl = ref(x)
    assert tuple(logits.shape) == (B, ncls, H, W)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), rl.detach().numpy(), rtol=1e-4,
                               atol=1e-4 * float(rl.detach().abs().max()))
    L.calc_loss(logits, lab.to(DEV), loss_type="dice_bce_mc").backward()
    ref_unet.dice_bce_mc(rl, lab, ncls).backward()
    errs = {k: rel_err(p.grad, rp.grad) for (k, p), (_, rp) in zip(m.named_parameters(), ref.named_parameters())}
    # (a ReLU input within rounding of zero can flip between two fp32 implementations; the bound
    #  leaves room for one such element in these small maps, a wrong pad / crop offset would be O(1))
    # (the 16x16 case has 1x1 .. 8x8 maps below the first level: a single flipped mask is a large share of such a tensor,
    #  and BatchNorm over 2 values per channel amplifies rounding by up to 1/sqrt(eps); measured 6e-2 there)
    tiny = H * W <= 256
    assert max(errs.values()) < (1e-1 if tiny else 3e-2), max(errs.items(), key=lambda kv: kv[1])
    assert float(np.median(list(errs.values()))) < (6e-2 if tiny else 2e-3)


# ---- a module applied twice in one tape (deferred gradient fills) ------------------------------------------
@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-3), ("fp16", 6e-2)])
def test_module_used_twice_in_one_tape_accumulates_both_gradients(dtype, tol):
    """The grouped end-of-backward launches WRITE their outputs; a second use of the same parameters must be added after
    them, not onto the still unfilled buffer.  Reference: torch autograd on CPU (weights shared by two applications)."""
    _need_gpu()
    import Model
    torch.manual_seed(5)
    dc = Model.DoubleConv(16, 16, compute_dtype=dtype)
    dc.load_state_dict(recipe.fill_state_dict(dc.state_dict(), seed=11))
    ref = torch.nn.Module()                     # plain PyTorch restatement of Model.DoubleConv (same state_dict keys)
    ref.double_conv = torch.nn.Sequential(
        torch.nn.Conv2d(16, 16, 3, padding=1, bias=False), torch.nn.BatchNorm2d(16), torch.nn.ReLU(),
        torch.nn.Conv2d(16, 16, 3, padding=1, bias=False), torch.nn.BatchNorm2d(16), torch.nn.ReLU())
    ref.load_state_dict(dc.state_dict())
    ref.forward = lambda v: ref.double_conv(v)
    dc = dc.to(DEV)

    class Twice(Model._UmiModule):
        def __init__(self, block):
            super().__init__()
            self.block = block
            self._compute_dtype = dtype

        def forward(self, x):
            return Model._run_tape(self, [x], lambda t, a: Model._build_double_conv(
                t, Model._build_double_conv(t, a, self.block), self.block))

    x = torch.randn(2, 16, 24, 40)
    gy = torch.randn(2, 16, 24, 40)
    m = Twice(dc).train()
    y = m(x.to(DEV))
    y.backward(gy.to(DEV))
    ref.train()
    yr = ref(ref(x))
    yr.backward(gy)
    assert max_err_scaled(y, yr) < tol
    got = dict(dc.named_parameters())
    for name, p in ref.named_parameters():
        assert rel_err(got[name].grad, p.grad) < tol, name


def test_fp16_path_trains_like_the_fp32_oracle():
    """the per-step fp16 gradient error (16-23 % rel-L2, bounded only against the quantised oracle's
    own noise floor) says nothing about TRAINING.  UNet(1,2,64) in fp16 storage with umi.optim.SGD against the fp32 CPU
    oracle with torch.optim.SGD, same weights, same 24 blob-structured batches (4 x 1 x 64 x 64; reference Trainer.py:697-727):
      * every step's loss within 2e-3 + 1 % of the oracle's (measured: <= 2e-4, profiles/r03_fp16_training_trajectory.json),
      * the loss falls by more than 5x over the run on both sides,
      * eval-mode IoU on a held-out batch within 0.01 of the oracle's and > 0.8; the two argmax masks agree on > 99 % of pixels."""
    _need_gpu()
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "check_fp16_training.py"), "24", "64", "4"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    m = json.loads([l for l in r.stdout.splitlines() if l.startswith("FP16_TRAINING ")][-1][len("FP16_TRAINING "):])
    print(m)
    lo, lh = m["loss_fp32_oracle"], m["loss_fp16_hip"]
    for s, (a, b) in enumerate(zip(lo, lh)):
        assert abs(a - b) <= 2e-3 + 0.01 * a, (s, a, b)
    assert lo[-1] < lo[0] / 5 and lh[-1] < lh[0] / 5
    assert abs(m["eval_iou_fp16_hip"] - m["eval_iou_fp32_oracle"]) < 0.01 and m["eval_iou_fp16_hip"] > 0.8
    assert m["eval_masks_agree"] > 0.99
