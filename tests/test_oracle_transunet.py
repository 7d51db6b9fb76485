"""Pins oracle/ref_transunet.py against outputs of the reference TransUNet (tests/golden/transunet_*.npz). CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import recipe, ref_transunet, ref_unet
from tests.test_oracle_golden import _sig_close, sig

CASES = {
    "transunet_small": (lambda: ref_transunet.small_config(2)),
    "transunet_small_rgb4": (lambda: ref_transunet.small_config(4)),
    "transunet_small_neg_gamma": (lambda: ref_transunet.small_config(2)),     # every 7th norm scale negative
    "transunet_r50_b16_224": (lambda: ref_transunet.r50_vit_b16_config(2, 3, dropout_rate=0.0)),
    # BASELINE configs[4] shape: 1,024 tokens, 127 -> 128 zero-pad of the 1/4-scale ResNet skip (resnet_skip.py:147-158)
    "transunet_r50_b16_512": (lambda: ref_transunet.r50_vit_b16_config(2, 3, dropout_rate=0.0)),
}


@pytest.mark.parametrize("name", list(CASES))
def test_transunet_oracle_matches_reference(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = CASES[name]()
    img, B, cin, seed = int(g["img"]), int(g["B"]), int(g["cin"]), int(g["seed"])
    torch.manual_seed(0)
    m = ref_transunet.RefTransUNet(cfg, img)
    sd = m.state_dict()
    assert len(sd) == int(g["n_keys"])
    assert sorted(sd.keys()) == sorted(g["keys"].tolist())          # identical key set as the reference
    m.load_state_dict(recipe.fill_state_dict(sd, seed=seed, negative_gamma=bool(int(g["neg_gamma"])) if "neg_gamma" in g else False))
    x, lab = recipe.synthetic_batch(B, cin, img, img, cfg["n_classes"], seed=seed)
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    m.train()
    logits = m(x)
    loss = ref_unet.dice_bce_mc(logits, lab, cfg["n_classes"])
    opt.zero_grad()
    loss.backward()
    _sig_close(sig(logits), g["logits_sig"], rtol=5e-4)
    if "logits" in g:
        np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=1e-3, atol=2e-5)
    assert abs(loss.item() - float(g["loss0"])) < 5e-6
    for k, p in m.named_parameters():
        if "position_embeddings" in k and float(g["grad_sig." + k][0]) == 0:
            continue
        _sig_close(sig(p.grad), g["grad_sig." + k], rtol=2e-3)
    opt.step()
    for k, v in m.state_dict().items():
        if ("after1." + k) in g:
            _sig_close(sig(v.float()), g["after1." + k], rtol=2e-3)
    m.eval()
    with torch.no_grad():
        ev = m(x)
    _sig_close(sig(ev), g["eval_logits_sig"], rtol=2e-3)


def test_r50_vit_b16_parameter_count():
    m = ref_transunet.RefTransUNet(ref_transunet.r50_vit_b16_config(2), 224)
    assert sum(p.numel() for p in m.parameters()) == 105_276_066
    assert len(m.state_dict()) == 409


@pytest.mark.parametrize("name,cls", [("transunet_small_multitask", "RefTransUNetMultitask"),
                                      ("transunet_small_multitask_em", "RefTransUNetMultitaskEM")])
def test_transunet_multitask_oracle_matches_reference(golden_dir, name, cls):
    """VisionTransformerMultitask / ...EM (reference vit_seg_modeling.py:444-638): shared encoder, 2 / 6
    CUP decoders and heads.  Same state_dict keys in the same order, logits of every head, loss, gradients."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = ref_transunet.small_config(2)
    img, B, cin, seed = int(g["img"]), int(g["B"]), int(g["cin"]), int(g["seed"])
    m = getattr(ref_transunet, cls)(cfg, img)
    assert list(m.state_dict().keys()) == g["keys"].tolist()
    m.load_state_dict(recipe.fill_state_dict(m.state_dict(), seed=seed, negative_gamma=False))
    x, _ = recipe.synthetic_batch(B, cin, img, img, 2, seed=seed)
    m.train()
    outs = m(x)
    labs = [recipe.synthetic_batch(B, cin, img, img, 2, seed=seed + 100 * i)[1] for i in range(len(outs))]
    loss = sum(ref_unet.dice_bce_mc(o, l, 2) for o, l in zip(outs, labs))
    loss.backward()
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.detach().numpy(), g[f"logits{i + 1}"], rtol=1e-3, atol=2e-5)
    assert abs(loss.item() - float(g["loss0"])) < 2e-5
    for k, p in m.named_parameters():
        if "position_embeddings" in k and float(g["grad_sig." + k][0]) == 0:
            continue
        _sig_close(sig(p.grad), g["grad_sig." + k], rtol=2e-3)
