"""TransUNet fixtures from the REFERENCE itself (called by tools/gen_golden.py; build container only).

Builds reference `TransUnet.vit_seg_modeling.VisionTransformer` from a ConfigDict equivalent to the oracle's plain
dict config, loads the build-owned recipe weights, runs forward + `dice_bce_mc` + backward (+1 SGD step), and stores
numeric outputs only.  Dropout is set to 0.0 in the config (CPU/GPU RNG streams cannot match).
"""
import os

import numpy as np
import torch

from oracle import recipe, ref_transunet


def _ref_config(cfg, img_size):
    import ml_collections
    C = ml_collections.ConfigDict
    g = img_size // 16
    c = C(dict(patches=C({"size": (16, 16), "grid": (g, g)}), hidden_size=cfg["hidden_size"],
               transformer=C(dict(mlp_dim=cfg["mlp_dim"], num_heads=cfg["num_heads"], num_layers=cfg["num_layers"],
                                  attention_dropout_rate=cfg["attention_dropout_rate"], dropout_rate=cfg["dropout_rate"])),
               classifier="seg", representation_size=None, resnet_pretrained_path=None, pretrained_path=None, patch_size=16,
               decoder_channels=tuple(cfg["decoder_channels"]), n_classes=cfg["n_classes"], activation="softmax",
               resnet=C(dict(num_layers=tuple(cfg["resnet_layers"]), width_factor=cfg["width_factor"])),
               skip_channels=list(cfg["skip_channels"]), n_skip=cfg["n_skip"]))
    return c


def _case(vsm, loss_mod, sig, meta, GOLD, name, cfg, img, B, cin, seed, full, neg_gamma=False):
    torch.manual_seed(0)
    loss_mod.CLASS_NUMBER = cfg["n_classes"]
    m = vsm.VisionTransformer(_ref_config(cfg, img), img_size=img, num_classes=cfg["n_classes"])
    out = dict(img=img, B=B, cin=cin, seed=seed, n_keys=len(m.state_dict()), neg_gamma=int(neg_gamma))
    out["keys"] = np.array(list(m.state_dict().keys()))
    for k, v in m.state_dict().items():                # the reference's own init under torch.manual_seed(0)
        out["init_sig." + k] = sig(v.float())
    m.load_state_dict(recipe.fill_state_dict(m.state_dict(), seed=seed, negative_gamma=neg_gamma))
    x, lab = recipe.synthetic_batch(B, cin, img, img, cfg["n_classes"], seed=seed)
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    m.train()
    logits = m(x)
    loss = loss_mod.calc_loss(logits, lab, loss_type="dice_bce_mc")
    opt.zero_grad()
    loss.backward()
    out["logits_sig"] = sig(logits)
    out["loss0"] = loss.item()
    if full:
        out["logits"] = logits.detach().numpy()
    for k, p in m.named_parameters():
        out["grad_sig." + k] = sig(p.grad)
    opt.step()
    for k, v in m.state_dict().items():
        if "running" in k or "num_batches" in k or full:
            out["after1." + k] = sig(v.float())
    m.eval()
    with torch.no_grad():
        ev = m(x)
    out["eval_logits_sig"] = sig(ev)
    if full:
        out["eval_logits"] = ev.numpy()
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out, **meta())
    print(f"wrote {name}.npz keys={out['n_keys']} loss0={out['loss0']:.6f}")


def _case_multitask(vsm, loss_mod, sig, meta, GOLD, name, cls_name, cfg, img, B, cin, seed):
    """VisionTransformerMultitask / ...EM (reference vit_seg_modeling.py:444-638), loss = sum of the heads' dice_bce_mc
    losses (Trainer.py:885-890 for two heads).  Seed chosen free of fp32 ReLU near-ties (see gen_golden.gen_unet_multitask)."""
    torch.manual_seed(0)
    loss_mod.CLASS_NUMBER = cfg["n_classes"]
    m = getattr(vsm, cls_name)(_ref_config(cfg, img), img_size=img, num_classes=cfg["n_classes"])
    out = dict(img=img, B=B, cin=cin, seed=seed, n_keys=len(m.state_dict()))
    out["keys"] = np.array(list(m.state_dict().keys()))
    m.load_state_dict(recipe.fill_state_dict(m.state_dict(), seed=seed, negative_gamma=False))
    x, _ = recipe.synthetic_batch(B, cin, img, img, cfg["n_classes"], seed=seed)
    m.train()
    logits = m(x)
    labs = [recipe.synthetic_batch(B, cin, img, img, cfg["n_classes"], seed=seed + 100 * i)[1] for i in range(len(logits))]
    loss = sum(loss_mod.calc_loss(o, l, loss_type="dice_bce_mc") for o, l in zip(logits, labs))
    loss.backward()
    out["loss0"] = loss.item()
    for i, o in enumerate(logits):
        out[f"logits{i + 1}"] = o.detach().numpy()
    for k, p in m.named_parameters():
        out["grad_sig." + k] = sig(p.grad)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out, **meta())
    print(f"wrote {name}.npz keys={out['n_keys']} loss0={out['loss0']:.6f}")


def _case_load_from(vsm, sig, meta, GOLD):
    """state_dict signatures after the REFERENCE's `load_from` (vit_seg_modeling.py:394-441 -> Block.load_from :189-224 ->
    PreActBottleneck.load_from resnet_skip.py:76-110) of a synthetic JAX-keyed checkpoint (oracle/recipe.py), for the two
    position-embedding paths a 4 x 4-token model can take: a 3 x 3 grid + class token (bilinear `ndimage.zoom` resize) and a
    4 x 4 grid + class token (class token dropped).  The real `.npz` is not in the reference tree."""
    cfg = ref_transunet.small_config(2)
    out = {}
    for tag, old_grid in (("zoom", 3), ("drop_cls", 4)):
        torch.manual_seed(0)
        m = vsm.VisionTransformer(_ref_config(cfg, 64), img_size=64, num_classes=2)
        w = recipe.synthetic_jax_checkpoint(m, cfg["hidden_size"], cfg["num_heads"], old_grid, seed=77)
        before = {k: v.clone() for k, v in m.state_dict().items()}
        m.load_from(w)
        out[tag + ".n_ckpt_keys"] = len(w)
        changed = []
        for k, v in m.state_dict().items():
            out[f"{tag}.sig." + k] = sig(v.float())
            if not torch.equal(v, before[k]):
                changed.append(k)
        out[tag + ".changed"] = np.array(changed)
    np.savez_compressed(os.path.join(GOLD, "transunet_small_load_from.npz"), **out, **meta())
    print("wrote transunet_small_load_from.npz", out["zoom.n_ckpt_keys"], len(out["zoom.changed"]))


def run(import_reference, sig, meta, GOLD, big=False, only512=False, only_load_from=False, only_neg=False):
    _, loss_mod, _ = import_reference()
    from TransUnet import vit_seg_modeling as vsm
    if only_load_from:
        _case_load_from(vsm, sig, meta, GOLD)
        return
    if only_neg:
        # every 7th GroupNorm / LayerNorm / BatchNorm scale negative (oracle/recipe.py): a fusion that assumes gamma > 0
        # (e.g. a ReLU or max taken before the affine) shows up here; VERDICT round 2, "TransUNet fixtures use positive gammas"
        _case(vsm, loss_mod, sig, meta, GOLD, "transunet_small_neg_gamma", ref_transunet.small_config(2), 64, 2, 1, 41, True,
              neg_gamma=True)
        return
    if only512:
        # BASELINE configs[4] shape: R50-ViT-B/16 at 512 x 512 (1,024 tokens; the ResNet's 127 -> 128 zero-pad fix-up of the
        # 1/4-scale skip, reference vit_seg_modeling_resnet_skip.py:147-158), B = 1, signatures only
        cfg = ref_transunet.r50_vit_b16_config(2, 3, dropout_rate=0.0)
        _case(vsm, loss_mod, sig, meta, GOLD, "transunet_r50_b16_512", cfg, 512, 1, 1, 34, False)
        return
    _case(vsm, loss_mod, sig, meta, GOLD, "transunet_small", ref_transunet.small_config(2), 64, 2, 1, 31, True)
    _case(vsm, loss_mod, sig, meta, GOLD, "transunet_small_rgb4", ref_transunet.small_config(4), 96, 1, 3, 32, True)
    _case_multitask(vsm, loss_mod, sig, meta, GOLD, "transunet_small_multitask", "VisionTransformerMultitask",
                    ref_transunet.small_config(2), 64, 2, 1, 35)
    _case_multitask(vsm, loss_mod, sig, meta, GOLD, "transunet_small_multitask_em", "VisionTransformerMultitaskEM",
                    ref_transunet.small_config(2), 64, 1, 1, 39)
    _case_load_from(vsm, sig, meta, GOLD)
    if big:
        cfg = ref_transunet.r50_vit_b16_config(2, 3, dropout_rate=0.0)
        _case(vsm, loss_mod, sig, meta, GOLD, "transunet_r50_b16_224", cfg, 224, 1, 1, 33, False)
