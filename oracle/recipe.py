"""Deterministic, build-owned seed recipes for weights and synthetic batches.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Fixtures under tests/golden/ store
only the recipe seed plus reference *outputs*; weights and inputs are regenerated
from these recipes on whichever box the test runs (same image => same torch CPU
RNG stream).

The fill statistics are deliberately "un-pretty" (BN gamma away from 1, beta and
running stats non-trivial, gamma allowed to go negative for a few channels) so a
wrong fusion order (e.g. max-pool before BN-affine, which is only valid for
gamma > 0) shows up as a parity failure.
"""
import zlib

import torch


def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) + 7919 * seed) & 0x7FFFFFFF)
    return g


def fill_state_dict(sd, seed: int = 0, negative_gamma: bool = True):
    """Return a new state_dict with every entry of `sd` refilled deterministically.

    conv / linear weights ~ N(0, sqrt(2/fan_in)); biases ~ N(0, 0.1);
    norm weight ~ U(0.5, 1.5) (every 7th channel negated when `negative_gamma`);
    norm bias ~ N(0, 0.1); running_mean ~ N(0, 0.2); running_var ~ U(0.5, 1.5);
    num_batches_tracked = 0; position embeddings ~ N(0, 0.02).
    """
    out = {}
    for k, v in sd.items():
        g = _gen(k, seed)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros_like(v)
        elif k.endswith("running_mean"):
            out[k] = 0.2 * torch.randn(v.shape, generator=g)
        elif k.endswith("running_var"):
            out[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif "position_embeddings" in k:
            out[k] = 0.02 * torch.randn(v.shape, generator=g)
        elif v.dim() >= 2:
            fan_in = v[0].numel()
            out[k] = torch.randn(v.shape, generator=g) * (2.0 / fan_in) ** 0.5
        elif k.endswith("weight"):      # 1-D weight => a norm layer's gamma
            w = 0.5 + torch.rand(v.shape, generator=g)
            if negative_gamma:
                w[::7] = -w[::7]
            out[k] = w
        else:                           # 1-D bias
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
        out[k] = out[k].to(v.dtype)
    return out


def synthetic_batch(batch, channels, height, width, n_classes, seed: int = 1234):
    """Inputs mimic the reference's Data_Binary output (DataLoader.py:661-679):
    per-image z-normalised float32 => x ~ N(0,1); labels are class indices, returned
    as float because Trainer casts labels with `.type(dtype)` (Trainer.py:701-702).
    Labels are blob-structured (thresholded box-blurred noise) so Dice is non-degenerate.
    """
    g = _gen("batch", seed)
    x = torch.randn(batch, channels, height, width, generator=g)
    noise = torch.randn(batch, 1, height, width, generator=g)
    k = 9 if min(height, width) >= 9 else 3
    blur = torch.nn.functional.avg_pool2d(noise, k, stride=1, padding=k // 2,
                                          count_include_pad=False)
    q = torch.quantile(blur.flatten(), torch.linspace(0, 1, n_classes + 1)[1:-1])
    labels = torch.bucketize(blur[:, 0], q).to(torch.float32)
    return x, labels


def synthetic_jax_checkpoint(model, hidden, heads, old_grid, seed=77, cls_token=True):
    """A seeded stand-in for the JAX ViT / R50 `.npz` checkpoint the reference's `load_from` consumes
    (`/root/reference/TransUnet/vit_seg_modeling.py:189-224,394-441`, `vit_seg_modeling_resnet_skip.py:76-110`): the key
    names and array layouts of the published checkpoints (HWIO conv kernels, [hidden, heads, head_dim] attention kernels,
    [1, 1 + grid^2, hidden] position embedding with a class token), values from numpy's generator.  Walks the MODEL's own
    module tree, which the reference and the product share, so both sides draw identical arrays.  The real checkpoint is
    not in the reference tree and cannot be downloaded (no network)."""
    import numpy as np
    g = np.random.default_rng(seed)
    w = {}

    def r(*shape):
        return (g.standard_normal(shape) * 0.1).astype(np.float32)

    def conv(key, weight):
        O, I, kh, kw = weight.shape
        w[key] = r(kh, kw, I, O)

    emb = model.transformer.embeddings
    conv("embedding/kernel", emb.patch_embeddings.weight)
    w["embedding/bias"] = r(emb.patch_embeddings.weight.shape[0])
    w["Transformer/encoder_norm/scale"], w["Transformer/encoder_norm/bias"] = r(hidden), r(hidden)
    w["Transformer/posembed_input/pos_embedding"] = r(1, (1 if cls_token else 0) + old_grid * old_grid, hidden)
    hd = hidden // heads
    for i, blk in enumerate(model.transformer.encoder.layer):
        root = f"Transformer/encoderblock_{i}"
        for n in ("query", "key", "value"):
            w[f"{root}/MultiHeadDotProductAttention_1/{n}/kernel"] = r(hidden, heads, hd)
            w[f"{root}/MultiHeadDotProductAttention_1/{n}/bias"] = r(heads, hd)
        w[f"{root}/MultiHeadDotProductAttention_1/out/kernel"] = r(heads, hd, hidden)
        w[f"{root}/MultiHeadDotProductAttention_1/out/bias"] = r(hidden)
        mlp = blk.ffn.fc1.weight.shape[0]
        w[f"{root}/MlpBlock_3/Dense_0/kernel"], w[f"{root}/MlpBlock_3/Dense_0/bias"] = r(hidden, mlp), r(mlp)
        w[f"{root}/MlpBlock_3/Dense_1/kernel"], w[f"{root}/MlpBlock_3/Dense_1/bias"] = r(mlp, hidden), r(hidden)
        for ln in ("LayerNorm_0", "LayerNorm_2"):
            w[f"{root}/{ln}/scale"], w[f"{root}/{ln}/bias"] = r(hidden), r(hidden)
    hm = emb.hybrid_model
    conv("conv_root/kernel", hm.root.conv.weight)
    C = hm.root.conv.weight.shape[0]
    w["gn_root/scale"], w["gn_root/bias"] = r(1, 1, 1, C), r(1, 1, 1, C)
    for bname, block in hm.body.named_children():
        for uname, unit in block.named_children():
            for c, gn in (("conv1", "gn1"), ("conv2", "gn2"), ("conv3", "gn3")):
                conv(f"{bname}/{uname}/{c}/kernel", getattr(unit, c).weight)
                C = getattr(unit, c).weight.shape[0]
                w[f"{bname}/{uname}/{gn}/scale"], w[f"{bname}/{uname}/{gn}/bias"] = r(1, 1, 1, C), r(1, 1, 1, C)
            if hasattr(unit, "downsample"):
                conv(f"{bname}/{uname}/conv_proj/kernel", unit.downsample.weight)
                C = unit.downsample.weight.shape[0]
                w[f"{bname}/{uname}/gn_proj/scale"], w[f"{bname}/{uname}/gn_proj/bias"] = r(1, 1, 1, C), r(1, 1, 1, C)
    return w
