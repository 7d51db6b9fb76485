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
