"""ResNetV2 hybrid backbone with skip features: drop-in for reference TransUnet/vit_seg_modeling_resnet_skip.py.

Same class names, constructor signatures, attribute names (=> `state_dict` keys) and construction order (=> identical
initial weights under the same seed).  The modules own parameters only; the arithmetic is emitted onto a libunetmi tape
by the `build_*` functions below (weight-standardised convs, GroupNorm(+residual)(+ReLU), 3x3/s2 max-pool).
"""
from collections import OrderedDict

import torch
import torch.nn as nn


def np2th(weights, conv=False):
    """JAX checkpoint array -> torch tensor (HWIO -> OIHW for conv kernels)."""
    if conv:
        weights = weights.transpose([3, 2, 0, 1])
    return torch.from_numpy(weights)


class StdConv2d(nn.Conv2d):
    """Conv2d whose weight is standardised per output channel at every forward (reference :18-25).  Only a parameter
    container here: `TUTape.std_conv` runs umi_wstd_fwd + the conv kernels and their backward."""

    def forward(self, x):
        """Weight-standardised convolution on its own (reference :20-25): a one-op tape."""
        return _run_block(self, [x], lambda t, a: t.std_conv(a, self))


def conv3x3(cin, cout, stride=1, groups=1, bias=False):
    return StdConv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=bias, groups=groups)


def conv1x1(cin, cout, stride=1, bias=False):
    return StdConv2d(cin, cout, kernel_size=1, stride=stride, padding=0, bias=bias)


class PreActBottleneck(nn.Module):
    """Bottleneck unit (reference :38-74): relu(gn1(conv1x1)) -> relu(gn2(conv3x3, stride)) -> gn3(conv1x1),
    residual = gn_proj(conv1x1(x, stride)) when the shape changes, out = relu(residual + y)."""

    def __init__(self, cin, cout=None, cmid=None, stride=1):
        super().__init__()
        cout = cout or cin
        cmid = cmid or cout // 4
        self.gn1 = nn.GroupNorm(32, cmid, eps=1e-6)
        self.conv1 = conv1x1(cin, cmid, bias=False)
        self.gn2 = nn.GroupNorm(32, cmid, eps=1e-6)
        self.conv2 = conv3x3(cmid, cmid, stride, bias=False)
        self.gn3 = nn.GroupNorm(32, cout, eps=1e-6)
        self.conv3 = conv1x1(cmid, cout, bias=False)
        self.relu = nn.ReLU(inplace=True)
        if stride != 1 or cin != cout:
            self.downsample = conv1x1(cin, cout, stride, bias=False)
            self.gn_proj = nn.GroupNorm(cout, cout)          # one group per channel, default eps 1e-5 (reference :58)

    def forward(self, x):
        return _run_block(self, [x], lambda t, a: build_unit(t, a, self))

    def load_from(self, weights, n_block, n_unit):
        def get(name, conv=False):
            return np2th(weights["/".join([n_block, n_unit, name])], conv=conv)
        with torch.no_grad():
            for i in (1, 2, 3):
                getattr(self, f"conv{i}").weight.copy_(get(f"conv{i}/kernel", conv=True))
                getattr(self, f"gn{i}").weight.copy_(get(f"gn{i}/scale").view(-1))
                getattr(self, f"gn{i}").bias.copy_(get(f"gn{i}/bias").view(-1))
            if hasattr(self, "downsample"):
                self.downsample.weight.copy_(get("conv_proj/kernel", conv=True))
                self.gn_proj.weight.copy_(get("gn_proj/scale").view(-1))
                self.gn_proj.bias.copy_(get("gn_proj/bias").view(-1))


class ResNetV2(nn.Module):
    """Root (7x7/s2 StdConv, GN32, ReLU) + three bottleneck stages (reference :112-160)."""

    def __init__(self, block_units, width_factor):
        super().__init__()
        width = int(64 * width_factor)
        self.width = width
        self.root = nn.Sequential(OrderedDict([
            ("conv", StdConv2d(3, width, kernel_size=7, stride=2, bias=False, padding=3)),
            ("gn", nn.GroupNorm(32, width, eps=1e-6)),
            ("relu", nn.ReLU(inplace=True)),
        ]))

        def stage(cin, cout, cmid, n, stride):
            units = [("unit1", PreActBottleneck(cin=cin, cout=cout, cmid=cmid, stride=stride))]
            units += [(f"unit{i:d}", PreActBottleneck(cin=cout, cout=cout, cmid=cmid)) for i in range(2, n + 1)]
            return nn.Sequential(OrderedDict(units))

        self.body = nn.Sequential(OrderedDict([
            ("block1", stage(width, width * 4, width, block_units[0], 1)),
            ("block2", stage(width * 4, width * 8, width * 2, block_units[1], 2)),
            ("block3", stage(width * 8, width * 16, width * 4, block_units[2], 2)),
        ]))

    def forward(self, x):
        """NCHW -> (1/16 feature, [skip 1/8, skip 1/4, skip 1/2]) (reference :142-160)."""
        def build(t, a):
            f, skips = build_resnet(t, a, self)
            return (f,) + tuple(skips)
        outs = _run_block(self, [x], build)
        return outs[0], list(outs[1:])


def _run_block(module, inputs, build):
    """Standalone forward of a ResNet piece: its own small tape (the full network runs them on VisionTransformer's tape)."""
    from Model import _resolve_dtype, _run_tape
    from umi.graph_tu import TUTape
    return _run_tape(module, inputs, build, tape_cls=TUTape, dtype=_resolve_dtype(getattr(module, "_compute_dtype", None)))


# ---- tape builders ---------------------------------------------------------------------------------------------------
def build_unit(t, x, u: PreActBottleneck):
    res = x
    if hasattr(u, "downsample"):
        res = t.group_norm(t.std_conv(x, u.downsample), u.gn_proj, relu=False)
    y = t.group_norm(t.std_conv(x, u.conv1), u.gn1, relu=True)
    y = t.group_norm(t.std_conv(y, u.conv2), u.gn2, relu=True)
    return t.group_norm(t.std_conv(y, u.conv3), u.gn3, relu=True, residual=res)


def build_resnet(t, x, net: ResNetV2):
    """Returns (1/16-resolution feature, [skip 1/8, skip 1/4, skip 1/2]) like reference :142-160."""
    in_size = x.shape[1]
    x = t.group_norm(t.std_conv(x, net.root.conv), net.root.gn, relu=True)
    features = [x]
    x = t.pool3s2(x)
    blocks = list(net.body.children())
    for i, block in enumerate(blocks[:-1]):
        for u in block.children():
            x = build_unit(t, x, u)
        right = int(in_size / 4 / (i + 1))
        if x.shape[1] != right:
            pad = right - x.shape[1]
            assert 0 < pad < 3, "x {} should {}".format(x.shape, right)
            features.append(t.pad_to(x, right))
        else:
            features.append(x)
    for u in blocks[-1].children():
        x = build_unit(t, x, u)
    return x, features[::-1]
