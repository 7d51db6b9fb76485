"""CPU restatement of the reference TransUNet (R50-ViT hybrid) hot path.  TEST INFRASTRUCTURE ONLY.

Restates with stock PyTorch CPU fp32 ops (citations are reference files under /root/reference/TransUnet):
  * vit_seg_modeling_resnet_skip.py:18-25   StdConv2d (weight standardisation, biased var, eps 1e-5)
  * vit_seg_modeling_resnet_skip.py:38-74   PreActBottleneck (GN32 eps 1e-6; gn_proj = GroupNorm(cout, cout) eps 1e-5)
  * vit_seg_modeling_resnet_skip.py:112-160 ResNetV2 (7x7/s2 root, 3x3/s2/p0 max-pool, zero-padded skip features)
  * vit_seg_modeling.py:122-165             Embeddings (1x1 patch conv + bias, + position embedding, dropout)
  * vit_seg_modeling.py:50-94, 97-119, 168-187, 227-244   Attention / Mlp (exact GELU) / Block (pre-LN, eps 1e-6) / Encoder
  * vit_seg_modeling.py:259-367             Conv2dReLU (conv, BN, ReLU), DecoderBlock (bilinear x2 align_corners=True,
                                            cat([x, skip])), DecoderCup, SegmentationHead (3x3 conv + bias)
  * vit_seg_modeling.py:370-392             VisionTransformer.forward (1 -> 3 channel repeat)

`RefTransUNet(cfg, img_size)` exposes the reference's state_dict keys (409 for R50-ViT-B_16) through a generated
module tree; `cfg` is a plain dict (see `r50_vit_b16_config`).  Pinned by tests/golden/transunet_*.npz (outputs of
the reference itself, tools/gen_golden_transunet.py).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .ref_unet import _Node, _sub


def r50_vit_b16_config(n_classes=2, n_skip=3, dropout_rate=0.1):
    """Values of reference vit_seg_configs.py:42-58 (get_r50_b16_config) as a plain dict."""
    return dict(hidden_size=768, mlp_dim=3072, num_heads=12, num_layers=12, attention_dropout_rate=0.0,
                dropout_rate=dropout_rate, resnet_layers=(3, 4, 9), width_factor=1,
                decoder_channels=(256, 128, 64, 16), skip_channels=[512, 256, 64, 16], n_classes=n_classes,
                n_skip=n_skip)


def small_config(n_classes=2):
    """A structurally identical miniature (for fast fixtures): width 32, hidden 64, 4 heads, 2 layers."""
    return dict(hidden_size=64, mlp_dim=128, num_heads=4, num_layers=2, attention_dropout_rate=0.0, dropout_rate=0.0,
                resnet_layers=(2, 2, 2), width_factor=0.5, decoder_channels=(64, 32, 32, 16),
                skip_channels=[256, 128, 32, 16], n_classes=n_classes, n_skip=3)


def _param(root, path, shape, init):
    mod, name = path.rsplit(".", 1)
    t = torch.empty(*shape)
    init(t)
    _sub(root, mod).register_parameter(name, nn.Parameter(t))


def _conv_w(t):
    nn.init.kaiming_uniform_(t, a=math.sqrt(5))


def _bias_for(fan_in):
    b = 1.0 / math.sqrt(fan_in)
    return lambda t: t.uniform_(-b, b)


class RefTransUNet(nn.Module):
    HEADS = (("decoder", "segmentation_head"),)                  # (decoder attribute, head attribute) per output

    def __init__(self, cfg, img_size=224):
        super().__init__()
        self.cfg, self.img_size = dict(cfg), img_size
        hid, mlp = cfg["hidden_size"], cfg["mlp_dim"]
        width = int(64 * cfg["width_factor"])
        self.width = width
        ones, zeros = (lambda t: t.fill_(1.0)), (lambda t: t.zero_())
        R = "transformer.embeddings.hybrid_model."
        _param(self, R + "root.conv.weight", (width, 3, 7, 7), _conv_w)
        _param(self, R + "root.gn.weight", (width,), ones)
        _param(self, R + "root.gn.bias", (width,), zeros)
        self.units = []                                   # (prefix, cin, cout, cmid, stride)
        cin = width
        for bi, (nunits, mult) in enumerate(zip(cfg["resnet_layers"], (4, 8, 16)), start=1):
            cout, cmid = width * mult, width * mult // 4
            for ui in range(1, nunits + 1):
                stride = 2 if (ui == 1 and bi > 1) else 1
                p = f"{R}body.block{bi}.unit{ui}."
                for g, c in (("gn1", cmid), ("gn2", cmid), ("gn3", cout)):
                    pass
                _param(self, p + "gn1.weight", (cmid,), ones); _param(self, p + "gn1.bias", (cmid,), zeros)
                _param(self, p + "conv1.weight", (cmid, cin, 1, 1), _conv_w)
                _param(self, p + "gn2.weight", (cmid,), ones); _param(self, p + "gn2.bias", (cmid,), zeros)
                _param(self, p + "conv2.weight", (cmid, cmid, 3, 3), _conv_w)
                _param(self, p + "gn3.weight", (cout,), ones); _param(self, p + "gn3.bias", (cout,), zeros)
                _param(self, p + "conv3.weight", (cout, cmid, 1, 1), _conv_w)
                if stride != 1 or cin != cout:
                    _param(self, p + "downsample.weight", (cout, cin, 1, 1), _conv_w)
                    _param(self, p + "gn_proj.weight", (cout,), ones); _param(self, p + "gn_proj.bias", (cout,), zeros)
                self.units.append((p, cin, cout, cmid, stride, bi))
                cin = cout
        self.res_out = cin
        E = "transformer.embeddings."
        _param(self, E + "patch_embeddings.weight", (hid, cin, 1, 1), _conv_w)
        _param(self, E + "patch_embeddings.bias", (hid,), _bias_for(cin))
        n_patches = (img_size // 16) ** 2
        _param(self, E + "position_embeddings", (1, n_patches, hid), zeros)
        for li in range(cfg["num_layers"]):
            p = f"transformer.encoder.layer.{li}."
            _param(self, p + "attention_norm.weight", (hid,), ones); _param(self, p + "attention_norm.bias", (hid,), zeros)
            _param(self, p + "ffn_norm.weight", (hid,), ones); _param(self, p + "ffn_norm.bias", (hid,), zeros)
            _param(self, p + "ffn.fc1.weight", (mlp, hid), nn.init.xavier_uniform_)
            _param(self, p + "ffn.fc1.bias", (mlp,), lambda t: nn.init.normal_(t, std=1e-6))
            _param(self, p + "ffn.fc2.weight", (hid, mlp), nn.init.xavier_uniform_)
            _param(self, p + "ffn.fc2.bias", (hid,), lambda t: nn.init.normal_(t, std=1e-6))
            for nm in ("query", "key", "value", "out"):
                _param(self, p + f"attn.{nm}.weight", (hid, hid), _conv_w)
                _param(self, p + f"attn.{nm}.bias", (hid,), _bias_for(hid))
        _param(self, "transformer.encoder.encoder_norm.weight", (hid,), ones)
        _param(self, "transformer.encoder.encoder_norm.bias", (hid,), zeros)
        skip = list(cfg["skip_channels"])
        for i in range(4 - cfg["n_skip"]):                      # vit_seg_modeling.py:343-345
            skip[3 - i] = 0
        if cfg["n_skip"] == 0:
            skip = [0, 0, 0, 0]
        dec = list(cfg["decoder_channels"])
        ins = [512] + dec[:-1]
        self.dec_blocks = [(f"blocks.{i}", sk) for i, sk in enumerate(skip)]
        for dname, _ in self.HEADS:                              # all decoders first, then all heads (reference order)
            self._bn((dname + ".conv_more", 512, hid))
            for i, (ci, co, sk) in enumerate(zip(ins, dec, skip)):
                self._bn((f"{dname}.blocks.{i}.conv1", co, ci + sk))
                self._bn((f"{dname}.blocks.{i}.conv2", co, co))
        for _, hname in self.HEADS:
            _param(self, hname + ".0.weight", (cfg["n_classes"], dec[-1], 3, 3), _conv_w)
            _param(self, hname + ".0.bias", (cfg["n_classes"],), _bias_for(dec[-1] * 9))

    def _bn(self, spec):
        path, co, ci = spec
        _param(self, path + ".0.weight", (co, ci, 3, 3), _conv_w)
        n = _sub(self, path + ".1")
        n.register_parameter("weight", nn.Parameter(torch.ones(co)))
        n.register_parameter("bias", nn.Parameter(torch.zeros(co)))
        n.register_buffer("running_mean", torch.zeros(co))
        n.register_buffer("running_var", torch.ones(co))
        n.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    # ---- pieces ---------------------------------------------------------------------------------------
    def _get(self, path):
        cur = self
        for part in path.split("."):
            cur = getattr(cur, part)
        return cur

    @staticmethod
    def std_conv(x, w, stride, pad):
        v, m = torch.var_mean(w, dim=[1, 2, 3], keepdim=True, unbiased=False)
        return F.conv2d(x, (w - m) / torch.sqrt(v + 1e-5), None, stride, pad)

    def _unit(self, x, spec):
        p, cin, cout, cmid, stride, _ = spec
        g = lambda nm: self._get(p + nm)
        res = x
        if stride != 1 or cin != cout:
            res = self.std_conv(x, g("downsample").weight, stride, 0)
            res = F.group_norm(res, cout, g("gn_proj").weight, g("gn_proj").bias, 1e-5)
        y = F.relu(F.group_norm(self.std_conv(x, g("conv1").weight, 1, 0), 32, g("gn1").weight, g("gn1").bias, 1e-6))
        y = F.relu(F.group_norm(self.std_conv(y, g("conv2").weight, stride, 1), 32, g("gn2").weight, g("gn2").bias, 1e-6))
        y = F.group_norm(self.std_conv(y, g("conv3").weight, 1, 0), 32, g("gn3").weight, g("gn3").bias, 1e-6)
        return F.relu(res + y)

    def resnet(self, x):
        hm = self.transformer.embeddings.hybrid_model
        in_size = x.shape[2]
        x = F.relu(F.group_norm(self.std_conv(x, hm.root.conv.weight, 2, 3), 32, hm.root.gn.weight, hm.root.gn.bias, 1e-6))
        feats = [x]
        x = F.max_pool2d(x, 3, 2, 0)
        nblocks = 3
        for bi in range(1, nblocks + 1):
            for spec in (u for u in self.units if u[5] == bi):
                x = self._unit(x, spec)
            if bi < nblocks:
                right = int(in_size / 4 / bi)                       # resnet_skip.py:150 (i = bi-1)
                if x.shape[2] != right:
                    feat = torch.zeros(x.shape[0], x.shape[1], right, right)
                    feat[:, :, :x.shape[2], :x.shape[3]] = x
                else:
                    feat = x
                feats.append(feat)
        return x, feats[::-1]

    def _bn_relu_conv(self, x, path):
        conv, bn = self._get(path + ".0"), self._get(path + ".1")
        y = F.conv2d(x, conv.weight, None, 1, 1)
        if self.training:
            bn.num_batches_tracked += 1
        return F.relu(F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, self.training, 0.1, 1e-5))

    def encoder(self, h):
        cfg = self.cfg
        nh = cfg["num_heads"]
        hd = cfg["hidden_size"] // nh
        for li in range(cfg["num_layers"]):
            L = self.transformer.encoder.layer
            blk = getattr(L, str(li))
            x = F.layer_norm(h, (h.shape[-1],), blk.attention_norm.weight, blk.attention_norm.bias, 1e-6)
            B, N, C = x.shape
            split = lambda t: t.view(B, N, nh, hd).permute(0, 2, 1, 3)
            q = split(F.linear(x, blk.attn.query.weight, blk.attn.query.bias))
            k = split(F.linear(x, blk.attn.key.weight, blk.attn.key.bias))
            v = split(F.linear(x, blk.attn.value.weight, blk.attn.value.bias))
            probs = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
            probs = F.dropout(probs, cfg["attention_dropout_rate"], self.training)
            ctxt = torch.matmul(probs, v).permute(0, 2, 1, 3).reshape(B, N, C)
            a = F.dropout(F.linear(ctxt, blk.attn.out.weight, blk.attn.out.bias), cfg["attention_dropout_rate"], self.training)
            h = a + h
            x = F.layer_norm(h, (C,), blk.ffn_norm.weight, blk.ffn_norm.bias, 1e-6)
            x = F.dropout(F.gelu(F.linear(x, blk.ffn.fc1.weight, blk.ffn.fc1.bias)), cfg["dropout_rate"], self.training)
            x = F.dropout(F.linear(x, blk.ffn.fc2.weight, blk.ffn.fc2.bias), cfg["dropout_rate"], self.training)
            h = x + h
        en = self.transformer.encoder.encoder_norm
        return F.layer_norm(h, (h.shape[-1],), en.weight, en.bias, 1e-6)

    def forward(self, x):
        if x.shape[1] == 1:
            x = x.repeat(1, 3, 1, 1)
        emb = self.transformer.embeddings
        f, feats = self.resnet(x)
        t = F.conv2d(f, emb.patch_embeddings.weight, emb.patch_embeddings.bias)
        B, C, gh, gw = t.shape
        h = t.flatten(2).transpose(-1, -2) + emb.position_embeddings
        h = F.dropout(h, self.cfg["dropout_rate"], self.training)
        h = self.encoder(h)
        tokens = h.permute(0, 2, 1).contiguous().view(B, C, gh, gw)
        outs = []
        for dname, hname in self.HEADS:                          # vit_seg_modeling.py:389-392 / :468-476 (shared encoder)
            y = self._bn_relu_conv(tokens, dname + ".conv_more")
            for i, (path, sk) in enumerate(self.dec_blocks):
                y = F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=True)
                if sk and i < self.cfg["n_skip"]:
                    y = torch.cat([y, feats[i]], dim=1)
                y = self._bn_relu_conv(y, f"{dname}.{path}.conv1")
                y = self._bn_relu_conv(y, f"{dname}.{path}.conv2")
            sh = getattr(getattr(self, hname), "0")
            outs.append(F.conv2d(y, sh.weight, sh.bias, 1, 1))
        return outs[0] if len(outs) == 1 else tuple(outs)


class RefTransUNetMultitask(RefTransUNet):
    """Reference VisionTransformerMultitask (vit_seg_modeling.py:444-476): one encoder, two CUP decoders, two heads."""
    HEADS = (("decoder1", "segmentation_head1"), ("decoder2", "segmentation_head2"))


class RefTransUNetMultitaskEM(RefTransUNet):
    """Reference VisionTransformerMultitaskEM (vit_seg_modeling.py:524-590): six decoders / heads over one encoder."""
    HEADS = tuple((f"decoder{i}", f"segmentation_head{i}") for i in range(1, 7))
