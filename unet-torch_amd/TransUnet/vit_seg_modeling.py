"""MI355X-native TransUNet: drop-in for reference TransUnet/vit_seg_modeling.py (hot-path classes).

`VisionTransformer(config, img_size=224, num_classes=21843, zero_head=False, vis=False)` keeps the reference
constructor (vit_seg_modeling.py:371), attribute tree (409 `state_dict` keys for R50-ViT-B_16) and construction order
(same initial weights under the same seed).  `forward` emits the whole network -- hybrid ResNetV2 with skips, 1x1 patch
embedding + position embedding, 12 pre-LN transformer blocks (softmax attention, exact-GELU MLP), CUP decoder
(bilinear x2 align_corners, cat([x, skip]), conv+BN+ReLU x2) and the 3x3 segmentation head -- onto a libunetmi tape
(umi/graph_tu.py) with a hand-written backward; nothing runs through torch.nn.

Keyword-only extra: `compute_dtype` ("fp16" default / "fp32" parity mode, env UMI_COMPUTE_DTYPE).
`VisionTransformerMultitask` / `VisionTransformerMultitaskEM` (reference :444-638): the same encoder with 2 / 6 CUP decoders
and heads on one tape.
Not supported: `vis=True` (attention maps are never materialised), the non-hybrid (pure ViT patch conv) variant.
"""
import copy
import logging

import numpy as np
import torch
import torch.nn as nn
from torch.nn import Conv2d, Dropout, LayerNorm, Linear
from torch.nn.modules.utils import _pair

from Model import _TapeFunction, _resolve_dtype, _run_tape
from umi import graph as G
from umi.graph_tu import TUTape

from . import vit_seg_configs as configs
from .vit_seg_modeling_resnet_skip import ResNetV2, build_resnet, np2th  # noqa: F401

logger = logging.getLogger(__name__)

_JAX = dict(q="MultiHeadDotProductAttention_1/query", k="MultiHeadDotProductAttention_1/key",
            v="MultiHeadDotProductAttention_1/value", o="MultiHeadDotProductAttention_1/out",
            fc0="MlpBlock_3/Dense_0", fc1="MlpBlock_3/Dense_1", ln0="LayerNorm_0", ln2="LayerNorm_2")


# ---- standalone forwards of the sub-modules ---------------------------------------------------------------------------------
# Inside VisionTransformer.forward the whole network is ONE tape.  Called on their own (as the reference's modules can be,
# vit_seg_modeling.py:73-94,113-119,154-165,177-187,237-244,309-315,355-367) the sub-modules build a small tape of their own
# through Model._run_tape: token tensors [B, N, C] travel as NCHW [B, C, 1, N] (the tape's token layout is NHWC with H = 1).
def _sub_dtype(module):
    return _resolve_dtype(getattr(module, "_compute_dtype", None))


def _tok_in(x):
    if x.dim() != 3:
        raise ValueError(f"expected tokens [B, N, C], got {tuple(x.shape)}")
    return x.permute(0, 2, 1).unsqueeze(2).contiguous()


def _tok_out(y):
    return y.squeeze(2).permute(0, 2, 1).contiguous()


def _run_tokens(module, x, build):
    return _tok_out(_run_tape(module, [_tok_in(x)], build, tape_cls=TUTape, dtype=_sub_dtype(module)))


class Attention(nn.Module):
    """12 x 64 multi-head softmax attention (reference :50-94); parameters only."""

    def __init__(self, config, vis):
        super().__init__()
        self.vis = vis
        self.num_attention_heads = config.transformer["num_heads"]
        self.attention_head_size = int(config.hidden_size / self.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = Linear(config.hidden_size, self.all_head_size)
        self.key = Linear(config.hidden_size, self.all_head_size)
        self.value = Linear(config.hidden_size, self.all_head_size)
        self.out = Linear(config.hidden_size, config.hidden_size)
        self.attn_dropout = Dropout(config.transformer["attention_dropout_rate"])
        self.proj_dropout = Dropout(config.transformer["attention_dropout_rate"])

    def forward(self, hidden_states):
        """[B, N, C] -> (attention output, None): `vis` is always False here, the fused kernel never materialises the weights."""
        return _run_tokens(self, hidden_states, lambda t, a: _build_attention(t, a, self)), None


class Mlp(nn.Module):
    """fc1 -> exact GELU -> dropout -> fc2 -> dropout (reference :97-119)."""

    def __init__(self, config):
        super().__init__()
        self.fc1 = Linear(config.hidden_size, config.transformer["mlp_dim"])
        self.fc2 = Linear(config.transformer["mlp_dim"], config.hidden_size)
        self.dropout = Dropout(config.transformer["dropout_rate"])
        self._init_weights()

    def _init_weights(self):
        nn.init.xavier_uniform_(self.fc1.weight)
        nn.init.xavier_uniform_(self.fc2.weight)
        nn.init.normal_(self.fc1.bias, std=1e-6)
        nn.init.normal_(self.fc2.bias, std=1e-6)

    def forward(self, x):
        return _run_tokens(self, x, lambda t, a: _build_mlp(t, a, self))


class Embeddings(nn.Module):
    """Hybrid ResNetV2 -> 1x1 patch conv + bias -> + position embedding -> dropout (reference :122-165)."""

    def __init__(self, config, img_size, in_channels=3):
        super().__init__()
        self.config = config
        img_size = _pair(img_size)
        if config.patches.get("grid") is None:
            raise NotImplementedError("only the hybrid (ResNet grid) TransUNet variant is on the MI355X hot path")
        grid = config.patches["grid"]
        patch = (img_size[0] // 16 // grid[0], img_size[1] // 16 // grid[1])
        if patch != (1, 1):
            raise NotImplementedError(f"patch size {patch}: the reference's train.py always sets grid = img_size/16 (patch 1x1)")
        n_patches = (img_size[0] // 16) * (img_size[1] // 16)
        self.hybrid = True
        self.hybrid_model = ResNetV2(block_units=config.resnet.num_layers, width_factor=config.resnet.width_factor)
        in_channels = self.hybrid_model.width * 16
        self.patch_embeddings = Conv2d(in_channels=in_channels, out_channels=config.hidden_size, kernel_size=patch, stride=patch)
        self.position_embeddings = nn.Parameter(torch.zeros(1, n_patches, config.hidden_size))
        self.dropout = Dropout(config.transformer["dropout_rate"])

    def forward(self, x):
        """NCHW image -> (tokens [B, N, hidden], [skip features, deepest first]) (reference :154-165)."""
        if x.size()[1] == 1:
            x = x.repeat(1, 3, 1, 1)

        def build(t, a):
            h, skips = _build_embeddings(t, a, self)
            return (h,) + tuple(skips)
        outs = _run_tape(self, [x], build, tape_cls=TUTape, dtype=_sub_dtype(self))
        return _tok_out(outs[0]), list(outs[1:])


class Block(nn.Module):
    """Pre-LN transformer block, LayerNorm eps 1e-6 (reference :168-187)."""

    def __init__(self, config, vis):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.attention_norm = LayerNorm(config.hidden_size, eps=1e-6)
        self.ffn_norm = LayerNorm(config.hidden_size, eps=1e-6)
        self.ffn = Mlp(config)
        self.attn = Attention(config, vis)

    def forward(self, x):
        return _run_tokens(self, x, lambda t, a: _build_block(t, a, self)), None

    def load_from(self, weights, n_block):
        root = f"Transformer/encoderblock_{n_block}"
        hs = self.hidden_size

        def arr(*parts):
            return np2th(weights["/".join((root,) + parts)])
        with torch.no_grad():
            for key, lin in (("q", self.attn.query), ("k", self.attn.key), ("v", self.attn.value), ("o", self.attn.out)):
                lin.weight.copy_(arr(_JAX[key], "kernel").view(hs, hs).t())
                lin.bias.copy_(arr(_JAX[key], "bias").view(-1))
            self.ffn.fc1.weight.copy_(arr(_JAX["fc0"], "kernel").t())
            self.ffn.fc2.weight.copy_(arr(_JAX["fc1"], "kernel").t())
            self.ffn.fc1.bias.copy_(arr(_JAX["fc0"], "bias").t())
            self.ffn.fc2.bias.copy_(arr(_JAX["fc1"], "bias").t())
            self.attention_norm.weight.copy_(arr(_JAX["ln0"], "scale"))
            self.attention_norm.bias.copy_(arr(_JAX["ln0"], "bias"))
            self.ffn_norm.weight.copy_(arr(_JAX["ln2"], "scale"))
            self.ffn_norm.bias.copy_(arr(_JAX["ln2"], "bias"))


class Encoder(nn.Module):
    def __init__(self, config, vis):
        super().__init__()
        self.vis = vis
        self.layer = nn.ModuleList()
        self.encoder_norm = LayerNorm(config.hidden_size, eps=1e-6)
        for _ in range(config.transformer["num_layers"]):
            self.layer.append(copy.deepcopy(Block(config, vis)))

    def forward(self, hidden_states):
        """[B, N, C] -> (encoded tokens, []) (reference :237-244; no attention maps: vis is False)."""
        def build(t, h):
            for blk in self.layer:
                h = _build_block(t, h, blk)
            return t.layer_norm(h, self.encoder_norm)
        return _run_tokens(self, hidden_states, build), []


class Transformer(nn.Module):
    def __init__(self, config, img_size, vis):
        super().__init__()
        self.embeddings = Embeddings(config, img_size=img_size)
        self.encoder = Encoder(config, vis)

    def forward(self, input_ids):
        """NCHW image -> (encoded tokens, [], skip features) (reference :253-256)."""
        x = input_ids
        if x.size()[1] == 1:
            x = x.repeat(1, 3, 1, 1)

        def build(t, a):
            h, skips = _build_embeddings(t, a, self.embeddings)
            for blk in self.encoder.layer:
                h = _build_block(t, h, blk)
            return (t.layer_norm(h, self.encoder.encoder_norm),) + tuple(skips)
        outs = _run_tape(self, [x], build, tape_cls=TUTape, dtype=_sub_dtype(self))
        return _tok_out(outs[0]), [], list(outs[1:])


class Conv2dReLU(nn.Sequential):
    """conv -> BatchNorm -> ReLU (module order of reference :259-281: conv, bn, relu)."""

    def __init__(self, in_channels, out_channels, kernel_size, padding=0, stride=1, use_batchnorm=True):
        conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=not use_batchnorm)
        relu = nn.ReLU(inplace=True)
        bn = nn.BatchNorm2d(out_channels)
        super().__init__(conv, bn, relu)


class DecoderBlock(nn.Module):
    def __init__(self, in_channels, out_channels, skip_channels=0, use_batchnorm=True):
        super().__init__()
        self.conv1 = Conv2dReLU(in_channels + skip_channels, out_channels, kernel_size=3, padding=1, use_batchnorm=use_batchnorm)
        self.conv2 = Conv2dReLU(out_channels, out_channels, kernel_size=3, padding=1, use_batchnorm=use_batchnorm)
        self.up = nn.UpsamplingBilinear2d(scale_factor=2)

    def forward(self, x, skip=None):
        """NCHW [, NCHW skip at twice the size] -> NCHW (reference :309-315: bilinear x2, cat([x, skip]), two conv+BN+ReLU)."""
        ins = [x] if skip is None else [x, skip]
        return _run_tape(self, ins, lambda t, a, sk=None: _build_decoder_block(t, a, sk, self), tape_cls=TUTape,
                         dtype=_sub_dtype(self))


class SegmentationHead(nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size=3, upsampling=1):
        conv2d = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, padding=kernel_size // 2)
        up = nn.UpsamplingBilinear2d(scale_factor=upsampling) if upsampling > 1 else nn.Identity()
        super().__init__(conv2d, up)


class DecoderCup(nn.Module):
    """conv_more (hidden -> 512) + four decoder blocks (reference :326-367)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        head_channels = 512
        self.conv_more = Conv2dReLU(config.hidden_size, head_channels, kernel_size=3, padding=1, use_batchnorm=True)
        decoder_channels = config.decoder_channels
        in_channels = [head_channels] + list(decoder_channels[:-1])
        if self.config.n_skip != 0:
            skip_channels = self.config.skip_channels      # shared list, mutated in place exactly like the reference
            for i in range(4 - self.config.n_skip):
                skip_channels[3 - i] = 0
        else:
            skip_channels = [0, 0, 0, 0]
        self.blocks = nn.ModuleList(DecoderBlock(i, o, s) for i, o, s in zip(in_channels, decoder_channels, skip_channels))

    def forward(self, hidden_states, features=None):
        """tokens [B, N, hidden] (N a square number) [, skip features] -> NCHW (reference :355-367)."""
        B, n_tok, _ = hidden_states.shape
        g = int(np.sqrt(n_tok))
        feats = list(features) if features is not None else []
        feats = feats[:self.config.n_skip]

        def build(t, h, *sk):
            return _build_decoder(t, h, list(sk) if sk else None, self, g, g)
        return _run_tape(self, [_tok_in(hidden_states)] + feats, build, tape_cls=TUTape, dtype=_sub_dtype(self))


# ---- tape builders -------------------------------------------------------------------------------------------------------
def _build_attention(t, x, attn: Attention, residual=None):
    """`residual`: the block's skip input, added by the projection dropout's kernel (Block.forward's `x + h`)."""
    if attn.attn_dropout.p > 0 and t.training:
        raise NotImplementedError("attention-probability dropout > 0 is not supported (all reference configs use 0.0)")
    ctx = t.qkv_attention(x, attn.query, attn.key, attn.value, attn.num_attention_heads)
    return t.linear_dropout(ctx, attn.out.weight, attn.out.bias, attn.proj_dropout.p, add=residual)


def _build_mlp(t, x, mlp: Mlp, residual=None):
    """`residual`: the block's skip input, added by the last dropout's kernel (Block.forward's `x + h`)."""
    x = t.linear_dropout(x, mlp.fc1.weight, mlp.fc1.bias, mlp.dropout.p, gelu=True)
    return t.linear_dropout(x, mlp.fc2.weight, mlp.fc2.bias, mlp.dropout.p, add=residual)


def _build_block(t, h, blk: Block, cfg=None):
    """Pre-LN block (reference :177-187).  Rates and head count come from the modules themselves (== cfg.transformer[...])."""
    h = _build_attention(t, t.layer_norm(h, blk.attention_norm), blk.attn, residual=h)
    return _build_mlp(t, t.layer_norm(h, blk.ffn_norm), blk.ffn, residual=h)


def _build_embeddings(t, a, emb: Embeddings):
    feat, skips = build_resnet(t, a, emb.hybrid_model)
    h = t.map_to_tokens(t.conv1x1_bias(feat, emb.patch_embeddings))
    return t.dropout(t.add_position(h, emb.position_embeddings), emb.dropout.p), skips


def _build_decoder_block(t, x, skip, blk: DecoderBlock):
    N, H, W, C = x.shape
    Cs = skip.shape[3] if skip is not None else 0
    if skip is not None:
        assert skip.shape[1:3] == (2 * H, 2 * W), f"skip {skip.shape} vs upsampled {(2 * H, 2 * W)}"
    cat = t.alloc(N, 2 * H, 2 * W, C + Cs, device=x.raw.device)
    up = t.bilinear2x_into(x, cat[..., :C])
    xin = t.concat(cat, [up, t.copy_into(skip, cat[..., C:])]) if skip is not None else up
    x = t.conv_bn(xin, blk.conv1[0].weight, blk.conv1[1])
    return t.conv_bn(x, blk.conv2[0].weight, blk.conv2[1])


def _build_decoder(t, tokens, features, dec: DecoderCup, gh, gw):
    x = t.tokens_to_map(tokens, gh, gw)
    x = t.conv_bn(x, dec.conv_more[0].weight, dec.conv_more[1])
    for i, blk in enumerate(dec.blocks):
        skip = features[i] if (features is not None and i < dec.config.n_skip and i < len(features)) else None
        x = _build_decoder_block(t, x, skip, blk)
    return x


class VisionTransformer(nn.Module):
    # (decoder attribute, head attribute) per output; the multitask variants below share one encoder between several
    HEADS = (("decoder", "segmentation_head"),)

    def __init__(self, config, img_size=224, num_classes=21843, zero_head=False, vis=False, *, compute_dtype=None):
        super().__init__()
        if vis:
            raise NotImplementedError("vis=True (returning attention maps) is not supported by the fused attention kernel")
        self.num_classes = num_classes
        self.zero_head = zero_head
        self.classifier = config.classifier
        self.transformer = Transformer(config, img_size, vis)
        for dname, _ in self.HEADS:                      # reference registration order: all decoders, then all heads
            setattr(self, dname, DecoderCup(config))
        for _, hname in self.HEADS:
            setattr(self, hname, SegmentationHead(in_channels=config['decoder_channels'][-1],
                                                  out_channels=config['n_classes'], kernel_size=3))
        self.config = config
        self._compute_dtype = compute_dtype
        self._step = 0

    def _umi_dtype(self):
        return _resolve_dtype(self._compute_dtype)

    def forward(self, x):
        if x.size()[1] == 1:
            x = x.repeat(1, 3, 1, 1)                        # reference :387-388
        params = list(self.parameters())
        dtype = self._umi_dtype()
        N, _, H, W = x.shape
        cfg = self.config
        # dropout streams: host seed drawn once per model from torch's generator + a device-side step counter that also
        # advances when this forward is replayed from a captured HIP graph (umi.graph.dropout_seeds)
        step, seed_dev = G.dropout_seeds(self, x.device, self.training)

        def run(record, in_needs):
            tape = TUTape(dtype, training=self.training, record=record, seed=step, seed_dev=seed_dev,
                          loss_scale=G.default_loss_scale(dtype, N * H * W),
                          grad_sink=getattr(self, "_umi_grad_sink", None) if record else None,
                          pack_cache=G.pack_cache_of(self))
            a = tape.input_nchw(x, needs_grad=False)
            emb = self.transformer.embeddings
            h, skips = _build_embeddings(tape, a, emb)
            gh = gw = int(np.sqrt(h.shape[2]))
            assert gh * gw == h.shape[2]
            # data-parallel runs: when the backward pass comes back to this point the encoder's and decoders' deferred gradient
            # fills run and their buckets go on the wire, under the hybrid ResNet's backward pass (a no-op without a sink)
            tape.flush_mark()
            for blk in self.transformer.encoder.layer:
                h = _build_block(tape, h, blk, cfg)
            h = tape.layer_norm(h, self.transformer.encoder.encoder_norm)
            outs = []
            for dname, hname in self.HEADS:             # every decoder reads the same tokens and skips: their gradients add up
                y = _build_decoder(tape, h, skips, getattr(self, dname), gh, gw)
                head = getattr(self, hname)[0]
                outs.append(tape.conv_bias(y, head.weight, head.bias, out_dtype=torch.float32, pad=head.padding[0]))
            tape.finish_forward()
            if len(outs) == 1:
                return tape, [a], outs[0], tape.output_nchw_plain(outs[0])
            return tape, [a], tuple(outs), tuple(tape.output_nchw_plain(o) for o in outs)

        record = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        return _TapeFunction.apply(run, record, 1, x, *params)

    def load_from(self, weights):
        """Load a JAX `.npz` ViT/R50 checkpoint (reference :394-441).  The reference's checkpoint file is not part of
        its tree; this mapping is therefore exercised by shape only."""
        from scipy import ndimage
        with torch.no_grad():
            emb = self.transformer.embeddings
            emb.patch_embeddings.weight.copy_(np2th(weights["embedding/kernel"], conv=True))
            emb.patch_embeddings.bias.copy_(np2th(weights["embedding/bias"]))
            enc = self.transformer.encoder
            enc.encoder_norm.weight.copy_(np2th(weights["Transformer/encoder_norm/scale"]))
            enc.encoder_norm.bias.copy_(np2th(weights["Transformer/encoder_norm/bias"]))
            posemb = np2th(weights["Transformer/posembed_input/pos_embedding"])
            new = emb.position_embeddings
            if posemb.size() == new.size():
                new.copy_(posemb)
            elif posemb.size()[1] - 1 == new.size()[1]:
                new.copy_(posemb[:, 1:])
            else:
                ntok = new.size(1)
                grid = posemb[0, 1:] if self.classifier == "seg" else posemb[0]
                gs_old, gs_new = int(np.sqrt(len(grid))), int(np.sqrt(ntok))
                logger.info("load_pretrained: position-embedding grid %s -> %s", gs_old, gs_new)
                grid = ndimage.zoom(grid.reshape(gs_old, gs_old, -1), (gs_new / gs_old, gs_new / gs_old, 1), order=1)
                new.copy_(np2th(grid.reshape(1, gs_new * gs_new, -1)))
            for uname, unit in enc.layer.named_children():
                unit.load_from(weights, n_block=uname)
            hm = emb.hybrid_model
            hm.root.conv.weight.copy_(np2th(weights["conv_root/kernel"], conv=True))
            hm.root.gn.weight.copy_(np2th(weights["gn_root/scale"]).view(-1))
            hm.root.gn.bias.copy_(np2th(weights["gn_root/bias"]).view(-1))
            for bname, block in hm.body.named_children():
                for uname, unit in block.named_children():
                    unit.load_from(weights, n_block=bname, n_unit=uname)


class VisionTransformerMultitask(VisionTransformer):
    """Reference vit_seg_modeling.py:444-522: one Transformer, `decoder1/2` + `segmentation_head1/2`, returns
    (logits1, logits2).  The encoder runs once per step; both decoders sit on the same tape."""
    HEADS = (("decoder1", "segmentation_head1"), ("decoder2", "segmentation_head2"))


class VisionTransformerMultitaskEM(VisionTransformer):
    """Reference vit_seg_modeling.py:524-638: six decoders / heads over one encoder, returns six logit maps."""
    HEADS = tuple((f"decoder{i}", f"segmentation_head{i}") for i in range(1, 7))


CONFIGS = {
    'ViT-B_16': configs.get_b16_config(),
    'ViT-B_32': configs.get_b32_config(),
    'ViT-L_16': configs.get_l16_config(),
    'ViT-L_32': configs.get_l32_config(),
    'ViT-H_14': configs.get_h14_config(),
    'R50-ViT-B_16': configs.get_r50_b16_config(),
    'R50-ViT-L_16': configs.get_r50_l16_config(),
    'testing': configs.get_testing(),
}
