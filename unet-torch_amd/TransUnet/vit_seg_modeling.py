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

from Model import _TapeFunction, _resolve_dtype
from umi import graph as G
from umi.graph_tu import TUTape

from . import vit_seg_configs as configs
from .vit_seg_modeling_resnet_skip import ResNetV2, build_resnet, np2th  # noqa: F401

logger = logging.getLogger(__name__)

_JAX = dict(q="MultiHeadDotProductAttention_1/query", k="MultiHeadDotProductAttention_1/key",
            v="MultiHeadDotProductAttention_1/value", o="MultiHeadDotProductAttention_1/out",
            fc0="MlpBlock_3/Dense_0", fc1="MlpBlock_3/Dense_1", ln0="LayerNorm_0", ln2="LayerNorm_2")


def _no_standalone(name):
    raise NotImplementedError(f"{name} runs inside VisionTransformer.forward (HIP tape); no standalone forward")


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
        _no_standalone("Attention")


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
        _no_standalone("Mlp")


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
        _no_standalone("Embeddings")


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
        _no_standalone("Block")

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
        _no_standalone("Encoder")


class Transformer(nn.Module):
    def __init__(self, config, img_size, vis):
        super().__init__()
        self.embeddings = Embeddings(config, img_size=img_size)
        self.encoder = Encoder(config, vis)

    def forward(self, input_ids):
        _no_standalone("Transformer")


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
        _no_standalone("DecoderBlock")


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
        _no_standalone("DecoderCup")


# ---- tape builders -------------------------------------------------------------------------------------------------------
def _build_block(t, h, blk: Block, cfg):
    heads = cfg.transformer["num_heads"]
    x = t.layer_norm(h, blk.attention_norm)
    ctx = t.qkv_attention(x, blk.attn.query, blk.attn.key, blk.attn.value, heads)   # attn_dropout rate is 0.0 in every config
    if cfg.transformer["attention_dropout_rate"] > 0 and t.training:
        raise NotImplementedError("attention-probability dropout > 0 is not supported (all reference configs use 0.0)")
    a = t.linear(ctx, blk.attn.out.weight, blk.attn.out.bias)
    h = t.add(a, h)
    x = t.layer_norm(h, blk.ffn_norm)
    x = t.dropout(t.gelu(t.linear(x, blk.ffn.fc1.weight, blk.ffn.fc1.bias)), cfg.transformer["dropout_rate"])
    x = t.dropout(t.linear(x, blk.ffn.fc2.weight, blk.ffn.fc2.bias), cfg.transformer["dropout_rate"])
    return t.add(x, h)


def _build_decoder(t, tokens, features, dec: DecoderCup, gh, gw):
    x = t.tokens_to_map(tokens, gh, gw)
    x = t.conv_bn(x, dec.conv_more[0].weight, dec.conv_more[1])
    for i, blk in enumerate(dec.blocks):
        skip = features[i] if (features is not None and i < dec.config.n_skip) else None
        N, H, W, C = x.shape
        Cs = skip.shape[3] if skip is not None else 0
        if skip is not None:
            assert skip.shape[1:3] == (2 * H, 2 * W), f"skip {skip.shape} vs upsampled {(2 * H, 2 * W)}"
        cat = t.alloc(N, 2 * H, 2 * W, C + Cs, device=x.raw.device)
        up = t.bilinear2x_into(x, cat[..., :C])
        xin = t.concat(cat, [up, t.copy_into(skip, cat[..., C:])]) if skip is not None else up
        x = t.conv_bn(xin, blk.conv1[0].weight, blk.conv1[1])
        x = t.conv_bn(x, blk.conv2[0].weight, blk.conv2[1])
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
            feat, skips = build_resnet(tape, a, emb.hybrid_model)
            gh, gw = feat.shape[1], feat.shape[2]
            h = tape.map_to_tokens(tape.conv1x1_bias(feat, emb.patch_embeddings))
            h = tape.dropout(tape.add_position(h, emb.position_embeddings), cfg.transformer["dropout_rate"])
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
