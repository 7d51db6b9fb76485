"""MI355X-native U-Net: drop-in for the reference `Model.py` (caki35/UNet-Torch).

Same class names, constructor signatures and `state_dict` keys as the reference
(`DoubleConv` Model.py:7-26, `Down` :29-47, `Up` :50-83, `OutConv` :86-92, `UNet` :95-169),
so `from Model import UNet`, `load_state_dict(torch.load('best.pt'))` and the reference
`Trainer` work unchanged.  The arithmetic does not go through torch.nn: `forward` builds a
tape of libunetmi HIP kernels (umi/graph.py) -- NHWC activations, BatchNorm statistics in
the conv epilogue, BN-apply+ReLU fused into the consumer's load, zero-copy skip
concatenation -- and a hand-written backward pass.  The nn.Conv2d / nn.BatchNorm2d children
only own parameters and buffers.

Build-specific, keyword-only extras (never positional, so reference call sites are
unaffected): `compute_dtype` = "fp16" (fp16 storage + fp32 accumulate, loss-scaled
gradients; default, env UMI_COMPUTE_DTYPE) or "fp32" (parity mode).

There is no CPU path here: inputs must live on the MI355X ("cuda" in PyTorch-ROCm naming).
"""
import os

import torch
import torch.nn as nn

from umi import graph as G


def _resolve_dtype(compute_dtype):
    name = compute_dtype or os.environ.get("UMI_COMPUTE_DTYPE", "fp16")
    if isinstance(name, torch.dtype):
        return name
    table = {"fp16": torch.float16, "float16": torch.float16, "half": torch.float16,
             "fp32": torch.float32, "float32": torch.float32}
    if name not in table:
        raise ValueError(f"compute_dtype must be fp16 or fp32, got {name!r}")
    return table[name]


class _TapeFunction(torch.autograd.Function):
    """Bridges a libunetmi tape into torch.autograd: one node for the whole block/network."""

    @staticmethod
    def forward(ctx, run, record, n_inputs, *tensors):
        inputs, params = tensors[:n_inputs], tensors[n_inputs:]
        tape, in_acts, out_act, out = run(record, [bool(t.requires_grad) for t in inputs])
        ctx.tape, ctx.in_acts, ctx.out_act, ctx.params, ctx.n_inputs = tape, in_acts, out_act, params, n_inputs
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gouts):
        tape = ctx.tape
        if tape is None or not tape.record:
            raise RuntimeError("backward called on a tape that was built without gradient recording")
        out_acts = ctx.out_act if isinstance(ctx.out_act, tuple) else (ctx.out_act,)
        for a, g in zip(out_acts, gouts):            # several heads (UNet_multitask): one seed per output
            tape.seed_grad_nchw(a, g)
        tape.backward()
        gin = [tape.input_grad_nchw(a) if need else None
               for a, need in zip(ctx.in_acts, ctx.needs_input_grad[3:3 + ctx.n_inputs])]
        gpar = []
        sink = tape.grad_sink
        for p, need in zip(ctx.params, ctx.needs_input_grad[3 + ctx.n_inputs:]):
            g = tape.param_grads.get(id(p))
            if g is None or not need:
                gpar.append(None)
                continue
            slot = sink.buffer_for(p) if sink is not None else None
            if (slot is not None and g[1].data_ptr() == slot.data_ptr() and p.dtype == torch.float32
                    and (p.grad is None or p.grad.data_ptr() == slot.data_ptr())):
                # Data-parallel runs: the gradient already sits in its bucket slot and the optimizer reads it there.  Handing a
                # VIEW of the bucket to autograd would make AccumulateGrad clone it -- one copy launch per parameter and step
                # (409 for a R50-ViT-B/16) of values that GradReducer.sync() then discards.  .grad becomes the slot itself
                # (semantics of zero_grad + backward; accumulating several backward passes into one step is not supported
                # under a gradient sink).
                p.grad = slot
                gpar.append(None)
            else:
                gpar.append(g[1].to(p.dtype))
        ctx.tape = None
        return (None, None, None, *gin, *gpar)


def _run_tape(module, inputs, build, tape_cls=None, dtype=None):
    """Run `build(tape, *input_acts) -> Act (or tuple of Acts)` for `module`; returns NCHW fp32 tensor(s).  `tape_cls`:
    G.Tape (default) or umi.graph_tu.TUTape for the TransUNet blocks."""
    params = [p for p in module.parameters()]
    dtype = dtype if dtype is not None else module._umi_dtype()
    N, _, H, W = inputs[0].shape
    tape_cls = tape_cls or G.Tape

    uses_dropout = any(isinstance(m_, nn.Dropout) for m_ in module.modules())
    seed, seed_dev = G.dropout_seeds(module, inputs[0].device, module.training and uses_dropout)

    def run(record, in_needs):
        tape = tape_cls(dtype, training=module.training, record=record,
                        loss_scale=G.default_loss_scale(dtype, N * H * W),
                        grad_sink=getattr(module, "_umi_grad_sink", None) if record else None,
                        pack_cache=G.pack_cache_of(module), seed=seed, seed_dev=seed_dev)
        acts = [tape.input_nchw(x, needs_grad=need) for x, need in zip(inputs, in_needs)]
        out_act = build(tape, *acts)
        tape.finish_forward()

        def nchw(o):
            if o.tx is None and o.raw.dtype == torch.float32:
                return tape.output_nchw_plain(o)
            from umi import ops
            return ops.materialize_nchw(o.raw, o.tx)
        out = tuple(nchw(o) for o in out_act) if isinstance(out_act, tuple) else nchw(out_act)
        return tape, acts, out_act, out

    # grad mode is off inside Function.forward, so decide here whether to record the tape
    record = torch.is_grad_enabled() and any(t.requires_grad for t in (*inputs, *params))
    return _TapeFunction.apply(run, record, len(inputs), *inputs, *params)


class _UmiModule(nn.Module):
    _compute_dtype = None

    def _umi_dtype(self):
        return _resolve_dtype(self._compute_dtype)


# ---- tape builders shared by the blocks and the full network -----------------------------------
def _build_double_conv(t, a, dc, out=None):
    seq = dc.double_conv
    y1 = t.conv_bn(a, seq[0].weight, seq[1])
    return t.conv_bn(y1, seq[3].weight, seq[4], out=out, input_exclusive=True)    # y1 has no other consumer


def _build_down(t, a, down, out=None):
    p = t.pool2(a)
    if down.dropout:                               # MaxPool2d -> Dropout(p) -> DoubleConv (reference Model.py:34-41)
        p = t.dropout(p, down.maxpool_conv[1].p)
    return _build_double_conv(t, p, down.maxpool_conv[-1], out=out)


def _build_up(t, x1, skip, cat, up):
    """`skip` already lives in cat[..., :C]; the transposed conv fills cat[..., C:]."""
    C = skip.shape[3]
    u = t.conv_transpose2x2(x1, up.up.weight, up.up.bias, cat[..., C:])
    c = t.concat(cat, [skip, u])
    if up.dropout_flag:                            # cat -> Dropout(p) -> DoubleConv (reference Model.py:79-83)
        c = t.dropout(c, up.dropout.p)
    return _build_double_conv(t, c, up.conv)


def _build_encoder(t, a, stages, f, into_concat=True):
    """inc + 4 x Down (reference Model.py:142-147).  Returns [(activation, concat buffer)] per level: every encoder output
    but the deepest is written straight into the lower channel half of the concat buffer its decoder stage will read
    (into_concat=False: plain outputs, for UNet_attention whose decoder concatenates the GATED skip instead)."""
    N, dev = a.shape[0], a.raw.device
    skips, cur = [], a
    for lvl, st in enumerate(stages):
        C = f * 2 ** lvl
        h, w = (cur.shape[1], cur.shape[2]) if lvl == 0 else (cur.shape[1] // 2, cur.shape[2] // 2)
        if into_concat and lvl < len(stages) - 1:
            cat = t.alloc(N, h, w, 2 * C, device=dev)
            out = cat[..., :C]
        else:
            cat, out = None, None
        cur = _build_double_conv(t, cur, st, out=out) if lvl == 0 else _build_down(t, cur, st, out=out)
        skips.append((cur, cat))
    return skips


def _build_decoder(t, skips, ups, own_buffers=False):
    """4 x Up from the deepest encoder output (reference Model.py:148-151).  own_buffers: a second decoder over the same
    encoder (UNet_multitask) gets its own concat buffers, the skips are copied into them."""
    y = skips[-1][0]
    for i, up in enumerate(ups):
        skip, cat = skips[len(skips) - 2 - i]
        if own_buffers:
            C = skip.shape[3]
            cat = t.alloc(*skip.shape[:3], 2 * C, device=skip.raw.device)
            skip = t.copy_into(skip, cat[..., :C])
        y = _build_up(t, y, skip, cat, up)
    return y


class DoubleConv(_UmiModule):
    """(convolution => [BN] => ReLU) * 2 -- reference Model.py:7-26."""

    def __init__(self, in_channels, out_channels, mid_channels=None, *, compute_dtype=None):
        super().__init__()
        mid_channels = mid_channels or out_channels
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(mid_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True))
        self._compute_dtype = compute_dtype

    def forward(self, x):
        return _run_tape(self, [x], lambda t, a: _build_double_conv(t, a, self))


class Down(_UmiModule):
    """MaxPool2d(2) [-> Dropout] -> DoubleConv -- reference Model.py:29-47."""

    def __init__(self, in_channels, out_channels, dropout=False, dropout_p=0.5, *, compute_dtype=None):
        super().__init__()
        layers = [nn.MaxPool2d(2)]
        if dropout:
            layers.append(nn.Dropout(p=dropout_p))
        layers.append(DoubleConv(in_channels, out_channels))
        self.maxpool_conv = nn.Sequential(*layers)
        self.dropout = dropout
        self._compute_dtype = compute_dtype

    def forward(self, x):
        return _run_tape(self, [x], lambda t, a: _build_down(t, a, self))


class Up(_UmiModule):
    """ConvTranspose2d(k2,s2) -> pad -> cat([skip, up]) [-> Dropout] -> DoubleConv -- Model.py:50-83."""

    def __init__(self, in_channels, out_channels, dropout_flag=False, dropout_p=0.5, *, compute_dtype=None):
        super().__init__()
        self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
        self.conv = DoubleConv(in_channels, out_channels)
        self.dropout_flag = dropout_flag
        if dropout_flag:
            self.dropout = nn.Dropout(p=dropout_p)
        self._compute_dtype = compute_dtype

    def forward(self, x1, x2):
        def build(t, a1, a2):
            N, H, W, C = a2.shape
            cat = t.alloc(N, H, W, C + self.up.out_channels, device=a2.raw.device)
            cat[..., :C].copy_(a2.raw)
            a2.raw = cat[..., :C]
            return _build_up(t, a1, a2, cat, self)
        return _run_tape(self, [x1, x2], build)


class OutConv(_UmiModule):
    """1x1 conv + bias -- reference Model.py:86-92."""

    def __init__(self, in_channels, out_channels, *, compute_dtype=None):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)
        self._compute_dtype = compute_dtype

    def forward(self, x):
        return _run_tape(self, [x], lambda t, a: t.conv_bias(a, self.conv.weight, self.conv.bias,
                                                             out_dtype=torch.float32))


class UNet(_UmiModule):
    """4-down/4-up U-Net -- reference Model.py:95-169 (same positional ctor arguments)."""

    def __init__(self, n_channels, n_classes, initial_feature_map=64, usa_cuda=True, dropout=False,
                 dropout_p=0.5, *, compute_dtype=None):
        super().__init__()
        self.usa_cuda = usa_cuda
        self.n_channels = {-2: 3, -1: 1}.get(n_channels, n_channels)
        self.n_classes = n_classes
        self.initial_feature_map = f = initial_feature_map
        self.dropout = dropout
        self.dropout_p = dropout_p
        self._compute_dtype = compute_dtype

        # Built stage by stage with `.apply(weights_init)` right after each one, like the reference
        # (Model.py:111-140), so the RNG stream -- and therefore `torch.manual_seed(s); UNet(...)` --
        # yields the same initial weights: kaiming-normal on nn.Conv2d only (ConvTranspose2d keeps
        # torch's default init).
        def stage(mod):
            mod.apply(self.weights_init)
            return mod
        self.inc = stage(DoubleConv(self.n_channels, f))
        self.down1 = stage(Down(f, f * 2, dropout, dropout_p))
        self.down2 = stage(Down(f * 2, f * 4, dropout, dropout_p))
        self.down3 = stage(Down(f * 4, f * 8, dropout, dropout_p))
        self.down4 = stage(Down(f * 8, f * 16, dropout, dropout_p))
        self.up1 = stage(Up(f * 16, f * 8, dropout, dropout_p))
        self.up2 = stage(Up(f * 8, f * 4, dropout, dropout_p))
        self.up3 = stage(Up(f * 4, f * 2, dropout, dropout_p))
        self.up4 = stage(Up(f * 2, f, dropout, dropout_p))
        self.outc = stage(OutConv(f, n_classes))

    def weights_init(self, m):
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight)

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != self.n_channels:
            raise ValueError(f"expected input [B,{self.n_channels},H,W], got {tuple(x.shape)}")

        def build(t, a):
            skips = _build_encoder(t, a, [self.inc, self.down1, self.down2, self.down3, self.down4],
                                   self.initial_feature_map)
            y = _build_decoder(t, skips, [self.up1, self.up2, self.up3, self.up4])
            return t.conv_bias(y, self.outc.conv.weight, self.outc.conv.bias, out_dtype=torch.float32)

        return _run_tape(self, [x], build)

    def use_checkpointing(self):
        return _no_checkpointing()


class UNet_multitask(_UmiModule):
    """One encoder, two decoders, two logit maps -- reference Model.py:172-254 (same ctor, same state_dict keys
    `up{1..4}_decod{1,2}`, `outc_decod{1,2}`; like the reference it builds Down/Up WITHOUT the dropout arguments).
    Both decoders run on one tape: the encoder activations are computed once and their gradients are the sum over the
    two decoders."""

    def __init__(self, n_channels, n_classes, initial_feature_map=64, usa_cuda=True, dropout=False,
                 dropout_p=0.5, *, compute_dtype=None):
        super().__init__()
        self.usa_cuda = usa_cuda
        self.n_channels = {-2: 3, -1: 1}.get(n_channels, n_channels)
        self.initial_feature_map = f = initial_feature_map
        self.dropout = dropout
        self.dropout_p = dropout_p
        self._compute_dtype = compute_dtype

        def stage(mod):
            mod.apply(self.weights_init)
            return mod
        self.inc = stage(DoubleConv(self.n_channels, f))
        self.down1 = stage(Down(f, f * 2))
        self.down2 = stage(Down(f * 2, f * 4))
        self.down3 = stage(Down(f * 4, f * 8))
        self.down4 = stage(Down(f * 8, f * 16))
        for d in (1, 2):                                 # reference order: all of decoder 1, then all of decoder 2
            for i in range(1, 5):
                setattr(self, f"up{i}_decod{d}", stage(Up(f * 2 ** (5 - i), f * 2 ** (4 - i))))
            setattr(self, f"outc_decod{d}", stage(OutConv(f, n_classes)))

    def weights_init(self, m):
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight)

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != self.n_channels:
            raise ValueError(f"expected input [B,{self.n_channels},H,W], got {tuple(x.shape)}")

        def build(t, a):
            skips = _build_encoder(t, a, [self.inc, self.down1, self.down2, self.down3, self.down4],
                                   self.initial_feature_map)
            outs = []
            for d in (1, 2):
                y = _build_decoder(t, skips, [getattr(self, f"up{i}_decod{d}") for i in range(1, 5)],
                                   own_buffers=(d == 2))
                head = getattr(self, f"outc_decod{d}").conv
                outs.append(t.conv_bias(y, head.weight, head.bias, out_dtype=torch.float32))
            return tuple(outs)

        return _run_tape(self, [x], build)


def _build_attention(t, q, x, ab, dest):
    """Attention_block.forward (reference Model.py:297-305): dest <- x * sigmoid(BN(psi(relu(BN(Wq up(q)) + BN(Wx x)))))."""
    N, h, w, Cq = q.shape
    if (2 * h, 2 * w) != tuple(x.shape[1:3]):
        raise ValueError(f"attention gate: upsampled query {2 * h}x{2 * w} != skip {x.shape[1]}x{x.shape[2]} "
                         "(the reference's Q1 + X1 fails the same way)")
    qu = t.conv_transpose2x2(q, ab.up.weight, ab.up.bias, t.alloc(N, 2 * h, 2 * w, Cq, device=q.raw.device),
                             bias_cancelled=True)       # up -> W_q (1x1) -> BatchNorm: see Tape.conv_transpose2x2
    q1 = t.conv_bn(qu, ab.W_q[0].weight, ab.W_q[1], pad=0, relu=False, bias=ab.W_q[0].bias)
    x1 = t.conv_bn(x, ab.W_x[0].weight, ab.W_x[1], pad=0, relu=False, bias=ab.W_x[0].bias)
    e = t.add_relu(q1, x1)
    p = t.conv_bn(e, ab.psi[0].weight, ab.psi[1], pad=0, relu=False, bias=ab.psi[0].bias)
    return t.gate(x, p, dest)


class Attention_block(_UmiModule):
    """Additive attention gate -- reference Model.py:265-305 (same ctor, same state_dict keys W_q.*, up.*, W_x.*, psi.*).
    The nn.Sigmoid / nn.ReLU children only keep the reference's module indices; the arithmetic is libunetmi kernels."""

    def __init__(self, C_q, C_x, C_hidden, *, compute_dtype=None):
        super().__init__()
        self.W_q = nn.Sequential(nn.Conv2d(C_q, C_hidden, kernel_size=1, stride=1, padding=0, bias=True),
                                 nn.BatchNorm2d(C_hidden))
        self.up = nn.ConvTranspose2d(C_q, C_q, kernel_size=2, stride=2)
        self.W_x = nn.Sequential(nn.Conv2d(C_x, C_hidden, kernel_size=1, stride=1, padding=0, bias=True),
                                 nn.BatchNorm2d(C_hidden))
        self.psi = nn.Sequential(nn.Conv2d(C_hidden, 1, kernel_size=1, stride=1, padding=0, bias=True),
                                 nn.BatchNorm2d(1), nn.Sigmoid())
        self.relu = nn.ReLU(inplace=True)
        self._compute_dtype = compute_dtype

    def forward(self, q, x):
        def build(t, aq, ax):
            N, H, W, C = ax.shape
            return _build_attention(t, aq, ax, self, t.alloc(N, H, W, C, device=ax.raw.device))
        return _run_tape(self, [q, x], build)


class UNet_attention(_UmiModule):
    """Attention U-Net -- reference Model.py:308-391: every skip connection is gated by the decoder state below it
    (attribute names `attenion{4..1}` as in the reference, typo included, so checkpoints load)."""

    def __init__(self, n_channels, n_classes, initial_feature_map=64, usa_cuda=True, dropout=False,
                 dropout_p=0.5, *, compute_dtype=None):
        super().__init__()
        self.usa_cuda = usa_cuda
        self.n_channels = {-2: 3, -1: 1}.get(n_channels, n_channels)
        self.initial_feature_map = f = initial_feature_map
        self.dropout = dropout
        self.dropout_p = dropout_p
        self._compute_dtype = compute_dtype

        def stage(mod):
            mod.apply(self.weights_init)
            return mod
        self.inc = stage(DoubleConv(self.n_channels, f))
        self.down1 = stage(Down(f, f * 2, dropout, dropout_p))
        self.down2 = stage(Down(f * 2, f * 4, dropout, dropout_p))
        self.down3 = stage(Down(f * 4, f * 8, dropout, dropout_p))
        self.down4 = stage(Down(f * 8, f * 16, dropout, dropout_p))
        # attention gates keep torch's default init (the reference never applies weights_init to them, Model.py:339-355)
        self.attenion4 = Attention_block(C_q=f * 16, C_x=f * 8, C_hidden=f * 4)
        self.attenion3 = Attention_block(C_q=f * 8, C_x=f * 4, C_hidden=f * 2)
        self.attenion2 = Attention_block(C_q=f * 4, C_x=f * 2, C_hidden=f)
        self.attenion1 = Attention_block(C_q=f * 2, C_x=f, C_hidden=int(f / 2))
        self.up1 = stage(Up(f * 16, f * 8, dropout, dropout_p))
        self.up2 = stage(Up(f * 8, f * 4, dropout, dropout_p))
        self.up3 = stage(Up(f * 4, f * 2, dropout, dropout_p))
        self.up4 = stage(Up(f * 2, f, dropout, dropout_p))
        self.outc = stage(OutConv(f, n_classes))

    def weights_init(self, m):
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight)

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != self.n_channels:
            raise ValueError(f"expected input [B,{self.n_channels},H,W], got {tuple(x.shape)}")

        def build(t, a):
            skips = _build_encoder(t, a, [self.inc, self.down1, self.down2, self.down3, self.down4],
                                   self.initial_feature_map, into_concat=False)
            y = skips[-1][0]
            gates = [self.attenion4, self.attenion3, self.attenion2, self.attenion1]
            for i, (att, up) in enumerate(zip(gates, [self.up1, self.up2, self.up3, self.up4])):
                skip = skips[3 - i][0]
                C = skip.shape[3]
                cat = t.alloc(*skip.shape[:3], 2 * C, device=skip.raw.device)
                gated = _build_attention(t, y, skip, att, cat[..., :C])       # written into the decoder's concat buffer
                y = _build_up(t, y, gated, cat, up)
            return t.conv_bias(y, self.outc.conv.weight, self.outc.conv.bias, out_dtype=torch.float32)

        return _run_tape(self, [x], build)

    def use_checkpointing(self):
        return _no_checkpointing()


def _no_checkpointing():
    # Dead code in the reference (Model.py:155-165: torch.utils.checkpoint is a module, the call
    # raises).  Saved activations are raw fp16 conv outputs here (~5.5 GB at B=16, 512^2 on a
    # 288 GB part), so activation checkpointing is unnecessary.
    raise NotImplementedError("use_checkpointing is broken in the reference and unnecessary here")
