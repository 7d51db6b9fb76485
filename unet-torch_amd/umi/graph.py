"""Tape engine: forward/backward of conv-BN-ReLU encoder-decoder graphs on libunetmi kernels.

Values are *lazily activated* NHWC tensors (`Act`): the stored tensor is the raw
convolution output and `tx` ([C,4] = {mean, gamma*rstd, beta, clamp}) is the BatchNorm
+ReLU the consumer applies while loading it.  Concatenation is free: producers write
into channel slices of one buffer and the concat `Act` just stacks the `tx` rows.

The backward pass is hand-written (no torch autograd inside the graph): every op
pushes a closure on the tape; `Tape.backward` replays them in reverse.  Gradients of
activations are NHWC in the compute dtype and carry the static loss scale (fp16 mode);
parameter gradients are produced in fp32, unscaled, in the parameter's own layout.

Reference semantics implemented here: Model.py:7-26 (DoubleConv), :29-47 (Down),
:50-83 (Up), :86-92 (OutConv); BatchNorm2d train/eval behaviour as torch.nn.
"""
import os

import torch

from . import lib as L
from . import ops


def _fuse_bnred():
    return os.environ.get("UMI_NO_BNRED_FUSION") != "1"          # tuning / A-B knob, read per call


def _defer_wgrad_reduce():
    return os.environ.get("UMI_NO_WGRAD_REDUCE_GROUP") != "1"    # A/B knob, read per call


def _fuse_bnapply(Ci):
    """Stage 3 of a conv's BatchNorm backward inside its weight-gradient kernel?  Every input-channel tile (64) of that kernel
    repeats the elementwise work on the gradient tile it stages, so the fusion only pays with a single tile: measured per
    layer at the bench shapes (tools/experiments/ab_bnapply.py, profiles/r02_wgrad_bnapply_ab.txt) -0.06..-0.09 ms at Ci = 64,
    +0.03..+0.24 ms from Ci = 128 up.  UMI_BNAPPLY_FUSION=0 / all: never / wherever the kernel applies, UMI_BNAPPLY_MAXCI=n: up to n
    input channels (A-B knobs; round 3, whole step on one box: 64 -> 23.19 ms, 128 -> 23.26 ms, all -> +1.3 ms)."""
    mode = os.environ.get("UMI_BNAPPLY_FUSION", "")
    return mode != "0" and (Ci <= int(os.environ.get("UMI_BNAPPLY_MAXCI", "64")) or mode == "all")


class Act:
    """Lazily-activated NHWC tensor.  `raw` is an [N,H,W,C] view, `tx` the consumer transform
    (None = consume as stored).  `grad` = d loss / d activated value (same layout/dtype)."""
    __slots__ = ("raw", "tx", "grad", "parts", "needs_grad", "bn_rstd", "bn_part", "bn_part_at", "gives")

    def __init__(self, raw, tx=None, parts=None, needs_grad=True):
        self.raw, self.tx, self.grad, self.parts, self.needs_grad = raw, tx, None, parts, needs_grad
        self.bn_rstd = None       # conv_bn outputs: 1/std of the batch statistics (BatchNorm backward)
        self.bn_part = None       # stage-1 partial sums of this layer's BatchNorm backward, when the kernel that wrote the
                                  # LAST contribution to `grad` produced them (its only consumer's data-gradient kernel,
                                  # conv_bn(..., input_exclusive=True), or the max-pool backward)
        self.gives = 0            # gradient contributions received so far; `bn_part` is valid iff bn_part_at == gives
        self.bn_part_at = -1

    @property
    def shape(self):
        return tuple(self.raw.shape)

    def full_tx(self):
        return self.tx


class Tape:
    def __init__(self, dtype, training, record, loss_scale=1.0, grad_sink=None, pack_cache=None, seed=0, seed_dev=None):
        self.dtype = dtype
        self._seed = int(seed)            # dropout streams: host part of the seed (per model, from torch's generator) ...
        self._seed_dev = seed_dev         # ... plus an int32 device counter the kernel adds in, so a step replayed from a
        self._drop_count = 0              # captured HIP graph (host part frozen) still draws fresh masks
        self.pack_cache = pack_cache      # ops.PackCache of the model that owns the parameters, or None (pack per use)
        self.grad_sink = grad_sink        # umi.ddp.GradReducer (flat buckets + overlapped all-reduce) or None
        self.training = training          # BatchNorm uses batch statistics
        self.record = record              # keep closures for backward
        self.loss_scale = float(loss_scale)
        # parameter gradients leave the kernels multiplied by `inv`: undoes the loss scale and, under
        # data parallelism, pre-divides by the world size so the all-reduce is a plain SUM
        self.inv = (grad_sink.grad_scale if grad_sink is not None else 1.0) / self.loss_scale
        self.steps = []
        self.param_grads = {}             # id(param) -> (param, grad tensor)
        self._deferred_unscale = []       # BatchNorm parameter gradients still carrying the loss scale
        self._deferred_zero = []          # gradients that are identically zero (conv bias under a BatchNorm): one fill at the end
        self._nbt = []
        self._wgrad_deferred = None
        # Deferred fills: a gradient buffer registered in param_grads whose CONTENT is only written (with `=`) by a grouped
        # launch at the end of the backward pass.  A later contribution to the same parameter (a module used twice in one
        # tape) must not be add_()-ed onto the still unfilled buffer: it is parked and added after the flush.
        self._pending_fill = []           # [lo, hi) address ranges of such buffers (gradients may be views into them)
        self._late_adds = []              # (destination, addend) pairs applied after every deferred fill has run
        self._late_ready = []             # parameters whose bucket slot (gradient sink) is a deferred fill: mark_ready after it
        self._inputs = []

    # ---- helpers -----------------------------------------------------------------------
    def alloc(self, N, H, W, C, dtype=None, zero=False, device=None):
        f = torch.zeros if zero else torch.empty
        return f((N, H, W, C), dtype=dtype or self.dtype, device=device)

    def _pack(self, kind, weight, wf, k8):
        """Kernel-layout copy of a convolution weight: from the model's PackCache when `weight` is a parameter."""
        c = self.pack_cache
        if c is not None and isinstance(weight, torch.nn.Parameter) and weight.dtype == torch.float32:
            return c.get(kind, weight, self.dtype, k8)
        return ops.PACKERS[kind](wf, self.dtype, k8=k8)

    def _defer_list(self, weight, gw):
        """The list a weight gradient's split-K reduction is deferred to (and `gw` marked as filled at the flush), or None."""
        if self._wgrad_deferred is None:
            return None
        self._mark_deferred_fill(gw)
        return self._wgrad_deferred

    def _mark_deferred_fill(self, *bufs):
        for b in bufs:
            self._pending_fill.append((b.data_ptr(), b.data_ptr() + b.numel() * b.element_size()))

    def _is_pending_fill(self, t):
        a = t.data_ptr()
        return any(lo <= a < hi for lo, hi in self._pending_fill)

    def _new_pgrad(self, p):
        """fp32 tensor the wgrad kernel writes into: a slot of the reducer's flat bucket when present."""
        v = self.grad_sink.buffer_for(p) if self.grad_sink is not None else None
        return v if v is not None else torch.empty_like(p, dtype=torch.float32)

    def _set_pgrad(self, p, g):
        key = id(p)
        if key in self.param_grads:
            dst = self.param_grads[key][1]
            if self.grad_sink is not None and self.grad_sink.buffer_for(p) is not None:
                # its bucket may already be on the wire (mark_ready at the first use)
                raise NotImplementedError("a parameter used twice in one tape under a gradient sink (umi.ddp.GradReducer)")
            if self._is_pending_fill(dst) or self._is_pending_fill(g):
                self._late_adds.append((dst, g))
            else:
                dst.add_(g)
            return
        v = self.grad_sink.buffer_for(p) if self.grad_sink is not None else None
        late = False
        if v is not None:
            late = self._is_pending_fill(g)
            if v.data_ptr() != g.data_ptr():
                if late:
                    raise RuntimeError("deferred gradient fills must target the sink's bucket slot (Tape._new_pgrad)")
                v.copy_(g)
            g = v
        self.param_grads[key] = (p, g)
        if v is not None:
            if late:
                self._late_ready.append(p)          # the slot is written by a grouped launch: ready once that has been issued
            else:
                self.grad_sink.mark_ready(p)

    def _set_pgrad_scaled(self, p, g, scale):
        """_set_pgrad(p, g * scale) with the product written straight into the sink's bucket slot when there is one (one
        launch instead of a multiply and a copy)."""
        v = self.grad_sink.buffer_for(p) if self.grad_sink is not None else None
        if v is not None and id(p) not in self.param_grads and v.shape == g.shape:
            torch.mul(g, scale, out=v)
            self._set_pgrad(p, v)
        else:
            self._set_pgrad(p, g * scale)

    def _give(self, act: Act, g):
        """Route gradient `g` (NHWC view) to `act`; concat acts forward slices to their parts."""
        if act.parts is not None:
            for part, c0, c1 in act.parts:
                self._give(part, g[..., c0:c1])
            return
        if not act.needs_grad:
            return
        act.gives += 1
        if act.grad is None:
            act.grad = g
        else:
            act.grad.add_(g)

    def _accumulate_target(self, act: Act, src, R, S, stride, pad):
        """The tensor a conv-shaped data gradient of `act` can be ADDED into by the kernel itself, or None.  An activation
        with two consumers (attention gate: g and x, reference Model.py:268-289) receives its second contribution this way
        instead of through a fresh tensor and an add pass; same rounding (stored fp16 + fp16)."""
        if (act.parts is not None or not act.needs_grad or act.grad is None or act.grad.dtype != torch.float16
                or os.environ.get("UMI_NO_DGRAD_ACCUMULATE") == "1"):
            return None
        return act.grad if ops.conv_accumulate_ok(src, act.grad, R, S, stride, pad) else None

    def finish_forward(self):
        """Host-side bookkeeping batched at the end of the forward pass (one launch instead of one per layer)."""
        if self._nbt:
            torch._foreach_add_(self._nbt, 1)
            self._nbt = []

    # ---- graph inputs / outputs --------------------------------------------------------
    def input_nchw(self, x: torch.Tensor, needs_grad=False):
        if not x.is_cuda:
            raise RuntimeError("unet-torch_amd: the HIP path needs the input on device 'cuda' (MI355X); "
                               "no CPU fallback exists in the product path")
        N, C, H, W = x.shape
        raw = torch.empty((N, H, W, C), dtype=self.dtype, device=x.device)
        if C == 1 and x.is_contiguous():
            raw.view(-1).copy_(x.view(-1))             # one channel: NCHW and NHWC are the same bytes -- a flat cast (the strided
                                                       # copy below takes 81 us for the bench's 16 x 1 x 512 x 512 batch, this 10)
        else:
            raw.copy_(x.permute(0, 2, 3, 1))           # NCHW -> NHWC + dtype cast in one pass
        a = Act(raw, None, needs_grad=needs_grad)
        self._inputs.append(a)
        return a

    def input_nhwc(self, raw, tx=None, needs_grad=False):
        a = Act(raw, tx, needs_grad=needs_grad)
        self._inputs.append(a)
        return a

    # ---- ops ---------------------------------------------------------------------------
    def conv_bn(self, a: Act, weight, bn, out=None, stride=1, pad=1, input_exclusive=False, relu=True, bias=None):
        """Conv2d(bias=False) -> BatchNorm2d -> ReLU, the last two deferred to the consumer.
        input_exclusive: this conv is the ONLY consumer of `a` (DoubleConv's second conv): its data-gradient kernel may then
        also emit stage 1 of `a`'s BatchNorm backward reduction (one pass over the gradient tensor less).
        relu=False: BatchNorm only (the W_q / W_x / psi branches of the attention gate, reference Model.py:268-289).
        bias: the conv's bias when a BatchNorm follows it (same place).  In training mode BatchNorm subtracts it again, so
        it is never added to the stored tensor: it only shifts running_mean, and its gradient is zero; in eval mode it is
        folded into the consumer transform."""
        Co, Ci, R, S = weight.shape
        N, H, W, Ca = a.shape
        assert Ca == Ci, f"conv expects {Ci} input channels, got {Ca}"
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
        if out is None:
            out = self.alloc(N, Ho, Wo, Co, device=a.raw.device)
        wf = weight.detach().float()
        if (not self.training and not self.record and _eval_fold() and self.dtype == torch.float16 and (R, S, stride, pad) == (3, 3, 1, 1)
                and ops.conv_plan(a.raw, out, 3, 3, 1, 1)[0] == 1):
            # inference (reference test_mc3serousv5.py:877-887): BatchNorm from running statistics + ReLU applied by the conv's
            # own epilogue, the tensor is stored activated -- no statistics pass, no transform in the consumers
            tx, _ = ops.eval_bn_tx(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
            if bias is not None:
                tx[:, 2] += tx[:, 1] * bias.detach().float()
            if not relu:
                tx[:, 3] = ops.NEG_INF
            if ops.conv3x3_fwd_act(a.raw, a.tx, self._pack("conv_fwd", weight, wf, True), tx, out):
                return Act(out, None)
        # the pointwise matrix-core kernel has no statistics epilogue: a 1x1 conv that feeds a BatchNorm (attention gates) runs
        # it without statistics and takes them in a separate HBM-bound pass over its (small: C_hidden channels) output; where
        # neither applies (odd channel counts) the generic kernel produces both
        part, flags, two_pass = None, 0, False
        if self.training and (R, S) == (1, 1):
            two_pass = (self.dtype == torch.float16 and Co % 8 == 0 and
                        ops.conv_plan(a.raw, out, R, S, stride, pad, 0, False)[0] == 1)
            # Co <= 8 (the gate's 1-channel psi conv): the narrow-output kernel, which has the statistics epilogue itself
            narrow = self.dtype == torch.float16 and Co <= 8
            flags = 0 if (two_pass or narrow) else L.CONV_FORCE_GENERIC
        res = ops.conv_fwd(a.raw, a.tx, lambda lay: self._pack("conv_fwd", weight, wf, bool(lay)), None, out,
                           R, S, stride, pad, want_stats=self.training and not two_pass, flags=flags)
        part = ops.bn_stats(out) if two_pass else res
        if self.training and part is None:
            raise RuntimeError("conv_bn: no BatchNorm statistics were produced for this layer")
        if self.training:
            mom = bn.momentum if bn.momentum is not None else 0.1
            tx, rstd = ops.bn_finalize(part, Co, N * Ho * Wo, bn.weight.detach(), bn.bias.detach(), bn.eps, mom,
                                       bn.running_mean if bn.track_running_stats else None,
                                       bn.running_var if bn.track_running_stats else None)
            if bn.track_running_stats and bn.num_batches_tracked is not None:
                self._nbt.append(bn.num_batches_tracked)          # += 1 for all layers in one launch (finish_forward)
            if bias is not None and bn.track_running_stats:
                bn.running_mean.add_(bias.detach().float(), alpha=mom)      # batch mean of (y + bias) = mean(y) + bias
        else:
            tx, rstd = ops.eval_bn_tx(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
            if bias is not None:
                tx[:, 2] += tx[:, 1] * bias.detach().float()
        if not relu:
            tx[:, 3] = ops.NEG_INF
        o = Act(out, tx)
        o.bn_rstd = rstd
        if self.record:
            def bwd():
                if o.grad is None:
                    return
                if not self.training:
                    raise RuntimeError("backward through BatchNorm in eval mode is not supported by the HIP path")
                inv = self.inv
                # stage-1 rows from the producer of o.grad are only valid if nothing was added to o.grad after them
                partials = o.bn_part if o.bn_part_at == o.gives else None
                # 3x3 layers on the matrix-core weight-gradient kernel: stage 3 of the BatchNorm backward (the elementwise
                # pass that turns o.grad into d(raw conv output)) is done by that kernel's producer waves
                fuse = (_fuse_bnapply(Ci) and self.dtype == torch.float16 and (R, S, stride, pad) == (3, 3, 1, 1)
                        and Ci % 8 == 0 and Co % 8 == 0)
                # the network's first conv (<= 4 input channels that take no gradient): only its weight gradient reads dz, and the
                # narrow-input kernel forms it on the fly (no apply pass, dz never stored)
                stem_fuse = (not fuse and self.dtype == torch.float16 and (R, S, stride, pad) == (3, 3, 1, 1) and Ci <= 4
                             and Co % 8 == 0 and not _wants_grad(a) and os.environ.get("UMI_NO_STEM_BNAPPLY") != "1")
                dbeta, dgamma = ops.bn_bwd(o.grad, out, tx, rstd, partials=partials, apply=not (fuse or stem_fuse))
                o.bn_part = None
                if self.grad_sink is None and inv != 1.0:
                    # un-scale all BatchNorm parameter gradients with one batched multiply at the end of the backward
                    # pass (36 tiny launches otherwise); with a gradient sink they must be final before they are handed over
                    self._deferred_unscale += [dgamma, dbeta]
                    self._set_pgrad(bn.weight, dgamma)
                    self._set_pgrad(bn.bias, dbeta)
                else:
                    self._set_pgrad_scaled(bn.weight, dgamma, inv)
                    self._set_pgrad_scaled(bn.bias, dbeta, inv)
                gw = self._new_pgrad(weight)
                if fuse:
                    dz = self.alloc(N, Ho, Wo, Co, device=out.device)
                    if ops.conv_wgrad_bnapply(a.raw, a.tx, o.grad, out, tx, rstd, dbeta, dgamma, dz, gw, Ci * R * S, R * S, 1,
                                              inv, R, S, stride, pad):
                        o.grad = dz
                    else:
                        fuse = False
                        ops.bn_bwd_apply(o.grad, out, tx, rstd, dbeta, dgamma)
                if stem_fuse:
                    if not ops.conv_wgrad_bnapply(a.raw, a.tx, o.grad, out, tx, rstd, dbeta, dgamma, None, gw, Ci * R * S, R * S, 1,
                                                  inv, R, S, stride, pad):
                        stem_fuse = False
                        ops.bn_bwd_apply(o.grad, out, tx, rstd, dbeta, dgamma)
                if not fuse and not stem_fuse:
                    ops.conv_wgrad(a.raw, a.tx, o.grad, None, gw, Ci * R * S, R * S, 1, inv, R, S, stride, pad,
                                   defer=self._defer_list(weight, gw))
                self._set_pgrad(weight, gw)
                if bias is not None:
                    gb = self._new_pgrad(bias)
                    # d/d bias of BatchNorm(conv + bias) vanishes identically: zero-filled with the other such gradients by ONE
                    # multi-tensor launch at the end of the backward pass (18 fills per U-Net step otherwise); with a gradient
                    # sink the slot must be final before it is handed over
                    if self.grad_sink is None:
                        self._deferred_zero.append(gb)
                        self._mark_deferred_fill(gb)     # a second use of the bias in this tape parks its addition behind the fill
                    else:
                        gb.zero_()
                    self._set_pgrad(bias, gb)
                if _wants_grad(a):
                    if stride != 1:
                        raise NotImplementedError("dgrad for strided conv_bn")
                    tgt = self._accumulate_target(a, o.grad, R, S, 1, R - 1 - pad)
                    if tgt is not None:
                        ops.conv_fwd(o.grad, None, lambda lay: self._pack("conv_dgrad", weight, wf, bool(lay)), None, tgt,
                                     R, S, 1, R - 1 - pad, flags=L.CONV_ACCUMULATE)
                        a.gives += 1
                        return
                    dx = self.alloc(N, H, W, Ci, device=out.device)
                    part = None
                    if (input_exclusive and _fuse_bnred() and a.grad is None and a.parts is None and a.bn_rstd is not None
                            and a.tx is not None and (R, S, pad) == (3, 3, 1) and self.dtype == torch.float16):
                        part = ops.conv_dgrad_bnred(o.grad, self._pack("conv_dgrad", weight, wf, True), dx, a.raw, a.tx,
                                                    a.bn_rstd)
                    if part is not None:
                        a.bn_part, a.bn_part_at = part, a.gives + 1      # valid after the _give below and until the next one
                    else:
                        ops.conv_fwd(o.grad, None, lambda lay: self._pack("conv_dgrad", weight, wf, bool(lay)), None, dx,
                                     R, S, 1, R - 1 - pad)
                    self._give(a, dx)
            self.steps.append(bwd)
        return o

    def conv_bias(self, a: Act, weight, bias, out_dtype=None, pad=0):
        """Conv2d with bias, output consumed as stored (OutConv / segmentation head)."""
        Co, Ci, R, S = weight.shape
        N, H, W, Ca = a.shape
        assert Ca == Ci
        out = self.alloc(N, H + 2 * pad - R + 1, W + 2 * pad - S + 1, Co, dtype=out_dtype or self.dtype,
                         device=a.raw.device)
        wf0 = weight.detach().float()
        ops.conv_fwd(a.raw, a.tx, lambda lay: self._pack("conv_fwd", weight, wf0, bool(lay)),
                     bias.detach().float() if bias is not None else None, out, R, S, 1, pad)
        o = Act(out, None)
        if self.record:
            def bwd():
                if o.grad is None:
                    return
                inv = self.inv
                g = o.grad
                gw = self._new_pgrad(weight)
                dx, part, fused_w = None, None, False
                if (_wants_grad(a) and (R, S, pad) == (1, 1, 0) and _fuse_bnred() and os.environ.get("UMI_NO_HEAD_BNRED") != "1"
                        and a.grad is None and a.parts is None and a.bn_rstd is not None and a.tx is not None
                        and self.dtype == torch.float16 and g.dtype == torch.float16):
                    # the head's data gradient is the only contribution to the last DoubleConv's gradient: stage 1 of that layer's
                    # BatchNorm backward rides on it, and so does the head's own weight gradient (its input is the activated tensor
                    # the kernel reads for the reduction): one pass over the 537 MB tensors at the bench shape instead of three
                    dx = self.alloc(N, H, W, Ci, device=out.device)
                    wf = weight.detach().float()
                    want_w = os.environ.get("UMI_NO_HEAD_WGRAD_FUSION") != "1"
                    part = ops.head_dgrad_bnred(g, self._pack("conv_dgrad", weight, wf, False), dx, a.raw, a.tx, a.bn_rstd,
                                                dW=gw if want_w else None, out_scale=inv)
                    fused_w = part is not None and want_w
                if not fused_w:
                    ops.conv_wgrad(a.raw, a.tx, g, None, gw, Ci * R * S, R * S, 1, inv, R, S, 1, pad,
                                   defer=self._defer_list(weight, gw))
                self._set_pgrad(weight, gw)
                if bias is not None:
                    gb = self._new_pgrad(bias)
                    ops.colsum(g, gb, inv)
                    self._set_pgrad(bias, gb)
                if _wants_grad(a):
                    if dx is None:
                        dx = self.alloc(N, H, W, Ci, device=out.device)
                    if part is not None:
                        a.bn_part, a.bn_part_at = part, a.gives + 1
                    else:
                        wf = weight.detach().float()
                        ops.conv_fwd(g, None, lambda lay: self._pack("conv_dgrad", weight, wf, bool(lay)), None, dx,
                                     R, S, 1, R - 1 - pad)
                    self._give(a, dx)
            self.steps.append(bwd)
        return o

    def pool2(self, a: Act):
        """MaxPool2d(2) of the activated tensor; result is stored activated."""
        N, H, W, C = a.shape
        out = self.alloc(N, H // 2, W // 2, C, device=a.raw.device)
        ops.pool2_fwd(a.raw, a.tx, out)
        o = Act(out, None)
        if self.record:
            def bwd():
                if o.grad is None or not _wants_grad(a):
                    return
                if a.parts is not None:
                    raise NotImplementedError("pool2 backward into a concat input")
                acc = a.grad is not None
                if not acc:
                    a.grad = self.alloc(N, H, W, C, device=out.device)
                part = None
                if (a.bn_rstd is not None and a.tx is not None and self.dtype == torch.float16 and _fuse_bnred()
                        and os.environ.get("UMI_NO_POOLRED_FUSION") != "1"):
                    # in the encoders built here the pool is the first consumer recorded after the layer it pools, so its
                    # backward is the LAST contribution to that layer's gradient: it can emit stage 1 of the BatchNorm
                    # backward reduction on the way (conv_bn's backward checks `gives` and ignores the rows otherwise)
                    part = ops.pool2_bwd_bnred(o.grad, a.raw, a.tx, a.bn_rstd, a.grad, acc)
                if part is None:
                    ops.pool2_bwd(o.grad, a.raw, a.tx, a.grad, acc)
                a.gives += 1
                a.bn_part, a.bn_part_at = part, a.gives
            self.steps.append(bwd)
        return o

    def conv_transpose2x2(self, a: Act, weight, bias, dest, out_hw=None, bias_cancelled=False):
        """ConvTranspose2d(k=2,s=2)+bias written into `dest` (a channel slice of a concat
        buffer, spatial size = the skip's), centred like F.pad in reference Model.py:69-73.
        bias_cancelled: the output feeds only an unpadded 1x1 conv followed by a training-mode BatchNorm (the attention gate's
        `up` -> `W_q`): the bias is a per-channel constant that the BatchNorm subtracts again, its gradient vanishes identically
        and is written as zeros instead of a column sum over the upsampled gradient."""
        Cin, Cout = weight.shape[:2]
        N, h, w, Ca = a.shape
        assert Ca == Cin
        _, Hd, Wd, Cd = dest.shape
        assert Cd == Cout
        dY, dX = Hd - 2 * h, Wd - 2 * w
        if dY < 0 or dX < 0:
            raise NotImplementedError("skip smaller than the upsampled map (negative F.pad) is not supported")
        oy, ox = dY // 2, dX // 2
        if dY or dX:
            dest.zero_()
        wf = weight.detach().float()
        ops.conv_fwd(a.raw, a.tx, lambda lay: self._pack("convT_fwd", weight, wf, bool(lay)),
                     bias.detach().float() if bias is not None else None, dest, 2, 2, 2, 0,
                     flags=L.CONV_UPSAMPLE2, up_offset=(oy, ox))
        o = Act(dest, None)
        if self.record:
            def bwd():
                if o.grad is None:
                    return
                inv = self.inv
                g = o.grad
                if dY or dX:
                    g = g[:, oy:oy + 2 * h, ox:ox + 2 * w, :].contiguous()
                gw = self._new_pgrad(weight)
                dlist = self._defer_list(weight, gw)
                fused = False
                if bias is not None:
                    gb = self._new_pgrad(bias)
                    if bias_cancelled and self.training and not (dY or dX):
                        gb.zero_()
                    elif os.environ.get("UMI_NO_CONVT_BIAS_FUSION") != "1" and ops.convT_wgrad_bias(g, a.raw, a.tx, gw, gb, inv,
                                                                                                    defer=dlist):
                        fused = True                     # the bias gradient rode on the weight gradient's pass over g
                    else:
                        ops.colsum(g, gb, inv)
                    self._set_pgrad(bias, gb)
                # dW[ci][co][t] = sum_p act(a)[p][ci] * g[2p+t][co]: a wgrad with the roles of x and dy swapped
                if not fused:
                    ops.conv_wgrad(g, None, a.raw, a.tx, gw, Cout * 4, 4, 1, inv, 2, 2, 2, 0, defer=dlist)
                self._set_pgrad(weight, gw)
                if _wants_grad(a):
                    tgt = self._accumulate_target(a, g, 2, 2, 2, 0)
                    if tgt is not None:
                        ops.conv_fwd(g, None, lambda lay: self._pack("convT_dgrad", weight, wf, bool(lay)), None, tgt,
                                     2, 2, 2, 0, flags=L.CONV_ACCUMULATE)
                        a.gives += 1
                        return
                    dx = self.alloc(N, h, w, Cin, device=dest.device)
                    part = None
                    if (_fuse_bnred() and os.environ.get("UMI_NO_CONVT_BNRED") != "1" and a.grad is None and a.parts is None
                            and a.bn_rstd is not None and a.tx is not None and self.dtype == torch.float16 and g.dtype == torch.float16):
                        # the transposed conv is the only consumer of the DoubleConv below it: stage 1 of that layer's BatchNorm
                        # backward rides on its data gradient
                        part = ops.conv_gather_bnred(g, self._pack("convT_dgrad", weight, wf, True), dx, a.raw, a.tx, a.bn_rstd,
                                                     2, 2, 2, 0)
                    if part is not None:
                        a.bn_part, a.bn_part_at = part, a.gives + 1
                    else:
                        ops.conv_fwd(g, None, lambda lay: self._pack("convT_dgrad", weight, wf, bool(lay)), None, dx,
                                     2, 2, 2, 0)
                    self._give(a, dx)
            self.steps.append(bwd)
        return o

    def dropout(self, a: Act, p):
        """nn.Dropout(p) of the *activated* `a` (reference Model.py:37 after the max-pool, :80-81 after the concat).  The
        consumer transform of `a`, if any, is applied inside the kernel; the result is stored activated.  Mask from the
        library's own counter-based stream: seed = a per-model value drawn once from torch's CPU generator (so
        torch.manual_seed reproduces a run) + the index of this dropout in the forward + a device-side step counter."""
        if not self.training or p <= 0.0:
            return a
        from . import ops_tu
        N, H, W, C = a.shape
        out = self.alloc(N, H, W, C, device=a.raw.device)
        mask = torch.empty(N * H * W * C, dtype=torch.uint8, device=a.raw.device)
        self._drop_count += 1
        ops_tu.dropout(a.raw, out, mask, False, p, (self._seed * 7919 + self._drop_count) & 0x7FFFFFFF, a.tx,
                       seed_dev=self._seed_dev)
        o = Act(out, None)
        if self.record:
            def bwd():
                if o.grad is None or not _wants_grad(a):
                    return
                dx = self.alloc(N, H, W, C, device=out.device)
                ops_tu.dropout(o.grad, dx, mask, True, p, 0)
                self._give(a, dx)
            self.steps.append(bwd)
        return o

    def copy_into(self, a: Act, dest):
        """A second home for `a` (same lazily-activated value, stored again in `dest`): lets one skip tensor sit in the
        concat buffers of two decoders (reference Model.py:244-254, UNet_multitask).  Pure data movement."""
        assert a.parts is None and tuple(dest.shape) == a.shape
        dest.copy_(a.raw)
        o = Act(dest, a.tx, needs_grad=a.needs_grad)
        if self.record:
            def bwd():
                if o.grad is not None and _wants_grad(a):
                    self._give(a, o.grad)
            self.steps.append(bwd)
        return o

    def add_relu(self, a: Act, b: Act):
        """relu(a + b) of two lazily-activated values, stored activated (attention gate, reference Model.py:302)."""
        N, H, W, C = a.shape
        assert b.shape == a.shape
        out = self.alloc(N, H, W, C, device=a.raw.device)
        ops.add2_relu(a.raw, a.tx, b.raw, b.tx, out)
        o = Act(out, None)
        if self.record:
            def bwd():
                if o.grad is None:
                    return
                da = self.alloc(N, H, W, C, device=out.device)      # two tensors: each branch's BatchNorm backward
                db = self.alloc(N, H, W, C, device=out.device)      # rewrites its gradient in place
                ops.add2_relu_bwd(o.grad, out, da, db)
                self._give(a, da)
                self._give(b, db)
            self.steps.append(bwd)
        return o

    def gate(self, x: Act, p: Act, dest):
        """dest <- x * sigmoid(p), p one channel broadcast over x's channels (reference Model.py:303-304); stored activated."""
        N, H, W, C = x.shape
        assert p.shape == (N, H, W, 1) and tuple(dest.shape) == x.shape
        ops.gate(x.raw, x.tx, p.raw, p.tx, dest)
        o = Act(dest, None)
        if self.record:
            def bwd():
                if o.grad is None:
                    return
                dx = self.alloc(N, H, W, C, device=dest.device)
                dp = self.alloc(N, H, W, 1, device=dest.device)
                ops.gate_bwd(o.grad, x.raw, x.tx, p.raw, p.tx, dx, dp)
                self._give(x, dx)
                self._give(p, dp)
            self.steps.append(bwd)
        return o

    def concat(self, buf, acts):
        """`acts` were produced into consecutive channel slices of `buf`."""
        txs, parts, c0 = [], [], 0
        for a in acts:
            C = a.shape[3]
            assert a.raw.data_ptr() == buf[..., c0:c0 + C].data_ptr(), "concat part is not a slice of the buffer"
            txs.append(a.tx if a.tx is not None else ops.passthrough_tx_const(C, buf.device))
            parts.append((a, c0, c0 + C))
            c0 += C
        assert c0 == buf.shape[3]
        if all(a.tx is None for a in acts):            # every part is stored activated (inference): nothing to apply on load
            return Act(buf, None, parts=parts)
        return Act(buf, torch.cat(txs, 0).contiguous(), parts=parts)

    # ---- outputs -----------------------------------------------------------------------
    def output_nchw_plain(self, a: Act):
        """Plain (tx-free) fp32 NHWC act -> NCHW contiguous fp32 torch tensor."""
        assert a.tx is None
        return a.raw.permute(0, 3, 1, 2).contiguous().float()

    def seed_grad_nchw(self, a: Act, g_nchw: torch.Tensor):
        N, C, H, W = g_nchw.shape
        g = g_nchw.permute(0, 2, 3, 1)
        if self.loss_scale != 1.0:
            g = g * self.loss_scale
        a.grad = torch.empty((N, H, W, C), dtype=self.dtype, device=g_nchw.device)
        a.grad.copy_(g)

    # Deferred gradient fills under a gradient sink: the U-Net tape turns them off (every weight gradient goes to its bucket
    # slot as the pass proceeds, the bucket's all-reduce overlaps the rest of the backward pass); TUTape keeps them and flushes
    # at marked points (flush_mark), so whole groups of layers still cost one launch.
    _defer_under_sink = False

    def _flush_deferred(self):
        """Run every deferred fill recorded so far, then the parked second contributions, then hand the filled bucket slots
        to the gradient sink."""
        if self._wgrad_deferred:
            ops.wgrad_reduce_flush(self._wgrad_deferred)     # the split-K reductions of all layers, 16 per launch
        if self._deferred_zero:
            torch._foreach_zero_(self._deferred_zero)        # (a deferred fill like the others: before the parked additions)
            self._deferred_zero = []
        self._finish_param_grads()
        for dst, src in self._late_adds:                    # second uses of a parameter whose first gradient was a deferred fill
            dst.add_(src)
        self._late_adds, self._pending_fill = [], []
        for p in self._late_ready:
            self.grad_sink.mark_ready(p)
        self._late_ready = []

    def backward(self):
        self._wgrad_deferred = [] if ((self.grad_sink is None or self._defer_under_sink) and _defer_wgrad_reduce()) else None
        for step in reversed(self.steps):
            step()
        self.steps = []
        self._flush_deferred()
        self._wgrad_deferred = None
        if self._deferred_unscale:
            torch._foreach_mul_(self._deferred_unscale, self.inv)
            self._deferred_unscale = []
        if self.grad_sink is not None:
            self.grad_sink.finish()

    def _finish_param_grads(self):
        """Hook: parameter-gradient work batched at the end of the backward pass (TUTape: StdConv2d standardisation)."""

    def input_grad_nchw(self, a: Act):
        if a.grad is None:
            return None
        g = a.grad.permute(0, 3, 1, 2).float()
        if self.loss_scale != 1.0:
            g = g / self.loss_scale
        return g.contiguous()


def _eval_fold():
    import os
    return os.environ.get("UMI_NO_EVAL_FOLD") != "1"


def dropout_seeds(module, device, training):
    """(host seed, device step counter) for the dropout streams of `module`'s forward.  The host part is drawn ONCE per model
    from torch's CPU generator (reproducible under torch.manual_seed, different between the seeds of a sweep and between a
    run and its resumption); the device counter advances by a device op on every training forward, also under HIP-graph
    replay, where anything computed on the host is frozen at capture time."""
    if not training or device.type != "cuda":
        return 0, None
    if getattr(module, "_drop_base", None) is None:
        module._drop_base = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    ctr = getattr(module, "_drop_step", None)
    if ctr is None or ctr.device != device:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("first training forward inside a HIP-graph capture; run one eager step first")
        ctr = module._drop_step = torch.zeros(1, dtype=torch.int32, device=device)
    ctr.add_(1)
    return module._drop_base, ctr


def pack_cache_of(module):
    """The module's cache of kernel-layout weight copies (ops.PackCache); stale entries (optimizer step, load_state_dict)
    are re-packed here, in one launch, before the tape runs.  UMI_NO_PACK_CACHE=1 packs per use (A/B switch)."""
    import os
    if os.environ.get("UMI_NO_PACK_CACHE") == "1":
        return None
    c = module.__dict__.get("_umi_pack_cache")
    if c is None:
        c = module.__dict__["_umi_pack_cache"] = ops.PackCache()
    c.refresh()
    return c


def _wants_grad(a: Act):
    if a.parts is not None:
        return any(_wants_grad(p) for p, _, _ in a.parts)
    return a.needs_grad


def default_loss_scale(dtype, n_pixels):
    """Static loss scale for fp16 gradients: activations' gradients are O(1/(N*H*W)), far below
    fp16's normal range, so scale them to ~2^-4 (power of two: exact, undone in fp32)."""
    if dtype != torch.float16:
        return 1.0
    import math
    return float(2 ** max(0, int(math.floor(math.log2(max(n_pixels, 1)))) - 3))
