"""Thin tensor-level wrappers over the libunetmi C ABI.

PyTorch is used for device memory and streams only; every arithmetic op below is a
HIP kernel in libunetmi.so.  Tensors are NHWC views `[N, H, W, C]` whose last dim is
contiguous and whose pixel stride (`stride(2)`) may exceed C (channel slice of a
concat buffer).
"""

import torch

from . import lib as L

NEG_INF = float("-inf")


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.UMI_F32
    if t.dtype == torch.float16:
        return L.UMI_F16
    raise TypeError(f"libunetmi supports float32/float16 storage, got {t.dtype}")


def torch_dtype(code: int):
    return torch.float32 if code == L.UMI_F32 else torch.float16


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("libunetmi ops need tensors on the MI355X (device 'cuda'); "
                               "there is no CPU fallback in the product path")


def _nhwc(t: torch.Tensor):
    """Validate an NHWC view; returns (N, H, W, C, ld)."""
    if t.dim() != 4 or (t.stride(3) != 1 and t.shape[3] != 1):
        raise ValueError(f"expected NHWC view with contiguous channels, got {tuple(t.shape)} / {t.stride()}")
    N, H, W, C = t.shape
    ld = t.stride(2) if W > 1 else (t.stride(1) if H > 1 else (t.stride(0) // max(H * W, 1) if N > 1 else C))
    ok = ld >= C and (H == 1 or t.stride(1) == W * ld) and (N == 1 or t.stride(0) == H * W * ld)
    if not ok:
        raise ValueError(f"NHWC view must be dense in N,H,W with one pixel stride: {tuple(t.shape)} / {t.stride()}")
    return N, H, W, C, ld


def _ptr(t):
    return None if t is None else t.data_ptr()


def passthrough_tx(C, device):
    """Transform rows for channels that are consumed as stored (no BN, no ReLU); a fresh tensor the caller may edit."""
    t = torch.zeros(C, 4, dtype=torch.float32, device=device)
    t[:, 1] = 1.0
    t[:, 3] = NEG_INF
    return t


_passthrough = {}


def passthrough_tx_const(C, device):
    """The same rows as a shared constant, built once per (C, device): for callers that only read / concatenate them."""
    key = (C, str(device))
    t = _passthrough.get(key)
    if t is None and torch.cuda.is_current_stream_capturing():
        return passthrough_tx(C, device)               # not cached: would live in the graph's private pool
    # (a constant built by an eager warm-up step is ordinary memory and is what a later capture reads: 3 launches per use less)
    if t is None:
        t = _passthrough[key] = passthrough_tx(C, device)
    return t


def eval_bn_tx(weight, bias, running_mean, running_var, eps):
    """Consumer transform for BatchNorm in eval mode (running statistics) + ReLU."""
    rstd = torch.rsqrt(running_var.float() + eps)
    scale = weight.float() * rstd
    return torch.stack([running_mean.float(), scale, bias.float() - running_mean.float() * scale,
                        torch.zeros_like(rstd)], dim=1).contiguous(), rstd


# ------------------------------------------------------------------------------------------
def pack_kn(src: torch.Tensor, T, K, N, st, sk, sn, flip_t, dtype, Kpad=None, Npad=None, k8=False):
    _need_cuda(src)
    assert src.dtype == torch.float32 and src.is_contiguous()
    Kpad, Npad = Kpad or K, Npad or N
    out = torch.empty(T * Kpad * Npad, dtype=dtype, device=src.device)
    f = L.fn("umi_pack_kn8" if k8 else "umi_pack_kn")
    L.check(f(src.data_ptr(), out.data_ptr(), T, K, N, st, sk, sn, int(flip_t), Kpad, Npad, _dt(out), _stream()),
            "umi_pack_kn")
    return out


def pack_conv_fwd(w: torch.Tensor, dtype, k8=False):
    """OIHW -> [R*S][Ci][Co]  (k8: the MFMA kernels' [R*S][Ci/8][Co][8])."""
    Co, Ci, R, S = w.shape
    return pack_kn(w, R * S, Ci, Co, 1, R * S, Ci * R * S, False, dtype, k8=k8)


def pack_conv_dgrad(w: torch.Tensor, dtype, k8=False):
    """OIHW -> rotated/transposed [R*S][Co][Ci] so dgrad is a plain forward conv (stride 1)."""
    Co, Ci, R, S = w.shape
    return pack_kn(w, R * S, Co, Ci, 1, Ci * R * S, R * S, True, dtype, k8=k8)


def pack_convT_fwd(w: torch.Tensor, dtype, k8=False):
    """ConvTranspose2d [Cin][Cout][2][2] -> [4][Cin][Cout]."""
    Cin, Cout = w.shape[:2]
    return pack_kn(w, 4, Cin, Cout, 1, Cout * 4, 4, False, dtype, k8=k8)


def pack_convT_dgrad(w: torch.Tensor, dtype, k8=False):
    """ConvTranspose2d [Cin][Cout][2][2] -> [4][Cout][Cin] (stride-2 2x2 conv over d(up))."""
    Cin, Cout = w.shape[:2]
    return pack_kn(w, 4, Cout, Cin, 1, 4, Cout * 4, False, dtype, k8=k8)


def _pack_args(kind, shape):
    """(T, K, N, st, sk, sn, flip_t) of the weight packings above."""
    if len(shape) == 2:                              # nn.Linear weight [Co, Ci] = a 1x1 convolution
        shape = (shape[0], shape[1], 1, 1)
    if kind == "conv_fwd":
        Co, Ci, R, S = shape
        return R * S, Ci, Co, 1, R * S, Ci * R * S, 0
    if kind == "conv_dgrad":
        Co, Ci, R, S = shape
        return R * S, Co, Ci, 1, Ci * R * S, R * S, 1
    if kind == "conv_dgrad_strided":                 # [R*S][Co][Ci], taps not flipped (UMI_CONV_DGRAD_STRIDED)
        Co, Ci, R, S = shape
        return R * S, Co, Ci, 1, Ci * R * S, R * S, 0
    if kind == "bias":                               # a vector as a 1 x C matrix (get_cat: biases end to end)
        return 1, 1, shape[0], 0, 0, 1, 0
    if kind == "convT_fwd":
        Cin, Cout = shape[:2]
        return 4, Cin, Cout, 1, Cout * 4, 4, 0
    if kind == "convT_dgrad":
        Cin, Cout = shape[:2]
        return 4, Cout, Cin, 1, 4, Cout * 4, 0
    raise KeyError(kind)


def pack_conv_dgrad_strided(w, dtype, k8=False):
    """OIHW -> [R*S][Co][Ci] unflipped, for UMI_CONV_DGRAD_STRIDED (k8: the MFMA kernels' [R*S][Co/8][Ci][8])."""
    Co, Ci, R, S = w.shape
    return pack_kn(w, R * S, Co, Ci, 1, Ci * R * S, R * S, False, dtype, k8=k8)


PACKERS = {"conv_fwd": pack_conv_fwd, "conv_dgrad": pack_conv_dgrad, "convT_fwd": pack_convT_fwd,
           "convT_dgrad": pack_convT_dgrad, "conv_dgrad_strided": pack_conv_dgrad_strided}


class PackCache:
    """Kernel-layout fp16/fp32 copies of a model's convolution weights, owned by the top-level module.

    The copies are caches of the fp32 OIHW masters (never serialised).  An entry is valid while the parameter's
    `_version` and storage are unchanged, so inference loops pack once; after an optimizer step every entry is stale and
    `refresh()` re-packs all of them with ONE umi_pack_kn_multi launch (a U-Net has 45 such packings per step).

    Two kinds of derived operands live here too (TransUNet):
      * `wstd(w)`: the standardised copy of a StdConv2d weight (resnet_skip.py:20-23) and its per-channel rstd, persistent
        tensors recomputed for ALL such convs by one umi_wstd_fwd_multi launch in `refresh()`; their kernel layouts are
        ordinary entries keyed on the standardised tensor, so they ride in the same pack launch;
      * `get_cat(kind, [w...])`: several matrices packed side by side into ONE operand (the Q/K/V projections run as one
        GEMM): one entry per source, each filling its slice of a shared destination."""

    class _Ent:
        __slots__ = ("w", "args", "dst", "ver", "k8", "ptr", "ldn")

    class _Wstd:
        __slots__ = ("w", "ws", "rstd", "eps", "ver", "off")

    def __init__(self):
        self.ents = {}
        self.cats = {}
        self.wstds = {}
        self.wstd_total = 0        # elements of all registered StdConv2d weights (size of the backward's flat buffers)
        self._tables = {}          # tuple(entry keys) -> (device address of the descriptor table, total_blocks, n)

    _CAP = 1 << 18                     # bytes of descriptor-table space (a U-Net's table is 3 KB, a TransUNet's 20 KB)

    def _buffers(self):
        """Pinned staging + device space for the descriptor tables, allocated once and OUTSIDE any HIP-graph capture (the
        first forward of a model is never captured: GraphedStep warms up first).  Tables are appended and never rewritten,
        and travel to the device with umi_table_upload (a kernel reading the device-mapped pinned memory): inside a capture
        that is an ordinary kernel node which re-reads the same, unchanged, bytes at every replay."""
        if getattr(self, "_dev", None) is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("PackCache: first use inside a HIP-graph capture; run one eager forward first")
            self._dev = torch.empty(self._CAP, dtype=torch.uint8, device="cuda")
            self._host = torch.empty(self._CAP, dtype=torch.uint8).pin_memory()
            self._used = 0
        return self._host, self._dev

    def _upload(self, arr):
        import numpy as np
        raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        nbytes = (raw.size + 15) // 16 * 16
        host, dev = self._buffers()
        if self._used + nbytes > self._CAP:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("PackCache: descriptor-table space exhausted during a HIP-graph capture")
            torch.cuda.current_stream().synchronize()          # nothing in flight may still read the old tables
            self._tables.clear()
            self._used = 0
        off = self._used
        host.numpy()[off:off + raw.size] = raw
        L.check(L.fn("umi_table_upload")(host.data_ptr() + off, dev.data_ptr() + off, nbytes, _stream()), "umi_table_upload")
        self._used = off + nbytes
        return dev.data_ptr() + off

    def __deepcopy__(self, memo):      # a copied / pickled module starts with an empty cache
        return PackCache()

    def __reduce__(self):
        return (PackCache, ())

    @staticmethod
    def _ver(w):
        return (w._version, w.data_ptr())

    def _new_ent(self, kind, w, k8, dst, ptr, ldn):
        import weakref
        e = PackCache._Ent()
        e.w, e.args, e.k8, e.ver = weakref.ref(w), _pack_args(kind, w.shape), bool(k8), None
        e.dst, e.ptr, e.ldn = dst, ptr, ldn
        if not torch.cuda.is_current_stream_capturing():
            self._buffers()
        # (tables built for the previous entry set stay where they are: a captured graph may still replay them)
        return e

    def get(self, kind, w, dtype, k8):
        key = (id(w), kind, dtype, bool(k8))
        e = self.ents.get(key)
        if e is not None and e.w() is not w:
            e = None                                   # id() reused by another tensor
        if e is None:
            T, K, N = _pack_args(kind, w.shape)[:3]
            dst = torch.empty(T * K * N, dtype=dtype, device=w.device)
            e = self.ents[key] = self._new_ent(kind, w, k8, dst, dst.data_ptr(), 0)
        ver = self._ver(w)
        if e.ver != ver:
            T, K, N, st, sk, sn, flip = e.args
            f = L.fn("umi_pack_kn8" if e.k8 else "umi_pack_kn")
            L.check(f(w.data_ptr(), e.dst.data_ptr(), T, K, N, st, sk, sn, flip, K, N, _dt(e.dst), _stream()),
                    "umi_pack_kn")
            e.ver = ver
        return e.dst

    def get_cat(self, kind, ws, dtype, k8):
        """One packed operand from several source matrices: "conv_fwd" concatenates along the output channels (columns of the
        GEMM's B operand), "conv_dgrad" along its rows, "bias" 1-D vectors end to end.  1x1 / Linear weights only."""
        key = (tuple(id(w) for w in ws), kind, dtype, bool(k8))
        c = self.cats.get(key)
        if c is not None and any(e.w() is not w for e, w in zip(c[1], ws)):
            c = None
        if c is None:
            args = [_pack_args(kind, w.shape) for w in ws]
            assert all(a[0] == 1 for a in args), "get_cat packs single-tap (1x1 / Linear) weights"
            esz = torch.empty(0, dtype=dtype).element_size()
            if kind == "conv_dgrad":                                   # [Ktot(/8)][N]([8]): row blocks, one after the other
                assert len({a[2] for a in args}) == 1 and (not k8 or all(a[1] % 8 == 0 for a in args))
                offs, tot, ldn = [], 0, 0
                for a in args:
                    offs.append(tot)
                    tot += a[1] * a[2]
            else:                                                      # [K(/8)][Ntot]([8]): column slices of rows of length Ntot
                assert len({a[1] for a in args}) == 1
                ntot = sum(a[2] for a in args)
                offs, n0 = [], 0
                for a in args:
                    offs.append(n0 * (8 if k8 else 1))
                    n0 += a[2]
                tot, ldn = args[0][1] * ntot, ntot
            dst = torch.empty(tot, dtype=dtype, device=ws[0].device)
            ents = []
            for w, off in zip(ws, offs):
                e = self._new_ent(kind, w, k8, dst, dst.data_ptr() + off * esz, ldn)
                self.ents[(id(w), kind + "@cat", dtype, bool(k8), id(dst))] = e
                ents.append(e)
            c = self.cats[key] = (dst, ents)
        stale = [(None, e, w) for e, w in zip(c[1], ws) if e.ver != self._ver(w)]
        if stale:
            self._repack(stale)
        return c[0]

    def wstd(self, w, eps):
        """The persistent (standardised weight, rstd) pair of a StdConv2d parameter; current when `refresh()` ran after the
        last change of `w` (else recomputed here, by the single-conv kernel)."""
        import weakref
        e = self.wstds.get(id(w))
        if e is not None and e.w() is not w:
            e = None
        ver = self._ver(w)
        if e is None:
            assert w.dtype == torch.float32 and w.is_contiguous()
            e = self.wstds[id(w)] = PackCache._Wstd()
            e.w, e.eps, e.ver, e.off = weakref.ref(w), float(eps), None, self.wstd_total
            e.ws = torch.empty(w.shape, dtype=torch.float32, device=w.device)
            e.rstd = torch.empty(w.shape[0], dtype=torch.float32, device=w.device)
            self.wstd_total += (w.numel() + 63) // 64 * 64         # 256-byte aligned slots
        if e.ver != ver:
            L.check(L.fn("umi_wstd_fwd")(w.data_ptr(), e.ws.data_ptr(), e.rstd.data_ptr(), w.shape[0], w[0].numel(), e.eps,
                                         _stream()), "umi_wstd_fwd")
            torch.autograd.graph.increment_version(e.ws)           # its packed copies are stale now
            e.ver = ver
        return e

    def _wstd_table(self, ents, dws=None):
        import numpy as np
        tkey = ("wstd",) + tuple((id(e), e.w().data_ptr()) for e in ents) + (tuple(t.data_ptr() for t in dws) if dws else ())
        tab = self._tables.get(tkey)
        if tab is None:
            arr = np.zeros(len(ents), dtype=_WSTD_DESC)
            b0 = 0
            for i, e in enumerate(ents):
                w = e.w()
                arr[i] = (w.data_ptr(), e.ws.data_ptr(), e.rstd.data_ptr(), e.off, w.shape[0], w[0].numel(), e.eps, b0,
                          dws[i].data_ptr() if dws else 0)
                b0 += w.shape[0]
            tab = self._tables[tkey] = (self._upload(arr), b0, len(ents))
        return tab

    def wstd_bwd(self, ents, g_flat, dw_flat, dws=None):
        """Backward of the standardisation for every entry, from the gradients w.r.t. the standardised weights in g_flat (at
        e.off), in one launch: into dw_flat[e.off:...], or -- `dws`: one fp32 tensor per entry, e.g. the slots of a gradient
        sink's buckets -- straight into those."""
        dev_ptr, rows, n = self._wstd_table(ents, dws)
        L.check(L.fn("umi_wstd_bwd_multi")(dev_ptr, n, rows, g_flat.data_ptr(), dw_flat.data_ptr() if dw_flat is not None else None,
                                           _stream()), "umi_wstd_bwd_multi")

    def _repack(self, items):
        """items: [(key, entry, source tensor)] -> one umi_pack_kn_multi launch per storage dtype."""
        import numpy as np
        by_dtype = {}
        for it in items:
            by_dtype.setdefault(it[1].dst.dtype, []).append(it)
        blk = None
        for dtype, its in by_dtype.items():
            tkey = tuple((id(e), w.data_ptr()) for _, e, w in its)
            tab = self._tables.get(tkey)
            if tab is None:
                blk = blk or L.fn("umi_pack_block_elems")()
                arr = np.zeros(len(its), dtype=_PACK_DESC)
                b0 = 0
                for i, (_, e, w) in enumerate(its):
                    T, K, N, st, sk, sn, flip = e.args
                    arr[i] = (w.data_ptr(), e.ptr, st, sk, sn, T, K, N, flip, K, N, int(e.k8), b0, e.ldn, 0)
                    b0 += (T * K * N + blk - 1) // blk
                tab = self._tables[tkey] = (self._upload(arr), b0, len(its))
            dev_ptr, total, n = tab
            L.check(L.fn("umi_pack_kn_multi")(dev_ptr, n, total, L.UMI_F32 if dtype == torch.float32 else L.UMI_F16,
                                              _stream()), "umi_pack_kn_multi")
            for _, e, w in its:
                e.ver = self._ver(w)

    def refresh(self):
        """Bring every stale entry up to date: one umi_wstd_fwd_multi launch for the standardised weights, then one
        umi_pack_kn_multi launch per storage dtype for the kernel layouts."""
        # inside a HIP-graph capture everything is redone: the replayed graph must refresh the copies itself, whatever the
        # parameter versions were when it was captured (e.g. a captured forward + backward whose optimizer step runs outside
        # the graph: nothing is stale at capture time, everything is at the second replay)
        force = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        wst = []
        for key, e in list(self.wstds.items()):
            w = e.w()
            if w is None:
                del self.wstds[key]          # (its slot in the flat gradient buffers stays reserved)
            elif force or e.ver != self._ver(w):
                wst.append(e)
        if wst:
            dev_ptr, rows, n = self._wstd_table(wst)
            L.check(L.fn("umi_wstd_fwd_multi")(dev_ptr, n, rows, _stream()), "umi_wstd_fwd_multi")
            torch.autograd.graph.increment_version([e.ws for e in wst])
            for e in wst:
                e.ver = self._ver(e.w())
        stale = []
        for key, e in list(self.ents.items()):
            w = e.w()
            if w is None:
                del self.ents[key]
                continue
            if force or e.ver != self._ver(w):
                stale.append((key, e, w))
        if stale:
            self._repack(stale)


def _np_dtypes():
    import numpy as np
    pack = np.dtype([("src", "u8"), ("dst", "u8"), ("st", "i8"), ("sk", "i8"), ("sn", "i8"), ("T", "i4"), ("K", "i4"),
                     ("N", "i4"), ("flip", "i4"), ("Kpad", "i4"), ("Npad", "i4"), ("k8", "i4"), ("blk0", "i4"),
                     ("ldn", "i4"), ("pad", "i4")])
    opt = np.dtype([("p", "u8"), ("g", "u8"), ("s0", "u8"), ("s1", "u8"), ("n", "i8"), ("blk0", "i4"), ("pad", "i4")])
    wstd = np.dtype([("w", "u8"), ("ws", "u8"), ("rstd", "u8"), ("off", "i8"), ("Co", "i4"), ("K", "i4"), ("eps", "f4"),
                     ("blk0", "i4"), ("dw", "u8")])
    # sizeof(umi_pack_desc) / sizeof(umi_optim_desc) / sizeof(umi_wstd_desc)
    assert pack.itemsize == 80 and opt.itemsize == 48 and wstd.itemsize == 56
    return pack, opt, wstd


_PACK_DESC, OPTIM_DESC, _WSTD_DESC = _np_dtypes()


# ------------------------------------------------------------------------------------------
def conv_plan(x, y, R, S, stride, pad, flags=0, has_bias=False):
    """(layout, stat_rows) libunetmi will use for this conv: layout 1 = weights must be packed k8."""
    import ctypes
    N, H, W, Ci, ldx = _nhwc(x)
    _, _, _, Co, ldy = _nhwc(y)
    lay, rows = ctypes.c_int(0), ctypes.c_int(0)
    L.check(L.fn("umi_conv_fwd_plan")(N, H, W, Ci, Co, R, S, stride, pad, ldx, ldy, _dt(x), _dt(y), flags,
                                      int(has_bias), ctypes.addressof(lay), ctypes.addressof(rows)),
            "umi_conv_fwd_plan")
    return lay.value, rows.value


def conv_accumulate_ok(x, y, R, S, stride, pad, flags=0):
    """True when libunetmi can ADD this conv's result into y (flags | CONV_ACCUMULATE): only its pointwise / tap-gather
    matrix-core kernel does; the caller otherwise computes into a fresh tensor and adds."""
    N, H, W, Ci, ldx = _nhwc(x)
    _, _, _, Co, ldy = _nhwc(y)
    if (x.data_ptr() | y.data_ptr()) & 15:
        return False
    return L.fn("umi_conv_fwd_plan")(N, H, W, Ci, Co, R, S, stride, pad, ldx, ldy, _dt(x), _dt(y),
                                     flags | L.CONV_ACCUMULATE, 0, None, None) == 0


def conv_fwd(x, tx, wp, bias, y, R, S, stride, pad, want_stats=False, flags=0, up_offset=(0, 0)):
    """y <- conv(tx(x), wp) [+ bias]; returns the stats-partials tensor when want_stats.
    `wp` is a packed weight tensor or a callable(layout) -> packed tensor (see conv_plan)."""
    _need_cuda(x, y)
    N, H, W, Ci, ldx = _nhwc(x)
    N2, oH, oW, Co, ldy = _nhwc(y)
    assert N2 == N
    if flags & L.CONV_UPSAMPLE2:
        Ho, Wo = H, W
    else:
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
        assert (oH, oW) == (Ho, Wo), f"output view {oH}x{oW} != conv output {Ho}x{Wo}"
    part = None
    layout, rows = conv_plan(x, y, R, S, stride, pad, flags, bias is not None)
    if callable(wp):
        wp = wp(layout)
    elif layout != 0:
        raise ValueError("this conv takes the MFMA path: pass a callable so the weights get the k8 packing")
    if want_stats:
        part = torch.empty(rows * 2 * Co, dtype=torch.float32, device=x.device)
    if tx is not None:
        assert tx.shape == (Ci, 4) and tx.dtype == torch.float32 and tx.is_contiguous()
    L.check(L.fn("umi_conv_fwd")(x.data_ptr(), ldx, _ptr(tx), wp.data_ptr(), _ptr(bias), y.data_ptr(), ldy,
                                 _ptr(part), N, H, W, Ci, Co, R, S, stride, pad, Ho, Wo,
                                 up_offset[0], up_offset[1], oH, oW, _dt(x), _dt(y), flags, _stream()),
            "umi_conv_fwd")
    return part


def conv3x3_fwd_act(x, tx, wp8, out_tx, y):
    """Inference conv3x3 + this layer's BatchNorm/ReLU on store (y stored activated).  False when the shape is not on the
    matrix-core path (nothing was launched)."""
    N, H, W, Ci, ldx = _nhwc(x)
    _, _, _, Co, ldy = _nhwc(y)
    st = L.fn("umi_conv3x3_fwd_act")(x.data_ptr(), ldx, _ptr(tx), wp8.data_ptr(), out_tx.data_ptr(), y.data_ptr(), ldy,
                                     N, H, W, Ci, Co, _dt(x), _stream())
    if st == -2:
        return False
    L.check(st, "umi_conv3x3_fwd_act")
    return True


def bn_finalize(part, C, count, gamma, beta, eps, momentum, running_mean, running_var):
    rows = part.numel() // (2 * C)
    tx = torch.empty(C, 4, dtype=torch.float32, device=part.device)
    rstd = torch.empty(C, dtype=torch.float32, device=part.device)
    L.check(L.fn("umi_bn_finalize")(part.data_ptr(), rows, C, float(count), _ptr(gamma), _ptr(beta), eps, momentum,
                                    _ptr(running_mean), _ptr(running_var), tx.data_ptr(), rstd.data_ptr(),
                                    _stream()), "umi_bn_finalize")
    return tx, rstd


def pool2_fwd(x, tx, y):
    N, H, W, C, ldx = _nhwc(x)
    _, Ho, Wo, _, ldy = _nhwc(y)
    assert (Ho, Wo) == (H // 2, W // 2)
    L.check(L.fn("umi_pool2_fwd")(x.data_ptr(), ldx, _ptr(tx), y.data_ptr(), ldy, N, H, W, C, _dt(x), _stream()),
            "umi_pool2_fwd")


def pool2_bwd(dpool, x, tx, da, accumulate):
    N, H, W, C, ldx = _nhwc(x)
    _, _, _, _, lddp = _nhwc(dpool)
    _, _, _, _, ldda = _nhwc(da)
    L.check(L.fn("umi_pool2_bwd")(dpool.data_ptr(), lddp, x.data_ptr(), ldx, _ptr(tx), da.data_ptr(), ldda,
                                  int(accumulate), N, H, W, C, _dt(x), _stream()), "umi_pool2_bwd")


def bn_stats(y):
    """Statistics partials [rows][2][C] of a stored tensor (for bn_finalize), or None when the kernel does not apply."""
    N, H, W, C, ldy = _nhwc(y)
    M = N * H * W
    rows = L.fn("umi_bn_stats_rows")(M, C) if y.dtype == torch.float16 else 0
    if rows <= 0:
        return None
    part = torch.empty(rows * 2 * C, dtype=torch.float32, device=y.device)
    st = L.fn("umi_bn_stats")(y.data_ptr(), ldy, part.data_ptr(), M, C, _dt(y), _stream())
    if st == -2:
        return None
    L.check(st, "umi_bn_stats")
    return part


def pool2_bwd_bnred(dpool, x, tx, rstd, da, accumulate):
    """pool2_bwd + stage 1 of the BatchNorm backward of the pooled layer; returns the partial rows, or None when the fused
    kernel does not take this problem (the caller then runs pool2_bwd and the separate reduction)."""
    N, H, W, C, ldx = _nhwc(x)
    rows = L.fn("umi_pool2_bwd_bnred_stat_rows")(N, H, W, C)
    if rows <= 0 or x.dtype != torch.float16 or tx is None or rstd is None:
        return None
    part = torch.empty(rows * 2 * C, dtype=torch.float32, device=x.device)
    st = L.fn("umi_pool2_bwd_bnred")(dpool.data_ptr(), _nhwc(dpool)[4], x.data_ptr(), ldx, tx.data_ptr(), rstd.data_ptr(),
                                     da.data_ptr(), _nhwc(da)[4], int(accumulate), part.data_ptr(), N, H, W, C, _dt(x),
                                     _stream())
    if st == -2:                                   # UMI_ERR_UNSUPPORTED (alignment / strides): not an error
        return None
    L.check(st, "umi_pool2_bwd_bnred")
    return part


_ws_cache = {}
_ws_high = {}          # device -> largest workspace requested so far
_ws_pinned = []        # blocks referenced by captured graphs


def workspace(nbytes, device):
    """Grow-only scratch buffer per device+stream (split-K slabs, reduction partials)."""
    key = (device, torch.cuda.current_stream().cuda_stream)
    buf = _ws_cache.get(key)
    need = max(int(nbytes), _ws_high.get(device, 1 << 20))
    _ws_high[device] = need                     # a new stream (e.g. a graph-capture stream) starts at the size already seen
    if buf is None or buf.numel() < nbytes:
        if buf is not None and torch.cuda.is_current_stream_capturing():
            _ws_pinned.append(buf)              # kernels already captured into a HIP graph keep pointing at the old block
        buf = torch.empty(need, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def conv_dgrad_bnred(dy, wp8, da, ybn, txbn, rstd):
    """3x3 data gradient da <- conv(dy, rotated weights) that also emits stage 1 of the BatchNorm backward reduction of the
    layer whose raw output is `ybn` (the layer `da` belongs to).  Returns the partial-sum tensor, or None when the shape
    is not on the MFMA path (nothing was launched)."""
    N, H, W, Ci, lddy = _nhwc(dy)
    _, _, _, Co, ldda = _nhwc(da)
    ldybn = _nhwc(ybn)[4]
    lay, rows = conv_plan(dy, da, 3, 3, 1, 1)
    if lay != 1:
        return None
    part = torch.empty(rows * 2 * Co, dtype=torch.float32, device=dy.device)
    st = L.fn("umi_conv_dgrad_bnred")(dy.data_ptr(), lddy, wp8.data_ptr(), da.data_ptr(), ldda, ybn.data_ptr(), ldybn,
                                      txbn.data_ptr(), rstd.data_ptr(), part.data_ptr(), N, H, W, Ci, Co, _dt(dy), _stream())
    if st == -2:
        return None
    L.check(st, "umi_conv_dgrad_bnred")
    return part


def conv_gather_bnred(x, wp8, y, ybn, txbn, rstd, R, S, stride, pad, flags=0):
    """y <- conv(x, wp8) on the pointwise / tap-gather matrix-core kernel (no transform, no bias) plus stage 1 of the BatchNorm
    backward reduction of the layer whose raw output is `ybn` (umi_conv_gather_bnred).  Returns the partial-sum tensor, or None when
    the problem is not on that kernel (nothing was launched)."""
    N, H, W, Ci, ldx = _nhwc(x)
    _, Ho, Wo, Co, ldy = _nhwc(y)
    ldybn = _nhwc(ybn)[4]
    if x.dtype != torch.float16 or y.dtype != torch.float16 or ybn.dtype != torch.float16:
        return None
    if (x.data_ptr() | y.data_ptr() | ybn.data_ptr()) & 15:
        return None
    rows = L.fn("umi_conv_gather_bnred_rows")(N, H, W, Ci, Co, R, S, stride, pad, Ho, Wo, ldx, ldy, _dt(x), flags)
    if rows <= 0:
        return None
    part = torch.empty(rows * 2 * Co, dtype=torch.float32, device=x.device)
    st = L.fn("umi_conv_gather_bnred")(x.data_ptr(), ldx, wp8.data_ptr(), y.data_ptr(), ldy, ybn.data_ptr(), ldybn,
                                       txbn.data_ptr(), rstd.data_ptr(), part.data_ptr(), N, H, W, Ci, Co, R, S, stride, pad,
                                       Ho, Wo, _dt(x), flags, _stream())
    if st == -2:
        return None
    L.check(st, "umi_conv_gather_bnred")
    return part


def head_dgrad_bnred(dl, wp, da, ybn, txbn, rstd, dW=None, out_scale=1.0):
    """Data gradient of a narrow pointwise conv (OutConv: <= 8 logit channels) that also emits stage 1 of the BatchNorm backward
    reduction of the layer `da` belongs to (umi_head_dgrad_bnred).  Returns the partial-sum tensor, or None when the shape is not
    taken (nothing was launched).  `wp`: the generic [1][Ci][Co] packing of the transposed weight.
    dW ([logit channels][feature channels][1][1] fp32): the head's weight gradient from the same pass (umi_head_bwd_fused)."""
    N, H, W, Ci, lddl = _nhwc(dl)
    _, _, _, Co, ldda = _nhwc(da)
    ldybn = _nhwc(ybn)[4]
    P = N * H * W
    if dl.dtype != torch.float16 or da.dtype != torch.float16 or ybn.dtype != torch.float16:
        return None
    rows = L.fn("umi_head_dgrad_bnred_rows")(P, Ci, Co, ldda, _dt(dl))
    if rows <= 0:
        return None
    part = torch.empty(rows * 2 * Co, dtype=torch.float32, device=dl.device)
    if dW is not None:
        assert dW.dtype == torch.float32 and dW.is_contiguous() and dW.numel() == Ci * Co
        ws = workspace(L.fn("umi_head_bwd_fused_ws_bytes")(P, Ci, Co), dl.device)
        st = L.fn("umi_head_bwd_fused")(dl.data_ptr(), lddl, wp.data_ptr(), da.data_ptr(), ldda, ybn.data_ptr(), ldybn,
                                        txbn.data_ptr(), rstd.data_ptr(), part.data_ptr(), dW.data_ptr(), Co, 1, out_scale,
                                        ws.data_ptr(), ws.numel(), P, Ci, Co, _dt(dl), _stream())
        if st == -2:
            return None
        L.check(st, "umi_head_bwd_fused")
        return part
    st = L.fn("umi_head_dgrad_bnred")(dl.data_ptr(), lddl, wp.data_ptr(), da.data_ptr(), ldda, ybn.data_ptr(), ldybn,
                                      txbn.data_ptr(), rstd.data_ptr(), part.data_ptr(), P, Ci, Co, _dt(dl), _stream())
    if st == -2:
        return None
    L.check(st, "umi_head_dgrad_bnred")
    return part


def bn_bwd(da, y, tx, rstd, partials=None, apply=True):
    """In place: da <- dy.  Returns (sum_dz, sum_dzx) = (d beta, d gamma) (still loss-scaled).  `partials`: stage-1 rows
    already produced by conv_dgrad_bnred for this layer (the reduction pass over `da` is skipped).  apply=False: the sums
    only, `da` stays the gradient of the activated output (conv_wgrad_bnapply does stage 3)."""
    N, H, W, C, ldy = _nhwc(y)
    _, _, _, _, ldda = _nhwc(da)
    M = N * H * W
    nb = L.fn("umi_bn_bwd_ws_bytes")(M, C)
    ws = workspace(nb, y.device)
    sums = torch.empty(2, C, dtype=torch.float32, device=y.device)
    if partials is not None:
        L.check(L.fn("umi_bn_bwd_from_partials")(partials.data_ptr(), partials.numel() // (2 * C), C, sums[0].data_ptr(),
                                                 sums[1].data_ptr(), _stream()), "umi_bn_bwd_from_partials")
    else:
        L.check(L.fn("umi_bn_bwd_reduce")(da.data_ptr(), ldda, y.data_ptr(), ldy, tx.data_ptr(), rstd.data_ptr(),
                                          sums[0].data_ptr(), sums[1].data_ptr(), M, C, _dt(y), ws.data_ptr(),
                                          ws.numel(), _stream()), "umi_bn_bwd_reduce")
    if apply:
        bn_bwd_apply(da, y, tx, rstd, sums[0], sums[1])
    return sums[0], sums[1]


def bn_bwd_apply(da, y, tx, rstd, sum_dz, sum_dzx):
    N, H, W, C, ldy = _nhwc(y)
    L.check(L.fn("umi_bn_bwd_apply")(da.data_ptr(), _nhwc(da)[4], y.data_ptr(), ldy, tx.data_ptr(), rstd.data_ptr(),
                                     sum_dz.data_ptr(), sum_dzx.data_ptr(), N * H * W, C, _dt(y), _stream()),
            "umi_bn_bwd_apply")


def conv_wgrad_bnapply(x, txa, da, y, tx_bn, rstd, sum_dz, sum_dzx, dz, dW, s_co, s_ci, s_t, out_scale, R, S, stride, pad):
    """Weight gradient fused with stage 3 of the following BatchNorm's backward (umi_conv_wgrad_bnapply); False where the
    fused kernel does not apply (nothing was launched)."""
    N, H, W, Ci, ldx = _nhwc(x)
    _, Ho, Wo, Co, ldda = _nhwc(da)
    if (Ho, Wo) != (H, W) or da.dtype != torch.float16:
        return False
    assert dW.dtype == torch.float32 and dW.is_contiguous() and (dz is None or dz.shape == da.shape)
    nb = L.fn("umi_conv_wgrad_ws_bytes")(N, Ho, Wo, Ci, Co, R, S, _dt(x), 0)
    ws = workspace(nb, x.device)
    # dz None: the layer's input takes no gradient, nothing else reads dz (the network's first conv: formed on the fly, never stored)
    st = L.fn("umi_conv_wgrad_bnapply")(x.data_ptr(), ldx, _ptr(txa), da.data_ptr(), ldda, y.data_ptr(), _nhwc(y)[4],
                                        tx_bn.data_ptr(), rstd.data_ptr(), sum_dz.data_ptr(), sum_dzx.data_ptr(),
                                        _ptr(dz), _nhwc(dz)[4] if dz is not None else 0, dW.data_ptr(), s_co, s_ci, s_t, out_scale,
                                        N, H, W, Ci, Co, R, S, stride, pad, _dt(x), 0, ws.data_ptr(), ws.numel(), _stream())
    if st == -2:                                   # UMI_ERR_UNSUPPORTED: not an error, the caller runs the two passes
        return False
    L.check(st, "umi_conv_wgrad_bnapply")
    return True


class _WgPending(__import__("ctypes").Structure):          # mirrors umi_wgrad_pending
    import ctypes as _c
    _fields_ = [("part", _c.c_void_p), ("dW", _c.c_void_p), ("s_co", _c.c_long), ("s_ci", _c.c_long), ("s_t", _c.c_long),
                ("scale", _c.c_float), ("splits", _c.c_int), ("RS", _c.c_int), ("Ci", _c.c_int), ("Co", _c.c_int)]


def conv_wgrad(x, txa, dy, txb, dW, s_co, s_ci, s_t, out_scale, R, S, stride, pad, flags=0, defer=None):
    """defer: a list.  The final split-K reduction is then recorded in it (with the call's own partial-sum buffer, kept alive by
    the list) instead of launched; wgrad_reduce_flush(defer) runs all recorded reductions, 16 per launch."""
    import ctypes
    N, H, W, Ci, ldx = _nhwc(x)
    _, Ho, Wo, Co, lddy = _nhwc(dy)
    assert dW.dtype == torch.float32 and dW.is_contiguous()
    nb = L.fn("umi_conv_wgrad_ws_bytes")(N, Ho, Wo, Ci, Co, R, S, _dt(x), flags)
    if defer is None:
        ws = workspace(nb, x.device)
        L.check(L.fn("umi_conv_wgrad")(x.data_ptr(), ldx, _ptr(txa), dy.data_ptr(), lddy, _ptr(txb), dW.data_ptr(),
                                       s_co, s_ci, s_t, out_scale, N, H, W, Ci, Co, R, S, stride, pad, Ho, Wo,
                                       _dt(x), flags, ws.data_ptr(), ws.numel(), _stream()), "umi_conv_wgrad")
        return
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=x.device)
    pend = _WgPending()
    L.check(L.fn("umi_conv_wgrad_deferred")(x.data_ptr(), ldx, _ptr(txa), dy.data_ptr(), lddy, _ptr(txb), dW.data_ptr(),
                                            s_co, s_ci, s_t, out_scale, N, H, W, Ci, Co, R, S, stride, pad, Ho, Wo,
                                            _dt(x), flags, ws.data_ptr(), ws.numel(), ctypes.addressof(pend), _stream()),
            "umi_conv_wgrad_deferred")
    if pend.part:
        defer.append((pend, ws, dW))


def convT_wgrad_bias(g, y, txy, dW, dbias, out_scale, defer=None):
    """Weight and bias gradient of ConvTranspose2d(2,2) in one pass over the upsampled map's gradient `g` (umi_conv_wgrad_bias);
    `y`/`txy` = the transposed conv's input and its transform.  False where the matrix-core kernel does not take the problem
    (nothing was launched: run colsum + conv_wgrad).  defer: as in conv_wgrad."""
    import ctypes
    N, H, W, Ci, ldx = _nhwc(g)
    _, Ho, Wo, Co, lddy = _nhwc(y)
    if g.dtype != torch.float16 or (H, W) != (2 * Ho, 2 * Wo):
        return False
    assert dW.dtype == torch.float32 and dW.is_contiguous() and dbias.dtype == torch.float32 and dbias.numel() == Ci
    nb = L.fn("umi_conv_wgrad_ws_bytes")(N, Ho, Wo, Ci, Co, 2, 2, _dt(g), 0)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=g.device) if defer is not None else workspace(nb, g.device)
    pend = _WgPending()
    assert dW.numel() == Ci * Co * 4                 # [Co][Ci][2][2] = ConvTranspose2d's [in][out][2][2]: in = y's channels, out = g's
    st = L.fn("umi_conv_wgrad_bias")(g.data_ptr(), ldx, y.data_ptr(), lddy, _ptr(txy), dW.data_ptr(), Ci * 4, 4, 1, dbias.data_ptr(),
                                     out_scale, N, H, W, Ci, Co, Ho, Wo, _dt(g), 0, ws.data_ptr(), ws.numel(),
                                     ctypes.addressof(pend) if defer is not None else None, _stream())
    if st == -2:
        return False
    L.check(st, "umi_conv_wgrad_bias")
    if defer is not None and pend.part:
        defer.append((pend, ws, dW))
    return True


def wgrad_reduce_flush(defer):
    """Run the reductions recorded by conv_wgrad(defer=...) and empty the list."""
    import ctypes
    if defer:
        arr = (_WgPending * len(defer))(*[d[0] for d in defer])
        L.check(L.fn("umi_wgrad_reduce_group")(len(defer), ctypes.addressof(arr), _stream()), "umi_wgrad_reduce_group")
        del defer[:]


def conv_wgrad_group(xs, dys, dWs, s_co, s_ci, out_scale):
    """umi_conv_wgrad_group: the weight gradients of len(xs) pointwise layers of one shape in one launch; False where the
    matrix-core kernel does not apply (nothing was launched)."""
    import ctypes
    n = len(xs)
    N, H, W, Ci, ldx = _nhwc(xs[0])
    _, _, _, Co, lddy = _nhwc(dys[0])
    for x, dy, dW in zip(xs, dys, dWs):
        assert _nhwc(x) == (N, H, W, Ci, ldx) and _nhwc(dy) == (N, H, W, Co, lddy) and x.dtype == xs[0].dtype
        assert dW.dtype == torch.float32
    arr = ctypes.c_void_p * n
    ptrs = [arr(*[t.data_ptr() for t in ts]) for ts in (xs, dys, dWs)]
    st = L.fn("umi_conv_wgrad_group")(n, ctypes.cast(ptrs[0], ctypes.c_void_p), ldx, ctypes.cast(ptrs[1], ctypes.c_void_p), lddy,
                                      ctypes.cast(ptrs[2], ctypes.c_void_p), s_co, s_ci, out_scale, N * H * W, Ci, Co,
                                      _dt(xs[0]), _stream())
    if st == -2:
        return False
    L.check(st, "umi_conv_wgrad_group")
    return True


def colsum(x, out, out_scale):
    N, H, W, C, ldx = _nhwc(x)
    M = N * H * W
    ws = workspace(L.fn("umi_colsum_ws_bytes")(M, C), x.device)
    L.check(L.fn("umi_colsum")(x.data_ptr(), ldx, out.data_ptr(), out_scale, M, C, _dt(x), ws.data_ptr(),
                               ws.numel(), _stream()), "umi_colsum")


def colsum_group(xs, outs, out_scale):
    """outs[i] <- out_scale * column sums of xs[i] for tensors of one shape, two launches per 16; False where the grouped
    kernel does not apply (nothing was launched)."""
    import ctypes
    n = len(xs)
    N, H, W, C, ldx = _nhwc(xs[0])
    M = N * H * W
    if xs[0].dtype != torch.float16 or C % 8 or ldx % 8:
        return False
    for x, o in zip(xs, outs):
        assert _nhwc(x) == (N, H, W, C, ldx) and o.dtype == torch.float32 and o.numel() == C
    ws = workspace(min(n, 16) * L.fn("umi_colsum_ws_bytes")(M, C), xs[0].device)
    arr = ctypes.c_void_p * n
    px, po = arr(*[t.data_ptr() for t in xs]), arr(*[t.data_ptr() for t in outs])
    st = L.fn("umi_colsum_group")(n, ctypes.cast(px, ctypes.c_void_p), ldx, ctypes.cast(po, ctypes.c_void_p), out_scale, M, C,
                                  _dt(xs[0]), ws.data_ptr(), ws.numel(), _stream())
    if st == -2:
        return False
    L.check(st, "umi_colsum_group")
    return True


def add2_relu(a, txa, b, txb, y):
    """y <- max(txa(a) + txb(b), 0)  (attention gate, reference Model.py:302)."""
    N, H, W, C, lda = _nhwc(a)
    _, _, _, _, ldb = _nhwc(b)
    _, _, _, _, ldy = _nhwc(y)
    assert a.shape == b.shape == y.shape
    L.check(L.fn("umi_add2_relu_fwd")(a.data_ptr(), lda, _ptr(txa), b.data_ptr(), ldb, _ptr(txb), y.data_ptr(), ldy,
                                      N * H * W, C, _dt(a), _stream()), "umi_add2_relu_fwd")


def add2_relu_bwd(dy, y, da, db):
    N, H, W, C, lddy = _nhwc(dy)
    L.check(L.fn("umi_add2_relu_bwd")(dy.data_ptr(), lddy, y.data_ptr(), _nhwc(y)[4], da.data_ptr(), _nhwc(da)[4],
                                      db.data_ptr(), _nhwc(db)[4], N * H * W, C, _dt(dy), _stream()), "umi_add2_relu_bwd")


def gate(x, txx, p, txp, y):
    """y <- txx(x) * sigmoid(txp(p)), p: [N,H,W,1] (reference Model.py:303-304)."""
    N, H, W, C, ldx = _nhwc(x)
    assert tuple(p.shape) == (N, H, W, 1) and p.is_contiguous() and y.shape == x.shape
    L.check(L.fn("umi_gate_fwd")(x.data_ptr(), ldx, _ptr(txx), p.data_ptr(), _ptr(txp), y.data_ptr(), _nhwc(y)[4],
                                 N * H * W, C, _dt(x), _stream()), "umi_gate_fwd")


def gate_bwd(dy, x, txx, p, txp, dx, dp):
    N, H, W, C, ldx = _nhwc(x)
    assert dp.is_contiguous() and tuple(dp.shape) == (N, H, W, 1)
    L.check(L.fn("umi_gate_bwd")(dy.data_ptr(), _nhwc(dy)[4], x.data_ptr(), ldx, _ptr(txx), p.data_ptr(), _ptr(txp),
                                 dx.data_ptr(), _nhwc(dx)[4], dp.data_ptr(), N * H * W, C, _dt(x), _stream()), "umi_gate_bwd")


def materialize_nchw(x, tx):
    N, H, W, C, ldx = _nhwc(x)
    y = torch.empty(N, C, H, W, dtype=torch.float32, device=x.device)
    L.check(L.fn("umi_materialize_nchw")(x.data_ptr(), ldx, _ptr(tx), y.data_ptr(), N, H, W, C, _dt(x), _stream()),
            "umi_materialize_nchw")
    return y
