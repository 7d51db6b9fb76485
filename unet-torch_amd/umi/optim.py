"""Fused multi-tensor optimizers behind the torch.optim façade.

`SGD` / `Adam` subclass `torch.optim.SGD` / `torch.optim.Adam`: same constructor, `param_groups`, `state` and
`state_dict()` layout (`momentum_buffer`; `step` / `exp_avg` / `exp_avg_sq`), so the reference's call sites
(train.py:341-347 builds the optimizer, Trainer.py:719-725 drives it and rewrites `param_group['lr']` for the poly
schedule) work unchanged.  `step()` updates every parameter of a group with ONE libunetmi launch
(umi_optim_sgd_multi / umi_optim_adam_multi) that follows torch's operation order, instead of torch's 4-10 foreach
launches per group.

Parameters must live on the MI355X; there is no CPU path here (the CPU oracle uses torch.optim itself).
"""
import math

import numpy as np
import torch

from . import lib as L
from . import ops


def _bump(p):
    # the kernels write through raw pointers: tell autograd (and ops.PackCache) that the parameter changed
    torch.autograd.graph.increment_version(p)


class _Table:
    """Device descriptor table of one param group, rebuilt only when a pointer changes.

    The upload is stream-capture safe: the pinned staging buffers and the device buffer are allocated once (outside any
    capture: `umi.graphs.GraphedStep` warms up first) and reused, and the copy is a kernel (`umi_table_upload`) reading the
    device-mapped pinned buffer, i.e. an ordinary kernel node of the graph.  That node re-reads the pinned buffer at every
    replay, so an optimizer that was captured into a graph must not also be stepped eagerly afterwards (GraphedStep owns
    it)."""

    def __init__(self):
        self.key, self.dev, self.blocks, self.n = None, None, 0, 0
        self.host, self.events, self.turn = [None, None], [None, None], 0
        self.captured = set()                            # staging buffers a captured graph replays from

    def _ensure(self, nbytes):
        if self.dev is None or self.dev.numel() < nbytes:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("umi.optim: first optimizer step inside a HIP-graph capture; run a warm-up step first "
                                   "(umi.graphs.GraphedStep does)")
            cap = max(nbytes, 16384)
            self.dev = torch.empty(cap, dtype=torch.uint8, device="cuda")
            self.host = [torch.empty(cap, dtype=torch.uint8).pin_memory() for _ in range(2)]
            self.events = [None, None]

    def get(self, rows):
        key = tuple(rows)
        if key != self.key:
            blk = L.fn("umi_optim_block_elems")()
            arr = np.zeros(len(rows), dtype=ops.OPTIM_DESC)
            b0 = 0
            for i, (p, g, s0, s1, n) in enumerate(rows):
                arr[i] = (p, g, s0, s1, n, b0, 0)
                b0 += (n + blk - 1) // blk
            raw = arr.view(np.uint8).reshape(-1)                 # 48 B per row: a multiple of 16
            self._ensure(raw.size)
            capturing = torch.cuda.is_current_stream_capturing()
            self.turn ^= 1                               # two staging buffers: the previous upload may still be in flight
            if capturing:
                # a captured upload node re-reads ITS staging buffer at every replay: a third capture (a third batch shape)
                # would overwrite the buffer the first graph still replays from, and that graph would step the parameters
                # with another graph's gradient addresses.  Trainer(graph=True) captures two shapes (full and ragged batch).
                if self.turn in self.captured:
                    raise RuntimeError("umi.optim: more than two HIP-graph captures share this optimizer's descriptor table "
                                       "(two staging buffers); use one optimizer per set of captured shapes")
                self.captured.add(self.turn)
            ev = self.events[self.turn]
            if ev is not None and not capturing:
                ev.synchronize()
            self.host[self.turn].numpy()[:raw.size] = raw
            L.check(L.fn("umi_table_upload")(self.host[self.turn].data_ptr(), self.dev.data_ptr(), raw.size, ops._stream()),
                    "umi_table_upload")
            if not capturing:
                self.events[self.turn] = torch.cuda.Event()
                self.events[self.turn].record()
            self.key, self.blocks, self.n = key, b0, len(rows)
        return self.dev.data_ptr(), self.n, self.blocks


def _table(tabs, key, cap, rows, also=()):
    """Descriptor table for (key, capturing?).  A step captured into a HIP graph gets its own table: the captured upload node
    re-reads its pinned staging buffer at every replay, so eager steps (other gradient addresses) must not share it.  The
    capture-side buffers are allocated during the eager warm-up step, never inside the capture."""
    if not cap:
        for k in (key,) + tuple(also):
            tabs.setdefault(k + (True,), _Table())._ensure(len(rows) * ops.OPTIM_DESC.itemsize)
    return tabs.setdefault(key + (cap,), _Table()).get(rows)


_HYPER = np.dtype([("lr", "f8"), ("base_lr", "f8"), ("iter", "f8"), ("max_iter", "f8"), ("power", "f8"), ("adam_t", "f8"),
                   ("beta1", "f8"), ("beta2", "f8"), ("lr_f", "f4"), ("step_size_f", "f4"), ("bc2_sqrt_f", "f4"), ("pad", "f4"),
                   ("pad2", "f8", 2)])
assert _HYPER.itemsize == 96           # sizeof(umi_optim_hyper)


class _DeviceHyper:
    """Mixin: learning rate (and Adam's step count) kept in device memory, so that a step captured into a HIP graph keeps
    following the schedule when it is replayed (reference Trainer.py:719-726 rewrites the LR after every step; a captured
    kernel argument would freeze it -- and torch.optim.Adam's bias corrections with it).

        opt.device_schedule()                                          constant LR (Adam: the step count advances on device)
        opt.device_schedule(poly=dict(base_lr=.., max_iterations=.., power=0.9, iter_num=0))
                                                                       + the reference's poly rule, applied by step() itself

    From then on `step()` reads the LR from the device block and ignores later edits of `param_groups[i]['lr']`;
    `sync_host()` copies lr / iteration / Adam step back into `param_groups` and `state` (it synchronises; `state_dict()`
    calls it).  Eager and graph-replayed steps are interchangeable in this mode: nothing step-dependent is a kernel argument."""

    def device_schedule(self, poly=None):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("umi.optim: call device_schedule() before the HIP-graph capture")
        blocks = []
        for group in self.param_groups:
            h = np.zeros(1, dtype=_HYPER)
            h["lr"] = float(group["lr"])
            if poly is not None:
                h["base_lr"], h["max_iter"] = float(poly["base_lr"]), float(poly["max_iterations"])
                h["power"], h["iter"] = float(poly.get("power", 0.9)), float(poly.get("iter_num", 0))
            if "betas" in group:
                h["beta1"], h["beta2"] = (float(b) for b in group["betas"])
                steps = {float(self.state[p]["step"]) for p in group["params"] if len(self.state.get(p, {}))}
                if len(steps) > 1:
                    raise RuntimeError("umi.optim: device_schedule() needs one common Adam step count per param group")
                h["adam_t"] = steps.pop() if steps else 0.0
            host = torch.empty(96, dtype=torch.uint8).pin_memory()
            host.numpy()[:] = h.view(np.uint8).reshape(-1)
            dev = torch.empty(96, dtype=torch.uint8, device="cuda")
            L.check(L.fn("umi_table_upload")(host.data_ptr(), dev.data_ptr(), 96, ops._stream()), "umi_table_upload")
            blocks.append((dev, host))
        torch.cuda.current_stream().synchronize()
        self._umi_hyper, self._umi_poly = blocks, poly is not None
        return self

    @property
    def device_hyper(self):
        return self.__dict__.get("_umi_hyper")

    def sync_host(self):
        """Device block -> param_groups[i]['lr'] (and Adam state['step']); returns the list of blocks as numpy records."""
        out = []
        if self.device_hyper is None or torch.cuda.is_current_stream_capturing():
            return out
        for group, (dev, _) in zip(self.param_groups, self._umi_hyper):
            h = dev.cpu().numpy().view(_HYPER)[0]
            group["lr"] = float(h["lr"])
            if "betas" in group:
                for p in group["params"]:
                    st = self.state.get(p)
                    if st is not None and "step" in st:
                        st["step"].fill_(float(h["adam_t"]))
            out.append(h)
        return out

    def state_dict(self):
        self.sync_host()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """Loading a state dict rewrites param_groups[i]['lr'] and Adam's step counts on the host; with a device schedule active
        the device block is what step() reads, so it is rebuilt from the loaded values (the poly rule keeps its base_lr /
        max_iterations / power and the ITERATION COUNT of the block: pass iter_num to device_schedule() to resume elsewhere)."""
        blocks = self.sync_host() if self.device_hyper is not None else []
        super().load_state_dict(state_dict)
        if self.device_hyper is not None:
            poly = None
            if self._umi_poly and blocks:
                h = blocks[0]
                poly = dict(base_lr=float(h["base_lr"]), max_iterations=float(h["max_iter"]), power=float(h["power"]),
                            iter_num=float(h["iter"]))
            self.device_schedule(poly=poly)


def _check(p):
    if not p.is_cuda:
        raise RuntimeError("umi.optim: parameters must be on the MI355X (device 'cuda'); use torch.optim on the CPU")
    if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse:
        raise RuntimeError("umi.optim: dense fp32 parameters and gradients only")
    if not p.is_contiguous():
        raise RuntimeError("umi.optim: parameters must be contiguous")


class SGD(_DeviceHyper, torch.optim.SGD):
    """torch.optim.SGD(lr, momentum, dampening, weight_decay, nesterov) with a single-launch step."""

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            if group.get("maximize") or group.get("differentiable"):
                raise NotImplementedError("umi.optim.SGD: maximize / differentiable are not supported")
            mom = float(group["momentum"])
            rows = {True: [], False: []}             # first step of a momentum buffer? -> rows
            touched = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                _check(p)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                first, buf = False, None
                if mom != 0.0:
                    st = self.state[p]
                    buf = st.get("momentum_buffer")
                    if buf is None:
                        buf = st["momentum_buffer"] = torch.empty_like(p, memory_format=torch.contiguous_format)
                        first = True
                rows[first].append((p.data_ptr(), g.data_ptr(), 0 if buf is None else buf.data_ptr(), 0, p.numel()))
                touched.append((p, g))
            tabs = self.__dict__.setdefault("_umi_tables", {})
            cap = torch.cuda.is_current_stream_capturing()       # a captured table-upload node re-reads its own staging buffers
            hyper = self.device_hyper
            if hyper is not None:
                L.check(L.fn("umi_optim_hyper_pre")(hyper[gi][0].data_ptr(), 0, ops._stream()), "umi_optim_hyper_pre")
            for first, rr in rows.items():
                if not rr:
                    continue
                ptr, n, blocks = _table(tabs, (gi, first), cap, rr, also=((gi, False),))   # the step after a first step
                if hyper is not None:
                    L.check(L.fn("umi_optim_sgd_multi_dev")(ptr, n, blocks, hyper[gi][0].data_ptr(), mom, float(group["dampening"]),
                                                            float(group["weight_decay"]), int(bool(group["nesterov"])),
                                                            int(first), ops._stream()), "umi_optim_sgd_multi_dev")
                else:
                    L.check(L.fn("umi_optim_sgd_multi")(ptr, n, blocks, float(group["lr"]), mom, float(group["dampening"]),
                                                        float(group["weight_decay"]), int(bool(group["nesterov"])), int(first),
                                                        ops._stream()), "umi_optim_sgd_multi")
            if hyper is not None and self._umi_poly:
                L.check(L.fn("umi_optim_hyper_poly")(hyper[gi][0].data_ptr(), ops._stream()), "umi_optim_hyper_poly")
            for p, _ in touched:
                _bump(p)
        return loss


class Adam(_DeviceHyper, torch.optim.Adam):
    """torch.optim.Adam(lr, betas, eps, weight_decay) (L2 weight decay, no amsgrad) with a single-launch step."""

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("maximize") or group.get("differentiable") or group.get("capturable"):
                raise NotImplementedError("umi.optim.Adam: amsgrad / maximize / differentiable / capturable are not supported")
            if isinstance(group["lr"], torch.Tensor):
                raise NotImplementedError("umi.optim.Adam: tensor learning rates are not supported")
            b1, b2 = (float(b) for b in group["betas"])
            hyper = self.device_hyper
            cap = torch.cuda.is_current_stream_capturing()
            if cap and hyper is None:
                # torch.optim.Adam refuses capture unless capturable=True; here the host-side `step += 1` and the bias
                # corrections baked into kernel arguments would silently repeat step t on every replay
                raise RuntimeError("umi.optim.Adam: a captured step needs the device-side step count; call "
                                   "optimizer.device_schedule() before capturing (umi.graphs.GraphedStep(optimizers=[...]) does)")
            by_step = {}
            touched = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                _check(p)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                st = self.state[p]
                if len(st) == 0:
                    if cap:
                        raise RuntimeError("umi.optim.Adam: first step inside a HIP-graph capture; run a warm-up step first")
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)            # host scalar, as torch keeps it
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                if hyper is None:
                    st["step"] += 1
                    t = int(st["step"].item()) if not st["step"].is_cuda else int(st["step"])
                else:
                    t = 0                                # the device block counts; sync_host() refreshes state['step']
                by_step.setdefault(t, []).append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                                  st["exp_avg_sq"].data_ptr(), p.numel()))
                touched.append((p, g))
            tabs = self.__dict__.setdefault("_umi_tables", {})
            if hyper is not None and touched:
                L.check(L.fn("umi_optim_hyper_pre")(hyper[gi][0].data_ptr(), 1, ops._stream()), "umi_optim_hyper_pre")
            for slot, (t, rr) in enumerate(sorted(by_step.items())):
                ptr, n, blocks = _table(tabs, (gi, slot), cap, rr)
                if hyper is not None:
                    L.check(L.fn("umi_optim_adam_multi_dev")(ptr, n, blocks, hyper[gi][0].data_ptr(), b1, b2, float(group["eps"]),
                                                             float(group["weight_decay"]), ops._stream()),
                            "umi_optim_adam_multi_dev")
                    continue
                bc1 = 1.0 - b1 ** t
                bc2 = 1.0 - b2 ** t
                L.check(L.fn("umi_optim_adam_multi")(ptr, n, blocks, float(group["lr"]) / bc1, b1, b2, math.sqrt(bc2),
                                                     float(group["eps"]), float(group["weight_decay"]), ops._stream()),
                        "umi_optim_adam_multi")
            if hyper is not None and self._umi_poly and touched:
                L.check(L.fn("umi_optim_hyper_poly")(hyper[gi][0].data_ptr(), ops._stream()), "umi_optim_hyper_poly")
            for p, _ in touched:
                _bump(p)
        return loss
