"""Data-parallel gradient exchange: one process per GPU, bucketed sum-all-reduce over RCCL/xGMI.

The reference has no multi-GPU path; this is new functionality with standard
DDP semantics: every rank holds a full replica, the global batch is split evenly, gradients are
averaged, BatchNorm statistics and the Dice sums stay per-replica.

Overlap: the tape's hand-written backward (umi/graph.py) writes each parameter gradient straight
into its slot of a flat fp32 bucket and tells the reducer; buckets are filled in reverse execution
order (outc, up4, ... inc), and a bucket's `all_reduce(async_op=True)` is issued the moment its last
gradient kernel has been enqueued.  torch's RCCL process group runs it on its own HIP stream, so
the decoder buckets travel over xGMI while the encoder backward is still computing; `finish()`
makes the compute stream wait for the outstanding collectives (no host sync).  The 1/world factor
is folded into the wgrad kernels' output scale, so the collective is a plain SUM.

`sync()` also supports models that do not use the tape (e.g. the CPU oracle in the gloo tests):
it then reduces `p.grad` through the same buckets after backward.
"""
import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("flat", "views", "pending", "work", "params")

    def __init__(self):
        self.flat, self.views, self.pending, self.work, self.params = None, {}, 0, None, []


class GradReducer:
    def __init__(self, model, world_size=None, bucket_mb=32.0, group=None):
        self.group = group
        self.world = world_size if world_size is not None else dist.get_world_size(group)
        params = [p for p in model.parameters() if p.requires_grad]
        self.params = params
        cap = int(bucket_mb * (1 << 20)) // 4
        self.buckets, cur, n = [], _Bucket(), 0
        for p in reversed(params):                      # reverse execution order
            if n and n + p.numel() > cap:
                self._seal(cur, n)
                cur, n = _Bucket(), 0
            cur.params.append(p)
            n += p.numel()
        if n:
            self._seal(cur, n)
        self.where = {id(p): b for b in self.buckets for p in b.params}
        self.grad_scale = 1.0 / self.world              # folded into the wgrad output scale by the tape
        self._marked = False
        # deferred mode (`flush()`): the tape only fills the buckets and launches nothing, so forward + backward can be
        # captured into a HIP graph (umi.graphs.GraphedStep); the collectives are issued after the replay.  Trades the
        # overlap with the backward pass (~0.1 ms per 16 MB bucket over xGMI) for a step whose ~350 launches no longer
        # depend on the host
        self.deferred = False
        self.reset()
        model._umi_grad_sink = self
        if self.world > 1:
            self.broadcast_parameters(model)

    def _seal(self, b, numel):
        p0 = b.params[0]
        b.flat = torch.zeros(numel, dtype=torch.float32, device=p0.device)
        off = 0
        for p in b.params:
            b.views[id(p)] = b.flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
        self.buckets.append(b)

    def broadcast_parameters(self, model):
        """Identical replicas: rank 0's parameters and buffers win (as torch DDP does at construction)."""
        with torch.no_grad():
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t, 0, group=self.group)

    def reset(self):
        for b in self.buckets:
            b.pending, b.work = len(b.params), None
        self._marked = False

    # ---- tape-facing API -------------------------------------------------------------------
    def buffer_for(self, p):
        b = self.where.get(id(p))
        return None if b is None else b.views[id(p)]

    def mark_ready(self, p):
        b = self.where.get(id(p))
        if b is None:
            return
        self._marked = True
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b):
        if self.deferred:
            return
        if self.world > 1 and b.work is None:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def flush(self):
        """Deferred mode, after the (replayed) backward filled the buckets: all-reduce them, make the stream wait, and point
        every .grad at its bucket slot (autograd left copies of the un-reduced values there)."""
        for b in self.buckets:
            if self.world > 1:
                # one collective at a time, in bucket order: RCCL runs them back to back on its stream either way (the call
                # returns once enqueued and makes the compute stream wait); several concurrent 32 MB gloo all-reduces, as used
                # by the one-GPU rehearsal, take seconds
                dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group)
            for p in b.params:
                p.grad = b.views[id(p)]
        self.reset()

    def finish(self):
        """Called at the end of the tape backward: flush stragglers, make the stream wait."""
        if self.deferred:
            return
        for b in self.buckets:
            if b.pending > 0 and b.pending < len(b.params):
                self._launch(b)                         # some parameter of this bucket got no gradient
            if b.work is not None:
                b.work.wait()
                b.work = None

    # ---- step-facing API --------------------------------------------------------------------
    def sync(self):
        """After loss.backward(): gradients are averaged when this returns (stream-ordered)."""
        if self._marked:
            self.finish()
            # autograd stores a COPY of each gradient (AccumulateGrad clones a tensor it cannot steal).  The tape's backward
            # ends with finish(), so that copy is taken after the all-reduce and already holds the reduced values; .grad is
            # re-pointed at the bucket slots anyway: no copy, and the slots are stable, so the fused optimizer's pointer table
            # never changes
            for b in self.buckets:
                for p in b.params:
                    if p.grad is not None:
                        p.grad = b.views[id(p)]
            self.reset()
            return
        # generic path: the model's backward did not go through the tape
        for b in self.buckets:
            for p in b.params:
                v = b.views[id(p)]
                if p.grad is None:
                    v.zero_()
                else:
                    v.copy_(p.grad)
            b.flat.mul_(self.grad_scale)
            self._launch(b)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                b.work = None
            for p in b.params:
                if p.grad is not None:
                    p.grad.copy_(b.views[id(p)])
        self.reset()
