"""HIP-graph capture of a whole training step (forward + loss + backward + optimizer step).

Every libunetmi entry point only enqueues kernels on the caller's stream, so a step built from `Model.UNet` /
`TransUnet.VisionTransformer`, `loss.calc_loss` and a `torch.optim` optimizer is capturable with `torch.cuda.graph` like
any torch program: replaying the graph issues the ~350 (U-Net) / ~1,850 (TransUNet) launches of a step without the Python
and dispatch cost.  Measured on one MI355X: U-Net 25.7 -> 24.7 ms/step, TransUNet R50-ViT-B/16 30.2 -> 23.2 ms/step.

Rules that make the capture valid (the class enforces what it can):
  * static shapes and static input buffers: `GraphedStep.__call__(x, y)` copies the batch into the captured tensors;
  * no tensor carrying the autograd graph of an EARLIER step may be alive at capture time (typically the loss an eager step
    returned).  Cause of the `hipStreamEndCapture` segfault seen in round 1 (tools/experiments/exp_graph_accgrad.py reproduces
    both orders): such a graph keeps the parameters' cached AccumulateGrad nodes alive, and those are bound to the stream of the
    step that created them; the captured backward then accumulates the gradients on that non-capturing stream, a
    cross-stream dependency the ROCm 7.2 runtime answers with a crash when the capture ends.  The warm-up here discards its
    results, and with `optimizers=[...]` the constructor probes every parameter for such a surviving node and raises a
    RuntimeError instead;
  * nothing in the step may synchronise with the host (`.item()`, prints of tensors): return tensors, read them later;
  * random streams must advance on the device: TransUNet's dropout kernels take their per-step offset from a device
    counter (`umi_dropout(seed_dev=...)`), so every replay draws fresh masks;
  * kernel ARGUMENTS are frozen at capture time.  With `optimizers=[...]` the umi.optim optimizers are switched to their
    device-resident hyper-parameter block first (`device_schedule()`: learning rate, poly schedule state and Adam's step
    count live in device memory and advance inside the captured step); a plain torch.optim optimizer keeps the LR of the
    capture (rebuild the GraphedStep to change it);
  * collectives are never captured: data parallel runs capture forward + backward only, with `GradReducer.deferred = True`
    (the tape fills the gradient buckets and launches nothing), and call `reducer.flush()` + `optimizer.step()` after each
    replay (bench.py, UMI_DDP_LAUNCH=graph); with a process group alive the capture uses the thread_local error mode.
"""
import torch


class GraphedStep:
    def __init__(self, step_fn, example_inputs, warmup=3, capture_error_mode=None, optimizers=(), poly=None):
        """step_fn(*static_inputs) -> tensor or tuple of tensors (e.g. the loss); it must run the whole step, including
        `optimizer.zero_grad(set_to_none=True)`, `backward()` and `optimizer.step()`.  warmup=0: the caller has already run
        the step eagerly (on a side stream, see the rules above) and the capture must not execute anything.
        optimizers / poly: see umi.optim._DeviceHyper.device_schedule (skipped for optimizers already in device mode)."""
        if not all(t.is_cuda for t in example_inputs):
            raise RuntimeError("GraphedStep needs device-resident example inputs")
        for opt in optimizers:
            if getattr(opt, "device_hyper", None) is None:
                if not hasattr(opt, "device_schedule"):
                    raise TypeError("GraphedStep(optimizers=...) takes umi.optim optimizers")
                opt.device_schedule(poly=poly)
        # a replay updates the parameters without passing through Python: their version counters must be bumped by hand or
        # version-keyed caches (ops.PackCache: the kernel-layout weight copies an eval-mode forward reuses) would go stale
        self._params = [p for opt in optimizers for g in opt.param_groups for p in g["params"]]
        self._check_no_foreign_grad_accumulators(self._params)
        self.static_inputs = [t.clone() for t in example_inputs]
        # The graph refers to everything the step touched by ADDRESS: parameters, optimizer state (momentum buffers, the fused
        # optimizer's descriptor tables in pinned host memory), weight-pack caches.  Keep the closure -- and through it the
        # model and the optimizer -- alive as long as the graph: an optimizer that only the closure referenced would be
        # collected after __init__, its buffers handed to the next allocation, and the replayed update would read them
        # (observed: the graphed model stops learning, or a memory fault).
        self._step_fn = step_fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step_fn(*self.static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if capture_error_mode is None:
            # with a process group alive, its watchdog thread polls HIP events of earlier collectives; in the default
            # "global" mode such a call from ANOTHER thread invalidates the capture.  Nothing captured here comes from
            # that thread, so only this thread's calls are checked.
            import torch.distributed as dist
            multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
            capture_error_mode = "thread_local" if multi else "global"
        self._check_no_foreign_grad_accumulators(self._params)        # (the warm-up above discarded its outputs)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=capture_error_mode):
            self.static_outputs = step_fn(*self.static_inputs)

    @staticmethod
    def _check_no_foreign_grad_accumulators(params):
        """A parameter's AccumulateGrad node is cached on the tensor (weakly) and bound to the stream it was created on.  If one
        survives here, something outside holds the autograd graph of an earlier step (e.g. its loss tensor); capturing would
        route the gradient accumulation through that step's stream and crash the runtime at the end of the capture.  Probe: tag
        the node, drop our reference, fetch it again -- a node nobody else holds is rebuilt without the tag."""
        import gc
        token = object()
        probed = []
        for p in params:
            if p.requires_grad and p.is_leaf:
                acc = p.expand_as(p).grad_fn.next_functions[0][0]
                acc.metadata["umi_capture_probe"] = token
                probed.append(p)
                del acc
        gc.collect()
        stale = [p for p in probed
                 if p.expand_as(p).grad_fn.next_functions[0][0].metadata.get("umi_capture_probe") is token]
        if stale:
            raise RuntimeError(
                f"GraphedStep: {len(stale)} parameter(s) still have the gradient-accumulator node of an earlier step "
                "(a tensor with that step's autograd graph is alive, e.g. the loss it returned).  Capturing now would run the "
                "gradient accumulation on that step's stream and crash hipStreamEndCapture; delete such tensors (or "
                ".detach() what you keep) before building the GraphedStep.")

    def __call__(self, *inputs):
        for dst, src in zip(self.static_inputs, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        if self._params:
            torch.autograd.graph.increment_version(self._params)
        return self.static_outputs
