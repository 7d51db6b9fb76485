"""HIP-graph capture of a whole training step (forward + loss + backward + optimizer step).

Every libunetmi entry point only enqueues kernels on the caller's stream, so a step built from `Model.UNet` /
`TransUnet.VisionTransformer`, `loss.calc_loss` and a `torch.optim` optimizer is capturable with `torch.cuda.graph` like
any torch program: replaying the graph issues the ~350 (U-Net) / ~1,850 (TransUNet) launches of a step without the Python
and dispatch cost.  Measured on one MI355X: U-Net 25.7 -> 24.7 ms/step, TransUNet R50-ViT-B/16 30.2 -> 23.2 ms/step.

Rules that make the capture valid (the class enforces what it can):
  * static shapes and static input buffers: `GraphedStep.__call__(x, y)` copies the batch into the captured tensors;
  * warm-up runs on a side stream and the capture follows IMMEDIATELY (an eager step on the default stream in between
    leaves autograd state that crashes `hipStreamEndCapture` on ROCm 7.2);
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
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=capture_error_mode):
            self.static_outputs = step_fn(*self.static_inputs)

    def __call__(self, *inputs):
        for dst, src in zip(self.static_inputs, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        if self._params:
            torch.autograd.graph.increment_version(self._params)
        return self.static_outputs
