#!/usr/bin/env python3
"""Benchmark of the hot path: U-Net training step (forward + dice_bce_mc + backward + SGD), images/sec.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload = BASELINE.json configs[1]: UNet(1, 2, 64), 512x512, batch 16 per GPU, fp16 storage with fp32
accumulation, synthetic data resident in HBM, random-init weights.  Weak scaling: per-GPU batch fixed,
gradients averaged with one RCCL all-reduce per bucket.  Prints ONE JSON line on rank 0 carrying the
`roofline` (dominant kernel = the 3x3 conv forward, timed live with HIP events) and `cpu_baseline`
(the oracle's torch-CPU restatement timed on this box's host cores, rank 0 at N=1 only) objects.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "unet-torch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

PEAK_F16_MFMA_TFLOPS = 2500.0     # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0


def unet_fwd_gflop_per_image(cin, ncls, f, H, W):
    """2*MACs of conv / convT / 1x1 only."""
    fl = 0.0
    chans = [f * 2 ** i for i in range(5)]
    h, w = H, W
    prev = cin
    for i, c in enumerate(chans):
        if i:
            h, w = h // 2, w // 2
        fl += 2.0 * h * w * 9 * (prev * c + c * c)
        prev = c
    for i in range(4):
        c = chans[3 - i]
        h, w = h * 2, w * 2
        fl += 2.0 * (h // 2) * (w // 2) * (2 * c) * c * 4            # ConvT 2x2
        fl += 2.0 * h * w * 9 * (2 * c * c + c * c)
    fl += 2.0 * H * W * f * ncls
    return fl / 1e9


def double_conv_shapes(cin, f, H, W, B):
    """(name, N, H, W, Ci, Co) of the 18 3x3 convs of the nine DoubleConvs at this config."""
    out, chans = [], [f * 2 ** i for i in range(5)]
    h, w, prev = H, W, cin
    for i, c in enumerate(chans):
        if i:
            h, w = h // 2, w // 2
        out += [(f"enc{i}.c1", B, h, w, prev, c), (f"enc{i}.c2", B, h, w, c, c)]
        prev = c
    for i in range(4):
        c = chans[3 - i]
        h, w = h * 2, w * 2
        out += [(f"dec{i}.c1", B, h, w, 2 * c, c), (f"dec{i}.c2", B, h, w, c, c)]
    return out


def measure_conv_roofline(device, dtype, cin, f, H, W, B, reps=5):
    """Times every DoubleConv 3x3 forward launch of the workload with HIP events on the launch stream."""
    from umi import ops
    per = []
    tot_fl, tot_t = 0.0, 0.0
    for name, n, h, w, ci, co in double_conv_shapes(cin, f, H, W, B):
        x = torch.randn(n, h, w, ci, device=device).to(dtype)
        wgt = torch.randn(co, ci, 3, 3, device=device) * (2.0 / (9 * ci)) ** 0.5
        tx = ops.passthrough_tx(ci, device)
        tx[:, 3] = 0.0                                           # BN-apply + ReLU on load, like the real step
        y = torch.empty(n, h, w, co, device=device, dtype=dtype)
        lay, _ = ops.conv_plan(x, y, 3, 3, 1, 1)
        wp = ops.pack_conv_fwd(wgt, dtype, k8=bool(lay))
        wpf = lambda _lay, wp=wp: wp
        ops.conv_fwd(x, tx, wpf, None, y, 3, 3, 1, 1, want_stats=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.conv_fwd(x, tx, wpf, None, y, 3, 3, 1, 1, want_stats=True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * n * h * w * 9 * ci * co
        per.append({"conv": name, "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 2)})
        if ci >= 16:                 # the Ci=1 stem is HBM-bound, not part of the MFMA roofline figure
            tot_fl += fl
            tot_t += ms
        del x, y, wgt, wp
    ach = tot_fl / tot_t / 1e9
    # HBM-side bytes per launch come from the committed PMC passes of the same 17 launches (tools/prof_traffic.py:
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 x2 correction on FETCH_SIZE); PMC collection
    # cannot run inside this process, so the figure is read from profiles/ and null when that file is absent
    traffic, traffic_note = None, None
    tf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_conv_fwd_traffic.json")
    if os.path.exists(tf) and (cin, f, H, W, B) == (1, 64, 512, 512, 16):
        with open(tf) as fh:
            tj = json.load(fh)
        traffic = tj["hbm_bytes_per_launch_avg"]
        traffic_note = (f"avg bytes/launch over the 17 launches from {os.path.basename(tf)}; algorithmic "
                        f"{tj['algorithmic_bytes_total'] // 17} B/launch (ratio {tj['ratio']}); counters include Infinity-Cache hits")
    return {"bound": "mfma", "kernel": "conv3x3 forward (+BN-stat epilogue, BN/ReLU-on-load), 17 DoubleConv launches",
            "achieved": round(ach, 2), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_F16_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_note": traffic_note, "per_launch": per}


def cpu_baseline(cin, ncls, f, H, W, budget_s=25.0):
    """The oracle (torch CPU fp32 restatement of the reference) timed on this box's host cores."""
    from oracle import recipe, ref_unet
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:                                   # container CPU quota (cgroup v2): "<quota> <period>" or "max <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    # every host core this process may use (affinity mask and cgroup quota honoured); UMI_CPU_BASELINE_THREADS caps it
    cores = min(cores, int(os.environ.get("UMI_CPU_BASELINE_THREADS", str(cores))))
    torch.set_num_threads(cores)
    m = ref_unet.RefUNet(cin, ncls, f, False).train()
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    B = 1
    x, lab = recipe.synthetic_batch(B, cin, H, W, ncls, seed=1234)
    ref_unet.train_step(m, opt, x, lab, ncls)                   # warm-up
    t0, n = time.time(), 0
    while n < 1 or (time.time() - t0 < budget_s and n < 3):
        ref_unet.train_step(m, opt, x, lab, ncls)
        n += 1
    dt = (time.time() - t0) / n
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(B / dt, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{n} train steps of UNet({cin},{ncls},{f}) batch {B} at {H}x{W}, fp32, torch CPU, after 1 warm-up",
            "cpu": model}


def self_launch(n, argv):
    """Parent of a multi-GPU run started from a plain shell: one child rank per GPU via torch.distributed.run."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("[bench] self-launch:", " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--features", type=int, default=64)
    ap.add_argument("--cin", type=int, default=1)
    ap.add_argument("--ncls", type=int, default=2)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="(default on one GPU) replay the step from a captured HIP graph")
    ap.add_argument("--eager", action="store_true",
                    help="issue every step's ~350 launches from Python instead of replaying a captured HIP graph of the step")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # plain `python bench.py --gpus N`: this process has not touched the GPU (and never will); it starts one fresh
        # rank per GPU under torch.distributed.run, relays their output (rank 0 prints the JSON line) and exits with
        # their status.  Never exec: the children are ordinary subprocesses.
        sys.exit(self_launch(a.gpus, sys.argv[1:]))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    # rehearsal knobs for a one-GPU box (not used by the driver): UMI_DDP_BACKEND=gloo + UMI_BENCH_SAME_GPU=1 run N ranks
    # on device 0 through the same GradReducer / bucket / sink code with gloo carrying the all-reduce
    backend = os.environ.get("UMI_DDP_BACKEND", "nccl")
    if os.environ.get("UMI_BENCH_SAME_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import Model
    import loss as L
    from umi import ddp
    L.CLASS_NUMBER = a.ncls
    torch.manual_seed(0)                         # identical initial weights on every rank
    model = Model.UNet(a.cin, a.ncls, a.features, compute_dtype=a.dtype).to(dev).train()
    from umi import optim as umi_optim
    # same arithmetic and state as torch.optim.SGD (reference config.yml:15-18), one launch per step; UMI_TORCH_OPTIM=1 -> torch's
    opt_cls = torch.optim.SGD if os.environ.get("UMI_TORCH_OPTIM") == "1" else umi_optim.SGD
    opt = opt_cls(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    reducer = ddp.GradReducer(model, world) if world > 1 else None

    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    x = torch.randn(a.batch, a.cin, a.size, a.size, device=dev, generator=g)
    labels = torch.randint(0, a.ncls, (a.batch, a.size, a.size), device=dev, generator=g).float()

    def step():
        logits = model(x)
        loss = L.calc_loss(logits, labels, loss_type="dice_bce_mc")
        opt.zero_grad()
        loss.backward()
        if reducer is not None:
            reducer.sync()
        opt.step()
        return loss

    # One GPU: the step (forward + loss + backward + optimizer) is captured once into a HIP graph and replayed, so the
    # measurement does not depend on how fast the host issues the launches (same kernels, same order; the replayed
    # trajectory is bit-identical to the eager one, tests/test_gpu_unet.py).  The W warm-up steps run eagerly before the
    # capture, which itself executes nothing.  Data parallel: eager (the RCCL all-reduce is not captured).
    graphed = None
    path_ms = {}
    # N > 1 reports the OVERLAPPED path by default (bucket all-reduces issued while the backward pass still computes: the
    # north_star's exchange); the deferred-graph path is timed beside it during the warm-up and reported only on request
    # (UMI_DDP_LAUNCH=auto: whichever is faster; =graph: always), labelled as not overlapped.
    ddp_launch = "eager" if (a.eager or os.environ.get("UMI_BENCH_GRAPH", "1") == "0") else os.environ.get("UMI_DDP_LAUNCH", "both")
    eager_step = step
    if world > 1 and ddp_launch in ("auto", "graph", "both"):
        # Data parallel has two launch paths:
        #   eager: every launch issued from Python, bucket all-reduces overlapped with the rest of the backward pass;
        #   graph: forward + loss + backward replayed from a HIP graph that writes the gradients straight into the reducer's
        #          flat buckets, then the bucket all-reduces (RCCL, eager) and the one-launch optimizer step.  ~10 host-issued
        #          launches per step instead of ~350, but the 124 MB of gradients are reduced after the backward pass.
        # Which one is faster depends on the host (measured on this pool at N=1: eager 24.8 .. 31.7 ms, graph 23.9 .. 25.2 ms):
        # three steps of each are timed during the warm-up (MAX over ranks) and both figures go into the JSON line.  The
        # timed region then runs the overlapped eager path ("both", the default), the faster of the two ("auto") or the
        # graph ("graph"); every rank takes the same decision.  The graph is captured FIRST: a loss tensor kept from an eager step would hold
        # that step's autograd graph alive and with it gradient accumulators bound to the default stream, which crashes
        # hipStreamEndCapture (umi/graphs.py).
        from umi.graphs import GraphedStep
        trace = os.environ.get("UMI_BENCH_TRACE") == "1"

        def fwd_bwd(xx, yy):
            logits = model(xx)
            loss = L.calc_loss(logits, yy, loss_type="dice_bce_mc")
            opt.zero_grad()
            loss.backward()
            if os.environ.get("UMI_BENCH_INJECT_CAPTURE_FAILURE_RANK") == str(rank) and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("injected capture failure on this rank (fallback rehearsal)")
            return loss

        def timed(fn, n):
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n

        t_graph = float("inf")
        graph_step = None
        gs = None
        try:
            reducer.deferred = True
            gs = GraphedStep(fwd_bwd, [x, labels], warmup=1)       # deferred reducer: no collective in here
        except Exception as e:
            print(f"[bench] rank {rank}: HIP-graph capture failed ({type(e).__name__}: {e})", file=sys.stderr)
            gs = None
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        # the ranks must agree on the path BEFORE the next collective: a rank that fell back alone would pair its eager
        # all-reduces with the others' flushes and hang at the first barrier
        ok = torch.tensor([1.0 if gs is not None else 0.0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() == 0.0:
            if gs is not None and rank == 0:
                print("[bench] another rank could not capture; all ranks use eager launches", file=sys.stderr)
            gs = None
            torch.cuda.empty_cache()
        if gs is not None:
            def graph_step():
                t0 = time.perf_counter()
                loss = gs(x, labels)
                t1 = time.perf_counter()
                reducer.flush()
                t2 = time.perf_counter()
                opt.step()
                if trace and rank == 0:
                    print(f"[trace] replay {1e3 * (t1 - t0):.1f} ms  flush {1e3 * (t2 - t1):.1f} ms  "
                          f"opt {1e3 * (time.perf_counter() - t2):.1f} ms (host times)", file=sys.stderr, flush=True)
                return loss
            for _ in range(max(1, a.warmup)):
                graph_step()
            t_graph = timed(graph_step, 3)
        reducer.deferred = False
        reducer.reset()                                  # a failed capture leaves buckets marked but never flushed
        t_eager = float("inf")
        if ddp_launch in ("auto", "both") or graph_step is None:
            for _ in range(max(1, a.warmup)):
                eager_step()
            t_eager = timed(eager_step, 3)
        tt = torch.tensor([min(t_graph, 1e9), min(t_eager, 1e9)], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_graph, t_eager = tt.tolist()
        path_ms = {"hipgraph_fwd_bwd_then_allreduce": None if t_graph >= 1e8 else round(1e3 * t_graph, 3),
                   "eager_overlapped_allreduce": None if t_eager >= 1e8 else round(1e3 * t_eager, 3)}
        if rank == 0:
            print(f"[bench] launch paths at N={world}: graph {1e3 * t_graph:.2f} ms/step, eager {1e3 * t_eager:.2f} ms/step",
                  file=sys.stderr)
        if graph_step is not None and ddp_launch != "both" and (ddp_launch == "graph" or t_graph <= t_eager):
            reducer.deferred = True
            step, graphed = graph_step, gs
        else:
            step, graphed = eager_step, None
    if world == 1 and not a.eager and os.environ.get("UMI_BENCH_GRAPH", "1") != "0":
        from umi.graphs import GraphedStep

        def step_xy(xx, yy):
            if os.environ.get("UMI_BENCH_INJECT_CAPTURE_FAILURE") == "1" and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("injected failure during capture (fallback rehearsal)")
            logits = model(xx)
            loss = L.calc_loss(logits, yy, loss_type="dice_bce_mc")
            opt.zero_grad()
            loss.backward()
            opt.step()
            return loss
        try:
            graphed = GraphedStep(step_xy, [x, labels], warmup=max(1, a.warmup))   # warm-up steps run eagerly on a side stream
            step = lambda: graphed(x, labels)                                      # noqa: E731
        except Exception as e:                                                     # capture refused: measure the eager path
            print(f"[bench] HIP-graph capture failed ({type(e).__name__}: {e}); falling back to eager launches", file=sys.stderr)
            graphed = None
            torch.cuda.synchronize()
            torch.cuda.empty_cache()                     # drop the failed capture's private pool
            for _ in range(3):                           # the side-stream warm-up cached its blocks for another stream
                step()
    if graphed is None and not (world > 1 and ddp_launch in ("auto", "graph", "both")):
        for _ in range(a.warmup):
            step()
    launch = ("eager" if world == 1 else "eager launches, bucket all-reduces overlapped with the backward pass") if graphed is None else (
        "hipgraph" if world == 1 else "hipgraph(fwd+bwd), then all-reduce (NOT overlapped) + optimizer")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    final_loss = float(loss.detach())

    if rank == 0:
        ms = dt / a.steps * 1e3
        value = a.batch * world * a.steps / dt
        gf = unet_fwd_gflop_per_image(a.cin, a.ncls, a.features, a.size, a.size)
        out = {
            "metric": "images/sec training step, 4-level U-Net 1ch\u21922cls 512\u00d7512, at 1/2/4/8 MI355X",   # BASELINE.json's metric, verbatim
            "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16" if a.dtype == "fp16" else "f32", "data": "synthetic",
            "config": {"workload": f"UNet({a.cin},{a.ncls},{a.features}) train step (fwd + dice_bce_mc + bwd + SGD), "
                                   f"{a.size}x{a.size}, batch {a.batch}/GPU, BASELINE configs[1]",
                       "global_batch": a.batch * world, "parallelism": f"dp{world}", "launch": launch,
                       "launch_paths_warmup_ms_per_step": path_ms or None,
                       "collective": None if world == 1 else {"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                                                             "library": "RCCL" if dist.get_backend() == "nccl" else dist.get_backend()},
                       "algorithmic_tflops_per_gpu": round(3 * gf * a.batch * a.steps / dt / 1e3, 2)},
            "final_loss": round(final_loss, 5),
        }
        if not a.no_roofline:
            out["roofline"] = measure_conv_roofline(dev, torch.float16 if a.dtype == "fp16" else torch.float32,
                                                    a.cin, a.features, a.size, a.size, a.batch)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cin, a.ncls, a.features, a.size, a.size)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
