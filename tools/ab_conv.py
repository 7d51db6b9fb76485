"""Same-process A/B of the conv3x3 forward kernels (umi_tune_conv3x3_impl) on the DoubleConv launches of the bench workload:
outputs compared bit for bit against impl 1, then interleaved timing rounds (median / min per layer and the 17-launch sum).
usage: python tools/ab_conv.py [arms, e.g. 1,2,nostage:2] [rounds] [reps]
An arm is IMPL or VARIANT:IMPL (VARIANT = a library built by tools/build_variant.py; its outputs are not compared when its
name starts with "x", the convention for timing-only ablations whose results are wrong by construction)."""
import ctypes
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import torch
import bench
from umi import lib as L, ops

arms = (sys.argv[1] if len(sys.argv) > 1 else "1,2,3").split(",")
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev, dt = "cuda", torch.float16


class Arm:
    def __init__(self, spec):
        self.name = spec
        var, _, im = spec.rpartition(":")
        self.impl = int(im)
        self.check = not var.startswith("x")
        if var:
            self.lib = ctypes.CDLL(os.path.join(REPO, "tools", "_ab", f"libunetmi_{var}.so"))
            self.conv = self.lib.umi_conv_fwd
            self.conv.restype = ctypes.c_int
            self.conv.argtypes = L.SIGNATURES["umi_conv_fwd"][1]
            self.tune = self.lib.umi_tune_conv3x3_impl
            self.plan = self.lib.umi_conv_fwd_plan
            self.plan.argtypes = L.SIGNATURES["umi_conv_fwd_plan"][1]
        else:
            self.conv, self.tune, self.plan = L.fn("umi_conv_fwd"), L.fn("umi_tune_conv3x3_impl"), L.fn("umi_conv_fwd_plan")

    def rows(self, n, h, w, ci, co):
        lay, rows = ctypes.c_int(0), ctypes.c_int(0)
        self.tune(self.impl)
        assert self.plan(n, h, w, ci, co, 3, 3, 1, 1, ci, co, 1, 1, 0, 0, ctypes.addressof(lay), ctypes.addressof(rows)) == 0
        return rows.value

    def run(self, x, tx, wp, y, part, n, h, w, ci, co, reps=1):
        self.tune(self.impl)
        args = (x.data_ptr(), ci, tx.data_ptr(), wp.data_ptr(), None, y.data_ptr(), co, part.data_ptr(), n, h, w, ci, co,
                3, 3, 1, 1, h, w, 0, 0, h, w, 1, 1, 0, torch.cuda.current_stream().cuda_stream)
        for _ in range(reps):
            st = self.conv(*args)
            assert st == 0, st


arms = [Arm(a) for a in arms]
gen = torch.Generator(device=dev).manual_seed(1)
layers = []
for name, n, h, w, ci, co in bench.double_conv_shapes(1, 64, 512, 512, 16):
    if ci < 16:
        continue
    x = torch.randn(n, h, w, ci, device=dev, generator=gen).to(dt)
    wgt = torch.randn(co, ci, 3, 3, device=dev, generator=gen) * (2.0 / (9 * ci)) ** 0.5
    if os.environ.get("AB_ZERO") == "1":        # clock experiment: all-zero operands draw less power (MI355X_MICROARCH.md, DVFS)
        x.zero_()
        wgt.zero_()
    tx = ops.passthrough_tx(ci, dev)
    tx[:, 1] = 0.5 + torch.rand(ci, device=dev, generator=gen)
    tx[:, 2] = 0.2 * torch.randn(ci, device=dev, generator=gen)
    tx[:, 3] = 0.0
    wp = ops.pack_conv_fwd(wgt, dt, k8=True)
    layers.append((name, n, h, w, ci, co, x, tx, wp))

res = {}
for name, n, h, w, ci, co, x, tx, wp in layers:
    outs = {}
    for a in arms:
        y = torch.empty(n, h, w, co, device=dev, dtype=dt)
        part = torch.zeros(a.rows(n, h, w, ci, co) * 2 * co, device=dev)
        a.run(x, tx, wp, y, part, n, h, w, ci, co)
        torch.cuda.synchronize()
        outs[a.name] = (y, part, part.view(-1, 2, co).double().sum(0))
    base = outs[arms[0].name]
    for a in arms[1:]:
        if not a.check:
            continue
        same = torch.equal(outs[a.name][0], base[0])
        serr = ((outs[a.name][2] - base[2]).abs().max() / base[2].abs().max()).item()
        if not same:        # a different MFMA shape sums in a different order: fp32 rounding, visible as rare 1-ulp fp16 flips
            d = (outs[a.name][0].float() - base[0].float()).abs()
            rel = d.max().item() / base[0].float().abs().max().item()
            print(f"  {name}: arm {a.name} vs {arms[0].name}: {(d > 0).float().mean().item():.2e} of outputs differ, max {rel:.1e} of scale")
            assert os.environ.get("AB_EXACT") != "1" and rel < 2e-3, f"{name}: arm {a.name} output differs from arm {arms[0].name}"
        assert os.environ.get("AB_ZERO") == "1" or serr < (1e-5 if same else 1e-3), (name, a.name, serr)
    t = {a.name: [] for a in arms}
    for r in range(rounds):
        for a in arms:
            y, part, _ = outs[a.name]
            a.run(x, tx, wp, y, part, n, h, w, ci, co)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            a.run(x, tx, wp, y, part, n, h, w, ci, co, reps)
            e1.record()
            torch.cuda.synchronize()
            t[a.name].append(e0.elapsed_time(e1) / reps)
    gf = 2.0 * n * h * w * 9 * ci * co / 1e9
    line = [f"{name:10s} {ci:4d}->{co:4d}@{h:3d}"]
    res[name] = {}
    for a in arms:
        tt = t[a.name]
        med = sorted(tt)[len(tt) // 2]
        res[name][a.name] = dict(ms_med=med, ms_min=min(tt), tf_med=gf / med, gflop=gf)
        line.append(f"{a.name}: {med:.3f} ms {gf / med:5.0f} TF")
    print("  ".join(line), flush=True)
tot = {a.name: sum(res[k][a.name]["ms_med"] for k in res) for a in arms}
gft = sum(res[k][arms[0].name]["gflop"] for k in res)
print("SUM", {k: f"{v:.3f} ms = {gft / v:.0f} TF" for k, v in tot.items()}, flush=True)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(REPO, "gpurun_out", "ab_conv.json"), "w"), indent=1)
