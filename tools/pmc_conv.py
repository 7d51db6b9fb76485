"""Hardware-counter passes over the conv3x3 forward launches of the bench workload.

  python tools/pmc_conv.py run                      (under rocprofv3: every MFMA-eligible DoubleConv conv twice, env UMI_CONV3X3_IMPL picks the kernel)
  python tools/pmc_conv.py parse DIR [out.json]     (DIR holds one sub-directory per pass with rocprofv3's counter_collection csv)

Every pass is its own rocprofv3 run (`--pmc ... --kernel-trace`), as gpurun requires; see tools/pmc_conv.sh."""
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]


def shapes():
    import bench
    return [s for s in bench.double_conv_shapes(1, 64, 512, 512, 16) if s[4] >= 16]


def run():
    import torch
    from umi import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    for name, n, h, w, ci, co in shapes():
        x = torch.randn(n, h, w, ci, device="cuda", generator=g).half()
        wgt = torch.randn(co, ci, 3, 3, device="cuda", generator=g) * (2.0 / (9 * ci)) ** 0.5
        tx = ops.passthrough_tx(ci, "cuda")
        tx[:, 3] = 0.0
        y = torch.empty(n, h, w, co, device="cuda", dtype=torch.float16)
        wp = ops.pack_conv_fwd(wgt, torch.float16, k8=True)
        for _ in range(2):
            ops.conv_fwd(x, tx, lambda _l, wp=wp: wp, None, y, 3, 3, 1, 1, want_stats=True)
        torch.cuda.synchronize()


def parse(d, out=None):
    sh = shapes()
    table = {}
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        rows = [r for r in csv.DictReader(open(f)) if "conv3x3" in r["Kernel_Name"]]
        by = {}
        for r in rows:
            by.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for cname, v in by.items():
            v.sort()
            vals = [x[1] for x in v]
            if len(vals) != 2 * len(sh):
                print("skip", cname, len(vals), f)
                continue
            table[cname] = [vals[2 * i + 1] for i in range(len(sh))]      # second launch of each pair
    names = sorted(table)
    res = {}
    for i, (name, n, h, w, ci, co) in enumerate(sh):
        res[name] = {"shape": [n, h, w, ci, co], **{c: table[c][i] for c in names}}
    hdr = f"{'layer':10s}" + "".join(f"{c[:22]:>24s}" for c in names)
    print(hdr)
    for k, v in res.items():
        print(f"{k:10s}" + "".join(f"{v[c]:24.4g}" for c in names))
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        parse(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
