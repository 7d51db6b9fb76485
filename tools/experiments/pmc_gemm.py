"""Hardware-counter passes over the pointwise matrix-core kernel on the ViT linear shapes (see tools/ab_gemm.py).
  python tools/experiments/pmc_gemm.py run            (under rocprofv3 --pmc ... --kernel-trace: each shape twice)
  python tools/experiments/pmc_gemm.py parse DIR      (one sub-directory per pass)"""
import csv, glob, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
SHAPES = [(768, 768), (3072, 768), (768, 3072), (768, 2304), (2304, 768)]
M = 4704


def run():
    import torch
    from umi import ops
    for K, N in SHAPES:
        x = torch.randn(1, 1, M, K, device="cuda").half()
        wp = ops.pack_conv_fwd(torch.randn(N, K, 1, 1, device="cuda") * K ** -0.5, torch.float16, k8=True)
        y = torch.empty(1, 1, M, N, device="cuda", dtype=torch.float16)
        for _ in range(2):
            ops.conv_fwd(x, None, lambda lay: wp, None, y, 1, 1, 1, 0)
        torch.cuda.synchronize()


def parse(d):
    table = {}
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        by = {}
        for r in csv.DictReader(open(f)):
            if "conv1x1_mfma" in r["Kernel_Name"]:
                by.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for c, v in by.items():
            v.sort()
            if len(v) == 2 * len(SHAPES):
                table[c] = [v[2 * i + 1][1] for i in range(len(SHAPES))]
    names = sorted(table)
    res = {f"{K}->{N}": {c: table[c][i] for c in names} for i, (K, N) in enumerate(SHAPES)}
    for k, v in res.items():
        print(k, json.dumps(v))
    json.dump(res, open(os.path.join(d, "pmc_gemm.json"), "w"), indent=1)


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else parse(sys.argv[2])
