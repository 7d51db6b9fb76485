"""HBM traffic of bench.py's roofline kernel (the 18 DoubleConv 3x3 forward launches of UNet(1,2,64) @512^2, batch 16).

  run   : launch every conv twice (under `rocprofv3 --pmc FETCH_SIZE ...` and again under `--pmc WRITE_SIZE ...`,
          separate passes as MI355X_MICROARCH.md's HBM section prescribes)
  parse : read the two counter_collection.csv files, apply the gfx950 correction (FETCH_SIZE reports half of a wide
          coalesced read: x2; both counters are in KiB... rocprofv3 reports KB), and write profiles/<out>.json with the
          measured bytes per launch next to the algorithmic bytes (x + packed weights read, y + BN partials written).
"""
import csv, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]


def shapes():
    import bench
    return bench.double_conv_shapes(1, 64, 512, 512, 16)


def run():
    import torch
    from umi import ops
    for name, n, h, w, ci, co in shapes():
        x = torch.randn(n, h, w, ci, device="cuda").half()
        wgt = torch.randn(co, ci, 3, 3, device="cuda") * (2.0 / (9 * ci)) ** 0.5
        tx = ops.passthrough_tx(ci, "cuda"); tx[:, 3] = 0.0
        y = torch.empty(n, h, w, co, device="cuda", dtype=torch.float16)
        lay, _ = ops.conv_plan(x, y, 3, 3, 1, 1)
        wp = ops.pack_conv_fwd(wgt, torch.float16, k8=bool(lay))
        for _ in range(2):
            ops.conv_fwd(x, tx, lambda _l, wp=wp: wp, None, y, 3, 3, 1, 1, want_stats=True)
        torch.cuda.synchronize()


def parse(fetch_csv, write_csv, out):
    def rows(f, cname):
        r = [x for x in csv.DictReader(open(f)) if x["Counter_Name"] == cname and
             ("conv3x3_mfma_kernel" in x["Kernel_Name"] or "stem3x3_fwd" in x["Kernel_Name"])]
        r.sort(key=lambda x: int(x["Dispatch_Id"]))
        return [float(x["Counter_Value"]) * 1024.0 for x in r]
    fe, wr = rows(fetch_csv, "FETCH_SIZE"), rows(write_csv, "WRITE_SIZE")
    sh = shapes()
    assert len(fe) == 2 * len(sh) == len(wr), (len(fe), len(wr), len(sh))
    per, tot_m, tot_a = [], 0.0, 0.0
    for i, (name, n, h, w, ci, co) in enumerate(sh):
        meas = 2.0 * fe[2 * i + 1] + wr[2 * i + 1]                  # second launch of the pair; FETCH_SIZE x2 (gfx950)
        alg = 2.0 * n * h * w * (ci + co) + 2.0 * 9 * ci * co
        per.append({"conv": name, "hbm_bytes": round(meas), "fetch_bytes_x2": round(2.0 * fe[2 * i + 1]), "write_bytes": round(wr[2 * i + 1]),
                    "algorithmic_bytes": round(alg), "ratio": round(meas / alg, 3)})
        if ci >= 16:
            tot_m += meas; tot_a += alg
    res = {"kernel": "conv3x3 forward, 17 MFMA DoubleConv launches (stem listed, not summed)", "hbm_bytes_total": round(tot_m),
           "algorithmic_bytes_total": round(tot_a), "ratio": round(tot_m / tot_a, 3), "hbm_bytes_per_launch_avg": round(tot_m / 17),
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950 128-B requests "
                     "tallied at 64 B); counters include Infinity-Cache hits", "per_launch": per}
    with open(os.path.join(REPO, "profiles", out), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "per_launch"}))
    for p in per:
        print(p)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        parse(sys.argv[2], sys.argv[3], sys.argv[4])
