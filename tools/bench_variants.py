"""Step time of the SURVEY 8(f)-3 model variants at the benchmark shape (B=16, 1x512x512, fp16, SGD), HIP-graph replay.
Informational: prints one JSON line per model."""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "unet-torch_amd")]
import torch
import Model, loss as L
from umi import optim as uo
from umi.graphs import GraphedStep

L.CLASS_NUMBER = 2
B, S = int(os.environ.get("B", "16")), int(os.environ.get("S", "512"))
x = torch.randn(B, 1, S, S, device="cuda")
lab = torch.randint(0, 2, (B, S, S), device="cuda").float()
names = sys.argv[1:] or ["UNet", "UNet_multitask", "UNet_attention"]
for name in names:
    torch.manual_seed(0)
    m = getattr(Model, name)(1, 2, 64, compute_dtype="fp16").cuda().train()
    opt = uo.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)

    def step(xx, yy):
        out = m(xx)
        outs = out if isinstance(out, tuple) else (out,)
        loss = sum(L.calc_loss(o, yy, loss_type="dice_bce_mc") for o in outs)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss
    eager = os.environ.get("EAGER") == "1"
    if eager:
        for _ in range(3):
            step(x, lab)
        run = lambda: step(x, lab)
    else:
        gs = GraphedStep(step, [x, lab], warmup=3)
        run = lambda: gs(x, lab)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        loss = run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(json.dumps({"model": name, "B": B, "size": S, "ms_per_step": round(ms, 2), "images_per_s": round(B / ms * 1e3, 1),
                      "launch": "eager" if eager else "hipgraph", "loss": float(loss.detach()),
                      "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}), flush=True)
    del m, opt
    if not eager:
        del gs
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
