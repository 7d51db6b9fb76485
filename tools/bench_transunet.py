"""Timing of the TransUNet R50-ViT-B/16 training step (config 4: 224x224) on one MI355X.  Not the headline bench."""
import os, sys, time, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import torch
import loss as L
from TransUnet.vit_seg_modeling import VisionTransformer, CONFIGS
import copy

B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
size = int(sys.argv[2]) if len(sys.argv) > 2 else 224
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dtype = sys.argv[4] if len(sys.argv) > 4 else "fp16"
cfg = copy.deepcopy(CONFIGS["R50-ViT-B_16"])
cfg.n_classes = 2
cfg.n_skip = 3
cfg.patches.grid = (size // 16, size // 16)
L.CLASS_NUMBER = 2
torch.manual_seed(0)
m = VisionTransformer(cfg, img_size=size, num_classes=2, compute_dtype=dtype).cuda().train()
from umi import optim as umi_optim
opt = (torch.optim.SGD if os.environ.get("UMI_TORCH_OPTIM") == "1" else umi_optim.SGD)(
    m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
x = torch.randn(B, 1, size, size, device="cuda")
lab = torch.randint(0, 2, (B, size, size), device="cuda").float()
def step():
    out = m(x); l = L.calc_loss(out, lab, loss_type="dice_bce_mc"); opt.zero_grad(); l.backward(); opt.step(); return l
if os.environ.get("UMI_BENCH_GRAPH") == "1":
    from umi.graphs import GraphedStep
    gs = GraphedStep(lambda xx, yy: step(), [x, lab], warmup=2)      # x / lab are the static buffers themselves
    step = lambda: gs(x, lab)                                         # noqa: E731
else:
    for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
gf = 58.35 if size == 224 else 336.27


def linear_roofline():
    """The dominant kernel family of the ViT half: the pointwise matrix-core kernel on the four linear shapes of a Block
    (tokens x hidden), timed with HIP events on this stream, 20 launches each (the same measurement as bench.py's roofline)."""
    from umi import ops
    M, hid, mlp = B * (size // 16) ** 2, cfg.hidden_size, cfg.transformer["mlp_dim"]
    rows, flops, secs = [], 0.0, 0.0
    for K, N in [(hid, 3 * hid), (hid, hid), (hid, mlp), (mlp, hid)]:
        xx = torch.randn(1, 1, M, K, device="cuda").half()
        wp = ops.pack_conv_fwd(torch.randn(N, K, 1, 1, device="cuda") * K ** -0.5, torch.float16, k8=True)
        yy = torch.empty(1, 1, M, N, device="cuda", dtype=torch.float16)
        for _ in range(3):
            ops.conv_fwd(xx, None, lambda lay: wp, None, yy, 1, 1, 1, 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            ops.conv_fwd(xx, None, lambda lay: wp, None, yy, 1, 1, 1, 0)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e-3
        rows.append({"linear": f"{K}->{N}", "us": round(t * 1e6, 1), "tflops": round(2.0 * M * K * N / t / 1e12, 1)})
        flops += 2.0 * M * K * N
        secs += t
    ach = flops / secs / 1e12
    return {"bound": "mfma", "kernel": "conv1x1_mfma on the four linears of a ViT Block (forward)", "achieved": round(ach, 1),
            "peak": 2500.0, "unit": "TFLOP/s", "frac": round(ach / 2500.0, 4), "traffic": None, "per_launch": rows}


print(json.dumps({"model": "TransUNet R50-ViT-B_16", "B": B, "size": size, "dtype": dtype, "ms_per_step": round(dt * 1e3, 2),
                  "images_per_s": round(B / dt, 2), "algorithmic_tflops": round(3 * gf * B / dt / 1e3, 2), "loss": float(l.detach()),
                  "launch": "hipgraph" if os.environ.get("UMI_BENCH_GRAPH") == "1" else "eager",
                  "roofline": linear_roofline() if os.environ.get("UMI_BENCH_ROOFLINE", "1") == "1" else None}))
