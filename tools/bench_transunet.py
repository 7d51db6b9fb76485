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
print(json.dumps({"model": "TransUNet R50-ViT-B_16", "B": B, "size": size, "dtype": dtype, "ms_per_step": round(dt * 1e3, 2),
                  "images_per_s": round(B / dt, 2), "algorithmic_tflops": round(3 * gf * B / dt / 1e3, 2), "loss": float(l.detach())}))
