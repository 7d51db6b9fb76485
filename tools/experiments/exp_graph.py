"""Experiment: capture one whole training step (forward + loss + backward + SGD) in a HIP graph (torch.cuda.graph) and
replay it -- every libunetmi launch goes to the current stream, so it is captured like any torch kernel.
usage: exp_graph.py unet|transunet [steps]"""
import copy, json, os, sys, time, faulthandler
faulthandler.enable()
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import torch
import loss as L

which = sys.argv[1] if len(sys.argv) > 1 else "unet"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L.CLASS_NUMBER = 2
torch.manual_seed(0)
if which == "unet":
    import Model
    B, size = 16, 512
    m = Model.UNet(1, 2, 64, compute_dtype="fp16").cuda().train()
else:
    from TransUnet.vit_seg_modeling import VisionTransformer, CONFIGS
    B, size = 24, 224
    cfg = copy.deepcopy(CONFIGS["R50-ViT-B_16"]); cfg.n_classes = 2; cfg.n_skip = 3; cfg.patches.grid = (size // 16, size // 16)
    m = VisionTransformer(cfg, img_size=size, num_classes=2, compute_dtype="fp16").cuda().train()
opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
x = torch.randn(B, 1, size, size, device="cuda")
lab = torch.randint(0, 2, (B, size, size), device="cuda").float()


def step():
    out = m(x)
    l = L.calc_loss(out, lab, loss_type="dice_bce_mc")
    opt.zero_grad(set_to_none=True)
    l.backward()
    opt.step()
    return l


def timeit(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize(); print("warm ok", flush=True)
# capture straight after the side-stream warm-up (an eager step on the default stream in between leaves autograd state
# that makes hipStreamEndCapture crash on ROCm 7.2), time the replays, then time eager last
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static_loss = step()
print("captured", flush=True)
graph_ms, _ = timeit(g.replay, steps)
l0 = float(static_loss)
g.replay(); torch.cuda.synchronize()
l1 = float(static_loss)
eager_ms, l_e = timeit(step, steps)
print(json.dumps({"model": which, "eager_ms": round(eager_ms, 2), "graph_ms": round(graph_ms, 2), "loss_after_replays": [l0, l1],
                  "eager_loss": float(l_e)}))
