"""Child-process body of tests/test_gpu_unet.py::test_graphed_step_matches_eager_trajectory."""
import copy, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import torch
import Model
import loss as L
from umi.graphs import GraphedStep

DEV = "cuda"
L.CLASS_NUMBER = 2
torch.manual_seed(21)
base = Model.UNet(1, 2, 8, compute_dtype="fp16").to(DEV).train()
x = torch.randn(2, 1, 64, 64, device=DEV)
lab = torch.randint(0, 2, (2, 64, 64), device=DEV).float()


def make(model):
    # the fused optimizer (one launch per step; its descriptor table is uploaded from pinned memory, also under capture)
    # unless "torch" is passed on the command line
    from umi import optim as umi_optim
    cls = torch.optim.SGD if "torch" in sys.argv[1:] else umi_optim.SGD
    opt = cls(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)

    def step(xx, yy):
        loss = L.calc_loss(model(xx), yy, loss_type="dice_bce_mc")
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss
    return step


m_g, m_e = copy.deepcopy(base), copy.deepcopy(base)
gs = GraphedStep(make(m_g), [x, lab], warmup=2)             # 2 warm-up steps have run on m_g (capturing executes nothing)
step_e = make(m_e)
for _ in range(2):
    le = step_e(x, lab)
for i in range(4):
    lg = gs(x, lab).clone()
    le = step_e(x, lab)
    assert torch.equal(lg, le.detach()), (i, float(lg), float(le))
for pg, pe in zip(m_g.parameters(), m_e.parameters()):
    assert torch.equal(pg, pe)
print("GRAPHED_STEP_OK", float(lg))
