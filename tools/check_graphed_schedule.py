"""Child-process body of tests/test_gpu_graph_lr.py: learning-rate schedule and Adam step count under HIP-graph replay.

  optim   : umi.optim.SGD / Adam with the device-resident hyper block (poly rule, reference Trainer.py:722-726), steps replayed
            from a GraphedStep, against the same optimizer class stepped eagerly with the host-side rule: same trajectory;
            an Adam step captured WITHOUT the device block raises.
  trainer : the product Trainer (model_type 'single', poly LR) driving the HIP U-Net, eagerly and with graph=True, against the
            reference's own run (tests/golden/trainer_single.npz): losses, iter_num, final LR.
"""
import copy, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import numpy as np
import torch
import Model
import loss as L
from oracle import recipe
from umi import optim as uo
from umi.graphs import GraphedStep

DEV = "cuda"
what = sys.argv[1]

if what == "optim":
    L.CLASS_NUMBER = 2
    torch.manual_seed(21)
    base = Model.UNet(1, 2, 8, compute_dtype="fp32").to(DEV).train()
    x = torch.randn(2, 1, 32, 32, device=DEV)
    lab = torch.randint(0, 2, (2, 32, 32), device=DEV).float()
    MAXIT, BASE = 10, {"sgd": 0.05, "adam": 2e-3}
    for kind in ("sgd", "adam"):
        def mk(model):
            if kind == "sgd":
                return uo.SGD(model.parameters(), lr=BASE[kind], momentum=0.9, weight_decay=1e-4)
            return uo.Adam(model.parameters(), lr=BASE[kind], weight_decay=1e-4)

        def step_fn(model, opt):
            def step(xx, yy):
                loss = L.calc_loss(model(xx), yy, loss_type="dice_bce_mc")
                opt.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()
                return loss.detach()
            return step

        m_g, m_e = copy.deepcopy(base), copy.deepcopy(base)
        o_g, o_e = mk(m_g), mk(m_e)
        poly = dict(base_lr=BASE[kind], max_iterations=MAXIT, power=0.9, iter_num=0)
        if kind == "adam":                                   # capture without the device block must be refused, not silently wrong
            m_x = copy.deepcopy(base)
            o_x = mk(m_x)
            try:
                GraphedStep(step_fn(m_x, o_x), [x, lab], warmup=1)
                raise SystemExit("Adam capture without device_schedule() did not raise")
            except RuntimeError as e:
                assert "device_schedule" in str(e), e
            torch.cuda.synchronize()
        gs = GraphedStep(step_fn(m_g, o_g), [x, lab], warmup=2, optimizers=[o_g], poly=poly)
        se = step_fn(m_e, o_e)
        losses_g, losses_e, it = [], [], 0
        for i in range(6):
            if i >= 2:
                losses_g.append(float(gs(x, lab)))
            losses_e.append(float(se(x, lab)))
            lr_ = BASE[kind] * (1.0 - it / MAXIT) ** 0.9           # host-side rule, pre-increment iter_num
            for g_ in o_e.param_groups:
                g_["lr"] = lr_
            it += 1
        # warm-up steps 0,1 ran inside GraphedStep (eagerly, device schedule already on), 2..5 were replays
        np.testing.assert_allclose(losses_g, losses_e[2:], rtol=2e-5, atol=1e-6)
        for pg, pe in zip(m_g.parameters(), m_e.parameters()):
            torch.testing.assert_close(pg, pe, rtol=2e-4, atol=2e-6)
        h = o_g.sync_host()[0]
        assert abs(o_g.param_groups[0]["lr"] - o_e.param_groups[0]["lr"]) < 1e-12 * BASE[kind] + 1e-15, (o_g.param_groups[0]["lr"], o_e.param_groups[0]["lr"])
        assert int(h["iter"]) == 6
        if kind == "adam":
            sd = o_g.state_dict()["state"]
            assert all(float(v["step"]) == 6.0 for v in sd.values()), [float(v["step"]) for v in sd.values()][:3]
        print(kind, "ok", losses_g[-1], o_g.param_groups[0]["lr"])
    print("GRAPHED_SCHEDULE_OK")

elif what == "guard":
    # the capture guard: a live loss of an eager step must give a RuntimeError, not the hipStreamEndCapture segfault
    L.CLASS_NUMBER = 2
    torch.manual_seed(0)
    m = Model.UNet(1, 2, 8, compute_dtype="fp16").to(DEV).train()
    opt = uo.SGD(m.parameters(), lr=0.01, momentum=0.9)
    x = torch.randn(2, 1, 64, 64, device=DEV)
    lab = torch.randint(0, 2, (2, 64, 64), device=DEV).float()

    def step(xx, yy):
        l = L.calc_loss(m(xx), yy, loss_type="dice_bce_mc")
        opt.zero_grad(set_to_none=True)
        l.backward()
        opt.step()
        return l
    kept = step(x, lab)                       # eager step on the default stream, its graph stays referenced
    torch.cuda.synchronize()
    try:
        GraphedStep(step, [x, lab], warmup=1, optimizers=[opt])
        raise SystemExit("capture with a live eager-step graph was not refused")
    except RuntimeError as e:
        assert "gradient-accumulator" in str(e), e
    kept = kept.detach()
    gs = GraphedStep(step, [x, lab], warmup=1, optimizers=[opt])
    l0 = float(gs(x, lab).detach())
    assert l0 == l0
    print("GRAPH_GUARD_OK", l0)

elif what == "trainer":
    import tempfile
    from torch.utils.data import DataLoader, TensorDataset
    from Trainer import Trainer
    g = np.load(os.path.join(REPO, "tests", "golden", "trainer_single.npz"))
    for graph in (False, True):
        L.CLASS_NUMBER = 2
        torch.manual_seed(0)
        m = Model.UNet(1, 2, 8, False, compute_dtype="fp32")
        m.load_state_dict(recipe.fill_state_dict(m.state_dict(), seed=21))
        m.to(DEV)
        xs, ls = recipe.synthetic_batch(6, 1, 32, 32, 2, seed=21)
        loaders = {"train": DataLoader(TensorDataset(xs[:4], ls[:4]), batch_size=2, shuffle=False),
                   "val": DataLoader(TensorDataset(xs[4:], ls[4:]), batch_size=1)}
        opt = uo.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        with tempfile.TemporaryDirectory() as td:
            tr = Trainer(m, "single", torch.cuda.FloatTensor, DEV, td, loaders, 2, opt, 25, 2, "dice_bce_mc", "dice_bce_mc",
                         lr_scheduler=True, graph=graph)
            tr.train()
            files = sorted(os.listdir(os.path.join(td, "models")))
        if graph:
            assert len(tr._graphs) == 1, "the graph path was not taken"
            opt.sync_host()
        np.testing.assert_allclose(tr.train_loss_list, g["train_loss"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(tr.val_loss_list, g["val_loss"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(tr.val_score_list, g["val_score"], rtol=2e-4, atol=2e-5)
        assert tr.iter_num == int(g["iter_num"])
        assert abs(opt.param_groups[0]["lr"] - float(g["final_lr"])) < 1e-12, (opt.param_groups[0]["lr"], float(g["final_lr"]))
        assert files == list(g["files"])
        print("trainer graph=%s ok" % graph, tr.train_loss_list, opt.param_groups[0]["lr"])
    print("GRAPHED_TRAINER_OK")

elif what == "trainer_mt":
    # the multi-task branch of the graph mode, a ragged last batch (its own capture from the second epoch on)
    # over 3 epochs, and a resume with iter_num > 0 feeding the device-side poly block -- graph=True must follow graph=False.
    import tempfile
    from torch.utils.data import DataLoader
    from Trainer import Trainer
    from tools.gen_golden import PairLabels, multitask_trainer_data
    runs = {}
    for graph in (False, True):
        torch.manual_seed(0)
        m = Model.UNet_multitask(1, 1, 8, False, compute_dtype="fp32")
        m.load_state_dict(recipe.fill_state_dict(m.state_dict(), seed=22))
        m.to(DEV)
        xs, l1, l2 = multitask_trainer_data()
        loaders = {"train": DataLoader(PairLabels(xs[:5], l1[:5], l2[:5]), batch_size=2, shuffle=False),     # batches of 2, 2, 1
                   "val": DataLoader(PairLabels(xs[5:], l1[5:], l2[5:]), batch_size=1)}
        opt = uo.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        with tempfile.TemporaryDirectory() as td:
            tr = Trainer(m, "multi_task", torch.cuda.FloatTensor, DEV, td, loaders, 2, opt, 25, 3, "mse", "mse",
                         lr_scheduler=True, graph=graph)
            tr.iter_num, tr.max_iterations = 4, 20    # a resumed run: the poly rule starts in the middle of its curve
            tr.train()
        if graph:
            assert len(tr._graphs) == 2, ("expected one graph per batch shape (full, ragged)", list(tr._graphs))
            opt.sync_host()
        runs[graph] = dict(train=list(tr.train_loss_list), t1=list(tr.train_loss_list_1), t2=list(tr.train_loss_list_2),
                           val=list(tr.val_loss_list), it=tr.iter_num, lr=opt.param_groups[0]["lr"],
                           w=torch.cat([p.detach().flatten() for p in m.parameters()]).double().cpu().numpy())
        print("trainer_mt graph=%s" % graph, runs[graph]["train"], runs[graph]["it"], runs[graph]["lr"])
    a, b = runs[False], runs[True]
    for k in ("train", "t1", "t2", "val"):
        np.testing.assert_allclose(b[k], a[k], rtol=2e-4, atol=2e-5)
    assert a["it"] == b["it"] == 4 + 9, (a["it"], b["it"])
    assert abs(a["lr"] - b["lr"]) < 1e-12, (a["lr"], b["lr"])
    rel = np.linalg.norm(a["w"] - b["w"]) / np.linalg.norm(a["w"])
    assert rel < 1e-5, rel
    print("GRAPHED_TRAINER_MT_OK", rel)
