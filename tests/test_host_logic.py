"""CPU tests of the host-side mirror (Trainer, loss) and of the drop-in module surface."""
import os
import re

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader, TensorDataset

from oracle import recipe, ref_unet
from tests.test_oracle_golden import _sig_close, sig


def test_loss_matches_oracle_and_reference_value(golden_dir):
    import loss as L
    g = np.load(os.path.join(golden_dir, "unet_1_2_8.npz"))
    logits = torch.from_numpy(g["logits"])
    _, lab = recipe.synthetic_batch(int(g["B"]), 1, int(g["H"]), int(g["W"]), 2, seed=int(g["seed"]))
    L.CLASS_NUMBER = 2
    v = L.calc_loss(logits, lab, loss_type="dice_bce_mc")
    assert abs(v.item() - float(g["loss0"])) < 2e-6          # the reference's own number
    assert abs(v.item() - ref_unet.dice_bce_mc(logits, lab, 2).item()) < 1e-7
    with pytest.raises(NotImplementedError):
        L.calc_loss(logits, lab, loss_type="HausdorffDTLoss")
    with pytest.raises(ValueError):
        L.calc_loss(logits, lab, loss_type="nope")


def test_trainer_reproduces_reference_run(golden_dir, tmp_path):
    """Product Trainer driving the CPU oracle model == reference Trainer driving reference UNet."""
    import loss as L
    from Trainer import Trainer
    g = np.load(os.path.join(golden_dir, "trainer_single.npz"))
    torch.manual_seed(0)
    L.CLASS_NUMBER = 2
    m = ref_unet.RefUNet(1, 2, 8, False)
    m.load_state_dict(recipe.fill_state_dict(m.state_dict(), seed=21))
    xs, ls = recipe.synthetic_batch(6, 1, 32, 32, 2, seed=21)
    loaders = {"train": DataLoader(TensorDataset(xs[:4], ls[:4]), batch_size=2, shuffle=False),
               "val": DataLoader(TensorDataset(xs[4:], ls[4:]), batch_size=1)}
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    tr = Trainer(m, "single", torch.FloatTensor, "cpu", str(tmp_path), loaders, 2, opt, 25, 2,
                 "dice_bce_mc", "dice_bce_mc", lr_scheduler=True)
    out = tr.train()
    assert out is m
    np.testing.assert_allclose(tr.train_loss_list, g["train_loss"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(tr.val_loss_list, g["val_loss"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(tr.val_score_list, g["val_score"], rtol=0, atol=5e-6)
    assert tr.iter_num == int(g["iter_num"])
    assert abs(opt.param_groups[0]["lr"] - float(g["final_lr"])) < 1e-12
    assert sorted(os.listdir(tmp_path / "models")) == list(g["files"])
    assert (tmp_path / "total.png").exists()
    log = (tmp_path / "logs.txt").read_text()
    ref_log = str(g["log"])
    pick = lambda s: re.findall(r"(Epoch \d+/\d+|LR [0-9.e-]+|saving best model)", s)
    assert pick(log) == pick(ref_log)
    for k, v in m.state_dict().items():
        _sig_close(sig(v.float()), g["final." + k], rtol=5e-4)
    # checkpoint compatibility (reference Trainer.py:759-765,808: bare state_dicts): what the Trainer wrote loads, with the
    # safe loader and strict key matching, into the reference-shaped oracle model and into the product model
    import Model
    for name in ("best.pt", "last_epoch.pt"):
        sd = torch.load(tmp_path / "models" / name, weights_only=True)
        assert list(sd.keys()) == list(m.state_dict().keys())
        ref_unet.RefUNet(1, 2, 8, False).load_state_dict(sd, strict=True)
        prod = Model.UNet(1, 2, 8, False)
        prod.load_state_dict(sd, strict=True)
        assert all(torch.equal(a, b) for a, b in zip(prod.state_dict().values(), sd.values()))


def test_trainer_rejects_out_of_scope(tmp_path):
    from Trainer import Trainer
    m = torch.nn.Linear(1, 1)
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    loaders = {"train": [0], "val": [0]}
    mk = lambda mt, lf="mse": Trainer(m, mt, torch.FloatTensor, "cpu", str(tmp_path), loaders, 1, opt, 1, 1, lf, lf)
    with pytest.raises(ValueError):
        mk("bogus").train()
    with pytest.raises(NotImplementedError):
        mk("CLTR").train()
    with pytest.raises(NotImplementedError):
        mk("single", "TopoLoss").train()


def test_product_unet_surface_matches_reference_contract():
    """Constructor signature, state_dict keys/shapes and init stream; no GPU needed to construct."""
    import inspect
    import Model
    sig_ = inspect.signature(Model.UNet.__init__)
    assert list(sig_.parameters)[:7] == ["self", "n_channels", "n_classes", "initial_feature_map", "usa_cuda",
                                         "dropout", "dropout_p"]
    assert sig_.parameters["initial_feature_map"].default == 64
    torch.manual_seed(0)
    m = Model.UNet(1, 2, 8, False)
    ref = ref_unet.RefUNet(1, 2, 8, False)
    sd, rsd = m.state_dict(), ref.state_dict()
    assert list(sd.keys()) == list(rsd.keys())
    assert all(sd[k].shape == rsd[k].shape and sd[k].dtype == rsd[k].dtype for k in sd)
    m.load_state_dict(rsd)          # reference-shaped checkpoints load
    assert "down1.maxpool_conv.2.double_conv.0.weight" in Model.UNet(-2, 4, 8, True, True).state_dict()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 16, 16))


def test_product_init_matches_reference_rng_stream(golden_dir):
    import Model
    g = np.load(os.path.join(golden_dir, "unet_1_2_8.npz"))
    if "init_sig.inc.double_conv.0.weight" not in g:
        pytest.skip("fixture predates init signatures")
    torch.manual_seed(0)
    m = Model.UNet(1, 2, 8, False)
    for k, v in m.state_dict().items():
        _sig_close(sig(v.float()), g["init_sig." + k], rtol=1e-6)


def test_trainer_multitask_reproduces_reference_run(golden_dir, tmp_path):
    """Product Trainer.multi_task_train driving the CPU oracle's UNet_multitask == reference Trainer.multi_task_train driving
    the reference UNet_multitask (Trainer.py:831-992): per-epoch total and per-task losses, poly-LR, checkpoints, final
    weights.  (The reference itself dies in its plot routine after the last epoch; the product finishes and plots.)"""
    from Trainer import Trainer
    from tools.gen_golden import PairLabels, multitask_trainer_data
    g = np.load(os.path.join(golden_dir, "trainer_multitask.npz"))
    m = ref_unet.RefUNetMultitask(1, 1, 8, False)
    m.load_state_dict(recipe.fill_state_dict(m.state_dict(), seed=22))
    xs, l1, l2 = multitask_trainer_data()
    loaders = {"train": DataLoader(PairLabels(xs[:4], l1[:4], l2[:4]), batch_size=2, shuffle=False),
               "val": DataLoader(PairLabels(xs[4:], l1[4:], l2[4:]), batch_size=1)}
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    tr = Trainer(m, "multi_task", torch.FloatTensor, "cpu", str(tmp_path), loaders, 2, opt, 25, 2, "mse", "mse",
                 lr_scheduler=True)
    assert tr.train() is m
    for mine, key in ((tr.train_loss_list, "train_loss"), (tr.val_loss_list, "val_loss"),
                      (tr.train_loss_list_1, "train_loss_1"), (tr.train_loss_list_2, "train_loss_2"),
                      (tr.val_loss_list_1, "val_loss_1"), (tr.val_loss_list_2, "val_loss_2")):
        np.testing.assert_allclose(mine, g[key], rtol=0, atol=5e-6)
    assert tr.iter_num == int(g["iter_num"]) and abs(opt.param_groups[0]["lr"] - float(g["final_lr"])) < 1e-12
    assert abs(tr.best_val_score - float(g["best_val_score"])) < 5e-6          # model selection on the validation loss
    assert sorted(os.listdir(tmp_path / "models")) == list(g["files"])
    assert (tmp_path / "total.png").exists()
    for k, v in m.state_dict().items():
        _sig_close(sig(v.float()), g["final." + k], rtol=5e-4)
    with pytest.raises(NotImplementedError):
        Trainer(m, "multi_task", torch.FloatTensor, "cpu", str(tmp_path), loaders, 2, opt, 25, 2, "multi_task_loss", "mse").train()
