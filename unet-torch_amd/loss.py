"""Losses of the hot path: drop-in for the reference `loss.py` entry point `calc_loss`.

`calc_loss(pred, target, bce_weight=0.5, loss_type='mse')` keeps the reference signature
(loss.py:442) and the module-level `CLASS_NUMBER` that `train.py:163` sets.  The branch on
the training hot path is 'dice_bce_mc' (loss.py:488-500): 0.5*CrossEntropy + 0.5*Dice of
the softmax (DiceLoss, loss.py:215-251: per-class `1 - (2*sum(p*t)+1e-5)/(sum(p*p)+sum(t*t)+1e-5)`
over the whole batch, mean over classes).  It runs as fp32 device ops on the logits the HIP
network returns; unlike the reference it does not `.item()`-sync once per class.

The dataset-specific research losses of the reference (Hausdorff, ActiveContour, Focal/Tversky,
TopK, ...) are out of the hot-path scope and raise.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

CLASS_NUMBER = 2      # overwritten by the caller, as train.py:163 does for the reference


class DiceLoss(nn.Module):
    """Multi-class soft Dice on probabilities (reference loss.py:215-251)."""

    def __init__(self, n_classes):
        super().__init__()
        self.n_classes = n_classes

    def forward(self, inputs, target, weight=None, softmax=False):
        if softmax:
            inputs = torch.softmax(inputs, dim=1)
        if weight is None:
            weight = [1] * self.n_classes
        if inputs.shape[1] != self.n_classes or inputs.shape[0] != target.shape[0] \
                or inputs.shape[2:] != target.shape[1:]:
            raise AssertionError(f"predict {tuple(inputs.shape)} & target {tuple(target.shape)} shape do not match")
        total = 0.0
        for c in range(self.n_classes):
            t = (target == c).float()
            p = inputs[:, c]
            dice = (2 * torch.sum(p * t) + 1e-5) / (torch.sum(p * p) + torch.sum(t * t) + 1e-5)
            total = total + (1 - dice) * weight[c]
        return total / self.n_classes


_OUT_OF_SCOPE = {"TopK", "BCE_HEM", "FL", "dice", "dice_bce", "dice_score", "log_cosh_dice_loss", "dice_score_mc",
                 "HausdorffDTLoss", "HausdorffERLoss", "ActiveContourLoss", "Tversky"}


def calc_loss(pred, target, bce_weight=0.5, loss_type='mse'):
    if loss_type == 'dice_bce_mc':
        loss_ce = F.cross_entropy(pred, target.long())
        loss_dice = DiceLoss(CLASS_NUMBER)(pred, target, softmax=True)
        return 0.5 * loss_ce + 0.5 * loss_dice
    if loss_type == 'CE':
        return F.cross_entropy(pred, target.long())
    if loss_type == 'BCE':
        return F.binary_cross_entropy_with_logits(pred.squeeze(1), target)
    if loss_type == 'mse':
        return F.mse_loss(pred.squeeze(1), target)
    if loss_type == 'mseMC':
        return F.mse_loss(pred, target)
    if loss_type == 'rmse':
        return torch.sqrt(F.mse_loss(pred, target))
    if loss_type == 'l1loss':
        return F.l1_loss(pred, target)
    if loss_type in _OUT_OF_SCOPE:
        raise NotImplementedError(f"loss_type {loss_type!r} is outside the MI355X hot-path scope")
    raise ValueError(f"unknown loss_type {loss_type!r}")
