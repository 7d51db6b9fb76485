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


_TARGET_DTYPES = {torch.int64: 0, torch.float32: 1, torch.uint8: 2, torch.int32: 3}


class _FusedDiceCE(torch.autograd.Function):
    """'dice_bce_mc' on device logits as two streaming libunetmi kernels per direction (csrc/loss_kernels.hip) instead of
    ~25 elementwise / reduction launches; same arithmetic as the composite below (fp64 final sums, deterministic)."""

    @staticmethod
    def forward(ctx, pred, target):
        from umi import lib as L, ops
        N, C = pred.shape[0], pred.shape[1]
        HW = pred[0, 0].numel()
        stats = torch.empty(3 * C + 2, dtype=torch.float32, device=pred.device)
        ws = ops.workspace(L.fn("umi_dice_ce_ws_bytes")(N, C, HW), pred.device)
        L.check(L.fn("umi_dice_ce_fwd")(pred.data_ptr(), target.data_ptr(), _TARGET_DTYPES[target.dtype], N, C, HW,
                                        stats.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()), "umi_dice_ce_fwd")
        ctx.save_for_backward(pred, target, stats)
        return stats[3 * C + 1].clone()

    @staticmethod
    def backward(ctx, gout):
        from umi import lib as L, ops
        pred, target, stats = ctx.saved_tensors
        N, C = pred.shape[0], pred.shape[1]
        HW = pred[0, 0].numel()
        g = gout.detach().to(torch.float32).contiguous()
        dl = torch.empty_like(pred)
        L.check(L.fn("umi_dice_ce_bwd")(pred.data_ptr(), target.data_ptr(), _TARGET_DTYPES[target.dtype], stats.data_ptr(),
                                        g.data_ptr(), N, C, HW, dl.data_ptr(), ops._stream()), "umi_dice_ce_bwd")
        return dl, None


def _fused_ok(pred, target):
    return (pred.is_cuda and pred.dtype == torch.float32 and pred.dim() >= 3 and pred.is_contiguous() and pred.shape[1] <= 8
            and pred.shape[1] == CLASS_NUMBER and target.is_cuda and target.is_contiguous() and target.dtype in _TARGET_DTYPES
            and target.shape[0] == pred.shape[0] and tuple(target.shape[1:]) == tuple(pred.shape[2:]))


def calc_loss(pred, target, bce_weight=0.5, loss_type='mse'):
    if loss_type == 'dice_bce_mc':
        if _fused_ok(pred, target):
            return _FusedDiceCE.apply(pred, target)
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
