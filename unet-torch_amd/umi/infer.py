"""Inference entry on the MI355X: the steps either side of the network in the reference's evaluation scripts
(test_mc3serousv5.py:100-127 `preprocess`, :877-887 forward -> softmax -> argmax -> uint8 mask), SURVEY 8(f) rank 4.

    x = infer.preprocess(img)                 # HWC uint8 / float image (numpy or tensor) -> [1,C,H,W] fp32 on the device
    mask = infer.predict_mask(model, x)       # eval-mode forward (BatchNorm from running statistics) + argmax, uint8 [N,H,W]

    x = infer.preprocess(img, input_size=(512, 512))   # + the reference's cubic scipy.ndimage.zoom resize, on the device

No CPU path: the arithmetic is libunetmi kernels (the resize restates SciPy's spline algorithm, oracle/ref_resize.py).
"""
import numpy as np
import torch

from . import lib as L
from . import ops


def zoom_cubic(img, input_size):
    """scipy.ndimage.zoom(img, (input_size[0] / H, input_size[1] / W[, 1]), order=3) of one HWC / HW image on the device
    (reference test_mc3serousv5.py:100-113); uint8 or float32 in, same type out.  `img`: numpy array or tensor."""
    if isinstance(img, np.ndarray):
        img = torch.from_numpy(np.ascontiguousarray(img))
    hw = img.dim() == 2
    if hw:
        img = img.unsqueeze(-1)
    if img.dim() != 3 or img.shape[2] > 4:
        raise ValueError(f"expected an HW or HWC image with at most 4 channels, got {tuple(img.shape)}")
    if img.dtype not in (torch.uint8, torch.float32):
        img = img.float()
    img = img.contiguous().to("cuda", non_blocking=True)
    H, W, C = img.shape
    oh, ow = int(round(H * (input_size[0] / H))), int(round(W * (input_size[1] / W)))
    out = torch.empty((oh, ow, C), dtype=img.dtype, device=img.device)
    nbytes = L.fn("umi_zoom_cubic_ws_bytes")(H, W, C)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=img.device)
    L.check(L.fn("umi_zoom_cubic_hwc")(img.data_ptr(), 0 if img.dtype == torch.uint8 else 1, out.data_ptr(), H, W, C, oh, ow,
                                       ws.data_ptr(), nbytes, ops._stream()), "umi_zoom_cubic_hwc")
    return out[..., 0] if hw else out


def preprocess(img, reverse_channels=None, input_size=None):
    """Per-channel z-normalisation of one image, HWC (or HW) -> [1,C,H,W] fp32 (reference `preprocess`: mean / np.std over
    H,W in fp64).  The reference reverses the channel order of EVERY 3-D (HWC) input (`transpose((2, 0, 1))[::-1]`,
    test_mc3serousv5.py:124: BGR -> RGB for cv2 images, but 2- and 4-channel inputs are reversed as well) and leaves 2-D
    (HW) inputs alone; `reverse_channels=None` follows that rule, True / False override it.
    `input_size` = (H, W) of the network input: an image of another size is first resized like the reference does, with the
    cubic `scipy.ndimage.zoom` (zoom_cubic above: on the device, same values)."""
    if input_size is not None:
        shp = img.shape
        if shp[0] != input_size[0] or shp[1] != input_size[1]:
            was_2d = len(shp) == 2
            img = zoom_cubic(img, input_size)
            if reverse_channels is None:
                reverse_channels = not was_2d
    if isinstance(img, np.ndarray):
        img = torch.from_numpy(np.ascontiguousarray(img))
    was_hwc = img.dim() == 3
    if img.dim() == 2:
        img = img.unsqueeze(-1)
    if img.dim() != 3 or img.shape[2] > 4:
        raise ValueError(f"expected an HW or HWC image with at most 4 channels, got {tuple(img.shape)}")
    if img.dtype not in (torch.uint8, torch.float32):
        img = img.float()
    img = img.contiguous().to("cuda", non_blocking=True)
    H, W, C = img.shape
    if reverse_channels is None:
        reverse_channels = was_hwc
    out = torch.empty((1, C, H, W), dtype=torch.float32, device=img.device)
    nbytes = L.fn("umi_znorm_ws_bytes")()
    ws = ops.workspace(nbytes, img.device)
    L.check(L.fn("umi_znorm_hwc")(img.data_ptr(), 0 if img.dtype == torch.uint8 else 1, out.data_ptr(), H * W, C,
                                  int(bool(reverse_channels)), ws.data_ptr(), nbytes, ops._stream()), "umi_znorm_hwc")
    return out


def argmax_mask(logits):
    """[N,C,H,W] fp32 logits -> uint8 [N,H,W] class mask (== softmax(dim=1).argmax(dim=1), first maximum wins)."""
    ops._need_cuda(logits)
    if logits.dim() != 4 or logits.dtype != torch.float32:
        raise ValueError("argmax_mask expects fp32 logits [N,C,H,W]")
    logits = logits.contiguous()
    N, C, H, W = logits.shape
    mask = torch.empty((N, H, W), dtype=torch.uint8, device=logits.device)
    L.check(L.fn("umi_argmax_mask")(logits.data_ptr(), mask.data_ptr(), N, C, H * W, ops._stream()), "umi_argmax_mask")
    return mask


@torch.no_grad()
def predict_mask(model, x):
    """Reference evaluation step (test_mc3serousv5.py:879-885): eval-mode forward, softmax, argmax -> uint8 mask.  The
    model's kernel-layout weight copies are cached between calls (ops.PackCache), so a loop over images packs once."""
    was_training = model.training
    model.eval()
    try:
        out = model(x.to("cuda"))
        if isinstance(out, tuple):
            return tuple(argmax_mask(o) for o in out)
        return argmax_mask(out)
    finally:
        model.train(was_training)
