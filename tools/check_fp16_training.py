"""Does the benchmarked fp16 path TRAIN like the fp32 reference?  (reference Trainer.py:697-727)

UNet(1,2,64) on blob-structured synthetic data: images = smooth random blobs + noise, labels = the blob mask.
Same initial weights and the same batches for
  * the HIP model in fp16 storage (the benchmarked kernels) stepped by umi.optim.SGD(0.01, 0.9, 1e-4), and
  * the CPU oracle (oracle/ref_unet.RefUNet, fp32) stepped by torch.optim.SGD with the same hyper-parameters.
Prints one line `FP16_TRAINING {json}`: both loss trajectories, the eval-mode IoU of both models on a held-out batch after
the last step, and the agreement of the two argmax masks.  usage: python tools/check_fp16_training.py [steps] [size] [batch]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import torch
import torch.nn.functional as F

from oracle import recipe, ref_unet

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4


def blobs(n, seed):
    """[n,1,S,S] z-normalised images and [n,S,S] {0,1} labels: low-pass filtered noise thresholded into blobs."""
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(n, 1, size // 8, size // 8, generator=g)
    field = F.interpolate(z, size=(size, size), mode="bicubic", align_corners=False)
    lab = (field[:, 0] > 0.3).float()
    img = 1.2 * lab.unsqueeze(1) + 0.6 * torch.randn(n, 1, size, size, generator=g)
    img = (img - img.mean(dim=(1, 2, 3), keepdim=True)) / img.std(dim=(1, 2, 3), keepdim=True)
    return img, lab


def iou(mask, lab):
    inter = ((mask == 1) & (lab == 1)).sum().item()
    union = ((mask == 1) | (lab == 1)).sum().item()
    return inter / max(union, 1)


def main():
    import Model
    import loss as L
    from umi import optim as uopt
    torch.manual_seed(0)
    ref = ref_unet.RefUNet(1, 2, 64)
    ref.load_state_dict(recipe.fill_state_dict(ref.state_dict(), seed=21, negative_gamma=False))
    hip = Model.UNet(1, 2, 64, compute_dtype="fp16")
    hip.load_state_dict(ref.state_dict())
    hip = hip.to("cuda")
    o_ref = torch.optim.SGD(ref.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    o_hip = uopt.SGD(hip.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    ref.train(), hip.train()
    lr_, lh_ = [], []
    for s in range(steps):
        x, y = blobs(batch, 100 + s)
        l_ref, _ = ref_unet.train_step(ref, o_ref, x, y, 2)
        out = hip(x.to("cuda"))
        l_hip = L.calc_loss(out, y.to("cuda"), loss_type="dice_bce_mc")
        o_hip.zero_grad()
        l_hip.backward()
        o_hip.step()
        lr_.append(float(l_ref))
        lh_.append(float(l_hip))
    xe, ye = blobs(2 * batch, 9999)
    ref.eval(), hip.eval()
    with torch.no_grad():
        m_ref = ref(xe).argmax(1)
        m_hip = hip(xe.to("cuda")).argmax(1).cpu()
    res = {"steps": steps, "size": size, "batch": batch, "loss_fp32_oracle": [round(v, 5) for v in lr_],
           "loss_fp16_hip": [round(v, 5) for v in lh_],
           "eval_iou_fp32_oracle": round(iou(m_ref, ye), 4), "eval_iou_fp16_hip": round(iou(m_hip, ye), 4),
           "eval_masks_agree": round((m_ref == m_hip).float().mean().item(), 4)}
    print("FP16_TRAINING " + json.dumps(res))


if __name__ == "__main__":
    main()
