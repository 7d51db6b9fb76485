"""TransUNet under the data-parallel gradient sink: same gradients, about the same launch count.
R50-ViT-B/16 @224, B = 2, fp16, world size 1 (the reducer's buckets are filled, no collective runs).
Prints `TU_SINK {json}`: worst relative gradient difference sink vs plain, kernel launches per step of both (torch.profiler)."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import torch

from oracle import recipe, ref_transunet


def main():
    import loss as L
    from umi import ddp
    from tests.test_gpu_transunet import product_config
    from TransUnet.vit_seg_modeling import VisionTransformer
    small = len(sys.argv) > 1 and sys.argv[1] == "small"
    img = 64 if small else 224
    cfg = ref_transunet.small_config(2) if small else ref_transunet.r50_vit_b16_config(2, 3, dropout_rate=0.0)
    L.CLASS_NUMBER = 2
    torch.manual_seed(0)
    m = VisionTransformer(product_config(cfg, img), img_size=img, num_classes=2, compute_dtype="fp16")
    m.load_state_dict(recipe.fill_state_dict(m.state_dict(), seed=3, negative_gamma=False))
    m.to("cuda").train()
    x, lab = recipe.synthetic_batch(2, 1, img, img, 2, seed=3)
    x, lab = x.cuda(), lab.cuda()

    def step():
        m.zero_grad(set_to_none=True)
        L.calc_loss(m(x), lab, loss_type="dice_bce_mc").backward()

    def launches():
        from torch.profiler import profile, ProfilerActivity
        step()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        return sum(e.count for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA)

    sd = {k: v.clone() for k, v in m.state_dict().items()}
    step()
    plain = [p.grad.detach().clone() for p in m.parameters()]
    n_plain = launches()
    m.load_state_dict(sd)
    red = ddp.GradReducer(m, world_size=1, bucket_mb=32.0)
    step()
    red.sync()
    worst, where = 0.0, ""
    for (k, p), g in zip(m.named_parameters(), plain):
        assert p.grad.data_ptr() == red.buffer_for(p).data_ptr(), k
        d = (p.grad.double() - g.double()).norm().item() / (g.double().norm().item() + 1e-12)
        if d > worst and g.double().norm().item() > 1e-6:
            worst, where = d, k
    n_sink = launches()
    red.sync()
    print("TU_SINK " + json.dumps({"config": "small" if small else "R50-ViT-B/16 @224 B=2 fp16", "buckets": len(red.buckets),
                                   "worst_rel_grad_diff": worst, "at": where, "launches_plain": n_plain, "launches_sink": n_sink}))


if __name__ == "__main__":
    main()
