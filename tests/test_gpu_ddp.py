"""Two ranks on ONE MI355X (gloo collectives on device tensors): the HIP tape's data-parallel path end to end --
gradients written into the reducer's buckets by the wgrad kernels, all-reduced, handed to the fused optimizer.
(RCCL needs one GPU per rank; the reducer only sees torch.distributed, so gloo exercises the same code.)"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make_model(kind):
    import Model
    if kind == "unet":
        return Model.UNet(1, 2, 8, compute_dtype="fp32").cuda().train(), 32
    # the small TransUNet in fp16: weight-standardised convs + GroupNorm, 2 ViT blocks, the CUP decoder -- every grouped
    # end-of-backward launch of umi/graph_tu.py writes into the reducer's bucket slots
    from oracle import ref_transunet
    from tests.test_gpu_transunet import product_config
    from TransUnet.vit_seg_modeling import VisionTransformer
    cfg = ref_transunet.small_config(2)
    return VisionTransformer(product_config(cfg, 64), img_size=64, num_classes=2, compute_dtype="fp16").cuda().train(), 64


def _worker(rank, world, port, q, kind="unet"):
    import sys
    import torch.distributed as dist
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [repo, os.path.join(repo, "unet-torch_amd")]
    import Model
    import loss as L
    from umi import ddp, optim as uo
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    L.CLASS_NUMBER = 2
    torch.manual_seed(100)                                         # same replica on both ranks (the broadcast is a no-op)
    m, S = _make_model(kind)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4, 1, S, S, generator=g)
    lab = torch.randint(0, 2, (4, S, S), generator=g).float()
    xs, ls = x[rank * 2:(rank + 1) * 2].cuda(), lab[rank * 2:(rank + 1) * 2].cuda()
    out = {}
    # this rank's own gradient, plain tape without the reducer (BatchNorm running statistics restored afterwards)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    L.calc_loss(m(xs), ls, loss_type="dice_bce_mc").backward()
    out["local"] = [p.grad.detach().cpu().numpy().copy() for p in m.parameters()]
    m.load_state_dict(sd)
    m.zero_grad(set_to_none=True)
    red = ddp.GradReducer(m, world, bucket_mb=0.05)
    opt = uo.SGD(m.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    for step in range(2):
        loss = L.calc_loss(m(xs), ls, loss_type="dice_bce_mc")
        opt.zero_grad()
        loss.backward()
        red.sync()
        if step == 0:
            out["synced"] = [p.grad.detach().cpu().numpy().copy() for p in m.parameters()]
            out["aliased"] = all(p.grad.data_ptr() == red.buffer_for(p).data_ptr() for p in m.parameters())
        opt.step()
    torch.cuda.synchronize()
    out["params"] = [p.detach().cpu().numpy() for p in m.parameters()]

    # deferred mode: forward + backward replayed from a HIP graph that only fills the buckets, then flush() + optimizer
    from umi.graphs import GraphedStep
    if kind != "unet":                                   # (the graph-replayed path is exercised with the U-Net)
        q.put((rank, out))
        dist.barrier()
        dist.destroy_process_group()
        return
    torch.manual_seed(100)
    m2 = Model.UNet(1, 2, 8, compute_dtype="fp32").cuda().train()
    red2 = ddp.GradReducer(m2, world, bucket_mb=0.05)
    red2.deferred = True
    opt2 = uo.SGD(m2.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)

    def fwd_bwd(xx, yy):
        loss = L.calc_loss(m2(xx), yy, loss_type="dice_bce_mc")
        opt2.zero_grad()
        loss.backward()
        return loss
    # (the warm-up pass inside GraphedStep runs forward + backward once: BatchNorm running statistics advance one extra
    #  time, nothing else changes -- no optimizer step is part of the captured function)
    gs = GraphedStep(fwd_bwd, [xs, ls], warmup=1)
    for step in range(2):
        gs(xs, ls)
        red2.flush()
        opt2.step()
    torch.cuda.synchronize()
    out["params_graph"] = [p.detach().cpu().numpy() for p in m2.parameters()]
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["unet", "transunet"])
def test_two_ranks_hip_tape_gradients_are_averaged_and_replicas_stay_identical(kind):
    import numpy as np
    if not torch.cuda.is_available():
        pytest.fail("needs an MI355X")
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = res[0], res[1]
    assert a["aliased"] and b["aliased"]                 # the optimizer reads the all-reduced bucket slots, not a stale copy
    n_diff = 0
    for la, lb, sa, sb in zip(a["local"], b["local"], a["synced"], b["synced"]):
        np.testing.assert_allclose(sa, 0.5 * (la + lb), rtol=2e-5, atol=1e-7)  # mean of the two ranks' own gradients
        np.testing.assert_array_equal(sa, sb)                              # both ranks hold the same reduced gradient
        n_diff += int(np.abs(la - lb).max() > 0)
    assert n_diff > 10                                   # the two shards really produced different local gradients
    for pa, pb in zip(a["params"], b["params"]):
        np.testing.assert_array_equal(pa, pb)            # replicas identical after two optimizer steps
    if kind != "unet":
        return
    for pa, pb, pe in zip(a["params_graph"], b["params_graph"], a["params"]):
        np.testing.assert_array_equal(pa, pb)            # ... also on the graph-replayed, deferred-all-reduce path,
        np.testing.assert_allclose(pa, pe, rtol=1e-6, atol=1e-7)   # which follows the eagerly overlapped path's trajectory
