"""World-size-2 gloo (CPU) test of the gradient reducer: bucketing, averaging, parameter broadcast,
and the tape-facing sink protocol (buffer_for / mark_ready / finish)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import recipe, ref_unet


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [repo, os.path.join(repo, "unet-torch_amd")]
    from umi import ddp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    torch.manual_seed(100 + rank)                      # deliberately different replicas before broadcast
    m = ref_unet.RefUNet(1, 2, 8, False).train()
    red = ddp.GradReducer(m, world, bucket_mb=0.25)     # several buckets
    assert len(red.buckets) > 2
    x, lab = recipe.synthetic_batch(4, 1, 32, 32, 2, seed=9)
    xs, ls = x[rank * 2:(rank + 1) * 2], lab[rank * 2:(rank + 1) * 2]
    loss = ref_unet.dice_bce_mc(m(xs), ls, 2)
    loss.backward()
    local = [p.grad.clone() for p in m.parameters()]
    red.sync()                                          # generic path (model does not use the tape)
    avg = [p.grad.clone() for p in m.parameters()]
    # sink protocol, as the tape drives it: write pre-scaled grads into the bucket slots in reverse order
    for p, g in reversed(list(zip(m.parameters(), local))):
        v = red.buffer_for(p)
        v.copy_(g * red.grad_scale)
        red.mark_ready(p)
    red.finish()
    sink = [red.buffer_for(p).clone() for p in m.parameters()]
    red.reset()
    # deferred mode (graph-replayed backward): filling the buckets launches nothing, flush() reduces them afterwards and
    # points .grad at the slots
    red.deferred = True
    for p, g in reversed(list(zip(m.parameters(), local))):
        red.buffer_for(p).copy_(g * red.grad_scale)
        red.mark_ready(p)
    red.finish()
    untouched = all(torch.equal(red.buffer_for(p), g * red.grad_scale) for p, g in zip(m.parameters(), local))
    red.flush()
    deferred = [p.grad.clone() for p in m.parameters()]
    aliased = all(p.grad.data_ptr() == red.buffer_for(p).data_ptr() for p in m.parameters())
    red.deferred = False
    q.put((rank, [t.numpy() for t in local], [t.numpy() for t in avg], [t.numpy() for t in sink],
           [p.detach().numpy() for p in m.parameters()], [t.numpy() for t in deferred], untouched and aliased))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_reducer_world2():
    import numpy as np
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, l0, a0, s0, w0, d0, ok0), (_, l1, a1, s1, w1, d1, ok1) = res
    assert ok0 and ok1                       # deferred mode: nothing reduced before flush(); .grad aliases the bucket slots
    for i in range(len(l0)):
        np.testing.assert_allclose(d0[i], (l0[i] + l1[i]) / 2, rtol=1e-6, atol=1e-8)
        np.testing.assert_array_equal(d0[i], d1[i])
        np.testing.assert_array_equal(w0[i], w1[i])                     # broadcast made replicas identical
        want = (l0[i] + l1[i]) / 2
        np.testing.assert_allclose(a0[i], want, rtol=1e-6, atol=1e-8)
        np.testing.assert_array_equal(a0[i], a1[i])
        np.testing.assert_allclose(s0[i], want, rtol=1e-6, atol=1e-8)
        np.testing.assert_array_equal(s0[i], s1[i])
