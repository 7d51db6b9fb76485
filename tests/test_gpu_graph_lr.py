"""Graph-safe learning rate (reference Trainer.py:719-727): the poly schedule and Adam's step count advance on
the device, so a HIP-graph-replayed step follows the eager / reference trajectory.  The bodies run in child processes
(tools/check_graphed_schedule.py): stream capture is sensitive to what ran before it in the process."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(what, marker):
    if not torch.cuda.is_available():
        pytest.fail("needs an MI355X")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "check_graphed_schedule.py"), what],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and marker in r.stdout, (r.returncode, r.stdout[-3000:], r.stderr[-3000:])


def test_graphed_sgd_poly_and_adam_follow_eager_trajectory():
    """umi.optim.SGD / Adam replayed from a HIP graph with the device-resident hyper block == the same optimizers stepped
    eagerly with the host-side poly rule; Adam refuses a capture without the block (its step count would freeze)."""
    _run("optim", "GRAPHED_SCHEDULE_OK")


def test_trainer_single_on_hip_model_eager_and_graphed_follow_reference_run():
    """Product Trainer ('single', adaptive_lr) on the HIP U-Net, eager and graph=True, vs the reference's own run
    (tests/golden/trainer_single.npz): epoch losses, validation scores, iter_num, final learning rate, checkpoint files."""
    _run("trainer", "GRAPHED_TRAINER_OK")


def test_capture_guard_refuses_live_eager_graph():
    """Round 1's `hipStreamEndCapture` segfault: a tensor holding the autograd graph of an eager step keeps the parameters'
    AccumulateGrad nodes bound to that step's stream (tools/experiments/exp_graph_accgrad.py reproduces crash and cure).
    GraphedStep(optimizers=[...]) detects it and raises; after the tensor is detached the capture works."""
    _run("guard", "GRAPH_GUARD_OK")


def test_trainer_multitask_graph_mode_with_ragged_batches_follows_eager_trainer():
    """'multi_task' Trainer with graph=True: two heads' losses from the static buffers, a ragged last batch (second capture from
    epoch 2 on) over 3 epochs and a resumed iter_num feeding the device-side poly block, against the same Trainer run eagerly:
    losses, iter_num, learning rate and final weights."""
    _run("trainer_mt", "GRAPHED_TRAINER_MT_OK")
