"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement (PyTorch fp32, CPU) of the reference hot path (caki35/UNet-Torch:
Model.py U-Net blocks, loss.py 'dice_bce_mc', the Trainer step, and the TransUNet
R50-ViT-B/16 path).  It exists to *check* the HIP product path and to be timed as
the `cpu_baseline` leg of bench.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything from this package.  Nothing under `unet-torch_amd/` imports it; the
product path raises if its HIP extension is missing instead of falling back here.

Parity pinning: the reference ships no tests, golden vectors or fixtures
(SURVEY.md section 4), so this restatement is pinned against outputs of the
reference itself, produced in the build container by tools/gen_golden.py
(imports /root/reference with three stub modules) and committed as small
fixtures under tests/golden/.  tests/test_oracle_golden.py replays them.
"""
