"""What shader clock does the card sustain while the conv3x3 MFMA kernel runs?  (the roofline peak assumes 2.4 GHz)
Samples rocm-smi in a side thread while one conv shape is launched back-to-back for a few seconds."""
import os, subprocess, sys, threading, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "unet-torch_amd")]
import torch
from umi import ops

samples, stop = [], False


def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            keep = [l.strip() for l in out.splitlines() if ("sclk" in l or "Power" in l or "fclk" in l or "mclk" in l) and "GPU[0]" in l]
            samples.append((time.time(), keep))
        except Exception as e:      # noqa
            samples.append((time.time(), [repr(e)]))
        time.sleep(0.25)


n, h, w, ci, co = 16, 64, 64, 512, 512
x = torch.randn(n, h, w, ci, device="cuda").half()
wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.02
y = torch.empty(n, h, w, co, device="cuda", dtype=torch.float16)
pk = ops.pack_conv_fwd(wt, torch.float16, k8=True)
dW = torch.empty(co, ci, 3, 3, device="cuda")
tx = ops.passthrough_tx(ci, "cuda"); tx[:, 3] = 0
MODE = os.environ.get("KERNEL", "fwd")


def launch():
    if MODE == "wgrad":
        ops.conv_wgrad(x, tx, y, None, dW, ci * 9, 9, 1, 1.0, 3, 3, 1, 1)
    else:
        ops.conv_fwd(x, None, lambda l: pk, None, y, 3, 3, 1, 1)


launch()
torch.cuda.synchronize()
th = threading.Thread(target=poll); th.start()
time.sleep(1.0)
t_idle = time.time()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.time()
iters = 0
e0.record()
while time.time() - t0 < 6.0:
    for _ in range(200):
        launch()
    iters += 200
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
t_end = time.time()
time.sleep(0.5)
stop = True; th.join()
ms = e0.elapsed_time(e1) / iters
print(f"{MODE} {ci}->{co}@{h}: {ms*1e3:.1f} us/launch, {2*n*h*w*ci*co*9/ms/1e9:.0f} TFLOP/s")
import re
for t, k in samples:
    tag = "idle" if t < t_idle else ("LOAD" if t < t_end else "after")
    print(tag, " ".join(re.findall(r"(sclk|Power)[^(]*\(?W?\)?:? *\(?([0-9.]+)", " | ".join(k)).__repr__().split()))
