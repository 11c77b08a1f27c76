"""Stress of the in-kernel hand-offs under UNEVEN load (GPU box only): the cfg2-size LSTM forward + backward is
repeated while a second stream streams large copies through HBM / L2 and occupies CUs at random moments; every
output must equal, bit for bit, the result of the quiet run, and no bounded wait may time out."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cerebralsignalnetworks_amd.lstm_model import HipLSTM

dev = torch.device("cuda:0")
B, T, C, H, L = 256, 500, 128, 768, 2
torch.manual_seed(1)
# (second argument "f32": the exact-float32 path's weight-stationary kernels, lstm_f32_persist.hip)
cdt = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else torch.bfloat16
m = HipLSTM(C, H, L, compute_dtype=cdt).to(dev)
x = torch.randn(B, T, C, device=dev)
dy = torch.randn(B, H, device=dev)

def run():
    for p in m.parameters():
        p.grad = None
    xt = x.clone().requires_grad_(True)
    y = m(xt)
    (y * dy).sum().backward()
    torch.cuda.synchronize()
    for plan in m.all_plans():
        assert plan.status() == 0, "a bounded in-kernel wait timed out"
    return [y.detach().clone()] + [p.grad.clone() for p in m.parameters()] + [xt.grad.clone()]

ref = run()
side = torch.cuda.Stream()
big = [torch.empty(64 << 20, dtype=torch.float32, device=dev) for _ in range(3)]      # 256 MB each
bad = 0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for rep in range(reps):
    stop = torch.zeros(1)
    with torch.cuda.stream(side):
        for k in range((60 + 20 * (rep % 3)) * (5 if cdt == torch.float32 else 1)):              # ~ the duration of one forward + backward
            big[(k + 1) % 3].copy_(big[k % 3])
            if k % 7 == rep % 7:
                big[2].mul_(1.0001)
    out = run()
    side.synchronize()
    diff = [int((a != b).sum().item()) for a, b in zip(out, ref)]
    if any(diff):
        bad += 1
        print("rep", rep, "MISMATCH", diff, flush=True)
    else:
        print("rep", rep, "identical", flush=True)
print("stress summary (%s, kernels %s): %d of %d repetitions differ" % (str(cdt).split(".")[-1], "/".join(m.all_plans()[0].kernel_names()), bad, reps))
sys.exit(1 if bad else 0)
