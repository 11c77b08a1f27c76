"""Two weight-stationary plans on two streams of one device, enqueued back to back so that their launches can overlap in time:
the kernels need all workgroups of a launch co-resident (one per CU), so two of them dispatched concurrently could starve each
other until the bounded waits time out.  Reports status words and agreement with the serial results.
      python tests/diag/two_streams.py [reps] [bf16|f32]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cerebralsignalnetworks_amd.lstm_model import HipLSTM      # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cdt = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else torch.bfloat16
B, T, C, H, L = 256, 200, 128, 768, 2
torch.manual_seed(0)
ms = [HipLSTM(C, H, L, compute_dtype=cdt).to(dev) for _ in range(2)]
xs = [torch.randn(B, T, C, device=dev) for _ in range(2)]
ws = [torch.randn(B, H, device=dev) for _ in range(2)]


def step(i):
    for p in ms[i].parameters():
        p.grad = None
    (ms[i](xs[i]) * ws[i]).sum().backward()
    return [p.grad for p in ms[i].parameters()]


ref = []
for i in range(2):
    g = step(i)
    torch.cuda.synchronize()
    ref.append([q.clone() for q in g])
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
bad = 0
for rep in range(reps):
    outs = [None, None]
    for i in range(2):
        streams[i].wait_stream(torch.cuda.current_stream())
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            outs[i] = step(i)
    torch.cuda.synchronize()
    st = [pl.status() for m in ms for pl in m.all_plans()]
    same = all(bool((a == b).all().item()) for i in range(2) for a, b in zip(outs[i], ref[i]))
    ok = not any(st) and same
    bad += 0 if ok else 1
    print(f"{'ok  ' if ok else 'FAIL'} rep {rep}: status {st}, gradients identical to the serial run: {same}", flush=True)
    if any(st):
        for m in ms:
            for pl in m.all_plans():
                pl.status_clear() if hasattr(pl, "status_clear") else None
print(f"{bad} of {reps} concurrent repetitions failed ({str(cdt).split('.')[-1]}, kernels {'/'.join(ms[0].all_plans()[0].kernel_names())})")
sys.exit(1 if bad else 0)
