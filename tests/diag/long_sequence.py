"""Long sequences: byte offsets beyond 2^31 inside the weight-stationary kernels' buffer resources (H = 1024, B = 256:
the float32 input projection is 4 MB per step, 2 GiB at T = 512).  Compares the default path with the per-diagonal
launches (pointer arithmetic in 64 bits) step by step.      python tests/diag/long_sequence.py [T]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cerebralsignalnetworks_amd.lstm_model import HipLSTM      # noqa: E402

dev = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 530
bad = 0
for (B, C, H, L) in ((256, 16, 1024, 1), (256, 16, 1024, 2), (256, 128, 768, 2)):
    outs = {}
    for name, env in (("default", {}), ("per-diagonal", {"CSN_NO_PERSIST": "1"})):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            torch.manual_seed(3)
            m = HipLSTM(C, H, L, compute_dtype=torch.bfloat16).to(dev)
            x = torch.randn(B, T, C, device=dev)
            with torch.no_grad():
                y_all, _ = m(x, want_all=True)
            torch.cuda.synchronize()
            outs[name] = y_all.float().cpu().numpy()
            st = [pl.status() for pl in m.all_plans()]
            assert not any(st), st
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    a, b = outs["default"], outs["per-diagonal"]
    per_t = np.linalg.norm((a - b).reshape(B, T, -1), axis=(0, 2)) / np.maximum(1e-9, np.linalg.norm(b.reshape(B, T, -1), axis=(0, 2)))
    worst = int(np.argmax(per_t))
    ok = per_t.max() < 2e-2
    bad += 0 if ok else 1
    print(f"{'ok  ' if ok else 'FAIL'} B{B} T{T} C{C} H{H} L{L}: worst step {worst} rel diff {per_t[worst]:.2e}; steps >= 510: {per_t[510:].max():.2e}", flush=True)
# training: weight gradients of the default path against the per-diagonal launches
for (B, C, H, L) in ((256, 16, 1024, 2), (256, 128, 768, 2)):
    grads = {}
    for name, env in (("default", {}), ("per-diagonal", {"CSN_NO_PERSIST": "1"})):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            torch.manual_seed(3)
            m = HipLSTM(C, H, L, compute_dtype=torch.bfloat16).to(dev)
            x = torch.randn(B, T, C, device=dev)
            w = torch.randn(B, H, device=dev)
            (m(x) * w).sum().backward()
            torch.cuda.synchronize()
            st = [pl.status() for pl in m.all_plans()]
            assert not any(st), st
            grads[name] = {n: q.grad.float().cpu().numpy() for n, q in m.named_parameters()}
            del m, x
            torch.cuda.empty_cache()
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    worst = max((float(np.linalg.norm(grads["default"][n] - grads["per-diagonal"][n]) / max(1e-12, np.linalg.norm(grads["per-diagonal"][n]))), n)
                for n in grads["default"])
    ok = np.isfinite(worst[0]) and worst[0] < 3e-2
    bad += 0 if ok else 1
    print(f"{'ok  ' if ok else 'FAIL'} training B{B} T{T} C{C} H{H} L{L}: worst gradient {worst[1]} rel diff {worst[0]:.2e}", flush=True)
print(f"{bad} failing case(s)")
sys.exit(1 if bad else 0)
