"""Long sequences: byte offsets beyond 2^31 inside the weight-stationary kernels' buffer resources (H = 1024, B = 256:
the float32 input projection is 4 MB per step, 2 GiB at T = 512, 4 GiB at T = 1024).
  1. the default path against the per-diagonal launches (pointer arithmetic in 64 bits), every row, step by step;
  2. the default path against the float64 ORACLE (oracle/lstm.py) on rows {0, B-1} -- rows are independent, so two rows
     of a 1100-step sequence cost seconds on the CPU: every step's output (reported separately for steps >= 1024) and
     the input gradient at every step; plus one large-batch case (B = 1024: 16 row tiles, one launch per layer).
A product-vs-product comparison alone would pass if both paths shared a defect; (2) is what pins the long-sequence path.
      python tests/diag/long_sequence.py [T]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cerebralsignalnetworks_amd.lstm_model import HipLSTM      # noqa: E402
from oracle import lstm as oracle_lstm                           # noqa: E402

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

# ---- (2) against the float64 oracle on two rows ------------------------------------------------------------------------
def rel(a, b):
    return float(np.linalg.norm(a - b) / max(1e-30, np.linalg.norm(b)))


# (float32 cases, round 4: the weight-stationary float32 kernels of lstm_f32_persist.hip -- fragment-major hand-off slots of
#  up to 4.6 GB per layer at T = 1100, and 16 row tiles walked in blocks of 5 / 4 at B = 1024)
for (B, C, H, L, Tn, cdt) in ((256, 16, 1024, 2, T, torch.bfloat16), (256, 128, 1024, 2, T, torch.bfloat16),
                              (256, 128, 768, 2, T, torch.bfloat16), (1024, 128, 768, 2, 500, torch.bfloat16),
                              (256, 128, 1024, 2, T, torch.float32), (1024, 128, 768, 2, 500, torch.float32)):
    torch.manual_seed(3)
    m = HipLSTM(C, H, L, compute_dtype=cdt).to(dev)
    x = torch.randn(B, Tn, C, device=dev, requires_grad=True)
    w = torch.randn(B, Tn, H, device=dev) / float(np.sqrt(Tn))
    y_all, _ = m(x, want_all=True)
    (y_all * w).sum().backward()
    torch.cuda.synchronize()
    st = [pl.status() for pl in m.all_plans()]
    assert not any(st), st
    path = [pl.path() for pl in m.all_plans() if pl.training][0]
    rows = [0, B - 1]
    params = {n: q.detach().double().cpu().numpy() for n, q in m.named_parameters()}
    y_ref, saved = oracle_lstm.lstm_forward(x.detach()[rows].double().cpu().numpy(), params, L, return_saved=True)
    dx_ref, _ = oracle_lstm.lstm_backward(w[rows].double().cpu().numpy(), params, saved, L)
    y_got = y_all.detach()[rows].double().cpu().numpy()
    dx_got = x.grad[rows].double().cpu().numpy()
    per_t = np.array([rel(y_got[:, t], y_ref[:, t]) for t in range(Tn)])
    late = per_t[1024:].max() if Tn > 1024 else float("nan")
    dx_t = np.array([rel(dx_got[:, t], dx_ref[:, t]) for t in range(Tn)])
    # bf16 operands (8 significant bits) through up to 1100 recurrent steps: the per-step outputs stay within a few 1e-3 of
    # the float64 oracle, the input gradient within a few 1e-2; a wrapped offset gives 1.0
    # (float32: 1e-4 / 1e-3 -- the recurrence amplifies float32 rounding over a thousand steps, a wrapped offset gives 1.0)
    f32 = cdt == torch.float32
    ok = (per_t.max() < (1e-4 if f32 else 2e-2) and dx_t.max() < (1e-3 if f32 else 6e-2) and rel(dx_got, dx_ref) < (1e-4 if f32 else 3e-2)
          and (Tn <= 1024 or late < (1e-4 if f32 else 2e-2)))
    bad += 0 if ok else 1
    path = ("f32 " if f32 else "") + str(path) + (" " + "/".join(m.all_plans()[0].kernel_names()) if f32 else "")
    print(f"{'ok  ' if ok else 'FAIL'} oracle rows {rows} B{B} T{Tn} C{C} H{H} L{L} (path {path}): y worst step {int(per_t.argmax())} "
          f"rel {per_t.max():.2e}, steps >= 1024: {late:.2e}; dx worst step {int(dx_t.argmax())} rel {dx_t.max():.2e}, whole {rel(dx_got, dx_ref):.2e}",
          flush=True)
    del m, x, w, y_all
    torch.cuda.empty_cache()
print(f"{bad} failing case(s)")
sys.exit(1 if bad else 0)
