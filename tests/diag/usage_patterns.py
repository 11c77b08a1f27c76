"""Usage patterns around the LSTM module that the shape fuzz does not reach, against the float64 oracle: inference plans
(no_grad), two forwards awaiting their backward, a non-default stream, changing batch sizes back to back, big batches that
fall off the weight-stationary path, eval after train.      python tests/diag/usage_patterns.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lstm      # noqa: E402
from cerebralsignalnetworks_amd import Model      # noqa: E402

dev = torch.device("cuda:0")
bad = 0


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(1e-12, np.linalg.norm(b)))


def report(name, err, tol):
    global bad
    ok = np.isfinite(err) and err < tol
    bad += 0 if ok else 1
    print(f"{'ok  ' if ok else 'FAIL'} {name}: {err:.2e} (tol {tol:g})", flush=True)


def make(C, H, L, dt, seed=3):
    p = lstm.init_params(C, H, L, 8, None, seed=seed)
    m = Model(input_size=C, lstm_size=H, lstm_layers=L, output_size=8, include_top=False, compute_dtype=dt)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    return p, {k[len("lstm."):]: v for k, v in p.items() if k.startswith("lstm.")}, m.to(dev)


rng = np.random.default_rng(5)
for dt, tol in ((torch.bfloat16, 2e-2), (torch.float32, 2e-5)):
    tag = str(dt)[6:]
    for (B, T, C, H, L) in ((64, 40, 128, 768, 2), (37, 9, 16, 128, 3), (256, 33, 128, 768, 2), (512, 12, 32, 256, 2), (1024, 6, 16, 128, 2)):
        p, lp, m = make(C, H, L, dt)
        x = rng.standard_normal((B, T, C)).astype(np.float32)
        y = lstm.lstm_forward(x, lp, L)
        y = y[0] if isinstance(y, tuple) else y
        # 1. inference plan
        with torch.no_grad():
            out = m.lstm(torch.from_numpy(x).to(dev))
        report(f"{tag} inference B{B} T{T} H{H} L{L}", rel(out.cpu().numpy(), y[:, -1]), tol)
        # 2. on a side stream
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            out2 = m.lstm(torch.from_numpy(x).to(dev))
        s.synchronize()
        report(f"{tag} side stream B{B} T{T} H{H} L{L}", rel(out2.cpu().numpy(), y[:, -1]), tol)
        # 3. two forwards awaiting one backward, second with another batch size
        B2 = max(1, B // 2 + 1)
        x2 = rng.standard_normal((B2, T, C)).astype(np.float32)
        ya, sa = lstm.lstm_forward(x, lp, L, return_saved=True)
        yb, sb = lstm.lstm_forward(x2, lp, L, return_saved=True)
        wa = rng.standard_normal((B, H)).astype(np.float32)
        wb = rng.standard_normal((B2, H)).astype(np.float32)
        dya = np.zeros((B, T, H)); dya[:, -1] = wa
        dyb = np.zeros((B2, T, H)); dyb[:, -1] = wb
        _, ga = lstm.lstm_backward(dya, lp, sa, L)
        _, gb = lstm.lstm_backward(dyb, lp, sb, L)
        m.zero_grad()
        o1 = m.lstm(torch.from_numpy(x).to(dev))
        o2 = m.lstm(torch.from_numpy(x2).to(dev))
        ((o1 * torch.from_numpy(wa).to(dev)).sum() + (o2 * torch.from_numpy(wb).to(dev)).sum()).backward()
        torch.cuda.synchronize()
        worst = max(rel(q.grad.cpu().numpy(), ga[n] + gb[n]) for n, q in m.lstm.named_parameters())
        report(f"{tag} two forwards, one backward B{B}+{B2} T{T} H{H} L{L}", worst, tol)
        st = [pl.status() for pl in m.lstm.all_plans()]
        if any(st):
            bad += 1
            print("FAIL status", st)
        # 4. same input twice through the same plan: identical bits
        with torch.no_grad():
            r1 = m.lstm(torch.from_numpy(x).to(dev)).clone()
            r2 = m.lstm(torch.from_numpy(x).to(dev)).clone()
        report(f"{tag} repeat bit-equal B{B} T{T} H{H} L{L}", float((r1 != r2).sum().item()), 0.5)
print(f"{bad} failing case(s)")
sys.exit(1 if bad else 0)
