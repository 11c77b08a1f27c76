"""Randomised LSTM plans (both dtypes, 1-4 layers, odd batch / length / channel counts, hidden sizes on and off the
weight-stationary list, chunk lengths, with and without per-step output gradients and dx) against the float64 oracle.
A bug hunt: one line per case, exit code 1 on a failure.      python tests/diag/fuzz_lstm.py [cases] [seed]"""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import lstm      # noqa: E402  (tools/ and tests/ may use the oracle; the product never does)
from cerebralsignalnetworks_amd import Model      # noqa: E402

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
bad = 0


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(1e-12, np.linalg.norm(b)))


for i in range(n_cases):
    H = int(rng.choice([128, 256, 384, 512, 768, 1024, 96, 160, 224, 64, 32]))
    big = H >= 768
    B = int(rng.integers(1, 40 if big else 140))
    T = int(rng.integers(1, 40 if big else 75))
    C = int(rng.choice([1, 3, 8, 16, 24, 32, 64, 100, 128, 130]))
    L = int(rng.integers(1, 5))
    chunk = str(int(rng.choice([1, 2, 3, 4, 7, 8, 16, 32, 64])))
    want_all, want_dx = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    dt = torch.bfloat16 if rng.integers(0, 3) else torch.float32
    name = f"B{B} T{T} C{C} H{H} L{L} chunk{chunk} all{int(want_all)} dx{int(want_dx)} {str(dt)[6:]}"
    old = os.environ.get("CSN_LSTM_CHUNK")
    os.environ["CSN_LSTM_CHUNK"] = chunk
    try:
        p = lstm.init_params(C, H, L, 8, None, seed=int(rng.integers(1, 1000)))
        lp = {k[len("lstm."):]: v for k, v in p.items() if k.startswith("lstm.")}
        x = rng.standard_normal((B, T, C)).astype(np.float32)
        dy_all = (rng.standard_normal((B, T, H)) * 0.1).astype(np.float32) if want_all else np.zeros((B, T, H), np.float32)
        dy_last = rng.standard_normal((B, H)).astype(np.float32)
        y, saved = lstm.lstm_forward(x, lp, L, return_saved=True)
        dy = dy_all.astype(np.float64).copy()
        dy[:, -1] += dy_last
        dx_ref, g_ref = lstm.lstm_backward(dy, lp, saved, L)
        m = Model(input_size=C, lstm_size=H, lstm_layers=L, output_size=8, include_top=False, compute_dtype=dt)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
        m = m.to(dev)
        xt = torch.from_numpy(x).to(dev).requires_grad_(want_dx)
        if want_all:
            y_all, y_last = m.lstm(xt, want_all=True)
            loss = (y_all * torch.from_numpy(dy_all).to(dev)).sum() + (y_last * torch.from_numpy(dy_last).to(dev)).sum()
        else:
            y_last = m.lstm(xt)
            loss = (y_last * torch.from_numpy(dy_last).to(dev)).sum()
        loss.backward()
        torch.cuda.synchronize()
        st = [pl.status() for pl in m.lstm.all_plans()]
        errs = {"y_last": rel(y_last.detach().cpu().numpy(), y[:, -1])}
        if want_all:
            errs["y_all"] = rel(y_all.detach().cpu().numpy(), y)
        if want_dx:
            errs["dx"] = rel(xt.grad.cpu().numpy(), dx_ref)
        for n, q in m.lstm.named_parameters():
            errs[n] = rel(q.grad.cpu().numpy(), g_ref[n])
        tol = 2e-2 if dt == torch.bfloat16 else 2e-5
        worst = max(errs, key=errs.get)
        ok = all(s == 0 for s in st) and all(np.isfinite(v) and v < tol for v in errs.values())
        print(f"{'ok  ' if ok else 'FAIL'} {name}: path {[pl.path() for pl in m.lstm.all_plans()]} worst {worst} {errs[worst]:.2e} status {st}", flush=True)
        bad += 0 if ok else 1
    except Exception as e:      # noqa: BLE001
        print(f"RAISE {name}: {type(e).__name__}: {str(e)[:300]}", flush=True)
        traceback.print_exc(limit=3)
        bad += 1
    finally:
        if old is None:
            os.environ.pop("CSN_LSTM_CHUNK", None)
        else:
            os.environ["CSN_LSTM_CHUNK"] = old
print(f"{bad} failing case(s)")
sys.exit(1 if bad else 0)
