"""Degenerate arguments: more layers than the weight-stationary path takes, empty batch / sequence, zero sizes -- the library
must either compute the right thing or refuse with an error, never fault."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lstm      # noqa: E402
from cerebralsignalnetworks_amd import cabi, Model      # noqa: E402

dev = torch.device("cuda:0")
bad = 0
rng = np.random.default_rng(2)
for (B, T, C, H, L) in ((64, 20, 16, 128, 5), (32, 12, 8, 256, 6), (64, 40, 128, 768, 5)):
    p = lstm.init_params(C, H, L, 8, None, seed=1)
    lp = {k[len("lstm."):]: v for k, v in p.items() if k.startswith("lstm.")}
    x = rng.standard_normal((B, T, C)).astype(np.float32)
    y = lstm.lstm_forward(x, lp, L)
    y = y[0] if isinstance(y, tuple) else y
    m = Model(input_size=C, lstm_size=H, lstm_layers=L, output_size=8, include_top=False)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    m = m.to(dev)
    xt = torch.from_numpy(x).to(dev)
    out = m.lstm(xt)
    out.sum().backward()
    torch.cuda.synchronize()
    err = float(np.linalg.norm(out.detach().cpu().numpy() - y[:, -1]) / np.linalg.norm(y[:, -1]))
    fin = all(bool(torch.isfinite(q.grad).all()) for q in m.lstm.parameters())
    ok = err < 2e-2 and fin
    bad += 0 if ok else 1
    print(f"{'ok  ' if ok else 'FAIL'} L={L} B{B} T{T} H{H}: y err {err:.2e}, grads finite {fin}, paths {[pl.path() for pl in m.lstm.all_plans()]}")
for (B, T, C, H, L) in ((0, 10, 8, 128, 1), (4, 0, 8, 128, 1), (4, 10, 0, 128, 1), (4, 10, 8, 0, 1), (4, 10, 8, 128, 0)):
    try:
        plan = cabi.LstmPlan(B, T, C, H, L, torch.bfloat16, dev)
        print(f"accepted B{B} T{T} C{C} H{H} L{L} (plan path {plan.path()})")
        bad += 1
    except Exception as e:      # noqa: BLE001
        print(f"refused B{B} T{T} C{C} H{H} L{L}: {type(e).__name__}: {str(e)[:100]}")
for fn, name in ((lambda: cabi.gemm_nt(torch.empty(0, 8, device=dev), torch.empty(4, 8, device=dev)), "gemm_nt M=0"),
                 (lambda: cabi.cosine_loss(torch.empty(0, 8, device=dev), torch.empty(0, 8, device=dev)), "cosine B=0"),
                 (lambda: cabi.l2_topk(torch.empty(0, 8, device=dev), torch.randn(3, 8, device=dev), 1), "topk Ng=0"),
                 (lambda: cabi.eeg_bandpass_znorm(torch.empty(0, 8, 100, device=dev), np.ones((1, 6))), "bandpass B=0")):
    try:
        fn()
        torch.cuda.synchronize()
        print(f"accepted {name} (no fault)")
    except Exception as e:      # noqa: BLE001
        print(f"refused {name}: {type(e).__name__}: {str(e)[:100]}")
print(f"{bad} failing case(s)")
sys.exit(1 if bad else 0)
