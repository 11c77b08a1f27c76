"""Diagnostic: the wave-specialised forward against the default one, and against itself (determinism)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as tp
from oracle import lstm
cuda = torch.device("cuda:0")
B, T, C, H, L = 64, 12, 128, 768, 2
rng = np.random.default_rng(5)
p = lstm.init_params(C, H, L, 8, None, seed=5)
x = rng.standard_normal((B, T, C)).astype(np.float32)
dy_all = (rng.standard_normal((B, T, H)) * 0.1).astype(np.float32)
dy_last = rng.standard_normal((B, H)).astype(np.float32)
run = lambda **env: tp._run_lstm(p, x, dy_all, dy_last, C, H, L, torch.bfloat16, cuda, {"CSN_LSTM_CHUNK": "4", **env})["y_all"]
ref = run()
a = run(CSN_FWD_WS="1"); b = run(CSN_FWD_WS="1")
for name, u, v in (("ws vs default", a, ref), ("ws vs ws", a, b)):
    d = np.abs(u - v)
    print(name, "max", d.max(), "rel", np.linalg.norm(u - v) / np.linalg.norm(v))
    per_t = d.reshape(B, T, H).max(axis=(0, 2))
    print("  per t:", np.array2string(per_t, precision=4))
    per_unit_slice = d.reshape(B, T, H // 24, 24).max(axis=(0, 1, 2))
    print("  per unit-in-slice:", np.array2string(per_unit_slice, precision=4))
    per_row = d.reshape(B, T, H).max(axis=(1, 2))
    print("  per row (first 32):", np.array2string(per_row[:32], precision=3))
