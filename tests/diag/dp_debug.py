"""Diagnostic: which configuration of the fast path times out / differs with the data-poll hand-off."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as tp
from oracle import lstm
cuda = torch.device("cuda:0")
B, T, C, H, L, chunk = 256, 24, 128, 768, 2, "8"
p = lstm.init_params(C, H, L, 8, None, seed=5)
rng = np.random.default_rng(B + T)
x = rng.standard_normal((B, T, C)).astype(np.float32)
dy_all = (rng.standard_normal((B, T, H)) * 0.1).astype(np.float32)
dy_last = rng.standard_normal((B, H)).astype(np.float32)
def run(**env):
    try:
        return tp._run_lstm(p, x, dy_all, dy_last, C, H, L, torch.bfloat16, cuda, {"CSN_LSTM_CHUNK": chunk, **env})["y_all"]
    except AssertionError as e:
        return str(e)[:80]
ref = run(CSN_FWD_FLAGS="1")
for env in ({}, {"CSN_NO_SIDE_STREAM": "1"}, {"CSN_NO_XCD_LOCAL": "1"}, {"CSN_NO_BESIDE": "1"}, {"CSN_NO_ROTATE": "1", "CSN_NO_FUSE_X": "1", "CSN_FWD_KSPLIT": "1"},
            {"CSN_PERSIST_STREAMS": "1", "CSN_NO_ROTATE": "1", "CSN_NO_FUSE_X": "1", "CSN_FWD_KSPLIT": "1"}, {"CSN_NO_PERSIST_BWD": "1"}, {"CSN_NO_FUSE_X": "1"}):
    r = run(**env)
    print(env, "->", r if isinstance(r, str) else ("same bits as flags" if np.array_equal(r, ref) else "max diff %.3g" % np.abs(r - ref).max()), flush=True)
