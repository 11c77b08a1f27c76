"""Diagnostic: repeat one LSTM forward/backward configuration under several env settings and count bit mismatches
against the per-diagonal launches (GPU box only)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import lstm
import test_gpu_parity as tg
cuda = torch.device("cuda:0")
B, T, C, H, L, chunk = [int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (64, 20, 32, 256, 1, 32))]
rng = np.random.default_rng(B * T + H)
p = lstm.init_params(C, H, L, 8, None, seed=5)
x = rng.standard_normal((B, T, C)).astype(np.float32)
dy_all = (rng.standard_normal((B, T, H)) * 0.1).astype(np.float32)
dy_last = rng.standard_normal((B, H)).astype(np.float32)
run = lambda **env: tg._run_lstm(p, x, dy_all, dy_last, C, H, L, torch.bfloat16, cuda, {"CSN_LSTM_CHUNK": str(chunk), **env})
exact = dict(CSN_NO_ROTATE="1", CSN_NO_FUSE_X="1")
if os.environ.get("PROBE_FIRST"):
    # the very first forward of the process is the default (rotated, fused) one, as in the test
    fast = run()
    norot = run(**exact)
    diag = run(CSN_NO_PERSIST="1", **exact)
    streams = run(CSN_PERSIST_STREAMS="1", **exact)
    anyp = run(CSN_NO_XCD_LOCAL="1", **exact)
    norot2 = run(**exact)
    for nm, a_, b_ in (("norot/diag", norot, diag), ("streams/diag", streams, diag), ("anyplace/diag", anyp, diag), ("norot2/diag", norot2, diag)):
        bad = {k: int((a_[k] != b_[k]).sum()) for k in a_ if (a_[k] != b_[k]).any()}
        print("first", nm, bad, flush=True)
    sys.exit(0)
ref = run(CSN_NO_PERSIST="1", **exact)
for name, env in (("local", {}), ("anyplace", {"CSN_NO_XCD_LOCAL": "1"}), ("streams", {"CSN_PERSIST_STREAMS": "1"}),
                  ("diag again", {"CSN_NO_PERSIST": "1"})):
    for rep in range(4):
        out = run(**env, **exact)
        bad = {k: int((out[k] != ref[k]).sum()) for k in out}
        ts = sorted(set(np.argwhere(out["y_all"] != ref["y_all"])[:, 1].tolist()))[:8]
        print(name, rep, {k: v for k, v in bad.items() if v}, "first bad t:", ts, flush=True)
# same order as tests/test_gpu_parity.py::test_fast_path_matches_oracle_and_v1
if os.environ.get("PROBE_ORDER"):
    for rep in range(3):
        fast = run()
        slow = tg._run_lstm(p, x, dy_all, dy_last, C, H, L, torch.bfloat16, cuda, {"CSN_CELL_V1": "1"})
        serial = run(CSN_NO_SIDE_STREAM="1")
        anyp = run(CSN_NO_XCD_LOCAL="1")
        norot = run(**exact)
        diag = run(CSN_NO_PERSIST="1", **exact)
        for nm, a_, b_ in (("fast/serial", fast, serial), ("fast/anyplace", fast, anyp), ("norot/diag", norot, diag), ("norot/ref", norot, ref), ("diag/ref", diag, ref)):
            bad = {k: int((a_[k] != b_[k]).sum()) for k in a_}
            ts = sorted(set(np.argwhere(a_["y_all"] != b_["y_all"])[:, 1].tolist()))[:8]
            print("order", rep, nm, {k: v for k, v in bad.items() if v}, "bad t:", ts, flush=True)
# stress: a "polluting" run with a different summation order before every checked run (same workspace addresses)
if os.environ.get("PROBE_STRESS"):
    n = int(os.environ["PROBE_STRESS"])
    counts = {}
    for rep in range(n):
        for name, env in (("local", {}), ("anyplace", {"CSN_NO_XCD_LOCAL": "1"}), ("streams", {"CSN_PERSIST_STREAMS": "1"}),
                          ("diag", {"CSN_NO_PERSIST": "1"})):
            run()                                   # pollute: rotated + fused
            out = run(**env, **exact)
            bad = {k: int((out[k] != ref[k]).sum()) for k in out if (out[k] != ref[k]).any()}
            if bad:
                ts = sorted(set(np.argwhere(out["y_all"] != ref["y_all"])[:, 1].tolist()))[:6]
                print("stress", rep, name, bad, "bad t:", ts, flush=True)
                counts[name] = counts.get(name, 0) + 1
    print("stress summary", n, counts, flush=True)
