"""GPU: the forward / scheduling variants that LOSE to the defaults and therefore live outside
the product library: `make -C cerebralsignalnetworks_amd/csrc experiments` builds lib/libcsn_hip_experiments.so, and
this module runs only against it:

    CSN_LIB_PATH=$PWD/cerebralsignalnetworks_amd/lib/libcsn_hip_experiments.so python -m pytest tests/test_gpu_experiments.py -m gpu

(the driver's `pytest -m gpu` skips it: the product library ignores these switches).  Each variant must give the bits of
the kernel it competes with, or agree to bf16 rounding where it sums in a different order."""
import os

import numpy as np
import pytest
import torch

from oracle import lstm

pytestmark = pytest.mark.gpu

if not os.environ.get("CSN_LIB_PATH", "").endswith("libcsn_hip_experiments.so"):
    pytest.skip("experiments library not selected (CSN_LIB_PATH)", allow_module_level=True)

from test_gpu_parity import _assert_same_bits, _rel, _run_lstm      # noqa: E402


@pytest.mark.parametrize("B,T,C,H,L,chunk", [(70, 75, 24, 128, 2, "32"), (130, 33, 16, 512, 2, "32"),
                                             (64, 40, 128, 768, 2, "8"), (256, 24, 128, 768, 2, "8")])
def test_losing_variants_match_the_defaults(cuda, B, T, C, H, L, chunk):
    rng = np.random.default_rng(B * T + H)
    p = lstm.init_params(C, H, L, 8, None, seed=5)
    x = rng.standard_normal((B, T, C)).astype(np.float32)
    dy_all = (rng.standard_normal((B, T, H)) * 0.1).astype(np.float32)
    dy_last = rng.standard_normal((B, H)).astype(np.float32)
    run = lambda **env: _run_lstm(p, x, dy_all, dy_last, C, H, L, torch.bfloat16, cuda, {"CSN_LSTM_CHUNK": chunk, **env})
    fast = run()
    # the upper layers' weight-gradient GEMMs beside the last backward launches on the plan's low-priority stream
    wg = run(CSN_WGRAD_OVERLAP="1")
    for k in fast:
        _assert_same_bits(fast[k], wg[k], f"wgrad_beside: {k}")
    ns = run(CSN_FWD_NSPLIT="1")
    ns_beside = run(CSN_FWD_NSPLIT="1", CSN_BESIDE_FWD="1")
    ns_halves = run(CSN_FWD_NSPLIT="1", CSN_FWD_HALVES="1")   # (H = 768: the K2 x N2 body pipelined over 32-row halves)
    for k in ns:
        _assert_same_bits(ns[k], ns_beside[k], f"ns beside: {k}")
        _assert_same_bits(ns[k], ns_halves[k], f"ns halves: {k}")
        assert _rel(ns[k], fast[k]) < 1e-2, (k, _rel(ns[k], fast[k]))
    ns_bf16x = run(CSN_FWD_NSPLIT="1", CSN_BESIDE_FWD="1", CSN_XPROJ_BF16="1")
    for k in ns:
        assert _rel(ns_bf16x[k], fast[k]) < 2e-2, (k, _rel(ns_bf16x[k], fast[k]))
    if H == 768:
        # the wave-specialised forward (experiments/lstm_fwd_ws.hip): four 16-row chains per tile, MFMA waves + gate waves
        ws = run(CSN_FWD_WS="1")
        ws_streams = run(CSN_FWD_WS="1", CSN_PERSIST_STREAMS="1")
        ws_anyplace = run(CSN_FWD_WS="1", CSN_NO_XCD_LOCAL="1")
        _assert_same_bits(ws["y_all"], ws_streams["y_all"], "ws streams: y_all")
        for k in ws:
            _assert_same_bits(ws[k], ws_anyplace[k], f"ws anyplace: {k}")
            assert _rel(ws[k], fast[k]) < 1e-2, (k, _rel(ws[k], fast[k]))
