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


@pytest.mark.parametrize("B,T,C,H,L,chunk", [(64, 130, 128, 256, 2, "32"),     # last chunk of 2 steps
                                             (128, 70, 128, 768, 2, "8"),      # the benchmark width, two M-tiles
                                             (64, 128, 128, 128, 1, "32"),     # one layer: no input-gradient GEMM at all
                                             (64, 129, 128, 512, 3, "16")])
def test_single_copy_of_gate_gradients_gives_the_same_bits(cuda, B, T, C, H, L, chunk):
    """CSN_BWD_SINGLE_COPY: when no dx of layer 0 is wanted (B % 64 == 0, every weight gradient on the
    256 x 256 kernel) the backward recurrence writes its gate gradients ONCE -- per-step fragment-major hand-off slabs,
    read in place by the weight-gradient GEMMs and by the input-gradient GEMMs inside the launches -- instead of a second,
    row-major copy.  Same values through the same MFMA sequences: every gradient must be bit-identical to the two-copy
    form, in every hand-off flavour (hinted, unhinted = every step through the re-read path, placement-independent)."""
    rng = np.random.default_rng(B * T + H + 1)
    p = lstm.init_params(C, H, L, 8, None, seed=6)
    x = rng.standard_normal((B, T, C)).astype(np.float32)
    dy_all = (rng.standard_normal((B, T, H)) * 0.1).astype(np.float32)
    dy_last = rng.standard_normal((B, H)).astype(np.float32)

    def run(want_dx=False, **env):
        info = {}
        out = _run_lstm(p, x, dy_all, dy_last, C, H, L, torch.bfloat16, cuda, {"CSN_LSTM_CHUNK": chunk, **env},
                        want_dx=want_dx, info=info)
        return out, info["dgates_copies"]

    with_dx, c2 = run(want_dx=True, CSN_BWD_SINGLE_COPY="1")      # dx wanted: its GEMM reads the row-major copy
    assert c2 == [2], c2
    two, c2 = run()
    assert c2 == [2], c2
    for env in ({}, {"CSN_DPOLL_NO_HINT": "1"}, {"CSN_NO_XCD_LOCAL": "1"}, {"CSN_DPOLL_NO_HINT": "1", "CSN_NO_XCD_LOCAL": "1"}):
        one, c1 = run(CSN_BWD_SINGLE_COPY="1", **env)
        assert c1 == [1], (env, c1)       # the form under test did run
        for k in one:
            _assert_same_bits(one[k], two[k], f"single copy {env} vs two copies: {k}")
            _assert_same_bits(one[k], with_dx[k], f"single copy {env} vs the dx run: {k}")
    # the flag hand-off has no per-step slabs to read in place: it keeps both copies
    flags, cf = run(CSN_BWD_SINGLE_COPY="1", CSN_BWD_FLAGS="1")
    assert cf == [2], cf
    for k in flags:
        _assert_same_bits(flags[k], two[k], f"flags: {k}")
