"""CPU: the C-ABI library is built, loads, and exports every symbol include/csn_hip.h declares
(no compute calls without a GPU); host-side argument checks reject bad shapes before any launch."""
import ctypes
import os
import re

import pytest

import __graft_entry__ as graft
from cerebralsignalnetworks_amd import cabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(cabi.LIB_PATH):
        graft.build()
    return cabi.load()


def test_header_symbols_are_exported_and_bound(lib):
    text = open(os.path.join(ROOT, "include", "csn_hip.h")).read()
    declared = set(re.findall(r"\b(csn_[a-z0-9_]+)\s*\(", text))
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in csn_hip.h but not exported"
    assert declared == set(cabi.SIGNATURES), declared ^ set(cabi.SIGNATURES)
    assert lib.csn_abi_version() == cabi.ABI_VERSION
    assert lib.csn_target_arch() == b"gfx950"


def test_argument_validation_happens_on_the_host(lib):
    d = cabi.LstmDesc(4, 8, 16, 48, 2, cabi.CSN_F32)        # H not a multiple of 32
    assert lib.csn_lstm_workspace_bytes(ctypes.byref(d), 1) == 0
    assert b"multiple of 32" in lib.csn_last_error()
    d = cabi.LstmDesc(256, 500, 128, 768, 2, cabi.CSN_BF16)  # cfg2
    nbytes = lib.csn_lstm_workspace_bytes(ctypes.byref(d), 1)
    assert 4e9 < nbytes < 20e9
    rc = lib.csn_eeg_bandpass_znorm(None, 1, 1, 8, None, 0, 0, None, 0, 0, None)
    assert rc == 1 and b"null" in lib.csn_last_error()
    rc = lib.csn_l2_topk(None, None, 10, 10, 4, 5, None, None, None, None)
    assert rc == 1


def test_product_refuses_cpu_tensors():
    import torch
    from cerebralsignalnetworks_amd import Model
    m = Model(input_size=16, lstm_size=32, lstm_layers=1, output_size=8, include_top=False)
    with pytest.raises(cabi.CsnError):
        m(torch.zeros(2, 5, 16))
    with pytest.raises(cabi.CsnError):
        cabi.eeg_bandpass_znorm(torch.zeros(1, 2, 8), None)


def test_state_dict_round_trips_with_stock_nn_lstm():
    import torch
    from cerebralsignalnetworks_amd import Model
    m = Model(input_size=16, lstm_size=32, lstm_layers=2, output_size=8, include_top=True)
    ref = torch.nn.LSTM(16, 32, num_layers=2, batch_first=True)
    sd = m.state_dict()
    assert {k[len("lstm."):] for k in sd if k.startswith("lstm.")} == set(ref.state_dict())
    ref.load_state_dict({k[len("lstm."):]: v for k, v in sd.items() if k.startswith("lstm.")})
    assert {"fc.weight", "fc.bias", "class_pred.weight", "class_pred.bias"} <= set(sd)
    # DINO-style checkpoint keys with a "backbone." prefix load with strict=False (Eval.py:310-313)
    ck = {"backbone." + k: v for k, v in sd.items()}
    missing = m.load_state_dict({k.replace("backbone.", ""): v for k, v in ck.items()}, strict=False)
    assert not missing.missing_keys
