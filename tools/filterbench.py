"""Filter kernel micro-benchmark (GPU box only): GB/s of algorithmic traffic (read f32 + write out)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cerebralsignalnetworks_amd import cabi, EEGFilters  # noqa: E402

dev = torch.device("cuda:0")
B, C, T = 256, 128, 500
x = torch.randn(4, B, C, T, device=dev)
sos = EEGFilters(1000, 3).sos
NREP = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for out_dtype, osz in ((torch.float32, 4), (torch.bfloat16, 2)):
    for _ in range(3):
        cabi.eeg_bandpass_znorm(x[0], sos, out_dtype=out_dtype)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = NREP
    e0.record()
    for i in range(n):
        cabi.eeg_bandpass_znorm(x[i % 4], sos, out_dtype=out_dtype)
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / n
    byts = B * C * T * (4 + osz)
    print(f"filter {out_dtype}: {t*1e6:.1f} us per call, {byts/t/1e9:.0f} GB/s algorithmic", flush=True)
for tm in (False, True):
    for _ in range(3):
        cabi.eeg_bandpass_znorm(x[0], sos, time_major=tm)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(NREP):
        cabi.eeg_bandpass_znorm(x[i % 4], sos, time_major=tm)
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / NREP
    print(f"filter f32 time_major={tm}: {t*1e6:.1f} us per call, {B*C*T*8/t/1e9:.0f} GB/s algorithmic", flush=True)
