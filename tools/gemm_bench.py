"""Diagnostic (GPU box): the two GEMM shapes of the cfg2 training step, timed warm with HIP events.
  weight gradient  C[3072, 768]  = A[128000, 3072]^T  B[128000, 768]   (TN, split-K slabs + reduction)
  input projection C[8192, 3072] = A[8192, 768]       Bt[3072, 768]^T  (NT, f32 out)
"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cerebralsignalnetworks_amd import cabi  # noqa: E402
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
def rnd(*s): return torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
def bench(name, fn, flops, reps=20):
    t_end = time.time() + 1.5
    while time.time() < t_end:          # warm clocks
        for _ in range(10): fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    print(f"{name}: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
K, M, N = 128000, 3072, 768
a, b = rnd(K, M), rnd(K, N)
bench("tn 3072x768x128000", lambda: cabi.gemm_tn(a, b), 2.0 * K * M * N)
b128 = rnd(K, 128)
bench("tn 3072x128x128000", lambda: cabi.gemm_tn(a, b128), 2.0 * K * M * 128)
a2, b2 = rnd(8192, 768), rnd(3072, 768)
bias = torch.randn(3072, device=dev)
out = torch.empty(8192, 3072, device=dev)
bench("nt 8192x3072x768 f32", lambda: cabi.gemm_nt(a2, b2, bias, out=out), 2.0 * 8192 * 3072 * 768)
