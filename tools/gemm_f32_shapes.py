"""Times the float32 GEMM calls of one exact-float32 training step at cfg2 shapes, one by one (csn_gemm_nt / csn_gemm_tn, f32)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cerebralsignalnetworks_amd import cabi
dev = torch.device("cuda:0")
TB, G, H, C = 128000, 3072, 768, 128


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, kind, (M, N, K) in (("projection l0", "nt", (TB, G, C)), ("projection l1", "nt", (TB, G, H)), ("dx l1", "nt", (TB, H, G)),
                              ("dW_hh", "tn", (G, H, TB)), ("dW_ih l1", "tn", (G, H, TB)), ("dW_ih l0", "tn", (G, C, TB))):
    if kind == "nt":
        a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev); out = torch.empty(M, N, device=dev)
        ms = t(lambda: cabi.gemm_nt(a, b, None, out=out))
    else:
        a = torch.randn(K, M, device=dev); b = torch.randn(K, N, device=dev)
        ms = t(lambda: cabi.gemm_tn(a, b))
    print(f"{name:14s} {kind} M{M} N{N} K{K}: {ms:7.3f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s", flush=True)
    del a, b
    torch.cuda.empty_cache()
