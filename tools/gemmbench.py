"""GEMM micro-benchmark at the shapes the LSTM path uses (GPU box only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cerebralsignalnetworks_amd import cabi  # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


def main():
    dev = torch.device("cuda:0")
    bf = torch.bfloat16
    shapes_nt = [("xproj chunk", 8192, 3072, 768), ("xproj full", 128000, 3072, 768), ("xproj1 full", 128000, 3072, 128),
                 ("dx chunk", 8192, 768, 3072)]
    for name, M, N, K in shapes_nt:
        a = torch.randn(M, K, device=dev).to(bf)
        b = torch.randn(N, K, device=dev).to(bf)
        bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev)
        t = timeit(lambda: cabi.gemm_nt(a, b, bias, out=out))
        print(f"NT {name:12s} M={M} N={N} K={K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)
    lib = cabi.load()
    for name, M, N, K in [("dW_hh", 3072, 768, 128000), ("dW_ih1", 3072, 128, 128000)]:
        a = torch.randn(K, M, device=dev).to(bf)
        b = torch.randn(K, N, device=dev).to(bf)
        scratch = torch.empty(lib.csn_gemm_tn_scratch_bytes(M, N, K), dtype=torch.uint8, device=dev)
        c = torch.empty(M, N, device=dev)

        def fn():
            cabi._check(lib.csn_gemm_tn(cabi._ptr(a), cabi._ptr(b), cabi._ptr(c), M, N, K, cabi.CSN_BF16,
                                        cabi._ptr(scratch), cabi._stream()))
        t = timeit(fn)
        print(f"TN {name:12s} M={M} N={N} K={K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)
    # reference point: torch (hipBLASLt) on the same shapes
    for name, M, N, K in [("xproj chunk", 8192, 3072, 768), ("xproj full", 128000, 3072, 768)]:
        a = torch.randn(M, K, device=dev).to(bf)
        b = torch.randn(N, K, device=dev).to(bf)
        t = timeit(lambda: torch.matmul(a, b.t()))
        print(f"torch NT {name:12s}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s (bf16 out)", flush=True)
    a = torch.randn(128000, 3072, device=dev).to(bf)
    b = torch.randn(128000, 768, device=dev).to(bf)
    t = timeit(lambda: torch.matmul(a.t(), b))
    print(f"torch TN dW_hh: {t*1e6:8.1f} us  {2*3072*768*128000/t/1e12:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
