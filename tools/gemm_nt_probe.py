"""Diagnostic: NT GEMM at the chunk shapes, f32 vs bf16 output, per kernel variant (GPU box only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cerebralsignalnetworks_amd import cabi
from tools.gemmbench import timeit
dev = torch.device("cuda:0"); bf = torch.bfloat16
for name, M, N, K in [("xproj chunk", 8192, 3072, 768), ("dx chunk", 8192, 768, 3072), ("big", 32768, 3072, 3072)]:
    a = torch.randn(M, K, device=dev).to(bf); b = torch.randn(N, K, device=dev).to(bf)
    for odt in (torch.float32, bf):
        out = torch.empty(M, N, device=dev, dtype=odt)
        for env in ({}, {"CSN_GEMM_NO_256": "1"}, {"CSN_GEMM_NO_DMA": "1"}):
            os.environ.update(env)
            t = timeit(lambda: cabi.gemm_nt(a, b, None, out=out))
            for k in env: del os.environ[k]
            print(f"{name:12s} out={str(odt)[6:]:9s} {str(env):28s} {t*1e6:8.1f} us {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)
