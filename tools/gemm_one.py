"""Runs one GEMM shape a few times (for PMC collection).  usage: gemm_one.py nt|tn"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cerebralsignalnetworks_amd import cabi
dev = torch.device("cuda:0"); bf = torch.bfloat16
if sys.argv[1] == "nt":
    M, N, K = 32768, 3072, 768
    a = torch.randn(M, K, device=dev).to(bf); b = torch.randn(N, K, device=dev).to(bf); out = torch.empty(M, N, device=dev)
    for _ in range(5): cabi.gemm_nt(a, b, None, out=out)
else:
    M, N, K = 3072, 768, 128000
    a = torch.randn(K, M, device=dev).to(bf); b = torch.randn(K, N, device=dev).to(bf)
    for _ in range(5): cabi.gemm_tn(a, b)
torch.cuda.synchronize()
