import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from cerebralsignalnetworks_amd import cabi
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
def rnd(*s): return torch.randn(*s, device=dev, generator=g)
for (M, N, K) in ((8192, 3072, 768), (8192, 4096, 1024), (8192, 768, 3072), (300, 960, 256), (500, 32768, 128), (257, 1920, 320)):
    a, b = rnd(M, K).to(torch.bfloat16), rnd(N, K).to(torch.bfloat16)
    bias = rnd(N)
    ref = a.double() @ b.double().t() + bias.double()
    out = cabi.gemm_nt(a, b, bias)
    e1 = float((out.double() - ref).norm() / ref.norm())
    os.environ["CSN_GEMM_NO_192"] = "1"
    old = cabi.gemm_nt(a, b, bias)
    del os.environ["CSN_GEMM_NO_192"]
    same = bool(torch.equal(out, old))
    o16 = cabi.gemm_nt(a, b, bias, out_dtype=torch.bfloat16)
    e2 = float((o16.double() - ref).norm() / ref.norm())
    acc = out.clone()
    cabi.gemm_nt(a, b, None, out=acc, accumulate=True)
    e3 = float((acc.double() - (2 * ref - bias.double())).norm() / ref.norm())
    print(f"{M}x{N}x{K}: f32 err {e1:.2e} bit-equal to the 128-wide kernel {same} | bf16 out {e2:.2e} | accumulate {e3:.2e}", flush=True)
def bench(name, fn, flops, reps=30):
    t_end = time.time() + 1.5
    while time.time() < t_end:
        for _ in range(10): fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    print(f"{name}: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
a2, b2 = rnd(8192, 768).to(torch.bfloat16), rnd(3072, 768).to(torch.bfloat16)
bias = rnd(3072); out = torch.empty(8192, 3072, device=dev)
bench("nt 8192x3072x768 f32 (256x192)", lambda: cabi.gemm_nt(a2, b2, bias, out=out), 2.0 * 8192 * 3072 * 768)
os.environ["CSN_GEMM_NO_192"] = "1"
bench("nt 8192x3072x768 f32 (256x128)", lambda: cabi.gemm_nt(a2, b2, bias, out=out), 2.0 * 8192 * 3072 * 768)
del os.environ["CSN_GEMM_NO_192"]
a3, b3 = rnd(8192, 3072).to(torch.bfloat16), rnd(768, 3072).to(torch.bfloat16)
out3 = torch.empty(8192, 768, device=dev)
bench("nt 8192x768x3072 f32 (256x192)", lambda: cabi.gemm_nt(a3, b3, None, out=out3), 2.0 * 8192 * 3072 * 768)
os.environ["CSN_GEMM_NO_192"] = "1"
bench("nt 8192x768x3072 f32 (old)", lambda: cabi.gemm_nt(a3, b3, None, out=out3), 2.0 * 8192 * 3072 * 768)
del os.environ["CSN_GEMM_NO_192"]
a4, b4 = rnd(8192, 1024).to(torch.bfloat16), rnd(4096, 1024).to(torch.bfloat16)
bias4 = rnd(4096); out4 = torch.empty(8192, 4096, device=dev)
bench("nt 8192x4096x1024 f32 (256x256)", lambda: cabi.gemm_nt(a4, b4, bias4, out=out4), 2.0 * 8192 * 4096 * 1024)
os.environ["CSN_GEMM_NO_192"] = "1"
bench("nt 8192x4096x1024 f32 (256x128)", lambda: cabi.gemm_nt(a4, b4, bias4, out=out4), 2.0 * 8192 * 4096 * 1024)
del os.environ["CSN_GEMM_NO_192"]
a5, b5 = rnd(112640, 128).to(torch.bfloat16), rnd(4096, 128).to(torch.bfloat16)
bias5 = rnd(4096); out5 = torch.empty(112640, 4096, device=dev)
ref5 = None
bench("nt 112640x4096x128 f32 (wide)", lambda: cabi.gemm_nt(a5, b5, bias5, out=out5), 2.0 * 112640 * 4096 * 128, reps=10)
w = out5.clone()
os.environ["CSN_GEMM_NO_192"] = "1"
bench("nt 112640x4096x128 f32 (old)", lambda: cabi.gemm_nt(a5, b5, bias5, out=out5), 2.0 * 112640 * 4096 * 128, reps=10)
print("bit-equal:", bool(torch.equal(w, out5)))
