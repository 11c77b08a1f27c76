import torch, time
dev = torch.device("cuda:0")
def bench(name, fn, flops, reps=20):
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
K, M, N = 128000, 3072, 768
a = torch.randn(K, M, device=dev).to(torch.bfloat16); b = torch.randn(K, N, device=dev).to(torch.bfloat16)
bench("hipblaslt tn 3072x768x128000 bf16 out", lambda: torch.mm(a.t(), b), 2.0*K*M*N)
b128 = torch.randn(K, 128, device=dev).to(torch.bfloat16)
bench("hipblaslt tn 3072x128x128000", lambda: torch.mm(a.t(), b128), 2.0*K*M*128)
a2 = torch.randn(8192, 768, device=dev).to(torch.bfloat16); b2 = torch.randn(3072, 768, device=dev).to(torch.bfloat16)
bench("hipblaslt nt 8192x3072x768 bf16 out", lambda: torch.mm(a2, b2.t()), 2.0*8192*3072*768)
A = torch.randn(8192, 8192, device=dev).to(torch.bfloat16); B = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
bench("hipblaslt nn 8192^3", lambda: torch.mm(A, B), 2.0*8192**3)
bench("hipblaslt nt 8192^3", lambda: torch.mm(A, B.t()), 2.0*8192**3)
bench("hipblaslt tn 8192^3", lambda: torch.mm(A.t(), B), 2.0*8192**3)
try:
    bench("hipblaslt nt 8192x3072x768 f32 out", lambda: torch.mm(a2, b2.t(), out_dtype=torch.float32), 2.0*8192*3072*768)
except Exception as e:      # noqa: BLE001
    print("mm(out_dtype=float32) not available:", str(e)[:120])
