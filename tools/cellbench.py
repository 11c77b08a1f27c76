"""Micro-benchmark of the per-timestep LSTM cell kernels (GPU box only).
Times N back-to-back launches with events on the launch stream; variants via env vars."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cerebralsignalnetworks_amd import cabi  # noqa: E402


def timeit(fn, n=300, warm=20):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = torch.cuda.current_stream()
    e0.record(st)
    for _ in range(n):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    B = int(os.environ.get("B", 256))
    H = int(os.environ.get("H", 768))
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    w = (torch.randn(4 * H, H, device=dev) / np.sqrt(H)).to(dt)
    wt = w.t().contiguous()
    T = 64
    hs = torch.randn(T, B, H, device=dev).to(dt)
    c = torch.randn(B, H, device=dev)
    xp = torch.randn(T, B, 4 * H, device=dev)
    dg = (torch.randn(T, B, 4 * H, device=dev) * 0.1).to(dt)
    gates = torch.rand(T, B, 4 * H, device=dev).to(dt)
    dcar = torch.zeros(B, H, device=dev)
    dy = torch.randn(B, H, device=dev)
    i = [0]

    def fwd():
        i[0] = (i[0] + 1) % T
        cabi.lstm_cell_forward(hs[i[0]], w, xp[i[0]], c)

    def fwd_nok():
        i[0] = (i[0] + 1) % T
        cabi.lstm_cell_forward(None, w, xp[i[0]], c)

    def bwd():
        i[0] = (i[0] + 1) % T
        cabi.lstm_cell_backward(dg[i[0]], wt, dy, gates[i[0]], c, c, dcar)

    def bwd_nok():
        i[0] = (i[0] + 1) % T
        cabi.lstm_cell_backward(None, wt, dy, gates[i[0]], c, c, dcar)

    def empty():
        cabi.cosine_loss(dy[:4], dy[:4], want_grad=False)

    only = os.environ.get("ONLY")
    res = {}
    for name, fn in (("fwd", fwd), ("fwd_no_k", fwd_nok), ("bwd", bwd), ("bwd_no_k", bwd_nok), ("2 tiny kernels", empty)):
        if only and name != only:
            continue
        res[name] = round(timeit(fn), 2)
    print(" ".join(f"{k}={v}us" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
