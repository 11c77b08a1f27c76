"""The data-parallel training step on the RCCL ("nccl") backend with the weight-stationary kernels, as far as ONE GPU can
show it: a one-rank communicator, the trainer told that the world has two ranks (so it broadcasts, all-reduces the 31 MB flat
gradient buffer on the device every step, scales by 1 / world and all-reduces the status verdict).  What this exercises that
the gloo rehearsals cannot: RCCL's kernels between the persistent forward / backward launches of consecutive steps, on the
same device, and the dtypes / reduce ops of every collective on the real backend.      python tests/diag/nccl_trainer_one_rank.py"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29633")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)

from cerebralsignalnetworks_amd import Model, EEGFilters      # noqa: E402
from cerebralsignalnetworks_amd import trainer as tr           # noqa: E402

tr.dist_info = lambda: (0, 2)          # (the communicator has one rank; the trainer behaves as rank 0 of 2)
torch.manual_seed(43)
B, C, T, H, L, D = 256, 128, 500, 768, 2, 384
m = Model(input_size=C, lstm_size=H, lstm_layers=L, output_size=D, include_top=False).to(dev)
t = tr.DistillTrainer(m, EEGFilters(1000, order=3).sos, loss="cosine", lr=1e-3, optimizer="rmsprop")
x = torch.randn(B, C, T, device=dev)
tg = torch.randn(B, D, device=dev)
t.grads.timing = []
losses = [float(t.train_step(x, tg)) for _ in range(6)]
dist.barrier()
torch.cuda.synchronize()
t.check_device_status()
assert all(l == l for l in losses), losses
assert losses[-1] < losses[0], losses
print("losses", [round(l, 4) for l in losses], "all-reduce ms per step", round(t.grads.all_reduce_ms(), 3), "backend", dist.get_backend())
dist.destroy_process_group()
print("nccl one-rank trainer ok")
