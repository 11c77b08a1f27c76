"""The collectives bench.py and the trainer issue on the RCCL ("nccl") backend, on a ONE-rank communicator (all a one-GPU
box can offer): dtypes and reduce ops must be supported by the backend -- float64 / int64 / int32 tensors, MAX / MIN / SUM,
barrier, broadcast.      python tests/diag/rccl_ops_one_rank.py"""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29631")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
t = torch.tensor([1.5], device=dev, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert t.item() == 1.5
o = torch.ones(1, device=dev, dtype=torch.float64); dist.all_reduce(o); assert o.item() == 1.0
c = torch.tensor([-464701186992624], device=dev, dtype=torch.int64)
lo, hi = c.clone(), c.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX); assert lo.item() == hi.item() == c.item()
s = torch.tensor([0, 1, 0], device=dev, dtype=torch.int32); dist.all_reduce(s, op=dist.ReduceOp.MAX); assert s.tolist() == [0, 1, 0]
g = torch.randn(31114752 // 4, device=dev); ref = g.clone(); dist.all_reduce(g, op=dist.ReduceOp.SUM); assert torch.equal(g, ref)
p = torch.randn(1000, device=dev); dist.broadcast(p, src=0)
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl one-rank ops ok")
