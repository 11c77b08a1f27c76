"""csn_l2_topk at its edges: k up to the gallery size, k > gallery size, one query, duplicates (ties -> lowest index)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cerebralsignalnetworks_amd import cabi      # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
bad = 0
for (Ng, Nq, D, k) in ((100, 7, 33, 100), (100, 7, 33, 64), (5000, 3, 384, 200), (3, 2, 8, 5), (1, 1, 1, 1), (257, 1, 384, 17)):
    g = rng.standard_normal((Ng, D)).astype(np.float32)
    q = rng.standard_normal((Nq, D)).astype(np.float32)
    g[Ng // 2] = g[0]                       # a duplicate gallery row: the tie goes to the lower index
    try:
        dist, idx = cabi.l2_topk(torch.from_numpy(g).to(dev), torch.from_numpy(q).to(dev), k)
        idx = idx.cpu().numpy()
        d2 = ((q.astype(np.float64)[:, None, :] - g.astype(np.float64)[None, :, :]) ** 2).sum(-1)
        ref = np.argsort(d2, axis=1, kind="stable")[:, :min(k, Ng)]
        ok = np.array_equal(idx[:, :min(k, Ng)], ref) and (k <= Ng or (idx[:, Ng:] == -1).all())
        print(f"{'ok  ' if ok else 'FAIL'} Ng{Ng} Nq{Nq} D{D} k{k}")
        bad += 0 if ok else 1
    except cabi.CsnError as e:
        print(f"refused Ng{Ng} Nq{Nq} D{D} k{k}: {str(e)[:120]}")
print(f"{bad} failing case(s)")
sys.exit(1 if bad else 0)
