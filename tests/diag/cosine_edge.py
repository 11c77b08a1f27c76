"""Cosine loss on degenerate rows (all-zero student / teacher rows, 1e-20 and 1e20 magnitudes) against torch in float64."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cerebralsignalnetworks_amd import cabi      # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
s = torch.randn(6, 33, device=dev)
t = torch.randn(6, 33, device=dev)
s[1] = 0
t[2] = 0
s[3] = 1e-20
t[4] = 1e20
st = s.clone().double().requires_grad_(True)
ref = (1.0 - torch.nn.functional.cosine_similarity(st, t.double(), dim=1)).mean()
ref.backward()
loss, g = cabi.cosine_loss(s, t)
print("loss", float(loss), float(ref), "finite grad", bool(torch.isfinite(g).all()))
print("row max|grad| reference", [f"{v:.3g}" for v in st.grad.abs().amax(1).tolist()])
print("row max|grad| hip      ", [f"{v:.3g}" for v in g.abs().amax(1).tolist()])
