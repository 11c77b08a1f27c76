"""Run against the DEBUG library `make tags` (lib/libcsn_hip_tags.so, -DCSN_SLAB_TAGS), in a process of its own:

    CSN_LIB_PATH=$PWD/cerebralsignalnetworks_amd/lib/libcsn_hip_tags.so python tools/tags_check.py [reps]

Every hand-off piece of the sentinel-armed rings carries bit 2 of its step; every consumer checks the tag of every
piece it multiplies.  A ring slot that serves its PREVIOUS occupant (data of step t -/+ 4: the case the sentinel proof
cannot see) raises CSN_STATUS_STALE_SLOT.  Prints one JSON line:
  clean:    status words of `reps` cfg2-shaped forward+backward passes, default / no-hint / placement-independent
            forms, with a second stream hammering HBM + L2 and occupying CUs at random moments  -> must all be 0
  injected: the same with CSN_TAGS_NO_REARM=1 (the kernels skip the re-arm stores, consumers do not wait): the
            detector MUST fire (status & 4), or the check above proves nothing."""
import json
import os
import sys
import threading

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cerebralsignalnetworks_amd import cabi, Model      # noqa: E402

assert cabi.LIB_PATH.endswith("libcsn_hip_tags.so"), "select the tags library with CSN_LIB_PATH"
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6


def run(shape, env, reps, noise):
    B, T, C, H, L = shape
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    stop = threading.Event()

    def hammer():
        s = torch.cuda.Stream(device=dev)
        a = torch.empty(64 << 20, dtype=torch.float32, device=dev)
        b = torch.empty_like(a)
        rng = np.random.default_rng(0)
        with torch.cuda.stream(s):
            while not stop.is_set():
                b.copy_(a)
                if rng.random() < 0.5:
                    torch.mm(a[:1 << 20].view(1024, 1024), b[:1 << 20].view(1024, 1024))
                s.synchronize()

    th = threading.Thread(target=hammer) if noise else None
    try:
        torch.manual_seed(1)
        m = Model(input_size=C, lstm_size=H, lstm_layers=L, output_size=32, include_top=False).to(dev)
        x = torch.randn(B, T, C, device=dev)
        if th:
            th.start()
        st = 0
        for _ in range(reps):
            m.zero_grad()
            m(x).square().mean().backward()
            torch.cuda.synchronize()
            for plan in m.lstm.all_plans():
                st |= plan.status(clear=True)
        return st
    finally:
        stop.set()
        if th:
            th.join()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


res = {"clean": {}, "injected": {}}
for name, shape in (("cfg2_T96", (256, 96, 128, 768, 2)), ("h512", (130, 70, 16, 512, 2)), ("cfg4_T64", (256, 64, 128, 1024, 2))):
    for form, env in (("default", {}), ("nohint", {"CSN_DPOLL_NO_HINT": "1"}), ("anyplace", {"CSN_NO_XCD_LOCAL": "1"}),
                      ("streams", {"CSN_PERSIST_STREAMS": "1"})):
        res["clean"][f"{name}/{form}"] = run(shape, env, reps, noise=True)
    res["injected"][name] = run(shape, {"CSN_TAGS_NO_REARM": "1", "CSN_DPOLL_NO_HINT": "1"}, 2, noise=False)
print(json.dumps(res))
