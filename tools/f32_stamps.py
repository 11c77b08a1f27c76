"""Diagnostic: phase breakdown of the float32 weight-stationary recurrence (diag library, `make diag`)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CSN_LIB_PATH"] = os.path.join(ROOT, "cerebralsignalnetworks_amd", "lib", "libcsn_hip_diag.so")
from cerebralsignalnetworks_amd import cabi, Model, EEGFilters  # noqa: E402
from cerebralsignalnetworks_amd.trainer import DistillTrainer  # noqa: E402

dev = torch.device("cuda:0")
B, C, T, H, L, D = 256, 128, 500, 768, 2, 384
if os.environ.get("CSN_STAMP_CFG4"):
    T, H = 440, 1024
m = Model(input_size=C, lstm_size=H, lstm_layers=L, output_size=D, include_top=False, compute_dtype=torch.float32).to(dev)
tr = DistillTrainer(m, EEGFilters(1000, 3).sos, loss="cosine")
x = torch.randn(B, C, T, device=dev); tg = torch.randn(B, D, device=dev)
lib = cabi.load()
lib.csn_debug_read_f32stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * 16)()
names = ("wait", "loads+mfma", "partials+barrier", "sums+math+store issue", "drain", "barrier+flag")
for it in range(3):
    tr.train_step(x, tg)
    torch.cuda.synchronize()
    lib.csn_debug_read_f32stamps(buf)
    for k, base in (("fwd", 0), ("bwd", 8)):
        per = [buf[base + i] * 0.01 / (T * L) for i in range(6)]      # (workgroup 5 stamps in both layers' launches)
        print("step %d %s per-step us: " % (it, k) + " | ".join("%s %.2f" % (n, v) for n, v in zip(names, per)) + " | sum %.2f" % sum(per), flush=True)
