"""Diagnostic: phase breakdown of the persistent forward INSIDE a real training step (diag library)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CSN_LIB_PATH"] = os.path.join(ROOT, "cerebralsignalnetworks_amd", "lib", os.environ.get("CSN_DIAG_LIB", "libcsn_hip_diag.so"))
from cerebralsignalnetworks_amd import cabi, Model, EEGFilters  # noqa: E402
from cerebralsignalnetworks_amd.trainer import DistillTrainer  # noqa: E402

dev = torch.device("cuda:0")
B, C, T, H, L, D = 256, 128, 500, 768, 2, 384
if os.environ.get("CSN_STAMP_CFG4"):      # Spampinato shapes: the N-split forward with the fused layer-0 projection
    T, H = 440, 1024
m = Model(input_size=C, lstm_size=H, lstm_layers=L, output_size=D, include_top=False).to(dev)
tr = DistillTrainer(m, EEGFilters(1000, 3).sos, loss="cosine")
x = torch.randn(B, C, T, device=dev); tg = torch.randn(B, D, device=dev)
lib = cabi.load()
lib.csn_debug_read_pstamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
lib.csn_debug_read_bstamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
lib.csn_debug_read_nstamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
has_ws = hasattr(lib, "csn_debug_read_wstamps")      # (wave-specialised forward: experiments library only)
if has_ws:
    lib.csn_debug_read_wstamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * 16)()
for it in range(4):
    tr.train_step(x, tg)
    torch.cuda.synchronize()
    # the stamping workgroup (blockIdx.x == 11) belongs to ONE layer's group: sums are over that layer's T steps
    if has_ws:
        lib.csn_debug_read_wstamps(buf)
    if has_ws and sum(buf) != 0:
        w = [buf[i] * 0.01 / T for i in range(16)]
        print("step %d fwd-ws per-step us, MFMA wave 0 (4 chains): wait ready %.2f | lds reads + mfma %.2f | tiles to lds + bump %.2f | sum %.2f"
              % (it, w[0], w[1], w[2], sum(w[:3])))
        print("          gate wave 0 (chain 0): input wait %.2f | x mfma %.2f | wait tiles %.2f | sum+gate math %.2f | stores %.2f | drain %.2f | flag + watch next line %.2f | h tile to LDS %.2f | sum %.2f"
              % (w[8], w[9], w[10], w[11], w[12], w[13], w[15], w[14], sum(w[8:16])), flush=True)
        print("          shader clock during the chunk: %.0f MHz" % (100.0 * buf[5] / max(buf[6], 1)), flush=True)
    for name, fn in (("fwd-ksplit", lib.csn_debug_read_pstamps), ("fwd-nsplit", lib.csn_debug_read_nstamps),
                     ("bwd", lib.csn_debug_read_bstamps)):
        fn(buf)
        per = [buf[i] * 0.01 / T for i in range(6)]
        if sum(buf[i] for i in range(7)) == 0:
            continue
        if name == "bwd" and any(buf[8 + i] for i in range(4)):
            print("   bwd data polls: MFMA phases redone per wave (all workgroups, this training step):", [int(buf[8 + i]) for i in range(4)])
        if name == "fwd-nsplit":
            print("   nsplit detail per-step us: dma issue %.2f | x-mfma + input request %.2f | wait g0 %.2f | g0 mfma + wait g1 %.2f | g1,g2 mfma + waits %.2f | (g3 mfma in loads+mfma rest)"
                  % tuple(buf[i] * 0.01 / T for i in (8, 9, 10, 11, 12)))
        if name == "fwd-ksplit" and buf[13]:
            n = buf[13]
            print("   fwd launches %d: waves alive (first entry -> last exit) %.1f us | entry skew %.1f | exit skew %.1f   (rocprofv3's dispatch duration also holds the dispatch and the end-of-kernel write-back)"
                  % (n, buf[10] * 0.01 / n, buf[11] * 0.01 / n, buf[12] * 0.01 / n))
        if name == "fwd-ksplit" and hasattr(lib, "csn_debug_read_grp_alive"):
            g = (ctypes.c_ulonglong * 32)()
            lib.csn_debug_read_grp_alive.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
            lib.csn_debug_read_grp_alive(g)
            print("   fwd workgroup alive per launch, by hand-off group (slice 0; us, launches): " +
                  " ".join("g%d %.1f (%d)" % (i, g[i] * 0.01 / max(1, g[8 + i]), g[8 + i]) for i in range(8)))
            print("   fwd step loop per launch / per step, by group: " +
                  " ".join("g%d %.1f / %.2f" % (i, g[16 + i] * 0.01 / max(1, g[8 + i]), g[16 + i] * 0.01 / max(1, g[24 + i])) for i in range(8)))
        if name == "fwd-ksplit":
            print("   fwd epilogue detail per-step us: pass 0 reads + math %.2f | pass 0 store issue %.2f | (rest of 'epilogue' = the split pass)" % (buf[8] * 0.01 / T, buf[9] * 0.01 / T))
        print("step %d %s per-step us: wait %.2f | loads+mfma %.2f | lds/gate-math %.2f | epilogue %.2f | drain %.2f | signal %.2f | sum %.2f | prologue per launch %.1f us"
              % (it, name, *per, sum(per), buf[6] * 0.01 / 16), flush=True)
