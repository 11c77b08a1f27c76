"""A/B of the inline-asm MFMA read hazard behind round 3's "wrong row groups" event.

    make -C cerebralsignalnetworks_amd/csrc hazard_demo        (build container; the .so files travel with the snapshot)
    python tests/diag/mfma_hazard_ab.py                         (GPU box)

Three libraries run the SAME fused layer-0 forward at H = 1024 (B 256 = 32 copies of 8 segments, so every 16-row MFMA row
group of every 64-row tile holds copies of the same rows and must produce the same bits):
  hazard1  round 3's operand constraints: the compiler copies 8 W_hh fragments VGPR -> AGPR straight in front of their
           first MFMA (`v_accvgpr_write_b32 a35, v219` / `v_mfma ... a[32:35]`, 0 wait states; tools/check_asm_hazards.py: 20)
  hazard2  the same constraints with `s_nop 1` in front of every such MFMA (the two wait states the hazard recogniser
           inserts for MFMAs it can see)
  product  W_ih as a VGPR operand: no copies at all
Expected: hazard1 -- copies in row group 0 differ from the copies in row groups 1..3 and from the other two libraries;
hazard2 and product -- all copies bit-equal, and equal to each other (same arithmetic, same order).
One child process per library (CSN_LIB_PATH is read at import); prints one JSON line per library and a verdict."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIBS = {"hazard1": "libcsn_hip_hazard1.so", "hazard2": "libcsn_hip_hazard2.so", "product": "libcsn_hip.so"}


def child(tag):
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from cerebralsignalnetworks_amd.lstm_model import HipLSTM
    dev = torch.device("cuda:0")
    B8, T, C, H, L = 8, 96, 128, 1024, 2
    rng = np.random.default_rng(5)
    x8 = rng.standard_normal((B8, T, C)).astype(np.float32)
    x = torch.from_numpy(np.tile(x8, (32, 1, 1))).to(dev)
    torch.manual_seed(9)
    m = HipLSTM(C, H, L, compute_dtype=torch.bfloat16).to(dev)
    with torch.no_grad():
        y = m(x)
    torch.cuda.synchronize()
    plan = m.all_plans()[0]
    assert plan.status() == 0
    y = y.float().cpu().numpy().reshape(32, B8, H)              # [copy, segment, unit]; copy k sits in rows 8k .. 8k+7
    rowgroup = (np.arange(32) * 8 // 16) % 4                     # 16-row MFMA row group (inside its 64-row tile) of each copy
    ref = y[2]                                                   # a copy in row group 1
    per_copy = np.abs(y - ref[None]).reshape(32, -1).max(axis=1)
    out = {"lib": tag, "path": plan.path(), "max_abs_diff_vs_copy2_by_rowgroup": {
        str(g): float(per_copy[rowgroup == g].max()) for g in range(4)},
        "elements_differing_in_copy0": int((y[0] != ref).sum()), "elements": int(ref.size)}
    np.save(os.path.join(ROOT, "gpurun_out", f"hazard_ab_{tag}.npy"), y[:4])
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    import numpy as np
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    res = {}
    for tag, so in LIBS.items():
        lib = os.path.join(ROOT, "cerebralsignalnetworks_amd", "lib", so)
        if not os.path.exists(lib):
            print(f"{tag}: {lib} missing (make hazard_demo)")
            sys.exit(2)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), tag], env=dict(os.environ, CSN_LIB_PATH=lib),
                             capture_output=True, text=True, timeout=600)
        if out.returncode != 0:
            print(out.stderr[-2000:])
            sys.exit(1)
        res[tag] = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        print(json.dumps(res[tag]))
    ys = {t: np.load(os.path.join(ROOT, "gpurun_out", f"hazard_ab_{t}.npy")) for t in LIBS}
    verdict = {
        "hazard1_rowgroup0_differs": res["hazard1"]["elements_differing_in_copy0"] > 0,
        "hazard1_rowgroups123_agree": all(res["hazard1"]["max_abs_diff_vs_copy2_by_rowgroup"][g] == 0.0 for g in "123"),
        "hazard2_all_copies_equal": all(v == 0.0 for v in res["hazard2"]["max_abs_diff_vs_copy2_by_rowgroup"].values()),
        "product_all_copies_equal": all(v == 0.0 for v in res["product"]["max_abs_diff_vs_copy2_by_rowgroup"].values()),
        "hazard2_equals_product_bitwise": bool((ys["hazard2"] == ys["product"]).all()),
        "hazard1_rowgroup1_equals_product_bitwise": bool((ys["hazard1"][2] == ys["product"][2]).all()),
    }
    print(json.dumps({"verdict": verdict}))
