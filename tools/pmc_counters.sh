#!/bin/bash
# usage (GPU box, repo root): tools/pmc_counters.sh <out.json> [extra bench.py args]
# Hardware counters per kernel launch of the bench workload, one rocprofv3 --pmc pass per counter GROUP (SQ has 8
# slots per pass, TCC 4, GRBM 2: MI355X_MICROARCH.md "rocprofv3 PMC slots"), never combined with any trace domain
# other than --kernel-trace.  A group whose pass fails (unknown counter name on this ROCm) is reported and skipped.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/pmc_counters.json}
shift
groups=(
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE"
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES"
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
  "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
  "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
)
i=0
: > gpurun_out/pmcc_index.txt
for g in "${groups[@]}"; do
  d=gpurun_out/pmcc_$i
  rm -rf $d
  if rocprofv3 --pmc $g --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-retrieval --no-f32-line --no-parity "$@" > $d.log 2>&1; then
    echo "$d" >> gpurun_out/pmcc_index.txt
    echo "pass $i ok: $g"
  else
    echo "pass $i FAILED: $g"; tail -3 $d.log
  fi
  i=$((i+1))
done
python3 - "$out" "$*" <<'PY'
import csv, glob, json, sys, collections
names = ["lstm_bwd_persist_kernel", "lstm_fwd_persist_kernel", "lstm_fwd_ns_kernel", "lstm_fwd_f32_persist_kernel", "lstm_bwd_f32_persist_kernel", "gemm_generic_kernel", "gemm_f32_128_kernel", "gemm_nt_wide_kernel", "gemm_nt_256_kernel", "gemm_nt_dma_kernel", "gemm_tn_256_kernel",
         "gemm_tn_bf16_kernel", "gemm_tn_n128_kernel", "eeg_filter_scan_kernel", "lstm_cell_fwd_il_kernel", "lstm_cell_bwd_il_kernel",
         "rmsprop_flat_kernel"]
res = {"bench_args": sys.argv[2], "command": "rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-retrieval --no-f32-line --no-parity (one pass per group)",
       "note": "per-launch averages; SQ_* cycle counters other than SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES count quad-cycles summed over waves; "
               "GRBM_GUI_ACTIVE is summed over the 8 XCDs", "kernels": {}}
for d in open("gpurun_out/pmcc_index.txt").read().split():
    fs = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    tot = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(fs[0])):
        for n in names:
            if n in r["Kernel_Name"]:
                k = (n, r["Counter_Name"])
                tot[k][0] += float(r["Counter_Value"]); tot[k][1] += 1
                break
    for (n, c), (v, k) in tot.items():
        res["kernels"].setdefault(n, {})[c] = v / k
        res["kernels"][n]["launches"] = k
for n, e in res["kernels"].items():
    der = {}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE"):
        # MFMA-busy cycles summed over the chip's 1024 SIMDs / (active cycles per XCD x 1024 SIMDs)
        der["mfma_util"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / ((e["GRBM_GUI_ACTIVE"] / 8.0) * 256 * 4)
    if "SQ_WAVE_CYCLES" in e and e.get("GRBM_GUI_ACTIVE"):
        der["waves_per_simd_avg"] = 4.0 * e["SQ_WAVE_CYCLES"] / ((e["GRBM_GUI_ACTIVE"] / 8.0) * 256 * 4)
    if e.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in e:
                der[c.lower() + "_share"] = e[c] / e["SQ_WAVE_CYCLES"]
    if e.get("TCC_REQ_sum"):
        der["l2_hit_rate"] = e.get("TCC_HIT_sum", 0.0) / max(1.0, e.get("TCC_HIT_sum", 0.0) + e.get("TCC_MISS_sum", 0.0))
    if e.get("SQ_LDS_IDX_ACTIVE"):
        der["lds_bank_conflict_share"] = e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"]
    e["derived"] = der
sys.path.insert(0, ".")
import bench
res["csrc_sha16"] = bench.csrc_sha16()     # bench.py drops these values once the kernel sources differ
json.dump(res, open(sys.argv[1], "w"), indent=1)
for n, e in res["kernels"].items():
    print(n, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in e["derived"].items()})
PY
for d in $(cat gpurun_out/pmcc_index.txt); do rm -rf $d; done
