#!/bin/bash
# usage (GPU box, repo root): tools/pmc_filter.sh <tag>
# Counters of the band-pass + z-score kernel alone (one rocprofv3 --pmc pass per group, kernel trace only).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-filter}
groups=(
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE"
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES"
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_BUSY_CU_CYCLES"
  "FETCH_SIZE"
  "WRITE_SIZE"
  "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum"
)
i=0
for g in "${groups[@]}"; do
  d=gpurun_out/pmcf_$i
  rm -rf $d
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d $d -- python3 tools/filterbench.py 6 > $d.log 2>&1 || { echo "pass $i FAILED: $g"; tail -3 $d.log; }
  i=$((i+1))
done
python3 - "$tag" <<'PY'
import csv, glob, json, sys, collections
out = collections.OrderedDict()
for d in sorted(glob.glob("gpurun_out/pmcf_*")):
    if not d[-1].isdigit():
        continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        tot = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "eeg_filter_scan_kernel" in r["Kernel_Name"] and "float" in r["Kernel_Name"] and "bf16" not in r["Kernel_Name"]:
                tot[r["Counter_Name"]][0] += float(r["Counter_Value"]); tot[r["Counter_Name"]][1] += 1
        for k, (v, n) in tot.items():
            out[k] = v / n
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        ds = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "eeg_filter_scan_kernel" in r["Kernel_Name"]]
        if ds:
            out.setdefault("kernel_ns", []).append(sorted(ds)[len(ds) // 2])
json.dump(out, open(f"gpurun_out/{sys.argv[1]}_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
rm -rf gpurun_out/pmcf_[0-9]
