#!/bin/bash
# usage (GPU box): tools/pmc_gemm.sh <tag> nt|tn  -> FETCH_SIZE / WRITE_SIZE per GEMM kernel launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; which=$2
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcg_${tag}_$c -- python3 tools/gemm_one.py $which > /dev/null 2>&1
  f=$(find gpurun_out/pmcg_${tag}_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$c" <<'PY'
import csv, sys, collections
tot = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != sys.argv[2]: continue
    k = r["Kernel_Name"][:40]
    tot[k][0] += float(r["Counter_Value"]); tot[k][1] += 1
for k, (v, n) in tot.items():
    if "gemm" in k: print(f"{sys.argv[2]} {k}: launches={n} per-launch raw={v/n:.4g}")
PY
  rm -rf gpurun_out/pmcg_${tag}_$c
done
