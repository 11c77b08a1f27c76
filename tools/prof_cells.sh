#!/bin/bash
# usage: tools/prof_cells.sh <tag> [variants...]  (on the GPU box) -> kernel durations from rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
vars=${@:-fwd fwd_no_k bwd bwd_no_k}
for v in $vars; do
  ONLY=$v rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pc_${tag}_$v -- python3 tools/cellbench.py > /dev/null 2>&1
  f=$(find gpurun_out/pc_${tag}_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$v" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "lstm_cell" in r["Name"]:
        print(f"{sys.argv[2]}: {r['Name'][:48]} calls={r['Calls']} avg_us={float(r['AverageNs'])/1e3:.2f} min_us={float(r['MinNs'])/1e3:.2f}")
PY
  rm -rf gpurun_out/pc_${tag}_$v
done
