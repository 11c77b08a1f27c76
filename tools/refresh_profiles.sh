#!/bin/bash
# usage (GPU box, from the repo root): tools/refresh_profiles.sh <tag> [extra bench.py args]
#   tools/refresh_profiles.sh r04                 -> gpurun_out/r04/...        (cfg2, the headline)
#   tools/refresh_profiles.sh r04_cfg4 --config cfg4
# Produces under gpurun_out/<tag>/ : bench.json (bench.py line), bench_under_rocprof.json + bench_kernel_stats.csv
# (rocprofv3 --kernel-trace --stats of the same command), pmc_traffic.json (separate --pmc FETCH_SIZE / WRITE_SIZE passes),
# pmc_counters.json (one --pmc pass per counter group).  Copy them to profiles/<tag>_* afterwards.
set -e
tag=${1:-rXX}
shift || true
out=gpurun_out/$tag
mkdir -p $out
python3 bench.py "$@" > $out/bench.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-retrieval --no-f32-line --no-parity "$@" > $out/bench_under_rocprof.json 2> $out/prof.log
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/bench_kernel_stats.csv
rm -rf $out/prof
tools/pmc_traffic.sh $out/pmc_traffic.json "$@" > $out/pmc_traffic.log 2>&1
tail -3 $out/pmc_traffic.log
tools/pmc_counters.sh $out/pmc_counters.json "$@" > $out/pmc_counters.log 2>&1
tail -3 $out/pmc_counters.log
cut -c1-200 $out/bench.json
