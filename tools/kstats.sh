#!/bin/bash
# usage (GPU box, repo root): [ENV=...] tools/kstats.sh <name> [bench.py args]  ->  gpurun_out/kstats_<name>.csv (rocprofv3 --kernel-trace --stats)
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
d=gpurun_out/kstats_$name
rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-retrieval --no-f32-line --no-parity "$@" > $d.json 2> $d.log
cp $(find $d -name "*kernel_stats.csv" | head -1) $d.csv && rm -rf $d
head -9 $d.csv | cut -d, -f1-4 | cut -c1-160
