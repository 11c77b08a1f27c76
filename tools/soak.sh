#!/bin/bash
# usage (GPU box, repo root): tools/soak.sh <out.log>   -- the round's soak: hand-off stress, stale-slot detector (debug library),
# long bench runs with the status word checked at the end.  ~3 minutes.
out=${1:-gpurun_out/soak.log}
mkdir -p $(dirname $out)
{
date
echo "handoff_stress 300 reps (cfg2 forward+backward under a noisy second stream; every output bit-equal to the quiet run)"
python3 tools/handoff_stress.py 300 2>&1 | tail -4
echo "handoff_stress 100 reps, exact-float32 path (weight-stationary float32 kernels)"
python3 tools/handoff_stress.py 100 f32 2>&1 | tail -2
echo "tags_check 40 reps per form (debug library: stale-slot detector)"
CSN_LIB_PATH=$PWD/cerebralsignalnetworks_amd/lib/libcsn_hip_tags.so python3 tools/tags_check.py 40 2>&1 | tail -1
echo "bench.py --steps 2000 (cfg2; status word checked after the timed region)"
python3 bench.py --steps 2000 --no-cpu-baseline --no-retrieval --no-f32-line --no-parity 2>/dev/null | cut -c1-420
echo "bench.py --config cfg4 --steps 1000 (fused layer-0 projection at H = 1024, input part behind the publish)"
python3 bench.py --config cfg4 --steps 1000 --no-cpu-baseline --no-retrieval --no-f32-line --no-parity 2>/dev/null | cut -c1-420
echo "bench.py --dtype f32 --steps 200 (weight-stationary float32 recurrence)"
python3 bench.py --dtype f32 --steps 200 --no-cpu-baseline --no-retrieval --no-parity 2>/dev/null | cut -c1-420
date
} > $out 2>&1
grep -v "^rep " $out
