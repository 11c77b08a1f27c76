#!/bin/bash
out=gpurun_out/ab/chunk.txt; mkdir -p gpurun_out/ab; : > $out
for cfg in cfg2 cfg4; do for c in 25 28 32 36 40 44 50 32; do
  CSN_LSTM_CHUNK=$c python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-retrieval --no-f32-line --no-parity 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$cfg chunk $c', round(d['ms_per_step'],3), {k:round(v,1) for k,v in r['us_per_launch'].items()}, r['launches'])" >> $out
done; done; cat $out
