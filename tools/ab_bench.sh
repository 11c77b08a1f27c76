#!/bin/bash
# usage: ab.sh <out> <bench args...> -- lib1 lib2 ...
out=$1; shift
args=()
while [[ "$1" != "--" ]]; do args+=("$1"); shift; done; shift
mkdir -p $(dirname $out)
for rep in 1 2; do
for lib in "$@"; do
  if [[ "$lib" == "product" ]]; then unset CSN_LIB_PATH; else export CSN_LIB_PATH=$PWD/cerebralsignalnetworks_amd/lib/libcsn_abl_$lib.so; fi
  python bench.py --steps ${ABSTEPS:-20} --warmup 3 --no-cpu-baseline --no-retrieval --no-f32-line --no-parity "${args[@]}" 2>>$out.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['ms_per_step'],3), {k:round(v,1) for k,v in d['roofline']['us_per_launch'].items()})" >> $out
done; done
cat $out
