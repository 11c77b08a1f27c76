#!/bin/bash
# usage (container, any cwd): tools/abl_build.sh <name> "<-D flags>" <file.hip> [...]
# An ablation library for timing experiments: the named sources recompiled with the flags, every other object taken from
# the product build -> cerebralsignalnetworks_amd/lib/libcsn_abl_<name>.so (select with CSN_LIB_PATH; never committed).
set -e
name=$1; flags=$2; shift 2
cd "$(dirname "$0")/../cerebralsignalnetworks_amd/csrc"
make -j7 all > /dev/null
B=../../build/csrc; A=../../build/abl_$name; mkdir -p $A
objs=""
for s in util eeg_filter gemm lstm_cell lstm_cell_blk lstm_fwd_persist lstm_fwd_ns lstm_bwd_persist lstm_f32_persist lstm loss retrieval; do
  if [[ " $* " == *" $s.hip "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on $flags -c $s.hip -o $A/$s.o &
    objs="$objs $A/$s.o"
  else objs="$objs $B/$s.o"; fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o ../lib/libcsn_abl_$name.so
echo built ../lib/libcsn_abl_$name.so
