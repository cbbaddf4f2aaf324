#!/bin/bash
# round 6: A/B of library builds on k_sweep_two's two classes: tools/run_r7_ab_lib.sh <tag> <lib or ""> [<lib> ...]
mkdir -p gpurun_out
tag=$1; shift
for lib in "$@"; do
  if [ "$lib" = product ]; then unset SBSIM_LIB; else export SBSIM_LIB=$PWD/tools/libexp_$lib.so; fi
  for lv in ${LEVELS:-2}; do
    export SBSIM_TWO_MAX_LEVEL=$lv
    echo "== $lib, level $lv" | tee -a gpurun_out/r7_ab_$tag.txt
    K=${K:-24} python tools/bench_two_rows.py 2>&1 | grep synth | tee -a gpurun_out/r7_ab_$tag.txt
  done
done
