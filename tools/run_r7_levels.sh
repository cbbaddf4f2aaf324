#!/bin/bash
# round 6: k_sweep_two's LDS levels (2: four buildings per CU, 1: three, 0: two) with and without measure-free periods
mkdir -p gpurun_out
for lv in 2 1 0; do for k in off on; do
  if [ $k = off ]; then export SBSIM_TWO_NO_SKIP=1; else unset SBSIM_TWO_NO_SKIP; fi
  export SBSIM_TWO_MAX_LEVEL=$lv
  echo "== level $lv, measure-free periods $k" | tee -a gpurun_out/r7_levels.txt
  K=${K:-24} python tools/bench_two_rows.py 2>&1 | grep synth | tee -a gpurun_out/r7_levels.txt
done; done
