#!/bin/bash
# round 6: k_sweep_two with measure-free periods -- parity tests, then A/B of the skip (off / kappa values) on the two classes
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "measure_free or block_kernels or two_rows or mixed_environment_gives" 2>&1 | tail -5 > gpurun_out/r7_two_tests.txt
cat gpurun_out/r7_two_tests.txt
for k in off 0.9 1.2 1.6 2.0; do
  if [ $k = off ]; then export SBSIM_TWO_NO_SKIP=1; unset SBSIM_DEBUG_SKIP_KAPPA; else unset SBSIM_TWO_NO_SKIP; export SBSIM_DEBUG_SKIP_KAPPA=$k; fi
  echo "== kappa $k" | tee -a gpurun_out/r7_two_ab.txt
  python tools/bench_two_rows.py 2>&1 | grep synth | tee -a gpurun_out/r7_two_ab.txt
done
for k in off 0.9 1.3; do
  if [ $k = off ]; then export SBSIM_TWO_NO_SKIP=1; unset SBSIM_DEBUG_SKIP_KAPPA; else unset SBSIM_TWO_NO_SKIP; export SBSIM_DEBUG_SKIP_KAPPA=$k; fi
  echo "== stamps build, kappa $k" | tee -a gpurun_out/r7_two_ab.txt
  SBSIM_LIB=$PWD/tools/libexp_stamps.so SBSIM_PHASE_TIMING=1 python tools/bench_two_rows.py 2>&1 | grep synth | tee -a gpurun_out/r7_two_ab.txt
done
