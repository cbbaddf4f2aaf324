#!/bin/bash
# GPU box: the roll-kernel parity tests on the product build, then an interleaved A/B of library builds on the bench
# batch (tools/exp_ab.py), then phase stamps.   tools/run_r6_ab.sh <tag> "<variants>"   e.g. "base product"
tag=${1:-ab}; variants=${2:-"base product"}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kats.py -m gpu -x -q -k "one_day or divergent or iteration_limit or idle_and_drawing or rectangular_building or full_size" 2>&1 | tail -4 | tee gpurun_out/$tag/tests.txt
ROUNDS=${ROUNDS:-3} STEPS=${STEPS:-60} timeout 1200 python tools/exp_ab.py $variants 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$tag/ab.txt
if [ -f tools/libexp_stamps.so ]; then
  SBSIM_LIB=$PWD/tools/libexp_stamps.so SBSIM_PHASE_TIMING=1 LIMS=2 timeout 300 python tools/prof_sweeps.py 2>&1 | grep -v amdgpu.ids | tail -17 | tee gpurun_out/$tag/phase.txt
fi
if [ -f tools/libexp_stamps0.so ]; then
  echo "--- stamps of the FIRST sweep's period"
  SBSIM_LIB=$PWD/tools/libexp_stamps0.so SBSIM_PHASE_TIMING=1 LIMS=2 timeout 300 python tools/prof_sweeps.py 2>&1 | grep -v amdgpu.ids | tail -16 | grep "second sweep" | tee gpurun_out/$tag/phase0.txt
fi
