#!/bin/bash
# GPU box, one development iteration on k_sweep_roll: the roll-kernel parity tests, the driver / steady windows,
# phase stamps (tools/libexp_stamps.so when present), the policy loop.   tools/run_r6_iter.sh <tag> [pytest -k expr]
tag=${1:-it}; kexpr=${2:-"one_day or divergent or iteration_limit or idle_and_drawing or rectangular_building or full_size"}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kats.py -m gpu -x -q -k "$kexpr" 2>&1 | tail -6 | tee gpurun_out/$tag/tests.txt
tools/quick_bench.sh $tag 2>&1 | tee gpurun_out/$tag/quick.txt
if [ -f tools/libexp_stamps.so ]; then
  SBSIM_LIB=$PWD/tools/libexp_stamps.so SBSIM_PHASE_TIMING=1 LIMS=2 timeout 300 python tools/prof_sweeps.py 2>&1 | grep -v amdgpu.ids | tail -17 | tee gpurun_out/$tag/phase.txt
fi
timeout 300 python bench.py --config policy 2>/dev/null | tail -1 > gpurun_out/$tag/policy.json
python -c "
import json; d=json.load(open('gpurun_out/$tag/policy.json')); print('policy ms/step %.4f value %.4g' % (d['ms_per_step'], d['value']))" | tee -a gpurun_out/$tag/quick.txt
