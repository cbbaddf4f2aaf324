#!/bin/bash
# round 6: k_sweep_roll's phase stamps with the decision's stamp fixed and the loop's back edge stamped (-DSB_STAMP_NEXT);
# then the stream tests on the experimental library
mkdir -p gpurun_out
(echo "== -DSB_PHASE_STAMPS -DSB_STAMP_NEXT ([15] = the next period's top)"; SB_STAMP_NEXT=1 SBSIM_LIB=$PWD/tools/libexp_rstamps.so SBSIM_PHASE_TIMING=1 LIMS=100 python tools/prof_sweeps.py 2>&1 | grep -v amdgpu.ids
 echo "== -DSB_PHASE_STAMPS ([15] = the ramp-up's end)"; SBSIM_LIB=$PWD/tools/libexp_rstamps0.so SBSIM_PHASE_TIMING=1 LIMS=100 python tools/prof_sweeps.py 2>&1 | grep -v amdgpu.ids) | tee gpurun_out/r7_phase_cycles.txt
SBSIM_LIB=$PWD/sbsim_amd/libsbsim_amd_exp.so python -m pytest tests/test_gpu_parity.py -q -m gpu -k "streaming_kernel or beyond_one_cu" 2>&1 | tail -3 | tee gpurun_out/r7_experimental_tests.txt
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "streaming_kernel or beyond_one_cu" 2>&1 | tail -3 | tee -a gpurun_out/r7_experimental_tests.txt
