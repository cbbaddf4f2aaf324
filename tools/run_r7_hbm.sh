#!/bin/bash
mkdir -p gpurun_out
tools/bin/ubench_hbm 2>&1 | tee gpurun_out/r7_hbm_ubench.txt
python tools/regimes_roll.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r7_roll_regimes.txt
