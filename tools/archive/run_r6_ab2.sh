#!/bin/bash
# GPU box: selected GPU tests, then exp_ab on the product build alone (base is an older ABI), quick bench.
tag=${1:-ab}; kexpr=${2:-"one_day or divergent or iteration_limit or idle_and_drawing or rectangular_building or full_size"}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "$kexpr" 2>&1 | tail -4 | tee gpurun_out/$tag/tests.txt
ROUNDS=3 STEPS=60 timeout 600 python tools/exp_ab.py product 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$tag/ab.txt
tools/quick_bench.sh $tag 2>&1 | tee gpurun_out/$tag/quick.txt
