#!/bin/bash
# Round 5 baseline (GPU box): driver window + steady window, phase stamps, policy loop with kernel statistics
tag=${1:-r6base}
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tools/quick_bench.sh $tag 2>&1 | tee gpurun_out/$tag/quick.txt
SBSIM_LIB=$PWD/tools/libexp_stamps.so SBSIM_PHASE_TIMING=1 LIMS=2 timeout 300 python tools/prof_sweeps.py 2>&1 | tail -20 | tee gpurun_out/$tag/phase.txt
timeout 300 python bench.py --config policy 2>/dev/null | tail -1 > gpurun_out/$tag/policy.json
python -c "
import json; d=json.load(open('gpurun_out/$tag/policy.json')); print('policy', d['ms_per_step'], d['value'])"
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/$tag/prof_policy -o pol -- python bench.py --config policy --steps 24 --warmup 12 > gpurun_out/$tag/policy_prof.log 2>&1
find gpurun_out/$tag/prof_policy -name '*kernel_stats.csv' | head -1 | xargs -I{} sh -c 'head -12 {}'
