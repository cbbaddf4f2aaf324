#!/bin/bash
# GPU box: kernel statistics of the policy loop (bench.py --config policy) -> gpurun_out/<tag>/policy_kernel_stats.csv
tag=${1:-pol}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
root=$PWD
cmd="python $root/bench.py --config policy --steps 24 --warmup 12"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pol_$tag -o pol -- $cmd > $root/gpurun_out/$tag/policy_under_rocprof.json 2> /tmp/prof_pol_$tag.err)
find /tmp/prof_pol_$tag -name "*kernel_stats.csv" -exec cp {} gpurun_out/$tag/policy_kernel_stats.csv \;
head -25 gpurun_out/$tag/policy_kernel_stats.csv | cut -c1-200
