#!/bin/bash
# round 6: kernel durations of k_sweep_two (rocprofv3 --kernel-trace --stats) with and without measure-free periods
export TMPDIR=/tmp
mkdir -p gpurun_out
for k in off on; do
  if [ $k = off ]; then export SBSIM_TWO_NO_SKIP=1; else unset SBSIM_TWO_NO_SKIP; fi
  out=/tmp/tr_$k; rm -rf $out
  (cd /tmp && B=21845 K=4 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/tools/bench_two_rows.py > $out.log 2>&1)
  echo "== measure-free periods $k" | tee -a gpurun_out/r7_trace_two.txt
  grep synth $out.log | tee -a gpurun_out/r7_trace_two.txt
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  grep -E "Name|k_sweep_two" $f | cut -c1-250 | tee -a gpurun_out/r7_trace_two.txt
done
rocm-smi --showclocks --showpower 2>&1 | head -30 | tee -a gpurun_out/r7_trace_two.txt
