#!/bin/bash
# round 6: shader clock and socket power while k_sweep_two runs, with and without measure-free periods
mkdir -p gpurun_out
for k in off on; do
  if [ $k = off ]; then export SBSIM_TWO_NO_SKIP=1; else unset SBSIM_TWO_NO_SKIP; fi
  echo "== measure-free periods $k" >> gpurun_out/r7_clk.txt
  B=21845 K=1500 python tools/bench_two_rows.py > /tmp/b_$k.log 2>&1 &
  pid=$!
  sleep 5
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | tr '\n' ' ' >> gpurun_out/r7_clk.txt; echo >> gpurun_out/r7_clk.txt
    sleep 0.5
  done
  grep synth /tmp/b_$k.log >> gpurun_out/r7_clk.txt
done
cat gpurun_out/r7_clk.txt
