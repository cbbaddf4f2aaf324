#!/bin/bash
# Runs on the GPU box: BASELINE.json configs[2] (bench.py --config mixed) -- the bench line, the
# rocprofv3 kernel statistics of the same command, and the per-class developer bench.
# Outputs go to gpurun_out/mixed_<tag>/ (copy what should be judged into profiles/).
tag=${1:-r03}
export TMPDIR=/tmp
out=$PWD/gpurun_out/mixed_$tag
mkdir -p $out
cmd="python $PWD/bench.py --config mixed --steps 12 --warmup 6"
echo "$cmd" > $out/command.txt
timeout 600 $cmd 2>/dev/null | tail -1 > $out/bench_mixed.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mixed_$tag -o bench -- $cmd > $out/bench_mixed_under_rocprof.json 2> /tmp/prof_mixed_$tag.err)
find /tmp/prof_mixed_$tag -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_mixed.csv \;
[ -f $PWD/tools/libexp_stamps.so ] && export SBSIM_LIB=$PWD/tools/libexp_stamps.so   # the stamps and block counters need a -DSB_PHASE_STAMPS build
SBSIM_PHASE_TIMING=1 timeout 600 python tools/bench_two_rows.py 2>/dev/null | tail -2 > $out/two_rows_classes.txt
unset SBSIM_LIB
head -6 $out/kernel_stats_mixed.csv | cut -c1-220
cat $out/two_rows_classes.txt | cut -c1-300
python - $out/bench_mixed.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"])
PY
