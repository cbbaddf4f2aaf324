#!/bin/bash
# GPU box, round 5's evidence in one call: the driver's command (with its "also" legs), the headline leg's kernel statistics, PMC
# traffic + SQ counters, phase stamps, the steady window; the mixed and policy lines with their kernel statistics.
tag=${1:-r06}
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/profiles_$tag
mkdir -p $out
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.time
tools/collect_profiles.sh $tag > $out/collect.log 2>&1
tail -25 $out/collect.log
timeout 600 python bench.py --config policy 2>/dev/null | tail -1 > $out/bench_policy.json
timeout 600 python bench.py --config mixed 2>/dev/null | tail -1 > $out/bench_mixed.json
cmdm="python $root/bench.py --config mixed --steps 12 --warmup 6"
echo "$cmdm" > $out/command_mixed.txt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mixed_$tag -o bench -- $cmdm > $out/bench_mixed_under_rocprof.json 2> /tmp/prof_mixed_$tag.err)
find /tmp/prof_mixed_$tag -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_mixed.csv \;
cmdp="python $root/bench.py --config policy --steps 24 --warmup 12"
echo "$cmdp" > $out/command_policy.txt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_policy_$tag -o bench -- $cmdp > $out/bench_policy_under_rocprof.json 2> /tmp/prof_policy_$tag.err)
find /tmp/prof_policy_$tag -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_policy.csv \;
[ -f gpurun_out/env_api_pairs.json ] && cp gpurun_out/env_api_pairs.json $out/env_api_pairs.json
python - $out <<'PY'
import json, sys
out = sys.argv[1]
for name in ("bench_driver_cmd", "bench_policy", "bench_mixed", "bench_steady"):
    try:
        d = None
        for line in open(f"{out}/{name}.json"):
            if line.startswith("{"): d = json.loads(line)
        print(name, "ms/step %.4f value %.4g frac %.4f" % (d["ms_per_step"], d["value"], d["roofline"]["frac"]))
    except Exception as e:
        print(name, "??", e)
PY
