#!/bin/bash
# Usage (on the GPU box via gpurun): tools/run_bench_only.sh <tag> [bench args] -- bench.py only, one summary line
tag=${1:-x}; shift
mkdir -p gpurun_out
timeout 600 python bench.py "$@" 2>&1 | tail -2 > gpurun_out/bench_${tag}.log
python - <<PY
import json
for line in open("gpurun_out/bench_${tag}.log"):
    if line.startswith("{"):
        d = json.loads(line)
        print("${tag}", {k: d[k] for k in ("value", "ms_per_step")}, "frac", round(d["roofline"]["frac"], 4), "kernel_ms", round(d["roofline"]["avg_kernel_ms"], 4),
              "sweeps", round(d["config"]["mean_sweeps_per_env_step"], 3), "parity", d.get("cpu_baseline", {}).get("parity_max_abs_dT_K"))
    else:
        print(line.rstrip())
PY
