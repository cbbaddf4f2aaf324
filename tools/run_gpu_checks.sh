#!/bin/bash
# Usage (on the GPU box via gpurun): tools/run_gpu_checks.sh <tag> [bench args]
tag=${1:-x}; shift
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
timeout 600 python bench.py "$@" 2>&1 | tail -2 > gpurun_out/bench_${tag}.log
python - <<PY
import json
for line in open("gpurun_out/bench_${tag}.log"):
    if line.startswith("{"):
        d = json.loads(line)
        print({k: d[k] for k in ("value", "ms_per_step", "env_steps_per_s")}, "frac", d["roofline"]["frac"],
              "sweeps", d["config"]["mean_sweeps_per_env_step"], "parity", d.get("cpu_baseline", {}).get("parity_max_abs_dT_K"),
              "cpu", d.get("cpu_baseline", {}).get("value"))
    else:
        print(line.rstrip())
PY
