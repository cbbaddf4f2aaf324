#!/bin/bash
# Usage (GPU box): tools/quick_bench.sh <tag>   -- the driver's bench window and a steady-state window, one line each
tag=${1:-x}
mkdir -p gpurun_out
for w in 5 100; do
  timeout 300 python bench.py --steps 20 --warmup $w --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/qb_${tag}_w$w.json
  python - "$tag" $w <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/qb_{sys.argv[1]}_w{sys.argv[2]}.json"))
r = d["roofline"]
print(f"warmup {sys.argv[2]:>3}: {d['ms_per_step']:.3f} ms/step  kernel {r.get('avg_kernel_ms', 0):.3f} ms  frac {r['frac']:.4f}  sweeps {d['config']['mean_sweeps_per_env_step']:.2f}  value {d['value']:.4g}")
PY
done
