#!/bin/bash
# Usage (GPU box): bash tools/run_r5_checks.sh  -- the full GPU suite, then the three bench configurations (JSON lines under gpurun_out/)
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r5_bench_default.json
timeout 600 python bench.py --steps 20 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_bench_steady.json
timeout 600 python bench.py --config policy --check-buildings 64 2>/dev/null | tail -1 > gpurun_out/r5_bench_policy.json
timeout 600 python bench.py --config mixed --check-buildings 64 2>/dev/null | tail -1 > gpurun_out/r5_bench_mixed.json
python - <<'PY'
import json
for n in ("default", "steady", "policy", "mixed"):
    try:
        d = json.load(open(f"gpurun_out/r5_bench_{n}.json"))
        print(n, {k: d[k] for k in ("value", "ms_per_step")}, "frac", d["roofline"]["frac"], d["roofline"].get("avg_kernel_ms"),
              "parity", d.get("cpu_baseline", {}).get("parity_max_abs_dT_K"), d.get("config", {}).get("parity_vs_oracle"))
    except Exception as e:
        print(n, "FAILED", e)
PY
