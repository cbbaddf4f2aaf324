#!/bin/bash
# Usage (GPU box): bash tools/run_r5_final.sh -- the full GPU suite, the soak of step_band.hip's plans, the bench lines
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
(for pr in 10,4,19,20 10,4,18,20 9,4,16,17; do PLAN_ROOMS=$pr B=2048 T=300 timeout 900 python tools/soak_two_kernels.py 2>&1 | grep -v amdgpu.ids | sed "s/^/PLAN_ROOMS=$pr: /"; done) | tee gpurun_out/soak_band.txt
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r5_bench_default.json
timeout 600 python bench.py --config mixed 2>/dev/null | tail -1 > gpurun_out/r5_bench_mixed.json
python - <<'PY'
import json
for n in ("default", "mixed"):
    try:
        d = json.load(open(f"gpurun_out/r5_bench_{n}.json"))
        print(n, {k: d[k] for k in ("value", "ms_per_step")}, "frac", d["roofline"]["frac"])
    except Exception as e:
        print(n, "FAILED", e)
PY
