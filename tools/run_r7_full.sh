#!/bin/bash
# round 6: the whole GPU suite, the soak of k_sweep_two's classes against the LDS-grid kernel, the planner fuzz, then the mixed line
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r7_gpu_tests.txt
(PLAN=SB2-synth B=8192 T=300 python tools/soak_two_kernels.py; PLAN=SB1-synth B=4096 T=250 python tools/soak_two_kernels.py) 2>&1 | grep steps | tee gpurun_out/r7_soak.txt
N=${FUZZ_N:-24} python tools/fuzz_planner.py 2>&1 | tail -4 | tee gpurun_out/r7_fuzz_planner.txt
python bench.py --config mixed --check-buildings 8 2>/dev/null | tail -1 > gpurun_out/r7_bench_mixed.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r7_bench_mixed.json"))
print("mixed", d["value"], d["ms_per_step"], {k: (v["sweep_kernel_ms_alone"], v["mean_sweeps_per_env_step"], v.get("parity_vs_oracle")) for k, v in d["config"]["classes"].items()})
PY
