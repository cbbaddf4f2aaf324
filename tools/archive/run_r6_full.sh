#!/bin/bash
# GPU box: the whole GPU suite, then the driver's bench command (with its "also" legs).   tools/run_r6_full.sh <tag>
tag=${1:-full}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/$tag/tests.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
tail -3 gpurun_out/$tag/bench.err
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
d = None
for line in open(f"gpurun_out/{tag}/bench.json"):
    if line.startswith("{"):
        d = json.loads(line)
if d is None:
    print("no bench line"); sys.exit(0)
r = d["roofline"]
print(f"headline: {d['ms_per_step']:.3f} ms/step kernel {r.get('avg_kernel_ms', 0):.3f} ms frac {r['frac']:.4f} value {d['value']:.4g} sweeps {d['config']['mean_sweeps_per_env_step']:.2f} traffic {r.get('traffic')} parity {d.get('cpu_baseline', {}).get('parity_max_abs_dT_K')}")
cb = d.get("cpu_baseline", {})
print("cpu_baseline:", {k: cb.get(k) for k in ("value", "cores", "threads_effective", "cpus_allowed", "host_loadavg_1min_before_after")})
for name, leg in d.get("also", {}).items():
    print(name, {k: v for k, v in leg.items() if k not in ("classes", "workload", "trace")})
    for cname, c in leg.get("classes", {}).items():
        print("   ", cname, c)
PY
