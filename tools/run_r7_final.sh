#!/bin/bash
# round 6, last check on the final tree: smoke(), the whole GPU suite, the default bench line (traffic replayed from the committed profile)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r7_final_smoke.txt
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r7_final_gpu_tests.txt
( time python bench.py ) > gpurun_out/r7_final_bench.json 2> gpurun_out/r7_final_bench.time
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r7_final_bench.json") if l.startswith("{")][-1])
r = d["roofline"]
print("value %.4g ms %.4f frac %.4f traffic %s measured_in_run %s" % (d["value"], d["ms_per_step"], r["frac"], r["traffic"], r.get("traffic_measured_in_run")))
a = d["also"]
print("steady frac %.4f | mixed %.4g | policy %.4g (sweep %.3f ms, %.2f TB/s) | large %s" % (a["steady"]["roofline_frac"], a["mixed"]["value"], a["policy"]["value"], a["policy"]["sweep_kernel_ms"], a["policy"]["hbm_TBps_real"], {k: v.get("parity_vs_oracle", {}).get("sweep_count_mismatches") for k, v in a["large_plans"].items() if isinstance(v, dict)}))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["parallel_efficiency"])
PY
tail -4 gpurun_out/r7_final_bench.time
