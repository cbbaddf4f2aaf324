#!/bin/bash
# Developer tool: rocprofv3 PMC passes over tools/bench_mid_plans.py (step_band.hip on PLANS=<i>); prints per-counter sums per launch.
# usage (GPU box): PLANS=0 tools/pmc_mid_plans.sh "<COUNTERS pass 1>" "<COUNTERS pass 2>" ...
export TMPDIR=/tmp BAND_ONLY=1 K=${K:-6}
i=0
for ctrs in "$@"; do
  i=$((i+1)); out=/tmp/pmc_mid_$i; rm -rf $out
  (cd /tmp && timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/tools/bench_mid_plans.py > $out.log 2>&1)
  python - "$out" <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_sweep_band" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-6:]     # the timed launches
    for r in rows:
        if int(r["Dispatch_Id"]) in ids: acc[r["Counter_Name"]] += float(r["Counter_Value"]) / len(ids)
print({k: "%.4g" % v for k, v in sorted(acc.items())})
PY
done
