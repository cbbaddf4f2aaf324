#!/bin/bash
# Developer tool: rocprofv3 PMC passes over a short bench run; prints per-counter sums for the sweep kernel.
# usage (GPU box): tools/pmc_sweep.sh "<COUNTERS pass 1>" "<COUNTERS pass 2>" ...
export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1)); out=/tmp/pmc_sweep_$i; rm -rf $out
  (cd /tmp && timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 > $out.log 2>&1)
  python - "$out" <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    # (not k_sweep_roll<96, true>: the float64 instantiation's near-empty launch on the redo list, one per step)
    rows = [r for r in csv.DictReader(open(f)) if "k_sweep" in r["Kernel_Name"] and "k_sweep_roll<96, true>" not in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-20:]     # the timed launches
    for r in rows:
        if int(r["Dispatch_Id"]) in ids: acc[r["Counter_Name"]] += float(r["Counter_Value"]) / len(ids)
print({k: "%.4g" % v for k, v in sorted(acc.items())})
PY
done
