#!/bin/bash
# Developer tool: rocprofv3 PMC passes over tools/bench_convection.py; per-launch means for k_convect.
# usage (GPU box): tools/pmc_convect.sh "<COUNTERS pass 1>" "<COUNTERS pass 2>" ...
export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1)); out=/tmp/pmc_conv_$i; rm -rf $out
  (cd /tmp && timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/tools/bench_convection.py > $out.log 2>&1)
  python - "$out" <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_convect" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    for r in rows:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]) / max(len(ids), 1)
print({k: "%.4g" % v for k, v in sorted(acc.items())})
PY
done
