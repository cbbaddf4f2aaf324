#!/bin/bash
# Developer tool: rocprofv3 PMC passes over any of the tools; per-launch means for the kernels whose name contains <substring>.
# usage (GPU box): tools/pmc_tool.sh <tools/script.py> <kernel substring> "<COUNTERS pass 1>" "<COUNTERS pass 2>" ...
export TMPDIR=/tmp
script=$1; sub=$2; shift 2
i=0
for ctrs in "$@"; do
  i=$((i+1)); out=/tmp/pmc_tool_$i; rm -rf $out
  (cd /tmp && timeout 900 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/$script > $out.log 2>&1)
  python - "$out" "$sub" <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
print({k: "%.4g" % (v / max(len(n[k]), 1)) for k, v in sorted(acc.items())}, "launches", {k: len(v) for k, v in n.items()})
PY
done
