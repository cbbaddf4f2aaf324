#!/bin/bash
# Developer tool: rocprofv3 PMC passes over a short run of the step kernel.
# usage: tools/pmc_run.sh <tag> "<COUNTERS pass 1>" "<COUNTERS pass 2>" ...
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_$tag
i=0
for ctrs in "$@"; do
  i=$((i+1))
  out=/tmp/pmc_${tag}_$i
  rm -rf $out
  (cd /tmp && LIMS=100 timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o p -- python /root/repo/tools/prof_sweeps.py > $out.log 2>&1)
  python - "$out" "$ctrs" <<'PY'
import sys, glob, csv, collections
out, ctrs = sys.argv[1], sys.argv[2]
files = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(list)
for f in files:
    for row in csv.DictReader(open(f)):
        if "k_step" in row.get("Kernel_Name", ""):
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k}: mean/dispatch={sum(v)/len(v):.4g} n={len(v)}")
if not agg:
    print("no counters found; files:", files[:3]); print(open(out + ".log").read()[-1500:])
PY
done
