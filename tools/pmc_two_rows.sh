#!/bin/bash
# Developer tool: rocprofv3 PMC passes over tools/bench_two_rows.py; per-counter means per launch for each
# k_sweep_two instantiation.  usage (GPU box): tools/pmc_two_rows.sh "<COUNTERS pass 1>" "<COUNTERS pass 2>" ...
export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1)); out=/tmp/pmc_two_$i; rm -rf $out
  (cd /tmp && B=21845 K=4 timeout 900 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/tools/bench_two_rows.py > $out.log 2>&1)
  python - "$out" <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_sweep_two" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("k_sweep_two")[1].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k].add(int(r["Dispatch_Id"]))
for k in acc:
    print("k_sweep_two" + k, len(n[k]), "launches:", {c: "%.4g" % (v / len(n[k])) for c, v in sorted(acc[k].items())})
PY
done
