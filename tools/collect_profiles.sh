#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats + HBM-traffic PMC passes for the bench command.
# Outputs go to gpurun_out/profiles_<tag>/ (copy the summaries you want judged into profiles/).
tag=${1:-r01}; shift
STEPS=${STEPS:-20}; WARM=${WARM:-5}   # the driver's bench command
export TMPDIR=/tmp
out=$PWD/gpurun_out/profiles_$tag
mkdir -p $out
cmd="python $PWD/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-also"   # (the headline leg alone: the default run's "also" legs launch other kernels)
echo "$cmd" > $out/command.txt
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats_$tag -o bench -- $cmd > $out/bench_under_rocprof.json 2> /tmp/prof_stats_$tag.err)
find /tmp/prof_stats_$tag -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find /tmp/prof_stats_$tag -name "*kernel_trace.csv" -exec sh -c 'head -1 "$1" > '"$out"'/kernel_trace_sb.csv; grep -E "k_sweep|k_pre|k_post|k_reset" "$1" >> '"$out"'/kernel_trace_sb.csv' _ {} \;
# PMC: one counter per pass (FETCH_SIZE needs 3 of the 4 TCC slots); plus a calibration run with
# iteration_limit=1 whose traffic is known by construction (the state is read once and written once)
for ctr in FETCH_SIZE WRITE_SIZE; do
  for variant in main calib; do
    extra=""; [ $variant = calib ] && extra="--iteration-limit 1"
    (cd /tmp && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_${ctr}_${variant}_$tag -o bench -- $cmd $extra > /dev/null 2> /tmp/prof_${ctr}_${variant}_$tag.err)
  done
done
python - "$tag" "$out" "$WARM" <<'PY'
import sys, glob, csv, json, re
tag, out, warm = sys.argv[1], sys.argv[2], int(sys.argv[3])
def vals(ctr, variant, kernel):
    v = []
    for f in sorted(glob.glob(f"/tmp/prof_{ctr}_{variant}_{tag}/**/*counter_collection.csv", recursive=True)):
        # (k_sweep_roll<96, true> is the float64 instantiation's launch on the redo list: a second, near-empty dispatch per step)
        rows = [r for r in csv.DictReader(open(f)) if kernel in r.get("Kernel_Name", "") and "k_sweep_roll<96, true>" not in r.get("Kernel_Name", "") and r["Counter_Name"] == ctr]
        rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
        v += [float(r["Counter_Value"]) for r in rows]
    return v
B = 65536
bench = {}
for line in open(out + "/bench_under_rocprof.json"):
    if line.startswith("{"):
        bench = json.loads(line)
li = bench.get("config", {}).get("launch", {})
Z, O = bench.get("config", {}).get("zones", 9), 46
# sb_launch_info.state_bytes_per_env_step = 16*state_doubles + per-zone/scalar/obs bytes (sbsim_hip.hip)
state_bytes = (li.get("state_bytes_per_env_step", 0) - ((8 * 4 + 4 * 2) * Z + 16 * 16 + 8 + 4 * O + 4)) // 2
kern = "k_sweep"
# the sweep kernel's real name, from the counter rows themselves
kernel_name = kern
for f in sorted(glob.glob(f"/tmp/prof_FETCH_SIZE_main_{tag}/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if kern in r.get("Kernel_Name", "") and "k_sweep_roll<96, true>" not in r.get("Kernel_Name", ""):
            m = re.search(r"k_sweep\w*(<[^>]*>)?", r["Kernel_Name"])
            kernel_name = m.group(0) if m else kern
            break
res = {"tag": tag, "units": "rocprofv3 FETCH_SIZE / WRITE_SIZE are in KiB", "kernel": kernel_name,
       "sweep_kernel_id": li.get("kernel"), "state_bytes_per_building": state_bytes}
f_main, w_main = vals("FETCH_SIZE", "main", kern), vals("WRITE_SIZE", "main", kern)
f_cal, w_cal = vals("FETCH_SIZE", "calib", kern), vals("WRITE_SIZE", "calib", kern)
res["fetch_kib_per_launch_all"] = f_main
res["write_kib_per_launch_all"] = w_main
ft, wt = f_main[warm:], w_main[warm:]
res["fetch_kib_timed_mean"] = sum(ft) / max(len(ft), 1)
res["write_kib_timed_mean"] = sum(wt) / max(len(wt), 1)
# calibration on known byte counts in this kernel's own access pattern (MI355X_MICROARCH.md, HBM):
# a sweep launch reads every building's state once (8 B/lane, 512-byte rows) + its g table
# (256 B), Bld record (112 B) and ring extremes (16 B), and writes the state + zone sums back --
# whatever the number of sweeps; the iteration_limit=1 run is the reference point.
exp_read = B * (state_bytes + 256 + 112 + 16)
exp_write = B * (state_bytes + 8 * Z + 12)
fc, wc = f_cal[warm:] or f_cal, w_cal[warm:] or w_cal
res["calib_fetch_kib_one_sweep"] = sum(fc) / max(len(fc), 1)
res["calib_write_kib_one_sweep"] = sum(wc) / max(len(wc), 1)
res["expected_read_bytes_per_launch"] = exp_read
res["expected_write_bytes_per_launch"] = exp_write
res["fetch_correction"] = exp_read / (res["calib_fetch_kib_one_sweep"] * 1024) if fc else None
res["write_correction"] = exp_write / (res["calib_write_kib_one_sweep"] * 1024) if wc else None
fcorr = res["fetch_correction"] or 2.0
wcorr = res["write_correction"] or 1.0
res["hbm_bytes_per_launch"] = res["fetch_kib_timed_mean"] * 1024 * fcorr + res["write_kib_timed_mean"] * 1024 * wcorr
res["source"] = (f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps ... --warmup {warm}`, "
                 f"sweep-kernel launches of the timed region only, corrected by x{fcorr:.3f} (reads) and x{wcorr:.3f} (writes), "
                 "both calibrated on an iteration_limit=1 run of the same command whose traffic is known by construction; "
                 "profiles/" + tag + "_traffic.json")
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if not k.endswith("_all")}, indent=1))
PY
head -3 $out/kernel_stats.csv | cut -c1-200
# SQ counters of the sweep kernel (instruction mix, wave cycles, LDS conflicts, instruction cache), per timed launch
bash $PWD/tools/pmc_sweep.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
  "SQ_WAVES SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_MISSES" > $out/pmc_sq.txt 2>&1
# cycle stamps of one building's phases inside the sweep kernel, and the per-sweep / fixed cost split
# (the stamps are compiled in with -DSB_PHASE_STAMPS only: tools/build_variant.sh stamps step_roll.hip,step_two_76.hip,step_two_80.hip,step_band_76.hip,...,step_band_96.hip -DSB_PHASE_STAMPS)
[ -f $PWD/tools/libexp_stamps.so ] && SBSIM_LIB=$PWD/tools/libexp_stamps.so SBSIM_PHASE_TIMING=1 LIMS=1,2,4 timeout 600 python $PWD/tools/prof_sweeps.py 2>&1 | grep -v amdgpu.ids > $out/phase_cycles.txt
# the same bench after 100 warm-up steps: the steady state beside the driver's transient window
timeout 600 python $PWD/bench.py --steps 20 --warmup 100 --no-cpu-baseline --no-also 2>/dev/null | tail -1 > $out/bench_steady.json
ls -la $out
