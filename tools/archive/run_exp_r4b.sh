# Usage (GPU box): bash tools/run_exp_r4b.sh <variant> ...   -- cycle stamps of one building per variant build (tools/build_variant.sh)
mkdir -p gpurun_out
for v in "$@"; do
  echo "== $v"
  SBSIM_LIB=$PWD/tools/libexp_$v.so SBSIM_PHASE_TIMING=1 LIMS=5 B=65536 python tools/prof_sweeps.py 2>&1 | grep -E "iter_limit|step [5-7]|wavefront 0"
done 2>&1 | tee gpurun_out/exp_r4b.txt
