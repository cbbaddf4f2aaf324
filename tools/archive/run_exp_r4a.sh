mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "divergent or roll_kernel or one_day or iteration_limit or full_size" 2>&1 | tail -4
export LIMS=5,6
for v in desync desync_1coef desync_nosr desync_nocopy desync_notrack desync_nomem desync_lean; do
  echo "== $v"; SBSIM_LIB=$PWD/tools/libexp_$v.so python tools/exp_fixed_sweeps.py 2>&1 | grep "sweeps="
done 2>&1 | tee gpurun_out/exp_r4a.txt
echo "== desync exact kernel"; SBSIM_ROLL_EXACT=1 SBSIM_LIB=$PWD/tools/libexp_desync.so python tools/exp_fixed_sweeps.py 2>&1 | grep "sweeps=" | tee -a gpurun_out/exp_r4a.txt
tools/quick_bench.sh a2 2>&1 | tail -2
