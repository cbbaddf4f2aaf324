#!/bin/bash
# Usage (GPU box): tools/exp_variants.sh <variant> [<variant> ...]   ("product" = the library itself)
# Each variant (tools/libexp_<variant>.so, tools/build_variant.sh): the fixed-sweep fit and the bench windows.
mkdir -p gpurun_out
for v in "$@"; do
  if [ $v = product ]; then unset SBSIM_LIB; else export SBSIM_LIB=$PWD/tools/libexp_$v.so; fi
  echo "== $v"
  [ -n "$NOFIT" ] || python tools/exp_fixed_sweeps.py 2>&1 | grep -v amdgpu.ids | tail -1
  bash tools/quick_bench.sh exp_$v
done 2>&1 | tee gpurun_out/exp_variants.txt
