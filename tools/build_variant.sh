#!/bin/bash
# Developer aid: tools/build_variant.sh <name> <source.hip>[,<source.hip>...] <hipcc flags...>
# Re-compiles the named translation units with extra flags and links them with the product's other objects into
# tools/libexp_<name>.so (git-ignored, travels to the GPU box); run with SBSIM_LIB=$PWD/tools/libexp_<name>.so.
# e.g. the cycle stamps of tools/prof_sweeps.py / tools/bench_two_rows.py (SBSIM_PHASE_TIMING=1):
#   tools/build_variant.sh stamps step_roll.hip,step_two_76.hip,step_two_80.hip,step_band_76.hip,step_band_80.hip,step_band_84.hip,step_band_88.hip,step_band_92.hip,step_band_96.hip -DSB_PHASE_STAMPS
set -e
name=$1; srcs=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
(cd $root && python -c "from sbsim_amd.build import build; build()")
obj=$root/sbsim_amd/csrc/_obj
bases=""
for src in ${srcs//,/ }; do
  base=$(basename $src .hip)
  bases="$bases $base"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -fPIC -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" \
    -I $root/include -I $root/sbsim_amd/csrc -c $root/sbsim_amd/csrc/$src -o $obj/${base}_$name.o &
done
wait
objs=""
for o in sbsim_hip step_reg step_roll step_two step_two_64 step_two_76 step_two_80 step_band step_band_68 step_band_72 step_band_76 step_band_80 step_band_84 step_band_88 step_band_92 step_band_96 step_stream step_lds generators floorplan episode; do
  case " $bases " in *" $o "*) objs="$objs $obj/${o}_$name.o";; *) objs="$objs $obj/$o.o";; esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $root/tools/libexp_$name.so
for b in $bases; do rm -f $obj/${b}_$name.o; done
echo $root/tools/libexp_$name.so
