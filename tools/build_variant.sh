#!/bin/bash
# Developer aid: tools/build_variant.sh <name> <source.hip> <hipcc flags...>
# Re-compiles ONE translation unit with extra flags and links it with the product's other objects into
# tools/libexp_<name>.so (git-ignored, travels to the GPU box); run with SBSIM_LIB=$PWD/tools/libexp_<name>.so.
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
(cd $root && python -c "from sbsim_amd.build import build; build()")
obj=$root/sbsim_amd/csrc/_obj
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -fPIC "$@" \
  -I $root/include -I $root/sbsim_amd/csrc -c $root/sbsim_amd/csrc/$src -o $obj/${base}_$name.o
objs=""
for o in sbsim_hip step_reg step_roll step_two step_two_76 step_two_80 step_band step_stream step_lds generators floorplan episode; do
  [ $o = $base ] && objs="$objs $obj/${base}_$name.o" || objs="$objs $obj/$o.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $root/tools/libexp_$name.so
rm -f $obj/${base}_$name.o
echo $root/tools/libexp_$name.so
