#!/bin/bash
# Kernel resource usage (registers, scratch, occupancy) of every HIP source, from the compiler's own
# remarks with the flags of sbsim_amd/build.py.  Usage: tools/resource_usage.sh > profiles/rNN_resource_usage.txt
cd "$(dirname "$0")/.."
echo "hipcc -Rpass-analysis=kernel-resource-usage, gfx950, $(/opt/rocm/bin/hipcc --version | grep -m1 -o 'HIP version.*'), flags of sbsim_amd/build.py"
echo
for f in sbsim_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -mllvm -amdgpu-atomic-optimizer-strategy=None -Iinclude -Isbsim_amd/csrc \
    -Rpass-analysis=kernel-resource-usage --cuda-device-only -c -o /dev/null "$f" 2>&1 |
    grep -E "remark:" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
    awk '/Function Name/{if (line) print line; line=$0; next} /VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size/{line=line "  | " $0} END{print line}'
done
