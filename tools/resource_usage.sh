#!/bin/bash
# kernel resource usage of the product library's kernels (hipcc -Rpass-analysis=kernel-resource-usage): VGPRs, SGPRs, scratch, LDS, occupancy
# usage: tools/resource_usage.sh [kernel-name-pattern] [extra -D flags]
pat=${1:-k_mesh_raster_fast}; shift
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize "$@" \
  -Rpass-analysis=kernel-resource-usage -c -o /tmp/ru.o mvtn_amd/csrc/mvr_kernels.hip 2>&1 | \
  awk -v pat="$pat" '/Function Name/ {on = ($0 ~ pat)} on && /remark/ {sub(/.*remark: [^ ]* /, ""); print}' | grep -v "^$"
