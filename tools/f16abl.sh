#!/bin/bash
# tools/f16abl.sh -- dev-only, ON THE GPU BOX: tools/f16bench.hip with each timing ablation.
set -e
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -I include -I mms_answer_selection_amd/csrc"
hipcc $F -DMMS_STAMPS tools/f16bench.hip -o /tmp/f16bench_miss && echo "=== miss count (MMS_STAMPS build)" && timeout -k 10 60 /tmp/f16bench_miss
for v in 0 1 2; do
  hipcc $F -DMMS_F16ABL=$v tools/f16bench.hip -o /tmp/f16bench$v
  echo "=== MMS_F16ABL=$v"; timeout -k 10 60 /tmp/f16bench$v
done
