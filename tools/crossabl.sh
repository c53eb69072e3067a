#!/bin/bash
# tools/crossabl.sh -- dev-only, ON THE GPU BOX: build tools/crossbench.hip with each timing ablation and run it.
set -e
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -I include -I mms_answer_selection_amd/csrc"
for v in 0 1 2 3; do
  hipcc $F -DMMS_XABL=$v tools/crossbench.hip -o /tmp/crossbench$v
  echo "=== MMS_XABL=$v"; timeout -k 10 60 /tmp/crossbench$v
done
