#!/bin/bash
# tools/ablate.sh -- dev-only: build timing-ablation variants of the library on the GPU box
# and run tools/membench against each (outputs are WRONG in ablated builds; only time matters).
set -e
ROOT=$(pwd)
SRCS=$(ls mms_answer_selection_amd/csrc/*.hip)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -I include -I mms_answer_selection_amd/csrc"
for v in "$@"; do
  mkdir -p /tmp/abl$v
  hipcc $FLAGS -DMMS_ABLATE=$v $SRCS -o /tmp/abl$v/libmms_hip.so
  hipcc --offload-arch=gfx950 -O3 -I include tools/membench.hip -o /tmp/abl$v/membench -L /tmp/abl$v -lmms_hip -Wl,-rpath,/tmp/abl$v
  echo "=== MMS_ABLATE=$v"
  /tmp/abl$v/membench | grep mms
done
