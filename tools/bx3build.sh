#!/bin/bash
# tools/bx3build.sh NAME [-D...]: builds tools/bx3bench.hip into tools/bin/bx3bench_NAME (dev harness)
set -e
name=$1; shift
cd /tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off "$@" -I /root/repo/include -I /root/repo/mms_answer_selection_amd/csrc \
  /root/repo/tools/bx3bench.hip -o /root/repo/tools/bin/bx3bench_$name 2>&1 | grep -v "warning\|^ *[0-9]* |\|^ *|" | grep -i "error" -A3 && exit 1
test -x /root/repo/tools/bin/bx3bench_$name
