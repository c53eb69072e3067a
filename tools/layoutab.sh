#!/bin/bash
# tools/layoutab.sh [CFG ...] -- dev-only, ON THE GPU BOX: A/B of the Euclidean G1 kernels' global-memory layout
# (pair: row-aligned | block: workgroup-dense | wave: wave-dense) and waves per workgroup, via tools/bin/membench
# (tools/membench.hip linked against the library).  CFG = fwdlayout,bwdlayout,fwdwaves,bwdwaves[,fusedlayout]
[ $# -eq 0 ] && set -- pair,pair,8,8 pair,block,8,8 block,block,8,8 block,block,8,4 pair,block,8,4 block,block,4,4 pair,block,4,4 block,block,16,16
for cfg in "$@"; do
  IFS=, read -r lf lb wf wb lfu <<< "$cfg"
  lfu=${lfu:-pair}
  echo "=== forward $lf x $wf waves, backward $lb x $wb waves, fused $lfu"
  MMS_EUCLID_LAYOUT_FWD=$lf MMS_EUCLID_LAYOUT_BWD=$lb MMS_PAIR32_WPB_FWD=$wf MMS_PAIR32_WPB_BWD=$wb MMS_EUCLID_LAYOUT_FUSED=$lfu timeout -k 10 120 tools/bin/membench | grep -E "mms fwd|mms bwd|SEQ mms" || exit 1
done
