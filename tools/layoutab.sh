#!/bin/bash
# tools/layoutab.sh -- dev-only, ON THE GPU BOX: A/B of the Euclidean G1 kernels' global-memory layout
# (MMS_EUCLID_LAYOUT_{FWD,BWD,FUSED} = pair: row-aligned | block: workgroup-dense) and waves per workgroup,
# via tools/bin/membench (tools/membench.hip linked against the library).
for cfg in "pair pair 8 8" "pair block 8 8" "block block 8 8" "block block 8 4" "pair block 8 4" "block block 4 4" "pair block 4 4" "block block 16 16"; do
  set -- $cfg
  echo "=== forward $1 x $3 waves, backward $2 x $4 waves"
  MMS_EUCLID_LAYOUT_FWD=$1 MMS_EUCLID_LAYOUT_BWD=$2 MMS_PAIR32_WPB_FWD=$3 MMS_PAIR32_WPB_BWD=$4 timeout -k 10 120 tools/bin/membench | grep -E "mms fwd  |mms bwd|SEQ mms" || exit 1
done
