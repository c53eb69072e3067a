#!/bin/bash
# tools/panel_pmc.sh -- dev-only, ON THE GPU BOX: MFMA-pipe and wave-state counters of cfg 3's panel GEMM launches
# (own --pmc passes, --kernel-trace only), averaged per launch per kernel.  Output: gpurun_out/panel_pmc/summary.txt
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/panel_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$tag -- python3 $ROOT/tools/gemm_probe.py 10 > $OUT/$tag.out 2> $OUT/$tag.err || { echo "pass $tag failed:"; tail -3 $OUT/$tag.err; continue; }
done
cd $ROOT
python3 - <<'PY' | tee gpurun_out/panel_pmc/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob('gpurun_out/panel_pmc/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'][:60], r['Counter_Name'])
        acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
for (kn, cn), (v, n) in sorted(acc.items()):
    if 'mms' in kn: print(f"{kn:60s} {cn:30s} {v/n:16.1f} per launch  n={n}")
PY
