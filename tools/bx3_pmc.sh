#!/bin/bash
# tools/bx3_pmc.sh -- dev-only, ON THE GPU BOX: kernel trace of cfg 3 on both matrix pipes (tools/cfg3_pipes_probe.py) and,
# in separate --pmc passes, the matrix-pipe / wave-state counters of its kernels, averaged per launch per kernel.
# Output: gpurun_out/r3/bx3_pmc/{kernel_stats.txt,summary.txt}
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3/bx3_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ROOT/tools/cfg3_pipes_probe.py > $OUT/kt.out 2> $OUT/kt.err || { tail -5 $OUT/kt.err; exit 1; }
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$tag -- python3 $ROOT/tools/cfg3_pipes_probe.py > $OUT/$tag.out 2> $OUT/$tag.err || { echo "pass $tag failed:"; tail -3 $OUT/$tag.err; continue; }
done
cd $ROOT
python3 - <<'PY' | tee gpurun_out/r3/bx3_pmc/summary.txt
import csv, glob, collections
for f in glob.glob('gpurun_out/r3/bx3_pmc/kt/**/*kernel_stats.csv', recursive=True):
    print("== rocprofv3 --kernel-trace --stats -- python3 tools/cfg3_pipes_probe.py")
    for r in list(csv.DictReader(open(f)))[:16]:
        print("%-100s calls %5s avg %9.2f us  min %9.2f  max %9.2f  %5s%%" % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, r['Percentage']))
acc = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob('gpurun_out/r3/bx3_pmc/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'][:70], r['Counter_Name'])
        acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
print("== separate --pmc passes, per launch")
for (kn, cn), (v, n) in sorted(acc.items()):
    if 'bx3' in kn or 'panel_gemm' in kn: print(f"{kn:70s} {cn:30s} {v/n:16.1f} per launch  n={n}")
PY
