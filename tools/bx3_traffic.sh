#!/bin/bash
# tools/bx3_traffic.sh -- dev-only, ON THE GPU BOX: HBM bytes per launch of cfg 3's kernels on both matrix pipes
# (FETCH_SIZE and WRITE_SIZE in separate --pmc passes with --kernel-trace only; corrections as tools/pmc_traffic.py).
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3/bx3_traffic; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $ROOT/tools/cfg3_pipes_probe.py > $OUT/$c.out 2> $OUT/$c.err || { echo "pass $c failed:"; tail -3 $OUT/$c.err; }
done
cd $ROOT
python3 - <<'PY' | tee gpurun_out/r3/bx3_traffic/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob('gpurun_out/r3/bx3_traffic/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'][:60], r['Counter_Name'])
        acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
ks = sorted({k for k, _ in acc})
print("HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: FETCH_SIZE counts half of a wide streaming read on gfx950; KiB units)")
for k in ks:
    if 'bx3' not in k and 'panel_gemm' not in k and 'splitk' not in k: continue
    f = acc.get((k, 'FETCH_SIZE'), [0, 1]); w = acc.get((k, 'WRITE_SIZE'), [0, 1])
    fb, wb = f[0] / max(f[1], 1) * 1024, w[0] / max(w[1], 1) * 1024
    print(f"{k:60s} FETCH_SIZE {fb/1e6:8.2f} MB (x2 = {2*fb/1e6:8.2f})  WRITE_SIZE {wb/1e6:8.2f} MB  -> {(2*fb+wb)/1e6:8.2f} MB per launch  (n={f[1]})")
PY
