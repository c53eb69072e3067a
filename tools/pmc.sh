#!/bin/bash
# tools/pmc.sh TAG "COUNTERS..." CMD... -- dev-only, ON THE GPU BOX: one rocprofv3 --pmc pass (with --kernel-trace
# only), per-kernel averages of every counter.
set -o pipefail
TAG=$1; CNT=$2; shift; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT -- "$@" > $OUT/run.out 2> $OUT/run.err || { tail -5 $OUT/run.err; exit 1; }
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'][:60], r['Counter_Name'])
        acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
    for (kn, cn), (v, n) in sorted(acc.items()):
        if 'mms' in kn: print(f"{cn:28s} {v/n:16.1f} per launch  n={n:3d}  {kn}")
PY
