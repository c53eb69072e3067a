#!/bin/bash
# tools/kt.sh TAG CMD... -- dev-only, ON THE GPU BOX: rocprofv3 --kernel-trace --stats of CMD, prints the kernel stats table.
set -o pipefail
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3/kt_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- "$@" > $OUT/run.out 2> $OUT/run.err || { tail -5 $OUT/run.err; exit 1; }
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print("%-90s calls %5s avg %9.2f us  min %9.2f  max %9.2f  %5s%%" % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, r['Percentage']))
PY
