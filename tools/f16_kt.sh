#!/bin/bash
# tools/f16_kt.sh -- dev-only, ON THE GPU BOX: rocprofv3 --kernel-trace --stats of cfg 5's shard (tools/f16_probe.py) in its
# three forms (ordered lane walk, ordered quad chain of round 2, tree sum) and of the cosine fp16 kernel.
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3/f16_kt; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lane -- python3 $ROOT/tools/f16_probe.py > $OUT/lane.out 2> $OUT/lane.err || tail -3 $OUT/lane.err
MMS_F16_CHAIN=quad rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/quad -- python3 $ROOT/tools/f16_probe.py > $OUT/quad.out 2> $OUT/quad.err || tail -3 $OUT/quad.err
MMS_F16_TREE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tree -- python3 $ROOT/tools/f16_probe.py > $OUT/tree.out 2> $OUT/tree.err || tail -3 $OUT/tree.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cos -- python3 $ROOT/tools/cos16_probe.py > $OUT/cos.out 2> $OUT/cos.err || tail -3 $OUT/cos.err
cd $ROOT
python3 - <<'PY' | tee gpurun_out/r3/f16_kt/summary.txt
import csv, glob
for tag in ("lane", "quad", "tree", "cos"):
    for f in glob.glob('gpurun_out/r3/f16_kt/%s/**/*kernel_stats.csv' % tag, recursive=True):
        for r in list(csv.DictReader(open(f)))[:3]:
            if 'mms' in r['Name']:
                print("%-5s %-110s calls %5s avg %8.2f us  min %8.2f  max %8.2f" % (tag, r['Name'][:110], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
