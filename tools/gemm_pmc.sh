#!/bin/bash
# tools/gemm_pmc.sh -- dev-only, ON THE GPU BOX: HBM/L2 counters of the cfg 3 GEMMs (own --pmc passes).
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/gemm_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$tag -- python3 $ROOT/tools/gemm_probe.py 10 > $OUT/$tag.out 2> $OUT/$tag.err || { tail -3 $OUT/$tag.err; continue; }
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/gemm_pmc/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'][:70], r['Counter_Name'])
        acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
    for (kn, cn), (v, n) in sorted(acc.items()):
        if 'mms' in kn: print(f"{cn:24s} {v/n:14.1f} per launch  n={n:3d}  {kn}")
PY
