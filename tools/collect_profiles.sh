#!/bin/bash
# tools/collect_profiles.sh ROUND -- run ON THE GPU BOX (via gpurun) from the repo root.
# Collects, for the default bench.py workload:
#   1. rocprofv3 --kernel-trace --stats            (per-kernel durations)
#   2. rocprofv3 --pmc FETCH_SIZE                  (own pass: TCC slots, MI355X_MICROARCH.md)
#   3. rocprofv3 --pmc WRITE_SIZE                  (own pass)
# into gpurun_out/prof_<ROUND>/ and distils them with tools/pmc_traffic.py into
# gpurun_out/prof_<ROUND>/summary/ (copy that into profiles/ to have it judged).
# PMC passes use --kernel-trace only (no sys/hip/hsa tracing), as the pool requires.
set -o pipefail
R=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT/summary
ARGS="--steps 1024 --warmup 64 --repeats 3 --no-cpu-baseline --no-variants $2"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ROOT/bench.py $ARGS > $OUT/kt.out 2> $OUT/kt.err || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py $ARGS > $OUT/fetch.out 2> $OUT/fetch.err || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py $ARGS > $OUT/write.out 2> $OUT/write.err || exit 1
cd $ROOT
python3 tools/pmc_traffic.py $OUT $R
