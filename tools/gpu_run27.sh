#!/bin/bash
# round 2: rocprofv3 kernel statistics of the JPEG-transcode workload (C3), asked for by the round-1 review
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run27
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-seq-compare --no-h2d > $OUT/bench_c3.json 2> $OUT/bench_c3.err
echo "rc=$?"; grep "^{" $OUT/bench_c3.json | cut -c1-600
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-200
cp "$f" $OUT/c3_kernel_stats.csv
