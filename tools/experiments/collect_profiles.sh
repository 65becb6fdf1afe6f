#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC HBM traffic of the bench.
# Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries into profiles/ afterwards.
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-seq-compare"
# 1) per-kernel time (same command shape as the default bench run)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH --steps 2 --warmup 1 > $OUT/bench_trace.json 2> $OUT/bench_trace.err
# 2) HBM traffic counters, one PMC pass each (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH --steps 1 --warmup 0 > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH --steps 1 --warmup 0 > $OUT/bench_write.json 2> $OUT/bench_write.err
find $OUT -name "*.csv" | head -40
python $ROOT/tools/experiments/summarize_profiles.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
