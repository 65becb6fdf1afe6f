#!/bin/bash
# Run on the GPU box (via gpurun): the rocprofv3 evidence of round 3 for the default bench workload (C2, 1024 x 4K).
#   1. kernel-trace + stats of `bench.py --steps 2 --warmup 1` (average duration of k_maniac_decode must agree with bench.py's HIP events)
#   2. FETCH_SIZE and WRITE_SIZE of the same launch, one --pmc pass each (they do not fit one pass; no trace domains next to --pmc)
#   3. the same counters on known byte counts in the kernel's TWO access patterns (tools/ubench_gather.hip: 64-byte leaf records read /
#      written as 2 bytes x 32 lanes; 512-byte supernodes read as 8 bytes x 64 lanes): one calibration factor per pattern
# Outputs: gpurun_out/prof_r3/ ; tools/experiments/summarize_profiles_r3.py condenses them into the files copied to profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-seq-compare --no-h2d"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH --steps 2 --warmup 1 > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH --steps 1 --warmup 0 > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH --steps 1 --warmup 0 > $OUT/bench_write.json 2> $OUT/bench_write.err
G=$ROOT/build/ubench_gather_bin
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_read_fetch -- $G read 20000 > $OUT/cal_read.json 2> $OUT/cal_read.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_snode_fetch -- $G snode 20000 > $OUT/cal_snode.json 2> $OUT/cal_snode.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write_write -- $G write 20000 > $OUT/cal_write.json 2> $OUT/cal_write.err
python $ROOT/tools/experiments/summarize_profiles_r3.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
