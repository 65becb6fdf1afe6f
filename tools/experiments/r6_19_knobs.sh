#!/bin/bash
# round 6, session 19: the scheduler's knobs re-measured on the final kernel (1024 x 4K, entropy launch alone): wavefront priorities, yield slack, long tiles per SIMD
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_19
mkdir -p $OUT
cd $ROOT
export FUIF_BENCH_CACHE=/tmp/fuif_bench_cache
run() { echo "## $1" | tee -a $OUT/knobs.txt; env $1 timeout 300 python tools/time_decode.py 1024 --reps 2 2>&1 | grep -v amdgpu | tee -a $OUT/knobs.txt; }
run "FUIFGPU_PRIO_BASE=2"
run "FUIFGPU_PRIO_BASE=3"
run "FUIFGPU_PRIO_BASE=4"
run "FUIFGPU_PRIO_BASE=-1"
run "FUIFGPU_YIELD_SLACK=2"
run "FUIFGPU_YIELD_SLACK=8"
run "FUIFGPU_LONG_PER_SIMD=4"
run "FUIFGPU_LONG_PER_SIMD=0"
run "FUIFGPU_PRIO_BASE=2"
