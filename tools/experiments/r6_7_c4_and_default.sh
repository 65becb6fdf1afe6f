#!/bin/bash
# round 6, session 7: the default build after the removal of the LDS-resident supernodes of the dense configuration; C4's launch on the wide configuration
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_7
mkdir -p $OUT
cd $ROOT
export FUIF_BENCH_CACHE=/tmp/fuif_bench_cache
timeout 300 python tools/time_decode.py 1024 --reps 3 --check 2>&1 | grep -v amdgpu | tee $OUT/time_1024_indexed.txt
timeout 1200 python tools/experiments/r6_7_c4_wide.py 8 8192 2>&1 | grep -v amdgpu | tee $OUT/c4_wide.txt
