#!/bin/bash
# round 6, session 8: C4's launch (8 and 64 pictures of 8192x8192x4) on the dense and on the wide configurations
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_8
mkdir -p $OUT
cd $ROOT
export FUIF_BENCH_CACHE=/tmp/fuif_bench_cache
timeout 900 python tools/experiments/r6_7_c4_wide.py 8 8192 2>&1 | grep -v amdgpu | tee $OUT/c4_wide_8.txt
timeout 1200 python tools/experiments/r6_7_c4_wide.py 64 8192 2>&1 | grep -v amdgpu | tee $OUT/c4_wide_64.txt
