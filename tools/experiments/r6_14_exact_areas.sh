#!/bin/bash
# round 6, session 14: context areas of the exact size (built in the scratch area, then moved): the headline launch, C4 at full size against round 5's library, the GPU suite
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_14
mkdir -p $OUT
cd $ROOT
export FUIF_BENCH_CACHE=/tmp/fuif_bench_cache
timeout 300 python tools/time_decode.py 1024 --reps 3 --check 2>&1 | grep -v amdgpu | tee $OUT/time_1024_indexed.txt
timeout 1500 python tools/experiments/r6_13_c4_ab.py 256 2>&1 | grep -v amdgpu | tee $OUT/c4_ab.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -16 | tee $OUT/gpu_tests.txt
