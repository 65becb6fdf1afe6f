#!/bin/bash
# Round 4, session 13: long tiles spread evenly over the SIMDs (FUIFGPU_LONG_PER_SIMD, default 3; 0 = off = the behaviour so far):
# parity of the group-parallel paths, then A/B on the 1024 x 4K launch.   gpurun --timeout 1200 -- bash tools/experiments/r4_13_long_per_simd.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r4_lps
mkdir -p $OUT
(time timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_parity.py tests/test_gpu_group_parallel.py tests/test_gpu_synthetic.py) > $OUT/parity.txt 2>&1; tail -4 $OUT/parity.txt
if ! grep -q " passed" $OUT/parity.txt || grep -q "failed\|error" $OUT/parity.txt; then echo "PARITY NOT GREEN: no timing"; exit 1; fi
{
for v in 0 3 0 3 4 2; do
  echo "== FUIFGPU_LONG_PER_SIMD=$v"
  FUIFGPU_LONG_PER_SIMD=$v timeout 200 python tools/time_decode.py 1024 3840 2160 --reps 2 --check
done
} 2>&1 | grep -v amdgpu | tee $OUT/times.txt
