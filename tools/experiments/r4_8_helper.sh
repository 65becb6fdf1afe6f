#!/bin/bash
# Round 4, session 8: the helper wavefront of the one-tile-per-image configuration (streams without a group index): parity of the
# non-indexed paths first, then the 1024 x 4K no-index launch with the helper off / on and a few settings of
# FUIFGPU_HELPER=exits,leaves,sleep.   gpurun --timeout 1500 -- bash tools/experiments/r4_8_helper.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r4_helper
mkdir -p $OUT
(time timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_parity.py tests/test_gpu_synthetic.py tests/test_fuzz.py) > $OUT/parity.txt 2>&1; tail -4 $OUT/parity.txt
if ! grep -q " passed" $OUT/parity.txt || grep -q "failed\|error" $OUT/parity.txt; then echo "PARITY NOT GREEN: no timing"; exit 1; fi
{
for h in 0 2,2,0 4,2,0 2,2,8 4,4,0 0; do
  echo "== FUIFGPU_HELPER=$h"
  FUIFGPU_HELPER=$h timeout 200 python tools/time_decode.py 1024 3840 2160 --no-index --reps 1 --check
done
} 2>&1 | grep -v amdgpu | tee $OUT/times.txt
