#!/bin/bash
# Round 4, session 12: FUIFGPU_YIELD_SLACK (rows a producer must be ahead before a suspended tile is resumed) below the default of 8:
# session 11 measured 4 -> 7.29 s, 8 -> 7.38 s, 16 -> 7.60 s, 32 -> 7.83 s.   gpurun --timeout 900 -- bash tools/experiments/r4_12_slack.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r4_slack
mkdir -p $OUT
{
for v in 8 4 2 1 0 3 6 4; do
  echo "== FUIFGPU_YIELD_SLACK=$v"
  FUIFGPU_YIELD_SLACK=$v timeout 200 python tools/time_decode.py 1024 3840 2160 --reps 2 --check
done
echo "== 128 pictures"
for v in 8 2; do FUIFGPU_YIELD_SLACK=$v timeout 200 python tools/time_decode.py 128 3840 2160 --reps 2; done
} 2>&1 | grep -v amdgpu | tee $OUT/times.txt
