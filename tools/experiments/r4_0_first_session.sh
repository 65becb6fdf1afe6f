#!/bin/bash
# First GPU session of round 4 (prepared at the end of round 3, when the GPU minutes were gone): what round 3 could only check on
# the wavefront emulator, plus the round's opening numbers.   gpurun --timeout 900 -- bash tools/experiments/r4_0_first_session.sh
#   1. the whole GPU suite (133 tests at the end of round 3; the batch-encode and sibling-lifetime cases have not run on hardware)
#   2. the writer: host / GPU one group at a time / GPU batch, same bytes (tools/time_encode.py), 1080p and 4K
#   3. the default bench line (about 2.5 minutes)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r4_first
mkdir -p $OUT
(time timeout 400 python -m pytest tests -m gpu -x -q) > $OUT/gpu_tests.txt 2>&1; tail -4 $OUT/gpu_tests.txt
{ timeout 200 python tools/time_encode.py 16 1920 1080; timeout 300 python tools/time_encode.py 8 3840 2160; } 2>&1 | grep -v amdgpu | tee $OUT/time_encode.txt
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json
