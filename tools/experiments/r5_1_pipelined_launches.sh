#!/bin/bash
# PREPARED in round 4 for the FIRST GPU session of round 5 (profiles/r4_pipelined_launches.txt measured the effect at 128 pictures):
# consecutive 1024-picture launches overlapped on two HIP streams.  Sequential vs pipelined, entropy only and with the inverse transforms behind every launch,
# with the second launch queued 3.6 s after the first (the first one's busy phase) and queued at once.
#   gpurun --timeout 900 -- bash tools/experiments/r5_1_pipelined_launches.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r5_pipe
mkdir -p $OUT
timeout 300 python tools/pipeline_decode.py 1024 --launches 4 --stagger 3.6 --rounds 1 2>&1 | grep -v amdgpu | tee $OUT/entropy_stagger.txt
timeout 300 python tools/pipeline_decode.py 1024 --launches 4 --stagger 0 --rounds 1 --only-pipelined 2>&1 | grep -v amdgpu | tee $OUT/entropy_at_once.txt
timeout 300 python tools/pipeline_decode.py 1024 --launches 4 --stagger 3.6 --rounds 1 --with-transforms 2>&1 | grep -v amdgpu | tee $OUT/with_transforms.txt
