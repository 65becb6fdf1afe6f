#!/bin/bash
# round 3, session: (a) SQ counters of the SHIPPED configuration (6 wavefronts per SIMD, context scheduler, priorities) on a launch in
# which every wavefront slot is busy: 1024 x 1920x1080 with the group index (the first regime of the headline launch);
# (b) the row slack a suspended tile waits for before it is resumed (FUIFGPU_YIELD_SLACK), 1024 x 4K.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/pmc_sq.sh r3_dense 1024 1920 1080 groups
mkdir -p gpurun_out/r3_final
for s in 2 8 32; do FUIFGPU_YIELD_SLACK=$s timeout 400 python tools/time_decode.py 1024 3840 2160 --reps 2 | sed "s/^/yield_slack $s: /"; done 2>&1 | grep -v amdgpu | tee gpurun_out/r3_final/yield_slack.txt
