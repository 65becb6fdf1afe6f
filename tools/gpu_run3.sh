#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run3
mkdir -p $OUT
cd $ROOT
timeout 300 python tools/tile_timeline.py 1024 3840 2160 > $OUT/timeline_simdq.txt 2>&1
FUIFGPU_TILE_ORDER=group timeout 300 python tools/tile_timeline.py 1024 3840 2160 > $OUT/timeline_group.txt 2>&1
grep -hv amdgpu $OUT/timeline_simdq.txt $OUT/timeline_group.txt
