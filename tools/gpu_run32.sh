#!/bin/bash
# round 2: idle wavefronts back off (1 / 4 / 32 naps between looks): a 128-picture launch and the headline launch
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run32
mkdir -p $OUT
cd $ROOT
timeout 200 python tools/tile_timeline.py 128 3840 2160 > $OUT/timeline128.txt 2>&1
grep "^launch\|^c54\|^c59\|^c60\|^scheduler" $OUT/timeline128.txt
timeout 200 python tools/tile_timeline.py 1024 3840 2160 > $OUT/timeline1024.txt 2>&1
grep "^launch\|^c54\|^c59\|^c60\|^scheduler\|per-SIMD" $OUT/timeline1024.txt
