#!/bin/bash
# Round 5: BASELINE config C4 at full size on the final library (the path did not change since round 4: a record that it still runs, 256 x 8192x8192x4 in ONE launch).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r5_10
mkdir -p $OUT
(time FUIFGPU_CTX_MB=32 timeout 800 python bench.py --workload c4 --width 8192 --height 8192 --batch 256 --chunk -1 --distinct 2 --steps 1 --warmup 0) > $OUT/bench_c4.json 2> $OUT/bench_c4.err
tail -c 2500 $OUT/bench_c4.json; grep -v "File\|^    \|amdgpu" $OUT/bench_c4.err | tail -5
