#!/bin/bash
# Round 6: BASELINE config C4 at full size on the final library (256 x 8192x8192x4 in ONE launch) and the small batches of VERDICT r5 weak 8 (1 / 16 / 128 pictures of 4K)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r6_12
mkdir -p $OUT
export FUIF_BENCH_CACHE=/tmp/fuif_bench_cache
for n in 1 16 128; do timeout 300 python tools/time_decode.py $n --reps 2 --check 2>&1 | grep -v amdgpu | tee -a $OUT/small_batches.txt; done
(time FUIFGPU_CTX_MB=32 timeout 900 python bench.py --workload c4 --width 8192 --height 8192 --batch 256 --chunk -1 --distinct 2 --steps 1 --warmup 0) > $OUT/bench_c4.json 2> $OUT/bench_c4.err
tail -c 2500 $OUT/bench_c4.json; grep -v "File\|^    \|amdgpu" $OUT/bench_c4.err | tail -5
