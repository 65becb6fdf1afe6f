#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r4_int16
mkdir -p $OUT
FUIFGPU_CTX_MB=32 timeout 900 python bench.py --workload c4 --width 8192 --height 8192 --batch 256 --chunk -1 --distinct 2 --steps 1 --warmup 0 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; tail -c 2500 $OUT/bench_c4.json; grep -v "File\|^    " $OUT/bench_c4.err | tail -5
