#!/bin/bash
# round 6, session 9: files without group index (narrow groups keep twice as many supernodes in LDS) alone and two side by side; 7 wavefronts per SIMD
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_9
mkdir -p $OUT
cd $ROOT
export FUIF_BENCH_CACHE=/tmp/fuif_bench_cache
timeout 300 python tools/time_decode.py 1024 --reps 2 2>&1 | grep -v amdgpu | tee $OUT/variants.txt
for v in w7c16 w6c16; do
  FUIF_AMD_LIB=$ROOT/build/libfuifgpu_$v.so timeout 300 python tools/time_decode.py 1024 --reps 2 2>&1 | grep -v amdgpu | tee -a $OUT/variants.txt
done
timeout 300 python tools/time_decode.py 1024 --no-index --reps 1 --check 2>&1 | grep -v amdgpu | tee $OUT/noindex.txt
timeout 600 python tools/pipeline_decode.py 1024 --launches 2 --stagger 0 --no-index --only-pipelined 2>&1 | grep -v amdgpu | tee -a $OUT/noindex.txt
