#!/bin/bash
# round 6, last session: the symbol decoder with its first four exponent decisions unrolled (per-exit mantissa and masks) against the library of HEAD f285df8
# on the same box, alternating: the asm against its C++ specification first (400 000 random states), then the 1024 x 4K launch, then 1024 x 1080p
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_20
mkdir -p $OUT
cd $ROOT
export FUIF_BENCH_CACHE=/tmp/fuif_bench_cache
timeout 300 build/test_fast_symbol_new 400000 2>&1 | tail -14 | tee $OUT/unit.txt
for v in r6head unroll r6head unroll; do
  FUIF_AMD_LIB=$ROOT/build/libfuifgpu_$v.so timeout 300 python tools/time_decode.py 1024 --reps 2 2>&1 | grep -v amdgpu | tee -a $OUT/variants.txt
done
for v in r6head unroll; do
  FUIF_AMD_LIB=$ROOT/build/libfuifgpu_$v.so timeout 300 python tools/time_decode.py 1024 1920 1080 --reps 2 2>&1 | grep -v amdgpu | tee -a $OUT/variants.txt
done
