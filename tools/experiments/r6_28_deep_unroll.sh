#!/bin/bash
# round 6, last session: chances 6..9 unrolled too, each behind its own existence test, their exits sharing one mantissa ladder (FUIF_FS_DEEP) against HEAD's
# library: the headline launch, C3's, and C4's (64 pictures of 8192x8192x4, 14 bit: the exponents that run past chance 5)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_28
mkdir -p $OUT
cd $ROOT
export FUIF_BENCH_CACHE=/tmp/fuif_bench_cache
timeout 300 build/test_fast_symbol_deep 400000 2>&1 | tail -3 | tee -a $OUT/unit.txt
for v in deep cand2b deep cand2b deep cand2b; do
  FUIF_AMD_LIB=$ROOT/build/libfuifgpu_$v.so timeout 300 python tools/time_decode.py 1024 --reps 2 --check 2>&1 | grep -v amdgpu | tee -a $OUT/variants.txt
done
for v in deep cand2b; do
  FUIF_AMD_LIB=$ROOT/build/libfuifgpu_$v.so timeout 300 python tools/time_decode.py 1024 --dct420 --reps 2 2>&1 | grep -v amdgpu | tee -a $OUT/variants.txt
done
C4_LIBS=build/libfuifgpu_deep.so,build/libfuifgpu_cand2b.so,build/libfuifgpu_deep.so,build/libfuifgpu_cand2b.so timeout 1200 python tools/experiments/r6_13_c4_ab.py 64 2>&1 | grep -v amdgpu | tee $OUT/c4_64.txt
