#!/bin/bash
# round 6, last session: the candidate decoder (four unrolled exponent decisions from ilast >= 4, looping renormalisation stubs, the zero decision's threshold kept
# in place) against the looped-stub build before it and HEAD's library; its asm against the specification first (with the ranges that end at chances 4 and 5)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_23
mkdir -p $OUT
cd $ROOT
export FUIF_BENCH_CACHE=/tmp/fuif_bench_cache
timeout 300 build/test_fast_symbol_cand 400000 2>&1 | tail -3 | tee -a $OUT/unit.txt
for v in cand unroll5L r6head cand unroll5L r6head; do
  FUIF_AMD_LIB=$ROOT/build/libfuifgpu_$v.so timeout 300 python tools/time_decode.py 1024 --reps 2 --check 2>&1 | grep -v amdgpu | tee -a $OUT/variants.txt
done
for v in cand r6head; do
  FUIF_AMD_LIB=$ROOT/build/libfuifgpu_$v.so timeout 300 python tools/time_decode.py 1024 --dct420 --reps 2 2>&1 | grep -v amdgpu | tee -a $OUT/variants.txt
  FUIF_AMD_LIB=$ROOT/build/libfuifgpu_$v.so timeout 300 python tools/time_decode.py 1024 --no-index --reps 1 2>&1 | grep -v amdgpu | tee -a $OUT/variants.txt
done
