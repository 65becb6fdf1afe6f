#!/bin/bash
# round 6, last session: code size of the unrolled decoder -- renormalisation stubs that loop for the second byte (12 instructions instead of 20) with four (K=5)
# and three (K=4) unrolled exponent decisions, against the first unrolled build; each asm against its specification first
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_22
mkdir -p $OUT
cd $ROOT
export FUIF_BENCH_CACHE=/tmp/fuif_bench_cache
for t in unroll5L unroll4L; do timeout 300 build/test_fast_symbol_$t 400000 2>&1 | tail -3 | sed "s/^/$t: /" | tee -a $OUT/unit.txt; done
for v in unroll unroll5L unroll4L unroll unroll5L unroll4L; do
  FUIF_AMD_LIB=$ROOT/build/libfuifgpu_$v.so timeout 300 python tools/time_decode.py 1024 --reps 2 2>&1 | grep -v amdgpu | tee -a $OUT/variants.txt
done
