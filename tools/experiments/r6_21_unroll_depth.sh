#!/bin/bash
# round 6, last session: four against six unrolled exponent decisions (FUIF_FS_UNROLL_K = 5 / 7), alternating on one box; the asm of the deeper build against its
# specification first
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_21
mkdir -p $OUT
cd $ROOT
export FUIF_BENCH_CACHE=/tmp/fuif_bench_cache
timeout 300 build/test_fast_symbol_k7 400000 2>&1 | tail -14 | tee $OUT/unit.txt
for v in unroll unroll7 unroll unroll7 r6head; do
  FUIF_AMD_LIB=$ROOT/build/libfuifgpu_$v.so timeout 300 python tools/time_decode.py 1024 --reps 2 2>&1 | grep -v amdgpu | tee -a $OUT/variants.txt
done
