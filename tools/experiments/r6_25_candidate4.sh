#!/bin/bash
# round 6, last session: candidate 4 (the sign as +1 / -1 in the unrolled part: value by one multiply, sign decision by one add; first mantissa decision of an
# exit computes its chance index and starts the bit collector itself) against HEAD's library (candidate 2)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_25
mkdir -p $OUT
cd $ROOT
export FUIF_BENCH_CACHE=/tmp/fuif_bench_cache
timeout 300 build/test_fast_symbol_cand4 400000 2>&1 | tail -3 | tee -a $OUT/unit.txt
for v in cand4 cand2b cand4 cand2b cand4 cand2b; do
  FUIF_AMD_LIB=$ROOT/build/libfuifgpu_$v.so timeout 300 python tools/time_decode.py 1024 --reps 2 --check 2>&1 | grep -v amdgpu | tee -a $OUT/variants.txt
done
