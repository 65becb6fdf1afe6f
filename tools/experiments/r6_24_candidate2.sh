#!/bin/bash
# round 6, last session: candidate 2 (window position carried through the chunk, masks with one instruction less) against candidate 1 (HEAD's library)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_24
mkdir -p $OUT
cd $ROOT
export FUIF_BENCH_CACHE=/tmp/fuif_bench_cache
timeout 300 build/test_fast_symbol_cand2 400000 2>&1 | tail -3 | tee -a $OUT/unit.txt
for v in cand2 cand cand2 cand; do
  FUIF_AMD_LIB=$ROOT/build/libfuifgpu_$v.so timeout 300 python tools/time_decode.py 1024 --reps 2 --check 2>&1 | grep -v amdgpu | tee -a $OUT/variants.txt
done
