#!/bin/bash
# round 6, last session: do the leaves (random 32 / 64-byte records, ~100 KB per long group, little reuse in L2) push the hot supernodes out of L2?  The leaf fetch and
# its write-back with the non-temporal hint (`nt`: stream through the caches), both / store only / load only, against HEAD's library
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_26
mkdir -p $OUT
cd $ROOT
export FUIF_BENCH_CACHE=/tmp/fuif_bench_cache
for v in cand2b leafnt leafnts leafntl cand2b leafnt leafnts leafntl; do
  FUIF_AMD_LIB=$ROOT/build/libfuifgpu_$v.so timeout 300 python tools/time_decode.py 1024 --reps 2 --check 2>&1 | grep -v amdgpu | tee -a $OUT/variants.txt
done
