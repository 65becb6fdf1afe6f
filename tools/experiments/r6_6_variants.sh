#!/bin/bash
# round 6, session 6: which of the round's changes pay in the WHOLE launch (all 6144 slots busy): the product build (11 LDS-resident narrow supernodes, compact
# leaves, reference loads issued together), the same without LDS-resident supernodes, 5 wavefronts per SIMD with 16 of them; per-phase cycles per channel group
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_6
mkdir -p $OUT
cd $ROOT
export FUIF_BENCH_CACHE=/tmp/fuif_bench_cache
for v in "" k0 w5k16 "" k0; do
  lib=$ROOT/fuif_amd/libfuifgpu.so; [ -n "$v" ] && lib=$ROOT/build/libfuifgpu_$v.so
  FUIF_AMD_LIB=$lib timeout 300 python tools/time_decode.py 1024 --reps 2 2>&1 | grep -v amdgpu | tee -a $OUT/variants.txt
done
FUIF_AMD_LIB=$ROOT/build/libfuifgpu_profch.so timeout 600 python tools/prof_by_channel.py 1024 2>&1 | grep -v amdgpu > $OUT/phases_by_channel_1024.txt; cat $OUT/phases_by_channel_1024.txt
