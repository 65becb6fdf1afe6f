#!/bin/bash
# round 6, session 4: per-phase cycles per channel group (narrow supernodes), the whole launch and the three long groups alone
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_4
mkdir -p $OUT
cd $ROOT
export FUIF_BENCH_CACHE=/tmp/fuif_bench_cache
FUIF_AMD_LIB=$ROOT/build/libfuifgpu_profch.so timeout 600 python tools/prof_by_channel.py 1024 2>&1 | grep -v amdgpu > $OUT/phases_by_channel_1024.txt; cat $OUT/phases_by_channel_1024.txt
FUIF_AMD_LIB=$ROOT/build/libfuifgpu_profch.so timeout 600 python tools/prof_by_channel.py 8 2>&1 | grep -v amdgpu > $OUT/phases_by_channel_8.txt; cat $OUT/phases_by_channel_8.txt
