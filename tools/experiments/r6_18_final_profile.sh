#!/bin/bash
# round 6, session 18: the final kernel's per-phase cycles per channel group (-DFUIF_PROF -DFUIF_PROF_BY_CHANNEL build) and the tile timeline of one 1024 x 4K launch (-DFUIF_TILELOG build)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_18
mkdir -p $OUT
cd $ROOT
export FUIF_BENCH_CACHE=/tmp/fuif_bench_cache
FUIF_AMD_LIB=$ROOT/build/libfuifgpu_profch.so timeout 600 python tools/prof_by_channel.py 1024 2>&1 | grep -v amdgpu | tee $OUT/phases_by_channel_1024.txt
FUIF_AMD_LIB=$ROOT/build/libfuifgpu_tilelog.so timeout 600 python tools/tile_timeline.py 1024 2>&1 | grep -v amdgpu | tee $OUT/tile_timeline_1024.txt | head -60
