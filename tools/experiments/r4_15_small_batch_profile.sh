#!/bin/bash
# Round 4, session 15: what bounds a SMALL batch -- one picture's chain of long channel groups -- with no contention:
# per-channel phase cycles at 64 pictures (the per-channel profile needs one row per channel: >= 61 pictures; 192 long tiles on 1024 SIMDs),
# and the tile timeline (start / end / running / waiting per group) at 16 pictures and at one.
#   gpurun --timeout 300 -- bash tools/experiments/r4_15_small_batch_profile.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r4_small
mkdir -p $OUT
FUIF_AMD_LIB=$ROOT/build/libfuifgpu_profch.so timeout 120 python tools/prof_by_channel.py 64 3840 2160 2>&1 | grep -v amdgpu | tee $OUT/by_channel_64.txt
FUIF_AMD_LIB=$ROOT/build/libfuifgpu_tilelog.so timeout 60 python tools/tile_timeline.py 16 3840 2160 2>&1 | grep -v amdgpu | tee $OUT/timeline_16.txt | head -70
FUIF_AMD_LIB=$ROOT/build/libfuifgpu_tilelog.so timeout 60 python tools/tile_timeline.py 1 3840 2160 2>&1 | grep -v amdgpu | tee $OUT/timeline_1.txt | head -70
