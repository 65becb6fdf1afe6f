#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run31
mkdir -p $OUT
cd $ROOT
timeout 200 python tools/tile_timeline.py 128 3840 2160 > $OUT/timeline128.txt 2>&1
grep "^launch\|^c4[0-9]\|^c5[0-9]\|^c60\|^total\|^scheduler\|per-SIMD\|t=" $OUT/timeline128.txt | head -60
