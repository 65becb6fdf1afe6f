#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run4
mkdir -p $OUT
cd $ROOT
for v in w8 w6 w5; do
  FUIF_AMD_LIB=$ROOT/fuif_amd/libfuifgpu_$v.so timeout 300 python tools/tile_timeline.py 1024 3840 2160 > $OUT/timeline_$v.txt 2>&1
  echo "=== $v"; grep -v amdgpu $OUT/timeline_$v.txt | head -3; grep -A3 "^c5[4-9]\|^c60\|total tile\|per-SIMD" $OUT/timeline_$v.txt | grep "^c5\|^c60\|total\|per-SIMD"
done
FUIF_AMD_LIB=$ROOT/fuif_amd/libfuifgpu_w8.so FUIFGPU_TILE_ORDER=group REPS=1 timeout 200 python tools/occupancy_probe.py 1024 3840 2160 2>&1 | grep -v amdgpu
