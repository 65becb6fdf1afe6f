#!/bin/bash
# round 2: 5 wavefronts per SIMD (96 VGPRs) against the shipped 6, with priorities and balanced queues
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run26
mkdir -p $OUT
cd $ROOT
for v in _w5 ""; do
  echo "=== libfuifgpu$v"
  FUIF_AMD_LIB=$ROOT/fuif_amd/libfuifgpu$v.so timeout 300 python tools/tile_timeline.py 1024 3840 2160 > $OUT/timeline$v.txt 2>&1
  grep "^launch\|^c54\|^c59\|^c60\|^scheduler\|Error" $OUT/timeline$v.txt
done
