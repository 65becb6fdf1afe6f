#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run7
mkdir -p $OUT
cd $ROOT
for sl in 8; do
  echo "=== slack $sl"
  FUIFGPU_YIELD_SLACK=$sl timeout 300 python tools/tile_timeline.py 1024 3840 2160 2>&1 | grep "launch\|^c5[49]\|^c60\|total tile\|scheduler\|per-SIMD"
done
