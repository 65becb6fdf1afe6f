#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run10
mkdir -p $OUT
cd $ROOT
for v in "" _w5 _w6 _w8; do
  case "$v" in "") n=4096;; _w5) n=5120;; _w6) n=6144;; _w8) n=8192;; esac
  echo "=== libfuifgpu$v"
  FUIF_AMD_LIB=$ROOT/fuif_amd/libfuifgpu$v.so REPS=1 timeout 300 python tools/occupancy_probe.py $n 1280 720 seq 2>&1 | grep -v amdgpu
  FUIF_AMD_LIB=$ROOT/fuif_amd/libfuifgpu$v.so timeout 300 python tools/tile_timeline.py 1024 3840 2160 2>&1 | grep "launch\|scheduler: busy"
done
