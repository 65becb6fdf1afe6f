#!/bin/bash
# round 2: wavefront priority by tile size class, 8 wavefronts per SIMD, streams with the writer's new tree rule
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run17
mkdir -p $OUT
cd $ROOT
run() {  # name lib prio
  echo "=== $1"
  FUIF_AMD_LIB=$ROOT/fuif_amd/$2 FUIFGPU_PRIO_BASE=$3 timeout 300 python tools/tile_timeline.py 1024 3840 2160 > $OUT/$1.txt 2>&1
  grep "^launch\|^c5[4-9]\|^c60\|^total tile-time\|^scheduler\|per-SIMD" $OUT/$1.txt
}
run w6_prio_off libfuifgpu.so -1
run w6_prio2 libfuifgpu.so 2
run w6_prio3 libfuifgpu.so 3
run w8_prio_off libfuifgpu_w8.so -1
run w8_prio2 libfuifgpu_w8.so 2
echo "=== one wavefront per image"
REPS=1 timeout 300 python tools/occupancy_probe.py 1024 3840 2160 seq 2>&1 | grep -v amdgpu | tee $OUT/seq.txt
