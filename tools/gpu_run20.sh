#!/bin/bash
# round 2 experiment: priority-0 (short) tiles sleep after every chunk -- does giving the long tiles more of the CU shorten the launch?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run20
mkdir -p $OUT
cd $ROOT
for n in 0 3 8 16; do
  echo "=== FUIFGPU_SHORT_NAPS=$n"
  FUIFGPU_SHORT_NAPS=$n timeout 300 python tools/tile_timeline.py 1024 3840 2160 > $OUT/naps$n.txt 2>&1
  grep "^launch\|^c54\|^c59\|^c60\|^scheduler\|per-SIMD" $OUT/naps$n.txt
done
