#!/bin/bash
# round 2: dense build without the LDS supernode probe; variant with the next pixel's property row prefetched
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run23
mkdir -p $OUT
cd $ROOT
for v in "" _pf; do
  echo "=== libfuifgpu$v"
  FUIF_AMD_LIB=$ROOT/fuif_amd/libfuifgpu$v.so timeout 300 python tools/tile_timeline.py 1024 3840 2160 > $OUT/timeline$v.txt 2>&1
  grep "^launch\|^c54\|^c59\|^c60\|^scheduler\|Error" $OUT/timeline$v.txt
done
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
