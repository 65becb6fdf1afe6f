#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run2
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
REPS=1 timeout 200 python tools/occupancy_probe.py 1024 3840 2160 > $OUT/simdq.txt 2>&1
FUIFGPU_TILE_ORDER=group REPS=1 timeout 200 python tools/occupancy_probe.py 1024 3840 2160 > $OUT/groupq.txt 2>&1
REPS=1 timeout 200 python tools/occupancy_probe.py 256,2048 1920 1080 > $OUT/simdq_small.txt 2>&1
grep -hv amdgpu $OUT/simdq.txt $OUT/groupq.txt $OUT/simdq_small.txt
