#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run8
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout 300 python tools/tile_timeline.py 1024 3840 2160 2>&1 | grep "launch\|^c5[49]\|^c60\|total tile\|scheduler\|per-SIMD"
REPS=1 timeout 300 python tools/occupancy_probe.py 1024,4096 1920 1080 seq 2>&1 | grep -v amdgpu
