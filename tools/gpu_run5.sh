#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run5
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout 300 python tools/tile_timeline.py 1024 3840 2160 > $OUT/timeline_ctx.txt 2>&1
grep -v "amdgpu\|^c[0-3]" $OUT/timeline_ctx.txt
REPS=2 timeout 200 python tools/occupancy_probe.py 256,2048 1920 1080 2>&1 | grep -v amdgpu
