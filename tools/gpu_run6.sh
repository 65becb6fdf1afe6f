#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run6
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_group_parallel.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
TILE_LOG_NPY=$OUT/tilelog.npy timeout 300 python tools/tile_timeline.py 1024 3840 2160 > $OUT/timeline_ctx.txt 2>&1
grep -v "amdgpu\|^c[0-3]" $OUT/timeline_ctx.txt
