#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run12
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
REPS=1 timeout 300 python tools/occupancy_probe.py 1024,6144 1920 1080 seq 2>&1 | grep -v amdgpu
timeout 300 python tools/tile_timeline.py 1024 3840 2160 2>&1 | grep "launch\|scheduler"
FUIF_AMD_LIB=$ROOT/fuif_amd/libfuifgpu_prof.so timeout 300 python tools/prof_kernel.py 8 1920 1080 > $OUT/phases.txt 2>&1; grep -v amdgpu $OUT/phases.txt
