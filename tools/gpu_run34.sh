#!/bin/bash
# round 2, HEAD: GPU test tier
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/run34
cd $ROOT
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/run34/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/run34/pytest.log
