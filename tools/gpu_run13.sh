#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run13
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
