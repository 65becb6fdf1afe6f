#!/bin/bash
# Round 4, session 16: the boundary front ends added after session 14 (fuif_gpu_index, fuif_amd.add_group_index) and the sharded parity tests on hardware.
#   gpurun --timeout 200 -- bash tools/experiments/r4_16_index_tool_tests.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r4_index
mkdir -p $OUT
(time timeout 150 python -m pytest -m gpu -x -q --durations=5 tests/test_boundary_cli.py tests/test_gpu_group_parallel.py) > $OUT/tests.txt 2>&1; tail -14 $OUT/tests.txt
