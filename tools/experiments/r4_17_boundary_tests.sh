#!/bin/bash
# Round 4, session 17: tests/test_boundary_cli.py on hardware after fuif_encode_file was bound (FUIFGPU_WRITE_INDEX).
#   gpurun --timeout 120 -- bash tools/experiments/r4_17_boundary_tests.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r4_boundary
(time timeout 100 python -m pytest -m gpu -x -q --durations=4 tests/test_boundary_cli.py) > gpurun_out/r4_boundary/tests.txt 2>&1; tail -12 gpurun_out/r4_boundary/tests.txt
