#!/bin/bash
# Round 5: the two raw-4:2:0 fixtures and the fused colour kernel's edge cases on hardware (+ the parity tests that walk over every fixture).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r5_12
mkdir -p $OUT
(time timeout 600 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_parity.py tests/test_gpu_synthetic.py tests/test_gpu_group_parallel.py) > $OUT/tests.txt 2>&1
tail -n 5 $OUT/tests.txt
