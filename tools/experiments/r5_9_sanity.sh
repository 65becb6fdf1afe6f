#!/bin/bash
# Round 5, after the last edits (upload-pipeline leg capped at 4 steps, boundary threads share a device's memory budget): the boundary tests and a short default line.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r5_9
mkdir -p $OUT
(time timeout 600 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_boundary_cli.py tests/test_gpu_rccl_world1.py) > $OUT/tests.txt 2>&1
tail -n 4 $OUT/tests.txt
(time timeout 900 python bench.py --steps 4 --warmup 1) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.json; grep real $OUT/bench_default.err
