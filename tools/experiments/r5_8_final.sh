#!/bin/bash
# Round 5, last GPU session: the library at HEAD -- the whole -m gpu suite, smoke(), and the driver's own bench command.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r5_8
mkdir -p $OUT
(time timeout 1100 python -m pytest tests/ -q -m gpu -p no:cacheprovider) > $OUT/gpu_tests.txt 2>&1
tail -n 6 $OUT/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -n 2 | tee $OUT/smoke.txt
(time timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5) > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err
tail -c 1200 $OUT/bench_driver_command.json; grep real $OUT/bench_driver_command.err
