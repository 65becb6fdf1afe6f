#!/bin/bash
# Round 5: both wide instantiations in one library, picked by fuifgpu_batch_set_in_flight: parity (the whole test_gpu_parity file) and, for 1024 x 4K without index,
# a launch alone with in_flight = 1 (58 LDS supernodes) and two launches side by side with in_flight = 2 (20).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r5_13
mkdir -p $OUT
(time timeout 400 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_parity.py tests/test_gpu_fast_symbol.py) > $OUT/tests.txt 2>&1
tail -n 4 $OUT/tests.txt
timeout 200 python tools/pipeline_decode.py 1024 --no-index --launches 2 --stagger 0 --rounds 1 --only-sequential --in-flight-1 2>&1 | grep -v amdgpu | tee $OUT/noindex_alone_58.txt
timeout 200 python tools/pipeline_decode.py 1024 --no-index --launches 2 --stagger 0 --rounds 1 --only-pipelined 2>&1 | grep -v amdgpu | tee $OUT/noindex_two_in_flight_20.txt
