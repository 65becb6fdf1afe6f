#!/bin/bash
# Round 4, session 10: dequantisation reads the int16 DCT coefficients and writes sample * q into the int32 copy (widening and scaling in
# one pass, Op::r16 on OP_QUANT): whole GPU suite, the C3 shape with and without it, the C3 bench line.
#   gpurun --timeout 1500 -- bash tools/experiments/r4_10_dequant.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r4_dequant
mkdir -p $OUT
(time timeout 1200 python -m pytest tests -m gpu -x -q) > $OUT/gpu_tests.txt 2>&1; tail -5 $OUT/gpu_tests.txt
if ! grep -q " passed" $OUT/gpu_tests.txt || grep -q "failed\|error" $OUT/gpu_tests.txt; then echo "GPU SUITE NOT GREEN: no timing"; exit 1; fi
{
timeout 200 python tools/time_decode.py 1024 3840 2160 --dct420 --reps 3
FUIFGPU_INT16_RESIDUALS=0 timeout 200 python tools/time_decode.py 1024 3840 2160 --dct420 --reps 3
} 2>&1 | grep -v amdgpu | tee $OUT/times.txt
timeout 600 python bench.py --workload c3 --no-live-traffic > $OUT/bench_c3.json 2> $OUT/bench_c3.err; tail -c 1500 $OUT/bench_c3.json
