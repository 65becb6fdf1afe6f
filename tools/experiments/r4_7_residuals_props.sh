#!/bin/bash
# Round 4, session 7: squeeze kernels reading int16 residuals straight from the coefficient slab (Op::r16; FUIFGPU_INT16_RESIDUALS=0 = widen
# everything, for A/B), up to 25 reference channels (-E 50), the leaf fetch issued before the write-back: whole GPU suite, C2 / no-index
# timing, the C5 line.   gpurun --timeout 1800 -- bash tools/experiments/r4_7_residuals_props.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r4_s7
mkdir -p $OUT
timeout 60 build/test_fast_symbol 400000 2>&1 | tail -3 | tee $OUT/unit.txt
(time timeout 1200 python -m pytest tests -m gpu -x -q) > $OUT/gpu_tests.txt 2>&1; tail -5 $OUT/gpu_tests.txt
if ! grep -q " passed" $OUT/gpu_tests.txt || grep -q "failed\|error" $OUT/gpu_tests.txt; then echo "GPU SUITE NOT GREEN: no timing"; exit 1; fi
{
timeout 200 python tools/time_decode.py 1024 3840 2160 --reps 3 --check
FUIFGPU_INT16_RESIDUALS=0 timeout 200 python tools/time_decode.py 1024 3840 2160 --reps 2 --check
timeout 200 python tools/time_decode.py 1024 3840 2160 --no-index --reps 2 --check
timeout 200 python tools/time_decode.py 1024 3840 2160 --dct420 --reps 2
} 2>&1 | grep -v amdgpu | tee $OUT/times.txt
timeout 600 python bench.py --workload c5 --batch 2048 --steps 1 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; tail -c 1800 $OUT/bench_c5.json; grep -v "File\|^    \|amdgpu.ids" $OUT/bench_c5.err | tail -3
