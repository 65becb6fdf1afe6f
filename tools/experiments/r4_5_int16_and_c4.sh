#!/bin/bash
# Round 4, session 5: the int16 coefficient slab (entropy kernel writes pixel_type samples; the inverse transforms run on a widened
# copy of a chunk of images) and the streaming batch (fuifgpu_batch_create_streaming / _undo_transforms_to): the whole GPU suite,
# C2 against the int32-slab library of session 4 on the same box, then BASELINE config C4 at full size in ONE entropy launch.
#   gpurun --timeout 2400 -- bash tools/experiments/r4_5_int16_and_c4.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r4_int16
mkdir -p $OUT
timeout 60 build/test_fast_symbol 400000 2>&1 | tail -3 | tee $OUT/unit.txt
(time timeout 1200 python -m pytest tests -m gpu -x -q) > $OUT/gpu_tests.txt 2>&1; tail -5 $OUT/gpu_tests.txt
if ! grep -q " passed" $OUT/gpu_tests.txt || grep -q "failed\|error" $OUT/gpu_tests.txt; then echo "GPU SUITE NOT GREEN: no timing"; exit 1; fi
{
for lib in build/libfuifgpu_trims2.so fuif_amd/libfuifgpu.so build/libfuifgpu_trims2.so fuif_amd/libfuifgpu.so; do
  FUIF_AMD_LIB=$ROOT/$lib timeout 200 python tools/time_decode.py 1024 3840 2160 --reps 2 --check
done
} 2>&1 | grep -v amdgpu | tee $OUT/times.txt
FUIFGPU_CTX_MB=32 timeout 900 python bench.py --workload c4 --width 8192 --height 8192 --batch 256 --chunk -1 --distinct 2 --steps 1 --warmup 0 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; tail -c 2500 $OUT/bench_c4.json; tail -5 $OUT/bench_c4.err
