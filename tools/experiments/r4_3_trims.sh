#!/bin/bash
# Round 4, session 3: the instruction trims of the pixel loop (decoder v2 with borrow-driven decisions and the lean mantissa loop, no
# LDS supernode slots in the dense configuration, asm leaf switch / commit with EXEC set by hand, mask patch, per-chunk stream-end
# test, the x == 0 rule peeled): parity first, then A/B against the previous library (build/libfuifgpu_base.so) on the same box.
#   gpurun --timeout 1500 -- bash tools/experiments/r4_3_trims.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r4_trims
mkdir -p $OUT
timeout 60 build/test_fast_symbol 400000 2>&1 | tail -8 | tee $OUT/unit.txt
(time timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_parity.py tests/test_gpu_group_parallel.py tests/test_gpu_synthetic.py tests/test_fuzz.py) > $OUT/parity.txt 2>&1; tail -5 $OUT/parity.txt
if ! grep -q " passed" $OUT/parity.txt || grep -q "failed\|error" $OUT/parity.txt; then echo "PARITY NOT GREEN: no timing"; exit 1; fi
{
for rep in 1 2; do
for lib in build/libfuifgpu_base.so fuif_amd/libfuifgpu.so; do
  FUIF_AMD_LIB=$ROOT/$lib timeout 200 python tools/time_decode.py 1024 3840 2160 --reps 2 --check
done; done
for lib in build/libfuifgpu_base.so fuif_amd/libfuifgpu.so; do
  FUIF_AMD_LIB=$ROOT/$lib timeout 200 python tools/time_decode.py 1024 3840 2160 --no-index --reps 1 --check
  FUIF_AMD_LIB=$ROOT/$lib timeout 200 python tools/time_decode.py 128 3840 2160 --reps 2
  FUIF_AMD_LIB=$ROOT/$lib timeout 200 python tools/time_decode.py 16 3840 2160 --reps 2
done
} 2>&1 | grep -v amdgpu | tee $OUT/times.txt
