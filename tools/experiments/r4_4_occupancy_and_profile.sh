#!/bin/bash
# Round 4, session 4: the second batch of trims (scalar-base supernode loads, running LDS address) against the first batch and round 3's
# kernel on one box; 5 and 7 wavefronts per SIMD with the trimmed kernel (LDS no longer limits 7: 4968 bytes per wavefront);
# per-channel phase profile and the tile timeline of the new kernel.   gpurun --timeout 1500 -- bash tools/experiments/r4_4_occupancy_and_profile.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r4_occ
mkdir -p $OUT
timeout 60 build/test_fast_symbol 400000 2>&1 | tail -3 | tee $OUT/unit.txt
(time timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_parity.py tests/test_gpu_group_parallel.py) > $OUT/parity.txt 2>&1; tail -4 $OUT/parity.txt
if ! grep -q " passed" $OUT/parity.txt || grep -q "failed\|error" $OUT/parity.txt; then echo "PARITY NOT GREEN: no timing"; exit 1; fi
{
for lib in build/libfuifgpu_base.so fuif_amd/libfuifgpu.so build/libfuifgpu_w7.so build/libfuifgpu_w5.so fuif_amd/libfuifgpu.so build/libfuifgpu_w7.so; do
  FUIF_AMD_LIB=$ROOT/$lib timeout 200 python tools/time_decode.py 1024 3840 2160 --reps 2 --check
done
FUIF_AMD_LIB=$ROOT/fuif_amd/libfuifgpu.so timeout 200 python tools/time_decode.py 1024 3840 2160 --no-index --reps 1 --check
} 2>&1 | grep -v amdgpu | tee $OUT/times.txt
FUIF_AMD_LIB=$ROOT/build/libfuifgpu_profch.so timeout 300 python tools/prof_by_channel.py 1024 3840 2160 2>&1 | grep -v amdgpu | tee $OUT/by_channel.txt
FUIF_AMD_LIB=$ROOT/build/libfuifgpu_tilelog.so timeout 300 python tools/tile_timeline.py 1024 3840 2160 2>&1 | grep -v amdgpu | tee $OUT/timeline.txt | tail -40
