#!/bin/bash
# Round 5, GPU session 7: launches that bring only 3 wavefronts per SIMD (FUIFGPU_WAVES_PER_SIMD=3: half of the device's slots), two of them side by side on two
# streams -- each launch is then bound by its WORK (37 k wavefront-seconds over 3072 slots = 12 s), not by its long groups (6.5 s), and the device is full all the time:
# the bound is 37.0 k / 6144 = 6.0 s per launch in steady state.  Per-launch completion times from one host thread per stream.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r5_7
mkdir -p $OUT
timeout 300 python tools/pipeline_decode.py 1024 --launches 8 --stagger 0 --per-launch 2>&1 | grep -v amdgpu | tee $OUT/full_device_launches.txt
FUIFGPU_WAVES_PER_SIMD=3 timeout 400 python tools/pipeline_decode.py 1024 --launches 10 --stagger 6 --per-launch 2>&1 | grep -v amdgpu | tee $OUT/half_device_long3.txt
FUIFGPU_WAVES_PER_SIMD=3 FUIFGPU_LONG_PER_SIMD=2 timeout 400 python tools/pipeline_decode.py 1024 --launches 10 --stagger 6 --per-launch 2>&1 | grep -v amdgpu | tee $OUT/half_device_long2.txt
