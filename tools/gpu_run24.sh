#!/bin/bash
# round 2: per-phase wavefront cycles of the dense launch under its real contention (6 wavefronts per SIMD, priorities), -DFUIF_PROF build
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run24
mkdir -p $OUT
cd $ROOT
FUIF_AMD_LIB=$ROOT/fuif_amd/libfuifgpu_prof.so timeout 300 python tools/prof_kernel.py 1024 3840 2160 2>&1 | grep -v amdgpu | tee $OUT/phases_1024x4k.txt
FUIF_AMD_LIB=$ROOT/fuif_amd/libfuifgpu_prof.so timeout 300 python tools/prof_kernel.py 8 1920 1080 2>&1 | grep -v amdgpu | tee $OUT/phases_8x1080p.txt
