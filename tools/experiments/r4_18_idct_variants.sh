#!/bin/bash
# Round 4, session 18 (the round's last GPU seconds): k_idct8x8 alone through its entry point (tools/idct_bench.py, one synthetic
# 1920 x 2160-block component = 1.06 GB of coefficients in, 1.06 GB of samples out), five builds back to back:
#   libfuifgpu.so        one lane per block, 166 VGPRs, 88 SGPRs spilled, occupancy 3 (shipped)
#   _idctb1              the same with a scheduling barrier per column: 0 spills
#   _idctpair            TWO lanes per block (4 columns / 4 rows each, 16 values swapped with the neighbour lane by DPP): 106 VGPRs, occupancy 4
#   _idctpairb1          + a barrier per column: 112 VGPRs, 0 spills
#   _idctpairw5          + amdgpu_waves_per_eu(5): 96 VGPRs, 7 spilled to scratch, occupancy 5
# every build prints a hash of 32 MB of its output (must be equal), then the paired builds run the golden fixtures and the JPEG-like parity test.
#   gpurun --timeout 85 -- bash tools/experiments/r4_18_idct_variants.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r4_idct
mkdir -p $OUT
for lib in fuif_amd/libfuifgpu.so build/libfuifgpu_idctb1.so build/libfuifgpu_idctpair.so build/libfuifgpu_idctpairb1.so build/libfuifgpu_idctpairw5.so fuif_amd/libfuifgpu.so; do
  FUIF_AMD_LIB=$ROOT/$lib timeout 20 python tools/idct_bench.py 1920 2160 12 2>&1 | grep -v amdgpu | tee -a $OUT/times.txt
done
for v in idctpairw5 idctpairb1 idctpair; do
  echo "== parity with build/libfuifgpu_$v.so" | tee -a $OUT/parity.txt
  FUIF_AMD_LIB=$ROOT/build/libfuifgpu_$v.so timeout 40 python -m pytest -m gpu -x -q tests/test_gpu_parity.py::test_golden_fixtures_bit_exact tests/test_gpu_group_parallel.py::test_jpeg_like_indexed 2>&1 | tail -n 2 | tee -a $OUT/parity.txt
done
