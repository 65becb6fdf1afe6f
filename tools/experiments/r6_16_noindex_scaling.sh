#!/bin/bash
# round 6, session 16: files WITHOUT group index (one wavefront per picture), 1 / 2 / 3 batches of 1024 x 4K in flight: where the path saturates
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_16
mkdir -p $OUT
cd $ROOT
export FUIF_BENCH_CACHE=/tmp/fuif_bench_cache
echo "## one batch in flight (58 slots = 116 narrow supernodes in LDS, one wavefront per SIMD)" | tee $OUT/noindex_scaling.txt
timeout 400 python tools/pipeline_decode.py 1024 --launches 2 --stagger 0 --no-index --in-flight-1 --batches 1 --only-pipelined 2>&1 | grep -v amdgpu | tee -a $OUT/noindex_scaling.txt
echo "## two in flight (20 slots = 40 narrow supernodes, two wavefronts per SIMD)" | tee -a $OUT/noindex_scaling.txt
timeout 400 python tools/pipeline_decode.py 1024 --launches 4 --stagger 0 --no-index --batches 2 --only-pipelined 2>&1 | grep -v amdgpu | tee -a $OUT/noindex_scaling.txt
echo "## three in flight, build with FUIF_LDS_WIDE=12 (12 slots = 24 narrow supernodes, three wavefronts per SIMD)" | tee -a $OUT/noindex_scaling.txt
FUIF_AMD_LIB=$ROOT/build/libfuifgpu_wide12.so timeout 500 python tools/pipeline_decode.py 1024 --launches 6 --stagger 0 --no-index --batches 3 --only-pipelined 2>&1 | grep -v amdgpu | tee -a $OUT/noindex_scaling.txt
echo "## two in flight on the same build (12 slots)" | tee -a $OUT/noindex_scaling.txt
FUIF_AMD_LIB=$ROOT/build/libfuifgpu_wide12.so timeout 400 python tools/pipeline_decode.py 1024 --launches 4 --stagger 0 --no-index --batches 2 --only-pipelined 2>&1 | grep -v amdgpu | tee -a $OUT/noindex_scaling.txt
