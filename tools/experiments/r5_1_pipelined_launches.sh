#!/bin/bash
# PREPARED in round 4 for the FIRST GPU session of round 5 (profiles/r4_pipelined_launches.txt measured the effect at 128 pictures):
# consecutive 1024-picture launches overlapped on two HIP streams.  Sequential vs pipelined, entropy only and with the inverse transforms behind every launch,
# with the second launch queued 3.6 s after the first (the first one's busy phase) and queued at once.
#   gpurun --timeout 900 -- bash tools/experiments/r5_1_pipelined_launches.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r5_pipe
mkdir -p $OUT
timeout 300 python tools/pipeline_decode.py 1024 --launches 4 --stagger 3.6 --rounds 1 2>&1 | grep -v amdgpu | tee $OUT/entropy_stagger.txt
timeout 300 python tools/pipeline_decode.py 1024 --launches 4 --stagger 0 --rounds 1 --only-pipelined 2>&1 | grep -v amdgpu | tee $OUT/entropy_at_once.txt
timeout 300 python tools/pipeline_decode.py 1024 --launches 4 --stagger 3.6 --rounds 1 --with-transforms 2>&1 | grep -v amdgpu | tee $OUT/with_transforms.txt
# the same through bench.py (opt-in mode written blind in round 4: first run = first test): every image of the warm-up pass is compared with its source picture
(time timeout 600 python bench.py --pipeline --steps 4 --warmup 2 --no-cpu-baseline) > $OUT/bench_pipeline.json 2> $OUT/bench_pipeline.err; tail -c 1500 $OUT/bench_pipeline.json; tail -n 5 $OUT/bench_pipeline.err
(time timeout 600 python bench.py --pipeline --pipeline-stagger 3.6 --steps 4 --warmup 2 --no-cpu-baseline) > $OUT/bench_pipeline_stagger.json 2> $OUT/bench_pipeline_stagger.err; tail -c 600 $OUT/bench_pipeline_stagger.json
# files WITHOUT the group index (one wavefront per picture): two launches side by side need two wide wavefronts per SIMD = fewer LDS supernodes per wavefront
# (build first, in the build container: tools/build_variant.sh wide20 -DFUIF_LDS_WIDE=20 ; the product keeps 58)
timeout 400 python tools/pipeline_decode.py 1024 --no-index --launches 2 --stagger 0 --rounds 1 2>&1 | grep -v amdgpu | tee $OUT/noindex_product.txt
if [ -f build/libfuifgpu_wide20.so ]; then
  FUIF_AMD_LIB=$ROOT/build/libfuifgpu_wide20.so timeout 400 python tools/pipeline_decode.py 1024 --no-index --launches 2 --stagger 0 --rounds 1 2>&1 | grep -v amdgpu | tee $OUT/noindex_wide20.txt
fi
# Under overlapped launches the limit is wavefront-slot-seconds of WORK, not the long groups' critical path: 7 wavefronts per SIMD (72 VGPRs, 43 spilled; "no
# difference" for a launch alone in round 4) may pay now (build first: tools/build_variant.sh w7 -DFUIF_WAVES=7)
if [ -f build/libfuifgpu_w7.so ]; then
  FUIF_AMD_LIB=$ROOT/build/libfuifgpu_w7.so timeout 300 python tools/pipeline_decode.py 1024 --launches 4 --stagger 3.6 --rounds 1 --only-pipelined 2>&1 | grep -v amdgpu | tee $OUT/entropy_w7.txt
fi
