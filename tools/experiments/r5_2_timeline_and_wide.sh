#!/bin/bash
# Round 5, GPU session 2: (a) where the second of two overlapped 1024-picture launches spends its time (tile log of both launches on one time axis);
# (b) files WITHOUT group index (one wavefront per picture): two launches side by side, product build (58 LDS supernodes: one wavefront per SIMD) against
# 20 / 12 LDS supernodes per wavefront (two / three wavefronts per SIMD); (c) the multi-device boundary tests on the one GPU of a box (--devices 0,0).
#   gpurun --timeout 1200 -- bash tools/experiments/r5_2_timeline_and_wide.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r5_2
mkdir -p $OUT
(time timeout 300 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_boundary_cli.py::test_batch_entry_spreads_files_over_a_device_list \
    tests/test_boundary_cli.py::test_batch_entry_decodes_many_files_in_one_launch tests/test_gpu_transform_exports.py::test_device_selection_and_peer_copy) > $OUT/tests.txt 2>&1
tail -n 5 $OUT/tests.txt
FUIF_AMD_LIB=$ROOT/build/libfuifgpu_tilelog.so TILE_LOG_NPY=$OUT/tile_logs.npy timeout 300 python tools/pipeline_decode.py 1024 --launches 4 --stagger 0 --rounds 1 --only-pipelined --tile-log 2>&1 | grep -v amdgpu > $OUT/overlap_timeline.txt
head -n 60 $OUT/overlap_timeline.txt
timeout 300 python tools/pipeline_decode.py 1024 --no-index --launches 2 --stagger 0 --rounds 1 2>&1 | grep -v amdgpu | tee $OUT/noindex_product.txt
for v in wide20 wide12; do
  FUIF_AMD_LIB=$ROOT/build/libfuifgpu_$v.so timeout 300 python tools/pipeline_decode.py 1024 --no-index --launches 2 --stagger 0 --rounds 1 2>&1 | grep -v amdgpu | tee $OUT/noindex_$v.txt
done
# three launches in flight need a third batch object: not built; 3 wavefronts per SIMD with wide12 = launches of 1536 pictures instead
FUIF_AMD_LIB=$ROOT/build/libfuifgpu_wide12.so timeout 300 python tools/pipeline_decode.py 1536 --no-index --launches 2 --stagger 0 --rounds 1 --only-pipelined 2>&1 | grep -v amdgpu | tee $OUT/noindex_wide12_1536.txt
