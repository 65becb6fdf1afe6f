#!/bin/bash
# round 3, session 1: idle-wavefront retirement + pinned tiles + statistics out of the release kernel.
#  (a) parity of the new library (group-parallel + golden tests), (b) the small-batch cliff (128 x 4K) old vs new,
#  (c) the headline launch (1024 x 4K) old vs new, with tile timelines from the -DFUIF_STATS builds.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_1
mkdir -p $OUT
cd $ROOT
export FUIF_BENCH_CACHE=/tmp/fuif_bench_cache
timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_group_parallel.py tests/test_gpu_parity.py tests/test_gpu_synthetic.py > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
FUIFGPU_CTX_KB=64 timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_group_parallel.py > $OUT/pytest_pinned.txt 2>&1; tail -3 $OUT/pytest_pinned.txt
for lib in r2 stats; do
  FUIF_AMD_LIB=$ROOT/build/libfuifgpu_$lib.so timeout 300 python tools/tile_timeline.py 128 3840 2160 > $OUT/timeline128_$lib.txt 2>&1
  grep "^launch\|^c54\|^c59\|^c60\|^scheduler" $OUT/timeline128_$lib.txt
done
timeout 300 python tools/time_decode.py 128 3840 2160 --check > $OUT/time128_release.txt 2>&1; tail -3 $OUT/time128_release.txt
timeout 300 python tools/time_decode.py 16 3840 2160 > $OUT/time16_release.txt 2>&1; tail -2 $OUT/time16_release.txt
for lib in stats r2; do
  FUIF_AMD_LIB=$ROOT/build/libfuifgpu_$lib.so timeout 400 python tools/tile_timeline.py 1024 3840 2160 > $OUT/timeline1024_$lib.txt 2>&1
  grep "^launch\|^c54\|^c59\|^c60\|^scheduler\|per-SIMD" $OUT/timeline1024_$lib.txt
done
timeout 400 python tools/time_decode.py 1024 3840 2160 --check > $OUT/time1024_release.txt 2>&1; tail -3 $OUT/time1024_release.txt
