#!/bin/bash
# round 3, session 3: records with leaf slots (the leaf arrives with the supernode that leads to it, by LDS-DMA).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_3
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_group_parallel.py tests/test_gpu_parity.py tests/test_gpu_synthetic.py > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
FUIFGPU_CTX_KB=64 timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_group_parallel.py > $OUT/pytest_pinned.txt 2>&1; tail -3 $OUT/pytest_pinned.txt
for slots in 16 8 0; do
  FUIFGPU_LEAF_SLOTS=$slots timeout 400 python tools/time_decode.py 1024 3840 2160 --check 2>&1 | grep -v amdgpu | sed "s/^/slots $slots: /" | tee -a $OUT/times.txt
done
for slots in 16 0; do
  FUIFGPU_LEAF_SLOTS=$slots timeout 400 python tools/time_decode.py 128 3840 2160 --check 2>&1 | grep -v amdgpu | sed "s/^/slots $slots: /" | tee -a $OUT/times.txt
done
timeout 300 python tools/time_decode.py 16 3840 2160 2>&1 | grep -v amdgpu | tee -a $OUT/times.txt
timeout 400 python tools/time_decode.py 1024 3840 2160 --no-index --reps 1 2>&1 | grep -v amdgpu | tee -a $OUT/times.txt
FUIF_AMD_LIB=$ROOT/build/libfuifgpu_stats.so timeout 400 python tools/tile_timeline.py 1024 3840 2160 > $OUT/timeline1024.txt 2>&1
grep "^launch\|^c54\|^c59\|^c60\|^scheduler\|per-SIMD" $OUT/timeline1024.txt
