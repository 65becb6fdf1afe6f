#!/bin/bash
# round 3, session 4: records with leaf slots, second version (nodes first, slots behind them in one asm statement; slots by run length)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_4
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_group_parallel.py tests/test_gpu_parity.py tests/test_gpu_synthetic.py > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
for slots in 16 8 0; do
  FUIFGPU_LEAF_SLOTS=$slots timeout 400 python tools/time_decode.py 1024 3840 2160 --check 2>&1 | grep -v amdgpu | sed "s/^/slots $slots: /" | tee -a $OUT/times.txt
done
FUIFGPU_LEAF_SLOTS=16 timeout 400 python tools/time_decode.py 128 3840 2160 --check 2>&1 | grep -v amdgpu | sed "s/^/slots 16: /" | tee -a $OUT/times.txt
for s in 16; do echo "== slots $s"; FUIFGPU_LEAF_SLOTS=$s FUIF_AMD_LIB=$PWD/build/libfuifgpu_prof.so timeout 400 python tools/prof_kernel.py 1024 3840 2160 2>&1 | grep -v amdgpu; done | tee $OUT/phases_slots_v2.txt
