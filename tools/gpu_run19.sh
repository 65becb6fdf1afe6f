#!/bin/bash
# round 2: sensitivity of the launch time to memory traffic (one extra 512-byte read per global supernode round); unrolled horizontal unsqueeze
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run19
mkdir -p $OUT
cd $ROOT
for v in "" _extra; do
  echo "=== libfuifgpu$v"
  FUIF_AMD_LIB=$ROOT/fuif_amd/libfuifgpu$v.so timeout 300 python tools/tile_timeline.py 1024 3840 2160 > $OUT/timeline$v.txt 2>&1
  grep "^launch\|^c54\|^c59\|^c60\|^scheduler" $OUT/timeline$v.txt
  FUIF_AMD_LIB=$ROOT/fuif_amd/libfuifgpu$v.so REPS=1 timeout 300 python tools/occupancy_probe.py 6144 1920 1080 seq 2>&1 | grep -v amdgpu | tee $OUT/independent$v.txt
done
echo "=== transforms"
timeout 300 python tools/transform_time.py 256 2>&1 | grep -v amdgpu | tee $OUT/transforms_rows32.txt
timeout 600 python -m pytest tests/test_gpu_transform_exports.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
