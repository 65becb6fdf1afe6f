#!/bin/bash
# round 2: size-balanced queues + priority (defaults), LDS-free horizontal unsqueeze, then the GPU test tier and the bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run18
mkdir -p $OUT
cd $ROOT
echo "=== balanced queues, priority base 2 (defaults)"
timeout 300 python tools/tile_timeline.py 1024 3840 2160 > $OUT/balanced_prio2.txt 2>&1
grep "^launch\|^c5[4-9]\|^c60\|^total tile-time\|^scheduler\|per-SIMD" $OUT/balanced_prio2.txt
echo "=== transforms"
FUIFGPU_HSQ=lds timeout 300 python tools/transform_time.py 256 2>&1 | grep -v amdgpu | tee $OUT/transforms_lds.txt
timeout 300 python tools/transform_time.py 256 2>&1 | grep -v amdgpu | tee $OUT/transforms_rows.txt
echo "=== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
echo "=== bench"
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; cat $OUT/bench_default.json; tail -3 $OUT/bench_default.err
