#!/bin/bash
# round 2: C3 (JPEG-transcode shape) with the 2x2 upsampling kernel: rocprofv3 kernel statistics, then the GPU test tier
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run28
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-seq-compare --no-h2d > $OUT/bench_c3.json 2> $OUT/bench_c3.err
echo "rc=$?"; grep "^{" $OUT/bench_c3.json | cut -c1-300
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
grep "fuifgpu" "$f" | cut -c1-160
cp "$f" $OUT/c3_kernel_stats.csv
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
