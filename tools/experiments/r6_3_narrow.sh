#!/bin/bash
# round 6, session 3: narrow supernodes (4 bytes per lane) + one tight context arena + scratch leaves behind the supernodes, on the MI355X
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_3
mkdir -p $OUT
cd $ROOT
export FUIF_BENCH_CACHE=/tmp/fuif_bench_cache
(timeout 200 build/ubench_context_bin 20 | grep -E "waves|scratch|tight" ) > $OUT/ubench_scratch_layout.txt 2>&1
timeout 600 python tools/time_decode.py 1024 --reps 3 --check 2>&1 | grep -v amdgpu > $OUT/time_1024_indexed.txt; cat $OUT/time_1024_indexed.txt
R6_QUICK=1 timeout 600 python tools/experiments/r6_1_upper_bounds.py 1024 2>&1 | grep -v amdgpu > $OUT/upper_bounds_quick.txt; cat $OUT/upper_bounds_quick.txt
timeout 600 python tools/time_decode.py 1024 --no-index --reps 1 --check 2>&1 | grep -v amdgpu > $OUT/time_1024_noindex.txt; cat $OUT/time_1024_noindex.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_group_parallel.py tests/test_gpu_synthetic.py -m gpu -x -q 2>&1 | tail -4 > $OUT/gpu_tests_subset.txt; cat $OUT/gpu_tests_subset.txt
cat $OUT/ubench_scratch_layout.txt
