#!/bin/bash
# round 2, HEAD: smoke() and a short bench line end to end
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run30
mkdir -p $OUT
cd $ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout 400 python bench.py --batch 128 --steps 1 --warmup 1 --no-cpu-all-cores > $OUT/bench_small.json 2> $OUT/bench_small.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench_small.json; tail -2 $OUT/bench_small.err
