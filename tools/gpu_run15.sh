#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run15
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 600 python bench.py --workload c5 --batch 512 --distinct 4 --steps 1 --warmup 1 > $OUT/c5_small.json 2> $OUT/c5_small.err; echo "c5 rc=$?"; cat $OUT/c5_small.json; tail -3 $OUT/c5_small.err
timeout 1500 ./tools/collect_profiles_r2.sh r2 > $OUT/collect.log 2>&1; tail -30 $OUT/collect.log
