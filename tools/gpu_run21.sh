#!/bin/bash
# round 2, state of the commit: GPU test tier, rocprofv3 evidence (kernel trace + PMC traffic + calibration), then the default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run21
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 1200 ./tools/collect_profiles_r2.sh r2b > $OUT/collect.log 2>&1; tail -25 $OUT/collect.log
cd $ROOT
if [ -s gpurun_out/prof_r2b/pmc_traffic.json ]; then cp gpurun_out/prof_r2b/pmc_traffic.json profiles/r2_pergroup_pmc_traffic.json; fi
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; cat $OUT/bench_default.json; tail -3 $OUT/bench_default.err
