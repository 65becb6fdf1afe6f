#!/bin/bash
# Round 4, last session: the evidence of the FINAL library (kernel names in the rocprofv3 CSV = the sources at HEAD):
#   1. the default `python bench.py` (C2; live PMC traffic, RCCL self-check, CPU legs) -> profiles/r4_final_bench_default.json
#   2. tools/collect_profiles_r4.sh: rocprofv3 --kernel-trace --stats of the bench command, one --pmc pass per counter, calibration
#   3. the whole GPU suite once more
#   gpurun --timeout 1800 -- bash tools/experiments/r4_9_final.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r4_final2
mkdir -p $OUT
(time timeout 900 python bench.py) > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1200 $OUT/bench_default.json; grep -v "File\|^    \|amdgpu.ids" $OUT/bench_default.err | tail -4
timeout 900 bash tools/collect_profiles_r4.sh > $OUT/collect.txt 2>&1; tail -28 $OUT/collect.txt
(time timeout 1200 python -m pytest tests -m gpu -x -q) > $OUT/gpu_tests.txt 2>&1; tail -5 $OUT/gpu_tests.txt
