#!/bin/bash
# Round 4, session 14: the whole GPU suite on hardware after the widening work of the round's second half (single-transform entry points for the
# Palette / Approximate / 2D-match inverses, soft matches, the stepwise front end of the binding, three soft-match fixtures).
#   gpurun --timeout 600 -- bash tools/experiments/r4_14_widening_tests.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r4_widen
mkdir -p $OUT
(time timeout 420 python -m pytest tests -m gpu -x -q --durations=8) > $OUT/gpu_tests.txt 2>&1; tail -16 $OUT/gpu_tests.txt
