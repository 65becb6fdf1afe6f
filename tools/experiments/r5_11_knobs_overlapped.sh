#!/bin/bash
# Round 5: the scheduler knobs of round 4 (tuned for launches alone) re-measured with overlapped launches (two streams, 8 launches of 1024 x 4K, per-launch completions):
# yield slack 4 (default) / 2 / 8, priorities off, long tiles per SIMD 3 (default) / 4.  One box, one after the other.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r5_11
mkdir -p $OUT
run() { name=$1; shift; env "$@" timeout 300 python tools/pipeline_decode.py 1024 --launches 8 --stagger 0 --per-launch 2>&1 | grep "steady state" | sed "s/^/$name: /" | tee -a $OUT/knobs.txt; }
run "default              " FUIFGPU_NOP=1
run "FUIFGPU_YIELD_SLACK=2" FUIFGPU_YIELD_SLACK=2
run "FUIFGPU_YIELD_SLACK=8" FUIFGPU_YIELD_SLACK=8
run "FUIFGPU_PRIO_BASE=-1 " FUIFGPU_PRIO_BASE=-1
run "FUIFGPU_LONG_PER_SIMD=4" FUIFGPU_LONG_PER_SIMD=4
run "default (again)      " FUIFGPU_NOP=1
