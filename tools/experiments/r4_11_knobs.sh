#!/bin/bash
# Round 4, session 11: the scheduler's two knobs re-measured on the trimmed kernel (FUIFGPU_PRIO_BASE: wavefront priority by tile size
# class, FUIFGPU_YIELD_SLACK: rows a producer must be ahead before a suspended tile is resumed), and __graft_entry__.smoke().
#   gpurun --timeout 1200 -- bash tools/experiments/r4_11_knobs.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r4_knobs
mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2 | tee $OUT/smoke.txt
{
for kv in "FUIFGPU_PRIO_BASE=2" "FUIFGPU_PRIO_BASE=-1" "FUIFGPU_PRIO_BASE=1" "FUIFGPU_PRIO_BASE=3" "FUIFGPU_YIELD_SLACK=4" "FUIFGPU_YIELD_SLACK=16" "FUIFGPU_YIELD_SLACK=32" "FUIFGPU_PRIO_BASE=2"; do
  echo "== $kv"
  env $kv timeout 200 python tools/time_decode.py 1024 3840 2160 --reps 2
done
} 2>&1 | grep -v amdgpu | tee $OUT/times.txt
