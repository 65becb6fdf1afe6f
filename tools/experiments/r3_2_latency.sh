#!/bin/bash
# round 3, session 2: (a) latency of the two dependent fetches per symbol vs the memory all wavefronts touch together
# (is the long tiles' 1.6 us per symbol in the 1024-picture launch, against 1.07 us in the 128-picture launch, the
# Infinity Cache running out?); (b) per-phase cycles of the SHIPPED 6-wavefront configuration, alone and under load.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_2
mkdir -p $OUT
cd $ROOT
timeout 300 build/ubench_latency_bin > $OUT/ubench_latency.txt 2>&1; cat $OUT/ubench_latency.txt
for n in 8 128 1024; do
  wh="3840 2160"; [ $n = 8 ] && wh="1920 1080"
  FUIF_AMD_LIB=$ROOT/build/libfuifgpu_prof.so timeout 400 python tools/prof_kernel.py $n $wh 2>&1 | grep -v amdgpu > $OUT/phases_$n.txt; cat $OUT/phases_$n.txt
done
