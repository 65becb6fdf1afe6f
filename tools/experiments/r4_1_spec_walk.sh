#!/bin/bash
# RECORD of the first GPU session of round 4 (result: profiles/r4_spec_walk_experiment.txt).  The -DFUIF_SPEC_WALK / -DFUIF_SPEC_LEAF code this
# script builds was DELETED from maniac_decode.hip right after it (both variants were slower); the script runs as written only at commit 6e77e44.
# Round 4, experiment prepared at the end of round 3 (no GPU minutes were left to run it): the speculative early walk,
# -DFUIF_SPEC_WALK (fuif_amd/csrc/maniac_decode.hip; profiles/r3_first_left_dependent_test.txt).
# The release kernel is bit-identical with and without the macro's code in the source (checked on the ISA); the variant is
# parity-green on the wavefront emulator and served 50.7 % of the walk rounds behind the root from LDS there (24.4 % with the
# static residents) on a deep-tree picture.  What is NOT known: whether it is faster.
#   gpurun --timeout 900 -- bash tools/experiments/r4_1_spec_walk.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r4_spec_walk
mkdir -p $OUT
[ -f build/libfuifgpu_spec.so ] || bash tools/build_variant.sh spec -DFUIF_SPEC_WALK
[ -f build/libfuifgpu_spec2.so ] || bash tools/build_variant.sh spec2 -DFUIF_SPEC_WALK=2     # speculate only while <= 4 wavefronts are alive on the SIMD (the tail)
{
for lib in fuif_amd/libfuifgpu.so build/libfuifgpu_spec.so build/libfuifgpu_spec2.so; do
  FUIF_AMD_LIB=$ROOT/$lib timeout 120 python tools/time_decode.py 128 3840 2160 --reps 2 --check
  FUIF_AMD_LIB=$ROOT/$lib timeout 200 python tools/time_decode.py 1024 3840 2160 --reps 3 --check
done
# per-phase cycles of both kernels (tree walk is where the supernode fetch sits): 8 streams (alone), 128 and 1024
[ -f build/libfuifgpu_prof.so ] || bash tools/build_variant.sh prof -DFUIF_PROF
[ -f build/libfuifgpu_spec_prof.so ] || bash tools/build_variant.sh spec_prof -DFUIF_PROF -DFUIF_SPEC_WALK
for lib in build/libfuifgpu_prof.so build/libfuifgpu_spec_prof.so; do for n in 8 1024; do
  echo "== $lib, $n streams"; FUIF_AMD_LIB=$ROOT/$lib timeout 300 python tools/prof_kernel.py $n 3840 2160
done; done
# streams WITHOUT a group index (one wavefront per image, the wide configuration): -DFUIF_SPEC_LEAF speculates on the leaf
[ -f build/libfuifgpu_specleaf.so ] || bash tools/build_variant.sh specleaf -DFUIF_SPEC_LEAF
for lib in fuif_amd/libfuifgpu.so build/libfuifgpu_specleaf.so; do
  FUIF_AMD_LIB=$ROOT/$lib timeout 200 python tools/time_decode.py 1024 3840 2160 --no-index --reps 2 --check
done
for lib in build/libfuifgpu_spec.so build/libfuifgpu_specleaf.so; do
  FUIF_AMD_LIB=$ROOT/$lib timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_group_parallel.py -m gpu -x -q
done
} 2>&1 | grep -v amdgpu | tee $OUT/times.txt
# per channel group (long tiles vs the chain in front of them), release against the speculative walk, 1024 pictures
{
for lib in build/libfuifgpu_profch.so build/libfuifgpu_spec_profch.so; do
  [ -f $lib ] && FUIF_AMD_LIB=$ROOT/$lib timeout 300 python tools/prof_by_channel.py 1024 3840 2160
done
} 2>&1 | grep -v amdgpu | tee $OUT/by_channel.txt
