#!/bin/bash
# Round 4, session 2: (a) the tests written since session 1 on hardware (RCCL at world size 1, the opt-in CPU route of the boundary,
# pinned tiles, the GPU encoder against the reference CLI's bytes); (b) SENSITIVITY PROBES on k_maniac_decode: builds that add N
# instructions of one kind per decoded symbol (-DFUIF_PROBE_S / _V / _B: scalar, vector, taken branch) next to the release library
# on the same box -- what one more instruction of each kind costs the 1024 x 4K launch (round 3's trims made the launch slower;
# before trimming again the derivative is measured).   gpurun --timeout 1200 -- bash tools/experiments/r4_2_probes.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r4_probes
mkdir -p $OUT
(time timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_rccl_world1.py tests/test_boundary_cli.py "tests/test_gpu_group_parallel.py::test_pinned_tiles_when_the_context_arena_runs_out" tests/test_zz_gpu_encoder.py) > $OUT/new_tests.txt 2>&1; tail -5 $OUT/new_tests.txt
{
for rep in 1 2; do
for lib in fuif_amd/libfuifgpu.so build/libfuifgpu_ps16.so build/libfuifgpu_pv16.so build/libfuifgpu_pb8.so; do
  FUIF_AMD_LIB=$ROOT/$lib timeout 200 python tools/time_decode.py 1024 3840 2160 --reps 2 --check
done; done
} 2>&1 | grep -v amdgpu | tee $OUT/times.txt
