#!/bin/bash
# Round 4, session 6: the evidence set of the FINAL library, by one script:
#   1. the default `python bench.py` (C2; live PMC traffic, RCCL self-check, CPU legs) -> profiles/r4_final_bench_default.json
#   2. rocprofv3 --kernel-trace --stats of the bench command + PMC passes + calibration (tools/collect_profiles_r4.sh)
#   3. per-phase cycles of the shipped 6-wavefront build alone (8 streams) and in the 1024-picture launch (-DFUIF_PROF)
#   4. one bench line each for C3 and C5 on one GPU with cpu_baseline and roofline (C4: tools/experiments/r4_5b_c4.sh)
#   5. the writer: host / GPU one group at a time / GPU batch (tools/time_encode.py); one picture alone by channel group
#   gpurun --timeout 2400 -- bash tools/experiments/r4_6_evidence.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r4_final
mkdir -p $OUT
(time timeout 900 python bench.py) > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 3000 $OUT/bench_default.json; grep -v "File\|^    \|amdgpu.ids" $OUT/bench_default.err | tail -4
timeout 900 bash tools/collect_profiles_r4.sh > $OUT/collect.txt 2>&1; tail -30 $OUT/collect.txt
{
for n in 8 1024; do echo "== build/libfuifgpu_prof.so, $n streams"; FUIF_AMD_LIB=$ROOT/build/libfuifgpu_prof.so timeout 300 python tools/prof_kernel.py $n 3840 2160; done
} 2>&1 | grep -v amdgpu | tee $OUT/phase_cycles.txt
timeout 600 python bench.py --workload c3 --no-live-traffic > $OUT/bench_c3.json 2> $OUT/bench_c3.err; tail -c 1800 $OUT/bench_c3.json
timeout 600 python bench.py --workload c5 --batch 2048 --steps 1 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; tail -c 1800 $OUT/bench_c5.json
{ timeout 200 python tools/time_encode.py 16 1920 1080; timeout 300 python tools/time_encode.py 8 3840 2160; } 2>&1 | grep -v amdgpu | tee $OUT/time_encode.txt
FUIF_AMD_LIB=$ROOT/build/libfuifgpu_profch.so timeout 200 python tools/prof_by_channel.py 1 3840 2160 2>&1 | grep -v amdgpu | tee $OUT/one_picture_by_channel.txt
