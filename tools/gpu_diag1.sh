#!/bin/bash
# GPU diagnostics, round 2 run 1: issue model, occupancy scaling on independent streams, tile order, SQ counters (baseline kernel)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/diag1
mkdir -p $OUT
cd $ROOT
./tools/ubench_issue_bin > $OUT/ubench_issue.txt 2>&1
cat $OUT/ubench_issue.txt
./tools/ubench_bin > $OUT/ubench.txt 2>&1
# per-SIMD throughput vs resident wavefronts on INDEPENDENT streams (no hand-off waits, no tail): 1, 2, 3, 4 per SIMD
REPS=1 timeout 300 python tools/occupancy_probe.py 1024,2048,3072,4096 1920 1080 seq > $OUT/occ_seq.txt 2>&1
cat $OUT/occ_seq.txt
# tile order on the headline shape (1024 x 4K, index)
REPS=1 timeout 200 python tools/occupancy_probe.py 1024 3840 2160 > $OUT/order_group.txt 2>&1
FUIFGPU_TILE_ORDER=image REPS=1 timeout 200 python tools/occupancy_probe.py 1024 3840 2160 > $OUT/order_image.txt 2>&1
FUIFGPU_TILE_ORDER=image:128 REPS=1 timeout 200 python tools/occupancy_probe.py 1024 3840 2160 > $OUT/order_image128.txt 2>&1
cat $OUT/order_*.txt
timeout 900 ./tools/pmc_sq.sh r2_base_groups 256 1920 1080 > $OUT/sq_groups.log 2>&1
tail -40 $OUT/sq_groups.log
