#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run14
mkdir -p $OUT
cd $ROOT
# streamed mode at a small size first (seconds)
timeout 600 python bench.py --workload c4 --width 1024 --height 1024 --batch 64 --chunk 16 --distinct 2 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/c4_small.json 2> $OUT/c4_small.err; echo "c4 small rc=$?"; cat $OUT/c4_small.json; tail -3 $OUT/c4_small.err
# the full-size C4 parity test (one 8192x8192x4 14-bit image against the oracle) in the background on the host cores ...
( FUIF_TEST_C4_FULL=1 timeout 1500 python -m pytest tests/test_gpu_synthetic.py::test_c4_full_size_image_matches_oracle -m gpu -x -q > $OUT/c4_full_test.log 2>&1; echo "c4 full test rc=$?" >> $OUT/c4_full_test.log ) &
# ... while the full-size C4 bench generates its inputs and runs: 256 x 8192x8192x4 in chunks of 32
timeout 1500 python bench.py --workload c4 --width 8192 --height 8192 --batch 256 --chunk 32 --distinct 2 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/c4_full.json 2> $OUT/c4_full.err; echo "c4 full rc=$?"; cat $OUT/c4_full.json; tail -3 $OUT/c4_full.err
wait
tail -4 $OUT/c4_full_test.log
