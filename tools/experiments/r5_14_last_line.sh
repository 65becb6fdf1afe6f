#!/bin/bash
# Round 5, last GPU seconds: a short default line on the final library (both wide instantiations; the overlapped legs call set_in_flight(2)).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r5_14
mkdir -p $OUT
(time timeout 420 python bench.py --steps 2 --warmup 1 --no-extra-legs --reference-encoded 0 --no-live-traffic --no-h2d --no-cpu-all-cores) > $OUT/bench_short.json 2> $OUT/bench_short.err
tail -c 700 $OUT/bench_short.json; grep real $OUT/bench_short.err
