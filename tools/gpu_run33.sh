#!/bin/bash
# round 2: GPU clock during a launch with few tiles per CU (128 pictures): is the 4x per-symbol slowdown a clock effect?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run33
mkdir -p $OUT
cd $ROOT
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Graphics Package Power"; sleep 0.5; done ) > $OUT/smi128.txt 2>&1 &
SMI=$!
timeout 200 python tools/tile_timeline.py 128 3840 2160 > $OUT/timeline128.txt 2>&1
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
grep "^launch\|^c54" $OUT/timeline128.txt
grep -i sclk $OUT/smi128.txt | sed 's/.*(\(.*\))/\1/' | tr '\n' ' '; echo
grep -i power $OUT/smi128.txt | sed 's/.*: //' | tr '\n' ' '; echo
