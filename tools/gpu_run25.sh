#!/bin/bash
# round 2: what clock does the GPU run at during the dense launch?  (cycle counters say ~1.25 GHz against ~1.9 GHz on a light launch)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run25
mkdir -p $OUT
cd $ROOT
sample() {  # name
  ( for i in $(seq 1 40); do date +%s.%N; rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | grep -i "sclk\|mclk\|fclk\|socclk\|Power\|Performance Level"; sleep 1; done ) > $OUT/smi_$1.txt 2>&1 &
  SMI=$!
  timeout 300 python tools/tile_timeline.py 1024 3840 2160 > $OUT/timeline_$1.txt 2>&1
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  grep "^launch\|^c54\|^scheduler: busy" $OUT/timeline_$1.txt
  grep -i "sclk" $OUT/smi_$1.txt | sort | uniq -c | sort -rn | head -6
  grep -i "power" $OUT/smi_$1.txt | sort | uniq -c | sort -rn | head -4
}
echo "=== default"
rocm-smi --showperflevel 2>/dev/null | grep -i level
sample default
echo "=== perf level high"
rocm-smi --setperflevel high 2>&1 | tail -2
sample high
rocm-smi --setperflevel auto 2>&1 | tail -1
