#!/bin/bash
# Round 5, GPU session 4: the whole -m gpu suite on the round's library; C3 as specified at 1024 x 4K under rocprofv3 (kernel statistics of the fused chain) and
# with the fusions switched off (A/B).
#   gpurun --timeout 1500 -- bash tools/experiments/r5_4_suite_and_c3.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r5_4
mkdir -p $OUT
(time timeout 1100 python -m pytest tests/ -q -m gpu -p no:cacheprovider) > $OUT/gpu_tests.txt 2>&1
tail -n 8 $OUT/gpu_tests.txt
C3="--workload c3 --no-overlap --steps 3 --warmup 1 --no-seq-compare --no-h2d --no-cpu-all-cores --no-rccl-selfcheck"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/c3_prof -- python $ROOT/bench.py $C3 > $ROOT/$OUT/bench_c3_under_rocprof.json 2> $ROOT/$OUT/bench_c3_under_rocprof.err
cd $ROOT
find $OUT/c3_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/c3_kernel_stats.csv
head -n 16 $OUT/c3_kernel_stats.csv
rm -rf $OUT/c3_prof
FUIFGPU_FUSE_DEQUANT=0 FUIFGPU_FUSE_YCBCR=0 timeout 300 python bench.py $C3 --no-cpu-baseline > $OUT/bench_c3_unfused.json 2> $OUT/bench_c3_unfused.err
FUIFGPU_FUSE_DEQUANT=1 FUIFGPU_FUSE_YCBCR=0 timeout 300 python bench.py $C3 --no-cpu-baseline > $OUT/bench_c3_dequant_only.json 2> $OUT/bench_c3_dequant_only.err
timeout 300 python bench.py --workload c3 --steps 6 --warmup 2 --no-seq-compare --no-h2d --no-cpu-all-cores --no-rccl-selfcheck --no-cpu-baseline > $OUT/bench_c3_overlapped.json 2> $OUT/bench_c3_overlapped.err
python - <<'P'
import json
for f in ("bench_c3_under_rocprof", "bench_c3_unfused", "bench_c3_dequant_only", "bench_c3_overlapped"):
    try:
        d = json.loads([l for l in open("gpurun_out/r5_4/%s.json" % f) if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"].get("kernel_ms", d["roofline"].get("launch_ms_alone")), d["roofline"]["transforms"]["ms"], d["config"].get("bits_per_pixel"), d["config"]["parity_roundtrip_ok"], d.get("overlap", {}).get("steps_identical_to_resident_outputs"))
    except Exception as e:
        print(f, "failed", e)
P
