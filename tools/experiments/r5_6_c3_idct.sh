#!/bin/bash
# Round 5, GPU session 6: the iDCT instantiation with int16 AC loads known at compile time (loads of a column issued together again): the JPEG parity tests on
# hardware, C3 as specified under rocprofv3 (resident path), and the same with the dequantisation fusion off for the kernel's own A/B.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r5_6
mkdir -p $OUT
(time timeout 600 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_synthetic.py::test_jpeg_like_chain_fused_and_unfused_match_oracle \
    tests/test_gpu_parity.py::test_golden_fixtures_bit_exact tests/test_gpu_group_parallel.py::test_jpeg_like_indexed tests/test_gpu_transform_exports.py tests/test_boundary_cli.py) > $OUT/tests.txt 2>&1
tail -n 4 $OUT/tests.txt
C3="--workload c3 --no-overlap --steps 3 --warmup 1 --no-seq-compare --no-h2d --no-cpu-all-cores --no-rccl-selfcheck --no-live-traffic --no-extra-legs --reference-encoded 0"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/c3_prof -- python $ROOT/bench.py $C3 > $ROOT/$OUT/bench_c3_under_rocprof.json 2> $ROOT/$OUT/bench_c3_under_rocprof.err
cd $ROOT
find $OUT/c3_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/c3_kernel_stats.csv
rm -rf $OUT/c3_prof
grep -i "idct\|ups2\|maniac\|dequant\|widen" $OUT/c3_kernel_stats.csv | cut -c1-200
FUIFGPU_FUSE_DEQUANT=0 timeout 300 python bench.py $C3 --no-cpu-baseline > $OUT/bench_c3_no_dequant_fusion.json 2> $OUT/bench_c3_no_dequant_fusion.err
python - <<'P'
import json
for f in ("bench_c3_under_rocprof", "bench_c3_no_dequant_fusion"):
    try:
        d = json.loads([l for l in open("gpurun_out/r5_6/%s.json" % f) if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"].get("kernel_ms"), d["roofline"]["transforms"], d["config"]["parity_roundtrip_ok"])
    except Exception as e:
        print(f, "failed", e)
P
