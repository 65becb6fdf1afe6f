#!/bin/bash
# Round 5, GPU session 3: (a) the fused JPEG-transcode chain (dequantisation in the iDCT loads, upsampling + YCbCr in one kernel) against the oracle on hardware;
# (b) C3 as specified (sigma-3 pixels, q90, 4:2:0) at 1024 x 4K: rocprofv3 kernel statistics of the resident path, A/B of the fusions; (c) the default bench line
# (overlapped steps, overlapped one-wavefront-per-picture leg on the 20-supernode wide configuration, reference-encoded leg, the small C3 / C4 / C5 legs).
#   gpurun --timeout 1500 -- bash tools/experiments/r5_3_c3_and_default_bench.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r5_3
mkdir -p $OUT
(time timeout 600 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_synthetic.py::test_jpeg_like_chain_fused_and_unfused_match_oracle \
    tests/test_gpu_parity.py::test_golden_fixtures_bit_exact tests/test_gpu_group_parallel.py::test_jpeg_like_indexed tests/test_gpu_parity.py::test_packed_output_is_the_pam_payload \
    "tests/test_gpu_transform_exports.py") > $OUT/tests.txt 2>&1
tail -n 5 $OUT/tests.txt
C3="--workload c3 --no-overlap --steps 3 --warmup 1 --no-seq-compare --no-h2d --no-cpu-all-cores --no-rccl-selfcheck"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/c3_prof -o c3 -- python $ROOT/bench.py $C3 > $ROOT/$OUT/bench_c3_under_rocprof.json 2> $ROOT/$OUT/bench_c3_under_rocprof.err
cd $ROOT
find $OUT/c3_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/c3_kernel_stats.csv
head -n 14 $OUT/c3_kernel_stats.csv
rm -rf $OUT/c3_prof
FUIFGPU_FUSE_DEQUANT=0 FUIFGPU_FUSE_YCBCR=0 timeout 300 python bench.py $C3 --no-cpu-baseline > $OUT/bench_c3_unfused.json 2> $OUT/bench_c3_unfused.err
python - <<'P'
import json
for f in ("bench_c3_under_rocprof", "bench_c3_unfused"):
    try:
        d = json.loads([l for l in open("gpurun_out/r5_3/%s.json" % f) if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"].get("kernel_ms"), d["roofline"]["transforms"], d["config"].get("bits_per_pixel"), d["config"]["parity_roundtrip_ok"])
    except Exception as e:
        print(f, "failed", e)
P
(time timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-all-cores) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 3000 $OUT/bench_default.json; tail -n 6 $OUT/bench_default.err | grep -v amdgpu
