#!/bin/bash
# Run on the GPU box (via gpurun): the rocprofv3 evidence of round 6.
#   1. kernel-trace + stats of the RESIDENT path (`bench.py --no-overlap --steps 2 --warmup 1`: launches alone on the device) -- the average duration of
#      k_maniac_decode must agree with `roofline.launch_ms_alone` (HIP events) of the default line;
#   2. the same of the DEFAULT timed region (overlapped steps: launches share the device with their neighbours, so their own durations are longer than ms_per_step);
#   3. C3 as specified (1024 x 4K, sigma-3 pixels, q90 4:2:0), resident path: the fused JPEG-transcode chain kernel by kernel;
#   4. the C5 shape on one GPU (2048 mixed 1080p pictures) and the default line itself (live PMC traffic inside it).
# Outputs: gpurun_out/prof_r6/ ; the summary printed at the end is what profiles/r6_rocprofv3_summary.txt holds.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-seq-compare --no-h2d --no-live-traffic --no-extra-legs --reference-encoded 0 --no-rccl-selfcheck"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/alone -- python $ROOT/bench.py $COMMON --no-overlap --steps 2 --warmup 1 > $OUT/bench_alone.json 2> $OUT/bench_alone.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/overlapped -- python $ROOT/bench.py $COMMON --steps 4 --warmup 1 --alone-steps 1 > $OUT/bench_overlapped.json 2> $OUT/bench_overlapped.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -- python $ROOT/bench.py --workload c3 --no-overlap --steps 3 --warmup 1 $COMMON > $OUT/bench_c3.json 2> $OUT/bench_c3.err
cd $ROOT
for d in alone overlapped c3; do find $OUT/$d -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${d}_kernel_stats.csv; rm -rf $OUT/$d; done
timeout 300 python bench.py --workload c5 --batch 2048 --steps 2 --warmup 1 --no-rccl-selfcheck > $OUT/bench_c5_2048.json 2> $OUT/bench_c5_2048.err
python - <<'P' | tee $OUT/summary.txt
import csv, json, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "prof_r6")
print("# tools/collect_profiles_r6.sh on the MI355X (round 6): rocprofv3 --kernel-trace --stats --output-format csv of bench.py, the library at HEAD")
for name, what in (("alone", "RESIDENT path, launches alone: bench.py --no-overlap --steps 2 --warmup 1"), ("overlapped", "DEFAULT timed region: bench.py --steps 4 --warmup 1 --alone-steps 1 (5 overlapped launches + 2 alone)"),
                   ("c3", "C3 (1024 x 4K, sigma 3, q90 4:2:0), resident path: bench.py --workload c3 --no-overlap --steps 3 --warmup 1")):
    print("== %s" % what)
    try:
        rows = list(csv.DictReader(open(os.path.join(out, name + "_kernel_stats.csv"))))
        for r in rows[:14]:
            if "at::native" in r["Name"] and float(r["Percentage"]) < 0.3:
                continue
            print("  %-70s calls %5s  total %14s ns  avg %16s ns  %7s %%" % (r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
        d = json.loads([l for l in open(os.path.join(out, "bench_%s.json" % name)) if l.startswith("{")][-1])
        rf = d["roofline"]
        print("  bench.py under the tracer: value %.1f Mpixels/s, ms_per_step %.1f, launch by HIP events %s ms, transforms %.1f ms" % (
            d["value"], d["ms_per_step"], rf.get("kernel_ms", rf.get("launch_ms_alone")), rf["transforms"]["ms"]))
    except Exception as e:
        print("  (failed: %s)" % e)
try:
    d = json.loads([l for l in open(os.path.join(out, "bench_c5_2048.json")) if l.startswith("{")][-1])
    print("== C5 shape on one GPU (2048 mixed 1080p pictures): %.1f Mpixels/s, entropy %.1f ms + transforms %.1f ms per step, CPU 1 thread %.2f" % (
        d["value"], d["config"]["entropy_kernel_ms"], d["config"]["transform_ms"], d.get("cpu_baseline", {}).get("value", 0)))
except Exception as e:
    print("== C5: failed", e)
P
(time timeout 900 python bench.py --steps 10 --warmup 2) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 1500 $OUT/bench_default.json
