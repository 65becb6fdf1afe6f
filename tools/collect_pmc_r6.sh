#!/bin/bash
# Run on the GPU box (via gpurun): round 6's calibration of FETCH_SIZE / WRITE_SIZE on the access patterns of the COMPACT context layout (tools/ubench_gather.hip:
# 256-byte narrow supernodes read as 4 bytes x 64 lanes, 32-byte compact leaves read / written as 2 bytes x 16 lanes, and round 3's 64-byte / 512-byte patterns
# again), and the two counters over one 1024 x 4K launch each.  tools/summarize_pmc_r6.py writes the JSON bench.py takes its read factor from
# (profiles/r6_pergroup_pmc_traffic.json).  Separate --pmc passes, no trace domains next to them.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_r6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
G=$ROOT/build/ubench_gather_bin
for pat in read snode snode4 read32; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_${pat} -- $G $pat 20000 > $OUT/cal_${pat}.json 2> $OUT/cal_${pat}.err
done
for pat in write write32; do
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_${pat} -- $G $pat 20000 > $OUT/cal_${pat}.json 2> $OUT/cal_${pat}.err
done
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-seq-compare --no-h2d --no-live-traffic --no-extra-legs --reference-encoded 0 --no-rccl-selfcheck --no-overlap"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH --steps 1 --warmup 0 > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH --steps 1 --warmup 0 > $OUT/bench_write.json 2> $OUT/bench_write.err
python $ROOT/tools/summarize_pmc_r6.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
