#!/bin/bash
# SQ-level PMC profile of k_maniac_decode on a small batch (issue vs wait breakdown)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/sq_counters.txt
wc -l $OUT/sq_counters.txt
B="python $ROOT/tools/quick_decode.py 64"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d $OUT/p1 -- $B > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU --output-format csv -d $OUT/p2 -- $B > $OUT/p2.log 2>&1
python - <<PY
import csv, glob
for d in ("p1","p2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = {}
        for r in csv.DictReader(open(f)):
            if "maniac" not in r.get("Kernel_Name",""): continue
            agg[r["Counter_Name"]] = agg.get(r["Counter_Name"],0.0) + float(r["Counter_Value"])
        for k,v in sorted(agg.items()): print("%-24s %18.0f" % (k, v))
PY
tail -3 $OUT/p1.log
