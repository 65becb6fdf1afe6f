#!/bin/bash
# Run on the GPU box (via gpurun): SQ issue/wait counters of k_maniac_decode on a batch small enough for the
# counter passes to finish.  One rocprofv3 --pmc pass per line (8 SQ slots each); no trace
# domains are combined with --pmc.   usage: tools/pmc_sq.sh <tag> <n_images> <w> <h> [seq]
set -u
TAG=${1:-r2}; N=${2:-256}; W=${3:-1920}; H=${4:-1080}; MODE=${5:-groups}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARG="$N $W $H"; [ "$MODE" = "seq" ] && ARG="$ARG seq"
CMD="python $ROOT/tools/occupancy_probe.py $ARG"
export REPS=1
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM" \
           "SQ_INST_CYCLES_SALU SQ_INSTS SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $SET --output-format csv -d $OUT/pass$i -- $CMD > $OUT/pass$i.log 2>&1
    echo "pass $i rc=$? : $SET"
done
python $ROOT/tools/summarize_sq.py $OUT k_maniac_decode > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
