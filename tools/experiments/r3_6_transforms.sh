#!/bin/bash
# round 3, session 6: the inverse-transform kernels of VERDICT r2 item 7, every variant timed on 128 x 3840x2160 in one session.
#   horizontal unsqueeze: one lane per row (round 2) against 64-row tiles through wave-private LDS (FUIFGPU_HSQUEEZE_TILES=1)
#   last chroma unsqueeze + YCoCg fused into one pass (FUIFGPU_FUSE_YCOCG, default on) against three ops
#   vertical unsqueeze: 1 / 4 / 8 row pairs per step (build/libfuifgpu_vs1.so, _vs4.so, the product)
#   iDCT with 22 shared products per 8-point pass, not contracted: the JPEG-transcode shape (--dct420)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/${OUT_TAG:-r3_transforms}
mkdir -p $OUT
T="timeout 300 python tools/time_decode.py 128 3840 2160 --reps 2"
{
echo "== C2 (YCoCg + Squeeze), 128 x 4K"
for fuse in ${FUSE_SET:-0 1}; do for tiles in 0 1; do
  echo "-- FUIFGPU_FUSE_YCOCG=$fuse FUIFGPU_HSQUEEZE_TILES=$tiles"
  FUIFGPU_FUSE_YCOCG=$fuse FUIFGPU_HSQUEEZE_TILES=$tiles $T --check 2>&1 | grep -v amdgpu
done; done
for v in ${VS_SET:-vs1 vs4}; do
  echo "-- $v (FUIFGPU_FUSE_YCOCG=1 FUIFGPU_HSQUEEZE_TILES=1)"
  FUIF_AMD_LIB=$ROOT/build/libfuifgpu_$v.so FUIFGPU_HSQUEEZE_TILES=1 $T --check 2>&1 | grep -v amdgpu
done
echo "== C3 (YCbCr + 4:2:0 + DCT + Quantize), 128 x 4K"
$T --dct420 2>&1 | grep -v amdgpu
} | tee $OUT/times.txt
cd /tmp && export TMPDIR=/tmp
trace() {   # name tiles fuse [flag]
  FUIFGPU_HSQUEEZE_TILES=$2 FUIFGPU_FUSE_YCOCG=$3 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_$1 -- python $ROOT/tools/time_decode.py 128 3840 2160 --reps 2 ${4:-} > $ROOT/$OUT/trace_$1.log 2>&1
  f=$(find $ROOT/$OUT/trace_$1 -name "*kernel_stats.csv" | head -1)
  echo "== kernel stats $1"; [ -n "$f" ] && head -12 "$f" | cut -c1-160
}
{
[ -z "${SKIP_UNFUSED_TRACE:-}" ] && trace c2_rows_unfused 0 0
trace c2_tiles_fused 1 1
trace c3 1 1 --dct420
} | tee $ROOT/$OUT/kernel_stats.txt
