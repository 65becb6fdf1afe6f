#!/bin/bash
# build a diagnostic variant of the library into build/ (git-ignored, travels to the GPU box):
#   tools/build_variant.sh stats -DFUIF_STATS            -> build/libfuifgpu_stats.so (scheduler statistics + tile log)
#   tools/build_variant.sh prof -DFUIF_PROF -DFUIF_STATS -> build/libfuifgpu_prof.so  (per-phase cycle counters)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $ROOT/build
cd $ROOT/fuif_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value "$@" -o $ROOT/build/libfuifgpu_$name.so plan.cpp index.cpp writer.cpp maniac_decode.hip maniac_encode.hip transforms.hip capi.hip
echo built $ROOT/build/libfuifgpu_$name.so
