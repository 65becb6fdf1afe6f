#!/bin/bash
# Round 5, GPU session 1: (a) the new parity tests -- reference-ENCODED streams at 1080p / 4K (trees > 4095 nodes), the 19 327-node golden fixture, the
# lifted node cap of the writer, the asm symbol decoder against its specification, the checksum export; (b) consecutive 1024-picture launches overlapped on two
# HIP streams, entropy only, staggered and at once; (c) the new default bench.py (overlapped steps, every step verified, reference-encoded leg).
#   gpurun --timeout 1500 -- bash tools/experiments/r5_1_overlap_and_refenc.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=gpurun_out/r5_1
mkdir -p $OUT
(time timeout 900 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_fast_symbol.py tests/test_gpu_reference_encoded.py \
    "tests/test_gpu_transform_exports.py::test_plane_checksums_export" tests/test_gpu_parity.py::test_golden_fixtures_bit_exact \
    tests/test_gpu_group_parallel.py::test_reference_written_files_indexed_after_the_fact tests/test_gpu_group_parallel.py::test_add_group_index_copies_refused_streams_through) > $OUT/tests.txt 2>&1
tail -n 6 $OUT/tests.txt
timeout 400 python tools/pipeline_decode.py 1024 --launches 6 --stagger 3.6 --rounds 1 2>&1 | grep -v amdgpu | tee $OUT/entropy_stagger36.txt
timeout 300 python tools/pipeline_decode.py 1024 --launches 6 --stagger 0 --rounds 1 --only-pipelined 2>&1 | grep -v amdgpu | tee $OUT/entropy_at_once.txt
(time timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-all-cores) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 6000 $OUT/bench_default.json; tail -n 8 $OUT/bench_default.err | grep -v amdgpu
