#!/bin/bash
# Round 5: the whole -m gpu suite on the FINAL library (both wide instantiations), with the round's last GPU minutes.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r5_15
(time timeout 270 python -m pytest tests/ -q -m gpu -p no:cacheprovider -x) > gpurun_out/r5_15/gpu_tests.txt 2>&1
tail -n 6 gpurun_out/r5_15/gpu_tests.txt
