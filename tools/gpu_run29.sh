#!/bin/bash
# round 2: forward YCoCg / Squeeze kernels and the writer's GPU path on hardware (GPU test tier), time of the forward path at 4K
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/run29
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu | tee $OUT/forward_time.txt
import time, numpy as np, fuif_amd
from fuif_amd.synth import photographic
img = photographic(3840, 2160, 3, 8, seed=1)
fuif_amd.encode_image(img[:, :64, :64], 8, tree_mode=0, gpu_forward=True)   # context
for gpu in (False, True, False, True):
    t = time.time(); b = fuif_amd.encode_image(img, 8, tree_mode=0, gpu_forward=gpu); dt = time.time() - t
    print("3840x2160x3 encode, tree_mode 0, forward transforms on the %s: %.2f s, %d bytes" % ("GPU" if gpu else "host", dt, len(b)))
PY
