#!/usr/bin/env python3
"""Round 6, session 7: BASELINE config C4's launch (its longest channel group = 33.5 M symbols on one range coder, the device mostly idle) on the dense
configuration (shipped) and on the wide configuration with 20 LDS-resident supernodes per wavefront + the context scheduler (FUIFGPU_EXP_CFG=0, an experiment
switch of capi.hip).  n pictures of 8192x8192x4, 14 bit (the launch lasts as long for 8 as for 256).  ANALYSIS TOOLING."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import make_inputs  # noqa: E402
import fuif_amd  # noqa: E402
from fuif_amd.synth import photographic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w = h = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
k = 2
inputs = make_inputs(k, w, h, 4, 14, 7000, os.environ.get("FUIF_BENCH_CACHE", "/tmp/fuif_bench_cache"), "squeeze_raw")
blobs = [inputs[i % k][1] for i in range(n)]
plan = fuif_amd.Plan(blobs[0])
batch = fuif_amd.Batch(plan, n, sum(len(b) for b in blobs))
for label, cfg in (("dense (shipped)", None), ("wide, 20 LDS supernodes (two wavefronts per SIMD)", "0"), ("wide, 58 LDS supernodes (one wavefront per SIMD)", "2")):
    if cfg is None:
        os.environ.pop("FUIFGPU_EXP_CFG", None)
    else:
        os.environ["FUIFGPU_EXP_CFG"] = cfg
    batch.upload(blobs)
    batch.decode(); batch.undo_transforms(); batch.sync()
    d, t = batch.timing()
    st, _ = batch.status()
    img = photographic(w, h, 4, 14, seed=inputs[(n - 1) % k][0])
    out = batch.out_planes(n - 1)
    ok = not st.any() and all(np.array_equal(out[c], img[c]) for c in range(4))
    print("%-50s n=%d %dx%dx4: entropy %.1f ms, transforms %.1f ms -> %.1f Mpixels/s at n = 256; last picture %s" % (
        label, n, w, h, d, t, 256 * w * h / 1e3 / (d + t * 256 / n), "== source pixels" if ok else "DIFFERS"), flush=True)
