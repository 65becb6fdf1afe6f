#!/usr/bin/env python3
"""Time of the inverse-transform schedule (fuifgpu_batch_undo_transforms) over a batch, from the library's own events.

  python tools/transform_time.py n_images [w h]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_inputs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
inputs = make_inputs(8, w, h, 3, 8, 1000, "/tmp/fuif_bench_cache")
import fuif_amd  # noqa: E402

blobs = [inputs[i % len(inputs)][1] for i in range(n)]
plan = fuif_amd.Plan(blobs[0])
batch = fuif_amd.Batch(plan, n, sum(len(b) for b in blobs))
batch.upload(blobs)
for rep in range(3):
    batch.decode(); batch.sync()      # undo_transforms consumes the coefficients: once per decode
    batch.undo_transforms(); batch.sync()
    t = batch.timing()
    px = n * w * h
    # every squeeze step reads and writes each sample of its output once (8 B), YCoCg 24 B per pixel
    print("%d x %dx%d  inverse transforms %.1f ms  (entropy %.1f ms)  %.1f Gpx/s" % (n, w, h, t[1], t[0], px / t[1] / 1e6), flush=True)
st, _ = batch.status()
assert not st.any()
