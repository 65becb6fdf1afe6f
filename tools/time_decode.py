#!/usr/bin/env python3
"""Time k_maniac_decode (+ the inverse transforms) on n replicas of the bench pictures, any build of the library:

  [FUIF_AMD_LIB=build/libfuifgpu_x.so] python tools/time_decode.py n [w h] [--no-index] [--reps r] [--check] [--dct420]

--dct420: the JPEG-transcode shape of BASELINE config C3 (YCbCr + 4:2:0 + DCT + Quantize) instead of YCoCg + Squeeze.

Prints the HIP-event time of each launch and Mpixels/s.  --check compares image 0 and the last image with the generator's pixels.
ANALYSIS TOOLING; the judged numbers come from bench.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_inputs  # noqa: E402
import fuif_amd  # noqa: E402

args = [a for i, a in enumerate(sys.argv[1:]) if not a.startswith("--") and sys.argv[i] != "--reps"]
flags = [a for a in sys.argv[1:] if a.startswith("--")]
n = int(args[0]) if args else 1024
w, h = (int(args[1]), int(args[2])) if len(args) > 2 else (3840, 2160)
reps = 2
for i, f in enumerate(sys.argv):
    if f == "--reps":
        reps = int(sys.argv[i + 1])
k = 8
kind = "dct420" if "--dct420" in flags else "squeeze"
inputs = make_inputs(k, w, h, 3, 8, 1000, os.environ.get("FUIF_BENCH_CACHE", "/tmp/fuif_bench_cache"), kind)
blobs = [inputs[i % k][1] for i in range(n)]
plan = fuif_amd.Plan(blobs[0])
batch = fuif_amd.Batch(plan, n, sum(len(b) for b in blobs))
if "--no-index" in flags:
    batch.set_group_parallel(False)
batch.upload(blobs)
for r in range(reps):
    batch.decode(); batch.undo_transforms(); batch.sync()
    d, t = batch.timing()
    print("%s n=%d %dx%d %s: entropy %.1f ms, transforms %.1f ms -> %.1f Mpx/s" % (
        os.path.basename(os.environ.get("FUIF_AMD_LIB", "libfuifgpu.so")), n, w, h, "per image" if "--no-index" in flags else "indexed",
        d, t, n * w * h / 1e3 / (d + t)), flush=True)
st, used = batch.status()
assert "--no-status" in flags or not st.any(), st[st != 0][:8]
if "--check" in flags and kind == "squeeze":
    from fuif_amd.synth import photographic
    for i in (0, n - 1):
        img = photographic(w, h, 3, 8, seed=inputs[i % k][0])
        out = batch.out_planes(i)
        assert all(np.array_equal(out[c], img[c]) for c in range(3)), "image %d differs from the source pixels" % i
    print("checked: images 0 and %d equal the generator's pixels" % (n - 1))
