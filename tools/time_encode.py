#!/usr/bin/env python3
"""Time the writer's paths on n synthetic pictures (analysis tooling; first thing to run on the MI355X in round 4):

  python tools/time_encode.py n [w h] [--tree-mode m]

  host      fuif_amd.encode_image, host entropy coder (one picture after the other, one thread)
  gpu-one   the same with gpu_entropy=True (one group per launch pair, synchronous)
  gpu-batch fuif_amd.encode_images: every group of every picture in one launch pair
All three must give the same bytes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fuif_amd  # noqa: E402
from fuif_amd.synth import photographic  # noqa: E402

args = [a for i, a in enumerate(sys.argv[1:]) if not a.startswith("--") and sys.argv[i] != "--tree-mode"]
n = int(args[0]) if args else 8
w, h = (int(args[1]), int(args[2])) if len(args) > 2 else (1920, 1080)
tree_mode = 1
for i, f in enumerate(sys.argv):
    if f == "--tree-mode":
        tree_mode = int(sys.argv[i + 1])
imgs = [photographic(w, h, 3, 8, seed=9000 + i) for i in range(n)]
mpx = n * w * h / 1e6


def timed(label, fn):
    t0 = time.perf_counter()
    out = fn()
    dt = time.perf_counter() - t0
    print("%-9s %d x %dx%d tree_mode %d: %.2f s -> %.2f Mpixels/s" % (label, len(out), w, h, tree_mode, dt, len(out) * w * h / 1e6 / dt), flush=True)
    return out


host = timed("host", lambda: [fuif_amd.encode_image(im, 8, tree_mode=tree_mode, index=True) for im in imgs])
one = timed("gpu-one", lambda: [fuif_amd.encode_image(im, 8, tree_mode=tree_mode, index=True, gpu_entropy=True) for im in imgs[: max(1, n // 8)]])
batch = timed("gpu-batch", lambda: fuif_amd.encode_images(imgs, 8, tree_mode=tree_mode, index=True))
assert one == host[: len(one)], "gpu_entropy differs from the host writer"
assert batch == host, "the batch differs from the host writer"
print("identical bytes: %d streams, %.2f MB" % (n, sum(len(b) for b in host) / 1e6))
