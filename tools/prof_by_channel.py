#!/usr/bin/env python3
"""Per-phase cycles of k_maniac_decode per CHANNEL GROUP (a -DFUIF_PROF -DFUIF_PROF_BY_CHANNEL build: tools/build_variant.sh profch
-DFUIF_PROF -DFUIF_PROF_BY_CHANNEL): where the long tiles (the launch's critical path) spend their cycles, as opposed to the average symbol.

  FUIF_AMD_LIB=build/libfuifgpu_profch.so python tools/prof_by_channel.py n_streams [w h]     ANALYSIS TOOLING"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FUIF_AMD_LIB", os.path.join(ROOT, "build", "libfuifgpu_profch.so"))
import fuif_amd  # noqa: E402
from bench import make_inputs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
inputs = make_inputs(min(n, 8), w, h, 3, 8, 1000, os.environ.get("FUIF_BENCH_CACHE", "/tmp/fuif_bench_cache"))
blobs = [inputs[i % len(inputs)][1] for i in range(n)]
plan = fuif_amd.Plan(blobs[0])
batch = fuif_amd.Batch(plan, n, sum(len(b) for b in blobs))
batch.upload(blobs)
batch.decode(); batch.sync()
ms = batch.timing()[0]
prof = batch.profile().astype(np.float64)
ch = plan.coded_channels
print("%s: %d streams, kernel %.1f ms" % (os.path.basename(os.environ["FUIF_AMD_LIB"]), n, ms))
print("%-5s %10s %8s %8s %8s %8s %8s %8s %8s %9s %7s" % ("group", "symbols", "vector", "patch", "walk", "switch", "decode", "rest", "total", "us/symbol", "GHz"))
for c in range(min(len(ch), n)):
    sym = ch[c]["w"] * ch[c]["h"] * n
    if sym < n * 200000:
        continue
    r = prof[c, :6] / sym
    wall_us = prof[c, 6] / 100.0 / sym          # slot 6: 100 MHz ticks of the run segments
    ghz = prof[c, 7] / (prof[c, 6] * 10.0) if prof[c, 6] else 0.0   # slot 7: shader cycles of the same segments
    print("c%-4d %10d %8.0f %8.0f %8.0f %8.0f %8.0f %8.0f %8.0f %9.3f %7.2f" % (c, sym // n, r[0], r[1], r[2], r[3], r[4], r[5], r.sum(), wall_us, ghz))
