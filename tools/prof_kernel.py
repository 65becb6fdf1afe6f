#!/usr/bin/env python3
"""Per-phase cycle breakdown of k_maniac_decode (diagnostic -DFUIF_PROF build).

  tools/build_variant.sh prof -DFUIF_PROF ;  python tools/prof_kernel.py [n_streams [w h]]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FUIF_AMD_LIB", os.path.join(ROOT, "build", "libfuifgpu_prof.so"))   # tools/build_variant.sh prof -DFUIF_PROF
import fuif_amd  # noqa: E402
from bench import make_inputs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
inputs = make_inputs(min(n, 8), w, h, 3, 8, 1000, os.environ.get("FUIF_BENCH_CACHE", "/tmp/fuif_bench_cache"))
blobs = [inputs[i % len(inputs)][1] for i in range(n)]
plan = fuif_amd.Plan(blobs[0])
batch = fuif_amd.Batch(plan, n, sum(len(b) for b in blobs))
batch.upload(blobs)
t0 = time.time(); batch.decode(); batch.sync(); dt = time.time() - t0
prof = batch.profile().astype(np.float64)
nsym = plan.info.coef_elems
names = ["vector phase", "property patch", "tree walk", "leaf switch", "symbol decode", "pixel rest"]
print("streams %d  kernel %.2f s  -> %.3f us/symbol/stream" % (n, dt, dt / nsym * 1e6))
tot = prof[:, :6].sum(axis=1).mean()
for k in range(6):
    c = prof[:, k].mean()
    print("  %-16s %8.1f cycles/symbol  %5.1f %%" % (names[k], c / nsym, 100 * c / tot))
print("  %-16s %8.1f cycles/symbol (instrumented)" % ("total", tot / nsym))
seg = prof[:, 7].mean()
print("  shader clock during run segments: %.2f GHz (cycle counter against the 100 MHz real-time counter)" % (prof[:, 7].sum() / (prof[:, 6].sum() * 10.0)))
try:
    ss = batch.sched_stats()
except fuif_amd.FuifGpuError:
    ss = np.zeros(8, np.uint64)   # a build without -DFUIF_STATS
print("  run segments (pick-up to suspension / end): %.1f cycles/symbol; scheduler busy %.0f wavefront-seconds -> cycle counter at %.2f GHz" % (
    seg / nsym, float(ss[6]) / 1e8, (prof[:, 7].sum() / (float(ss[6]) / 1e8) / 1e9) if ss[6] else 0.0))
