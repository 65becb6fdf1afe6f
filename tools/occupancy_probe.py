#!/usr/bin/env python3
"""How does k_maniac_decode throughput change with the number of resident waves per SIMD?

  FUIF_AMD_LIB=<lib built with -DFUIF_LDS_SUPER=k> python tools/occupancy_probe.py n_streams [w h]
prints kernel ms and Mpx/s for one launch over n_streams streams (8 distinct images, replicated)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_inputs  # noqa: E402

w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
inputs = make_inputs(8, w, h, 3, 8, 1000, "/tmp/fuif_bench_cache")
import fuif_amd  # noqa: E402

for n in [int(a) for a in sys.argv[1].split(",")]:
    blobs = [inputs[i % len(inputs)][1] for i in range(n)]
    plan = fuif_amd.Plan(blobs[0])
    batch = fuif_amd.Batch(plan, n, sum(len(b) for b in blobs))
    batch.upload(blobs)
    for rep in range(2):
        batch.decode(); batch.sync()
        t = batch.timing()
        print("lib %s streams %5d  %dx%d  decode kernel %9.1f ms  %8.1f Mpx/s" % (
            os.path.basename(os.environ.get("FUIF_AMD_LIB", "libfuifgpu.so")), n, w, h, t[0], n * w * h / t[0] / 1e3), flush=True)
    st, _ = batch.status()
    assert not st.any(), st[:8]
    del batch
