#!/usr/bin/env python3
"""Throughput of k_maniac_decode against resident wavefronts per SIMD and LDS-resident supernodes.

  FUIF_AMD_LIB=<lib built with -DFUIF_LDS_SUPER=k> python tools/occupancy_probe.py n_streams[,n..] [w h] [seq]
one launch over n_streams streams (8 distinct images, replicated); `seq` ignores the group index."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_inputs  # noqa: E402

w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
modes = [False] if "seq" in sys.argv else ([True, False] if "both" in sys.argv else [True])
inputs = make_inputs(8, w, h, 3, 8, 1000, "/tmp/fuif_bench_cache")
import fuif_amd  # noqa: E402

for n in [int(a) for a in sys.argv[1].split(",")]:
    blobs = [inputs[i % len(inputs)][1] for i in range(n)]
    plan = fuif_amd.Plan(blobs[0])
    batch = fuif_amd.Batch(plan, n, sum(len(b) for b in blobs))
    for par in modes:
        batch.set_group_parallel(par)
        batch.upload(blobs)
        for rep in range(int(os.environ.get("REPS", "2"))):
            batch.decode(); batch.sync()
            t = batch.timing()
            print("lib %s streams %5d %dx%d %s  decode kernel %9.1f ms  %8.1f Mpx/s" % (
                os.path.basename(os.environ.get("FUIF_AMD_LIB", "libfuifgpu.so")), n, w, h, "groups" if par else "images",
                t[0], n * w * h / t[0] / 1e3), flush=True)
        st, _ = batch.status()
        assert not st.any(), st[:8]
    batch.close()
