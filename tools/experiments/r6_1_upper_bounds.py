#!/usr/bin/env python3
"""Round 6, session 1: upper bound of a hybrid launch -- what the wave-serial kernel delivers when the tiles below a size are FREE
(their planes are left in place from a previous decode of the same batch, their progress words preset to 'final':
FUIFGPU_EXP_SKIP_SAMPLES, an experiment switch of capi.hip that only exists for this session) and at 3..6 wavefronts per SIMD.
ANALYSIS TOOLING."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import make_inputs  # noqa: E402
import fuif_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w, h, k = 3840, 2160, 8
inputs = make_inputs(k, w, h, 3, 8, 1000, os.environ.get("FUIF_BENCH_CACHE", "/tmp/fuif_bench_cache"), "squeeze")
blobs = [inputs[i % k][1] for i in range(n)]
plan = fuif_amd.Plan(blobs[0])
batch = fuif_amd.Batch(plan, n, sum(len(b) for b in blobs), streaming=True, tmp_images=8)


def run(label, skip, waves, reps=2):
    for key, v in (("FUIFGPU_EXP_SKIP_SAMPLES", skip), ("FUIFGPU_EXP_WAVES_PER_SIMD", waves)):
        if v:
            os.environ[key] = str(v)
        else:
            os.environ.pop(key, None)
    batch.upload(blobs)
    ts = []
    for r in range(reps):
        batch.decode(); batch.sync()
        ts.append(batch.timing()[0])
    st, _ = batch.status()
    ok = not st.any()
    # the coefficient planes of the first and the last picture against the full decode's
    same = all(np.array_equal(a, b) for i in (0, n - 1) for a, b in zip(batch.coef_planes(i), ref[i])) if ref else True
    print("%-58s %s ms  status %s  planes %s" % (label, " ".join("%8.1f" % t for t in ts), "ok" if ok else "FLAGGED", "equal" if same else "DIFFER"), flush=True)


ref = {}
run("all 61 tiles per picture, 6 per SIMD (the shipped launch)", 0, 0)
ref = {i: batch.coef_planes(i) for i in (0, n - 1)}
quick = bool(os.environ.get("R6_QUICK"))
if quick:
    run("tiles <= 1.04 M samples free (6 per picture left), 6 per SIMD", 1100000, 0)
    run("tiles <= 2.07 M samples free (3 per picture left), 6 per SIMD", 2100000, 0)
    sys.exit(0)
for waves in (0, 5, 4, 3):
    wl = "%d per SIMD" % waves if waves else "6 per SIMD"
    run("tiles <= 0.52 M samples free (9 per picture left), " + wl, 600000, waves)
    run("tiles <= 1.04 M samples free (6 per picture left), " + wl, 1100000, waves)
    run("tiles <= 2.07 M samples free (3 per picture left), " + wl, 2100000, waves)
run("all tiles, 5 per SIMD", 0, 5)
run("all tiles, 4 per SIMD", 0, 4)
