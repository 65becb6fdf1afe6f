#!/usr/bin/env python3
"""Schedule of one dense k_maniac_decode launch from the kernel's tile log (fuifgpu_batch_tile_log).

  FUIF_AMD_LIB=build/libfuifgpu_stats.so python tools/tile_timeline.py n_images [w h]   (tools/build_variant.sh stats -DFUIF_STATS:
  the release library records no tile log; FUIFGPU_TILE_ORDER=group for the round-1 list order)
Prints, per channel group, when its tiles start / end and how long they waited for rows of other tiles; the number
of tiles running over time; and how evenly the SIMDs finish."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_inputs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
inputs = make_inputs(8, w, h, 3, 8, 1000, os.environ.get("FUIF_BENCH_CACHE", "/tmp/fuif_bench_cache"))
import fuif_amd  # noqa: E402

blobs = [inputs[i % len(inputs)][1] for i in range(n)]
plan = fuif_amd.Plan(blobs[0])
batch = fuif_amd.Batch(plan, n, sum(len(b) for b in blobs))
batch.upload(blobs)
batch.tile_log()            # arms logging
batch.decode(); batch.sync()
ms = batch.timing()[0]
log = batch.tile_log().astype(np.float64)
raw = batch.tile_log()
if os.environ.get("TILE_LOG_NPY"):
    np.save(os.environ["TILE_LOG_NPY"], raw)
img = (raw[:, 0] >> np.uint64(32)).astype(np.int64)
ch = (raw[:, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64)
t0 = raw[:, 1].astype(np.float64); t1 = raw[:, 2].astype(np.float64)
running = (raw[:, 3] & np.uint64(0xFFFFFFFFFFFF)).astype(np.float64)   # ticks some wavefront was running the tile
waited = (t1 - t0) - running                                             # suspended, or ready and not picked up yet
key = ((raw[:, 3] >> np.uint64(48)) & np.uint64(0xFFF)).astype(np.int64)      # CU; bits 60..61: the SIMD of the tile's last run segment
base = t0.min()
t0 = (t0 - base) / 1e5; t1 = (t1 - base) / 1e5; waited /= 1e5      # ms (100 MHz ticks)
print("launch %.1f ms by HIP events; tile log spans %.1f ms; %d tiles, %d images, %d SIMD keys" % (ms, t1.max(), len(raw), n, len(set(key.tolist()))))
print("%-6s %6s %9s %9s %9s %9s %9s" % ("group", "tiles", "start", "end", "run ms", "waited", "last end"))
for c in sorted(set(ch.tolist())):
    m = ch == c
    if (t1[m] - t0[m]).mean() < 20 and c < 40:
        continue
    print("c%-5d %6d %9.0f %9.0f %9.0f %9.0f %9.0f" % (c, m.sum(), t0[m].mean(), t1[m].mean(), (t1[m] - t0[m]).mean(), waited[m].mean(), t1[m].max()))
print("total tile-time %.0f s, of which waiting for other tiles %.0f s (%.1f %%)" % ((t1 - t0).sum() / 1e3, waited.sum() / 1e3, 100 * waited.sum() / (t1 - t0).sum()))
T = t1.max()
print("tiles running over time (of %d wavefront slots):" % min(len(raw), 4096))
for q in np.linspace(0, T, 21)[:-1]:
    print("  t=%7.0f ms  %5d" % (q, int(((t0 <= q) & (t1 > q)).sum())))
ends = {}
for k, e in zip(key.tolist(), t1.tolist()):
    ends[k] = max(ends.get(k, 0.0), e)
e = np.array(sorted(ends.values()))
print("per-SIMD finish time: min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f ms" % (e.min(), np.percentile(e, 10), np.median(e), np.percentile(e, 90), e.max()))
try:
    ss = batch.sched_stats()
except fuif_amd.FuifGpuError:   # a -DFUIF_TILELOG / -DFUIF_PROF build: the tile log without the scheduler counters
    ss = np.zeros(8, np.uint64)
print("scheduler: idle %.0f wavefront-seconds, picking %.0f, spinning inside tiles %.0f; %d pick-ups, %d suspensions, %d tiles without a context area" % (float(ss[0]) / 1e8, float(ss[3]) / 1e8, float(ss[4]) / 1e8, int(ss[1]), int(ss[2]), int(ss[5])))
print("scheduler: busy (pick-up to next look) %.0f wavefront-seconds, wavefront lifetime %.0f" % (float(ss[6]) / 1e8, float(ss[7]) / 1e8))
st, _ = batch.status()
assert not st.any()
