#!/usr/bin/env python3
"""EXPERIMENT (round 4, measured at 128 pictures with the round's last GPU seconds: profiles/r4_pipelined_launches.txt, two launches overlap without
slowing each other; 1024 pictures = the first session of round 5): cross-batch pipelining of the entropy launch.

The 1024-picture launch is as long as its long channel groups (6.4-7 s); from t ~ 4.1 s on they are all that is left and HALF of every SIMD's
wavefront slots are empty (profiles/r4_occupancy_profile_timeline.txt: 34.7 k of 44.2 k wavefront-slot-seconds are used).  A batch cannot be
shorter than its long groups -- but the NEXT batch's launch, queued on a second HIP stream, can take the slots the retiring wavefronts of
the current one give up (idle wavefronts leave once every tile of their launch has started).  Two batch objects (own coefficient slabs and
context arenas: streaming batches, no output slab), launches alternating between them and their two streams; the second one starts
`--stagger` seconds after the first, and from then on each starts when its predecessor on the same stream ends.

  python tools/pipeline_decode.py [n_images] [--launches K] [--stagger S] [--distinct D] [--rounds R] [--size WxH] [--with-transforms] [--only-pipelined] [--no-index] [--no-streams]

--no-index: one wavefront per picture.  There a launch fills one wavefront slot per SIMD for its whole length (29 KB of LDS supernodes per wavefront:
one fits a SIMD's share); two launches can only run side by side in a build with fewer LDS supernodes (tools/build_variant.sh wide20 -DFUIF_LDS_WIDE=20:
two wavefronts per SIMD) -- if a lone wavefront's chain is latency-bound, as profiles/r4_small_batch_profile.txt says, that is up to 2x for such files.

Prints the wall time of K sequential launches and of K pipelined ones.  If the slots fill as hoped, a launch every ~5.7 s instead of
7.2 s (+25 % Mpixels/s); what it costs is a second coefficient slab + context arena (~85 GB per 1024 x 4K batch).  ANALYSIS TOOLING."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_inputs  # noqa: E402
import fuif_amd  # noqa: E402

argv = sys.argv[1:]
pos = [a for i, a in enumerate(argv) if not a.startswith("--") and (i == 0 or argv[i - 1] not in ("--launches", "--stagger", "--size", "--distinct", "--rounds", "--batches"))]
n = int(pos[0]) if pos else 1024


def opt(name, default, cast):
    return cast(argv[argv.index(name) + 1]) if name in argv else default


K = opt("--launches", 4, int)
stagger = opt("--stagger", 3.6, float)
w, h = (int(x) for x in opt("--size", "3840x2160", str).split("x"))
distinct = opt("--distinct", 8, int)
rounds = opt("--rounds", 2, int)
NB = opt("--batches", 2, int)     # batch objects / HIP streams the launches alternate between (round 6: 3 for streams without group index)
inputs = make_inputs(distinct, w, h, 3, 8, 1000, os.environ.get("FUIF_BENCH_CACHE", "/tmp/fuif_bench_cache"))
blobs = [inputs[i % distinct][1] for i in range(n)]
plan = fuif_amd.Plan(blobs[0])
L = fuif_amd.lib()
streams = [None] * NB
if "--no-streams" not in argv:
    hip = fuif_amd.hip_runtime()            # (the runtime the library itself runs on: torch's bundled copy when torch is installed)
    for k in range(NB):
        s = C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0     # hipStreamNonBlocking
        streams[k] = s
cap = sum(len(b) for b in blobs) + 4096 * n
batches = [fuif_amd.Batch(plan, n, cap, streaming=True) for _ in range(NB)]
for b, s in zip(batches, streams):
    b.set_in_flight(1 if "--in-flight-1" in argv else 2)     # (1: the 58-supernode wide configuration of a launch alone)
    if "--no-index" in argv:
        b.set_group_parallel(False)        # files as the reference CLI writes them: one wavefront per picture (the wide configuration, one per SIMD)
    b.upload(blobs, stream=s)
    b.sync(s)
px = n * w * h
# --with-transforms: every launch is followed, on its own stream, by the inverse transforms of its batch in slices into one slice-sized output
# buffer per batch (fuifgpu_batch_undo_transforms_to) -- the whole step of bench.py, nothing waits on the host; the last slice is hashed at the end
with_tr = "--with-transforms" in argv
L.fuifgpu_dev_alloc.restype = C.c_void_p
n_slice = max(1, min(n, (8 << 30) // (4 * max(plan.info.out_elems, 1))))
outs = [L.fuifgpu_dev_alloc(C.c_size_t(n_slice * plan.info.out_elems * 4)) for _ in range(NB)] if with_tr else [None] * NB
assert not with_tr or all(outs), "device allocation failed"


def transforms(b, s, out):
    for s0 in range(0, n, n_slice):
        b.undo_transforms_to(s0, min(n_slice, n - s0), out, s)


def last_slice_hash():
    import hashlib
    import numpy as np
    hs = []
    for out in outs:
        buf = np.zeros(min(n_slice * plan.info.out_elems, 4 << 20), np.int32)
        L.fuifgpu_dev_download(buf.ctypes.data_as(C.c_void_p), C.c_void_p(out), C.c_size_t(buf.size * 4))
        hs.append(hashlib.sha256(buf.tobytes()).hexdigest()[:12])
    return hs



def run(pipelined):
    t0 = time.perf_counter()
    for i in range(K):
        b, s = batches[i % NB], streams[i % NB]
        if pipelined and i == 1 and stagger > 0:
            time.sleep(stagger)                # the second launch is queued while the first is in its busy phase
        b.decode(s)                            # asynchronous: a launch waits for its predecessor on the same stream only
        if with_tr:
            transforms(b, s, outs[i % NB])
        if not pipelined:
            b.sync(s)
    for b, s in zip(batches, streams):
        b.sync(s)
    dt = time.perf_counter() - t0
    for b in batches:
        st, _ = b.status()
        assert not st.any(), st[st != 0][:8]
    return dt


if "--per-launch" in argv:
    # one host thread per stream: queue a launch, wait for it, note the time -- the spacing of the completions in the middle of the run is the steady-state
    # time per launch (the totals above include the ramp: the first launch has the device to itself, the last one too)
    import threading
    done = []
    t_start = time.perf_counter()

    def worker(k):
        b, s = batches[k], streams[k]
        if k == 1 and stagger > 0:
            time.sleep(stagger)
        for i in range(k, K, NB):
            b.decode(s)
            if with_tr:
                transforms(b, s, outs[k])
            b.sync(s)
            done.append((time.perf_counter() - t_start, i))
    th = [threading.Thread(target=worker, args=(k,)) for k in range(NB)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    done.sort()
    ends = [t for t, _ in done]
    gaps = [b - a for a, b in zip(ends, ends[1:])]
    mid = gaps[len(gaps) // 4: len(gaps) - len(gaps) // 4] or gaps
    print("per-launch completions (s): " + " ".join("%.2f" % t for t in ends))
    print("steady state: %.2f s per launch in the middle half of the run (%d gaps) -> %.1f Mpixels/s; whole run %.2f s for %d launches -> %.1f Mpixels/s (%s)" % (
        sum(mid) / len(mid), len(mid), px / (sum(mid) / len(mid)) / 1e6, ends[-1], K, K * px / ends[-1] / 1e6, "entropy + inverse transforms" if with_tr else "entropy only"), flush=True)
    for b in batches:
        st, _ = b.status()
        assert not st.any(), st[st != 0][:8]
    sys.exit(0)
tile_log = "--tile-log" in argv      # (a -DFUIF_TILELOG build: tools/build_variant.sh tilelog -DFUIF_TILELOG)
if tile_log:
    for b in batches:
        b.tile_log()                 # arms logging
for mode in ((True,) if "--only-pipelined" in argv else (False,) if "--only-sequential" in argv else (False, True)) * rounds:
    dt = run(mode)
    print("%s: %d launches of %d x %dx%d in %.2f s -> %.2f s per launch, %.1f Mpixels/s (%s)" % (
        "pipelined (%d streams, stagger %.1f s)" % (NB, stagger) if mode else "sequential", K, n, w, h, dt, dt / K, K * px / dt / 1e6,
        "entropy + inverse transforms; last output slices %s" % "/".join(last_slice_hash()) if with_tr else "entropy only"), flush=True)
print("last launches by their own events: " + " / ".join("%.0f" % b.timing()[0] for b in batches) + " ms")

if tile_log:
    # the LAST launch of each batch object on one time axis (the log's ticks are the device's 100 MHz real-time counter): per launch and per
    # channel group class when its tiles start / end and how long they ran; wavefront slots in use by each launch over time
    import numpy as np
    logs = [b.tile_log() for b in batches]
    base = min(int(l[:, 1].min()) for l in logs)
    nch = plan.info.nb_coded_channels
    sizes = np.array([c["w"] * c["h"] for c in plan.coded_channels], np.float64)
    long_ch = [c for c in range(nch) if sizes[c] >= sizes.max() * 0.99]
    print("long channel groups:", long_ch)
    rows = []
    for k, l in enumerate(logs):
        ch = (l[:, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64)
        t0 = (l[:, 1].astype(np.int64) - base) / 1e5
        t1 = (l[:, 2].astype(np.int64) - base) / 1e5
        run = (l[:, 3] & np.uint64(0xFFFFFFFFFFFF)).astype(np.float64) / 1e5
        is_long = np.isin(ch, long_ch)
        rows.append((t0, t1, run, is_long, ch))
        print("launch on batch %d: tiles %d, first start %.0f ms, last end %.0f ms" % (k, len(l), t0.min(), t1.max()))
        for name, m in (("long", is_long), ("short", ~is_long)):
            print("   %-5s tiles: start mean %.0f (min %.0f max %.0f)  end mean %.0f (max %.0f)  running %.0f ms mean of %.0f ms lifetime; sum of running time %.0f wavefront-s" % (
                name, t0[m].mean(), t0[m].min(), t0[m].max(), t1[m].mean(), t1[m].max(), run[m].mean(), (t1[m] - t0[m]).mean(), run[m].sum() / 1e3))
        for c in long_ch + [c for c in range(nch) if sizes[c] >= sizes.max() * 0.2 and c not in long_ch]:
            m = ch == c
            print("   c%-3d start %.0f end %.0f (max %.0f) running %.0f of %.0f ms" % (c, t0[m].mean(), t1[m].mean(), t1[m].max(), run[m].mean(), (t1[m] - t0[m]).mean()))
    T = max(r[1].max() for r in rows)
    print("tiles alive (started, not ended) over time: launch 0 long/short | launch 1 long/short")
    for q in np.linspace(0, T, 41)[:-1]:
        parts = []
        for t0, t1, run, is_long, ch in rows:
            alive = (t0 <= q) & (t1 > q)
            parts.append("%5d /%6d" % (int((alive & is_long).sum()), int((alive & ~is_long).sum())))
        print("  t=%7.0f ms   %s" % (q, "  |  ".join(parts)))
    if os.environ.get("TILE_LOG_NPY"):
        np.save(os.environ["TILE_LOG_NPY"], np.stack([np.pad(l, ((0, max(len(x) for x in logs) - len(l)), (0, 0))) for l in logs]))
