#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the MI355X FUIF decode path on BASELINE.json's headline config.

Workload (config.workload = "C2"): a batch of 1024 3840x2160 8-bit photographic RGB images,
YCoCg+Squeeze lossless .fuif, per GPU (weak scaling: every rank decodes its own 1024 streams; the
path shards by independent images and has no data-path collective -- the only exchange is the
final gather of the decoded pictures, packed to 8-bit RGB on the GPU, to rank 0 over RCCL, outside
the timed region and reported separately).  `--workload c5` is BASELINE config 5:
8192 mixed Squeeze / DCT 1920x1080 images sharded over the ranks (strong scaling).

One "step" = one pass of the hot path over the batch with the compressed streams already resident
in HBM: entropy kernel (k_maniac_decode) + inverse-transform schedule, ending with all int32
output planes in HBM.  Inputs are synthetic (fuif_amd/synth.py, seeded) and are encoded on the
host by the product's own FUIF writer (csrc/writer.cpp) before the timed region; K distinct
images are replicated to the batch size and every replica is decoded independently.

Output: ONE JSON line on rank 0 (see the task contract), with `roofline` for the dominant kernel
and `cpu_baseline` = the reference decoder (oracle/_ref, kind "reference") or the oracle
restatement (kind "port") timed single-threaded on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


WORKLOADS = {
    # BASELINE.json configs[1] -- the headline config
    "c2": dict(channels=3, bits=8, kind="squeeze", lossless=True,
               desc="C2: batch of %d %dx%d 8-bit photographic YCoCg+Squeeze lossless per GPU"),
    # BASELINE.json configs[2] -- JPEG-transcode shape (YCbCr + 4:2:0 + 8x8 DCT + Quantize + Squeeze of DC), lossy q90
    "c3": dict(channels=3, bits=8, kind="dct420", lossless=False,
               desc="C3: batch of %d %dx%d JPEG-transcode-like (YCbCr+4:2:0+DCT+Quantize q90) lossy per GPU"),
    # BASELINE.json configs[3] -- deep-bit raw-sensor shape: 4 channels, 14 bit, Squeeze only (pass --width 8192 --height 8192
    # --batch <what fits>: one 8192x8192x4 image needs 2.2 GB of planes)
    "c4": dict(channels=4, bits=14, kind="squeeze_raw", lossless=True,
               desc="C4: batch of %d %dx%d 14-bit 4-channel Squeeze-only lossless per GPU"),
}


def _encode_one(args):
    seed, w, h, channels, bits, kind = args
    import fuif_amd
    from fuif_amd.synth import photographic
    if kind == "dct420":
        from fuif_amd.jpeglike import encode_jpeg_like
        # the C2 pixels (sigma = 3) through JPEG quality 90, 4:2:0 (rounds 1-4 used sigma = 1: a third of the bytes per stream)
        img = photographic(w, h, channels, bits, seed=seed)
        return seed, encode_jpeg_like(img, 90, True, index=True)
    img = photographic(w, h, channels, bits, seed=seed)
    return seed, fuif_amd.encode_image(img, bits, ycocg=(kind != "squeeze_raw"), tree_mode=1, index=True)


def _stream_path(cache_dir, kind, w, h, channels, bits, seed):
    return os.path.join(cache_dir, "synth_idx3_%s_%dx%dx%d_%dbit_seed%d.fuif" % (kind, w, h, channels, bits, seed))   # streams carry the group index trailer (csrc/index.cpp)


def make_inputs_many(specs, cache_dir):
    """specs: [(k, w, h, channels, bits, seed0, kind)] -> one list of (seed, stream bytes) per spec; everything that is not in the cache of this box
    session is encoded by ONE pool of host processes (largest pictures first)."""
    os.makedirs(cache_dir, exist_ok=True)
    jobs, blobs = [], {}
    for k, w, h, channels, bits, seed0, kind in specs:
        for i in range(k):
            key = (kind, w, h, channels, bits, seed0 + i)
            path = _stream_path(cache_dir, *key)
            if os.path.exists(path):
                blobs[key] = open(path, "rb").read()
            elif key not in [(j[5], j[1], j[2], j[3], j[4], j[0]) for j in jobs]:
                jobs.append((seed0 + i, w, h, channels, bits, kind))
    if jobs:
        import multiprocessing as mp
        jobs.sort(key=lambda j: -j[1] * j[2] * j[3])
        nproc = max(1, min(len(jobs), (os.cpu_count() or 2)))
        with mp.get_context("fork").Pool(nproc) as pool:
            for key, blob in pool.imap_unordered(_encode_keyed, jobs, chunksize=1):
                blobs[key] = blob
                try:    # (atomic: the ranks of a multi-GPU run share the directory, and a late rank must never read half a file)
                    final = _stream_path(cache_dir, *key)
                    part = "%s.%d.part" % (final, os.getpid())
                    with open(part, "wb") as f:
                        f.write(blob)
                    os.replace(part, final)
                except OSError:
                    pass
    return [[(seed0 + i, blobs[(kind, w, h, channels, bits, seed0 + i)]) for i in range(k)] for k, w, h, channels, bits, seed0, kind in specs]


def _encode_keyed(job):
    seed, w, h, channels, bits, kind = job
    return (kind, w, h, channels, bits, seed), _encode_one(job)[1]


def make_inputs(k, w, h, channels, bits, seed0, cache_dir, kind="squeeze"):
    """K distinct encoded streams (+ their seeds); cached on local disk inside one box session."""
    return make_inputs_many([(k, w, h, channels, bits, seed0, kind)], cache_dir)[0]


def reference_encodes_start(k, w, h, channels, bits, seed0, cache_dir):
    """`reference_encoded_streams` leg: K pictures of the headline workload encoded by the REFERENCE encoder -- the unmodified CLI
    oracle/_ref/fuif with its default flags, one background process per picture (~20 s each for 3840x2160), started before anything
    else so that they finish while the timed region runs.  Returns [(seed, path, Popen or None)] or None when the CLI is not there
    (it is built by oracle/Makefile where /root/reference exists and travels to the GPU box prebuilt)."""
    import subprocess
    from fuif_amd.synth import photographic, write_pnm
    cli = os.path.join(ROOT, "oracle", "_ref", "fuif")
    if not os.path.exists(cli) or channels != 3 or bits != 8:
        return None
    os.makedirs(cache_dir, exist_ok=True)
    env = dict(os.environ)
    if os.path.exists("/opt/conda/lib/libjpeg.so.9"):
        env["LD_PRELOAD"] = "/opt/conda/lib/libjpeg.so.9"
    jobs = []
    for i in range(k):
        seed = seed0 + i
        out = os.path.join(cache_dir, "refenc_%dx%dx%d_%dbit_seed%d.fuif" % (w, h, channels, bits, seed))
        proc = None
        if not os.path.exists(out):
            src = out[:-5] + ".ppm"
            write_pnm(src, photographic(w, h, channels, bits, seed=seed), (1 << bits) - 1)
            part = out + ".%d.part.fuif" % os.getpid()
            proc = (subprocess.Popen([cli, src, part], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL), part, src)
        jobs.append((seed, out, proc))
    return jobs


def reference_encodes_wait(jobs, timeout_s=240.0):
    """[(seed, stream bytes)] of reference_encodes_start's jobs"""
    res = []
    for seed, out, proc in jobs:
        if proc is not None:
            p, part, src = proc
            p.wait(timeout=timeout_s)
            if p.returncode != 0 or not os.path.exists(part):
                raise RuntimeError("the reference encoder failed on seed %d" % seed)
            os.replace(part, out)
            try:
                os.unlink(src)
            except OSError:
                pass
        with open(out, "rb") as f:
            res.append((seed, f.read()))
    return res


# The other BASELINE.json configs at sizes that take seconds, as extra keys of the default line (the driver's record, not only
# profiles/, shows every config with a rate, a roofline and a CPU baseline).  Full-size runs of each: --workload c3 / c4 / c5.
EXTRA_LEGS = {
    # C3 and C5 at BASELINE size since round 6 (VERDICT r5 "missing" 5): a step of 1024 JPEG-transcoded 4K pictures is half a second on the device
    "c3": dict(desc="C3 as specified: %d x 3840x2160 JPEG-transcode-like (YCbCr + 4:2:0 + 8x8 DCT + Quantize q90 + Squeeze of DC), sigma-3 pixels", n=1024,
               parts=[dict(kind="dct420", w=3840, h=2160, channels=3, bits=8, k=4, seed0=2000)], steps=3),
    # C4 at its REAL picture size since round 6: a launch is bounded by its longest channel group (33.5 M symbols on one range coder)
    # and by the memory latency all resident tiles add up to -- 8 pictures: 37.6 s, 64: 50.2 s, 256: the full-size run of profiles/r6_c4_full_size.txt
    # (profiles/r6_c4_wide_configurations.txt).  One timed launch, no warm-up launch (most of a minute each).
    "c4": dict(desc="C4 at its real picture size: %d x 8192x8192 14-bit 4-channel Squeeze-only lossless (a quarter of the 256-picture batch; one launch, no warm-up)", n=64,
               parts=[dict(kind="squeeze_raw", w=8192, h=8192, channels=4, bits=14, k=1, seed0=7000)], steps=1, warmup=0),
    "c5": dict(desc="C5 on one GPU: %d x 1920x1080 mixed Squeeze / DCT pictures (alternating; a quarter of the 8192 of the 8-GPU configuration), one launch per kind", n=2048,
               parts=[dict(kind="squeeze", w=1920, h=1080, channels=3, bits=8, k=2, seed0=3000), dict(kind="dct420", w=1920, h=1080, channels=3, bits=8, k=2, seed0=4000)], steps=2),
}


def run_extra_leg(name, spec, streams, dev):
    """one small configuration on the resident path: every part (kind) gets its own Batch, a step = decode + inverse transforms of all parts; 1 warm-up +
    spec['steps'] timed steps; parity: lossless parts against the generator's pixels (every image), lossy parts MSE < 40 for the first replica and all
    replicas identical; roofline of the entropy kernel from its HIP events; the reference decoder on one thread on a bounded sample of the same streams"""
    import torch
    import fuif_amd
    from fuif_amd.synth import photographic
    n_parts = len(spec["parts"])
    parts = []
    for pi, part in enumerate(spec["parts"]):
        mine = [i for i in range(spec["n"]) if i % n_parts == pi]
        blobs = [streams[pi][j % len(streams[pi])][1] for j in range(len(mine))]
        plan = fuif_amd.Plan(blobs[0])
        out = torch.empty(len(blobs) * plan.info.out_elems, dtype=torch.int32, device=dev)
        batch = fuif_amd.Batch(plan, len(blobs), sum(len(b) for b in blobs), out_ptr=out.data_ptr())
        batch.upload(blobs)
        parts.append(dict(part=part, blobs=blobs, plan=plan, out=out, batch=batch))

    def step():
        for p in parts:
            p["batch"].decode()
            p["batch"].undo_transforms()

    for _ in range(spec.get("warmup", 1)):
        step()
    torch.cuda.synchronize()
    dec, tr = [], []
    t0 = time.perf_counter()
    for _ in range(spec["steps"]):
        step()
        d_sum = t_sum = 0.0
        for p in parts:
            p["batch"].sync()
            d, t = p["batch"].timing()
            d_sum += d; t_sum += t
        dec.append(d_sum); tr.append(t_sum)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ok, px, alg, alg_tr = True, 0, 0.0, 0.0
    for p in parts:
        part, plan, info = p["part"], p["plan"], p["plan"].info
        W, H, C, BITS = part["w"], part["h"], part["channels"], part["bits"]
        st, _ = p["batch"].status()
        ok = ok and not st.any()
        view = p["out"].view(len(p["blobs"]), info.out_elems)
        chans = plan.output_channels
        K = len(streams[spec["parts"].index(part)])
        for k in range(K):
            src = torch.from_numpy(photographic(W, H, C, BITS, seed=streams[spec["parts"].index(part)][k][0])).to(dev)
            first = None
            for i in range(k, len(p["blobs"]), K):
                for c, oc in enumerate(chans[:C]):
                    got = view[i, oc["offset"]: oc["offset"] + oc["w"] * oc["h"]].view(oc["h"], oc["w"])
                    if part["kind"] != "dct420":
                        ok = ok and bool(torch.equal(got, src[c]))
                    elif first is None:
                        ok = ok and (got[:H, :W].to(torch.float32) - src[c].to(torch.float32)).pow(2).mean().item() < spec.get("mse_max", 40.0)
                if part["kind"] == "dct420":
                    if first is None:
                        first = view[i].clone()
                    else:
                        ok = ok and bool(torch.equal(view[i], first))
        px += len(p["blobs"]) * W * H
        alg += sum(len(b) for b in p["blobs"]) + 2.0 * info.coef_elems * len(p["blobs"])
        alg_tr += (2.0 * info.coef_elems + 4.0 * info.out_elems) * len(p["blobs"])
    # The REAL reference decodes stream 0 of every kind (entropy + inverse transforms on the host) and every output plane of picture 0 is compared with it:
    # what makes a lossy leg bit-exact evidence instead of an MSE bound.  Pictures above 40 M samples (the 4-channel raw ones) are
    # lossless and compared with their source pixels above; the reference would take a minute on them.
    ref_compared, ref_ok = [], True
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from oracle_py import Ref
        ref_lib = Ref() if Ref.available() else None
    except Exception:
        ref_lib = None
    for p in parts:
        part, plan, info = p["part"], p["plan"], p["plan"].info
        if ref_lib is None or part["w"] * part["h"] * part["channels"] > 40e6:
            continue
        d_ref = ref_lib.decode(p["blobs"][0])
        host = p["out"].view(len(p["blobs"]), info.out_elems)[0].cpu().numpy()
        chans = plan.output_channels
        same = bool(d_ref.ok) and len(d_ref.channels) == len(chans) and all(
            np.array_equal(host[oc["offset"]: oc["offset"] + oc["w"] * oc["h"]], np.asarray(rc["data"]).reshape(-1)) for oc, rc in zip(chans, d_ref.channels))
        ref_ok = ref_ok and same
        ref_compared.append("%s %dx%d: %d planes %s" % (part["kind"], part["w"], part["h"], len(chans), "equal" if same else "DIFFER"))
    ok = ok and ref_ok
    d_avg, t_avg = float(np.mean(dec)) / 1e3, float(np.mean(tr)) / 1e3
    res = {"workload": spec["desc"] % spec["n"], "value": round(px * spec["steps"] / 1e6 / elapsed, 3), "unit": "Mpixels/s", "steps": spec["steps"], "warmup": spec.get("warmup", 1),
           "ms_per_step": round(elapsed / spec["steps"] * 1e3, 3), "entropy_kernel_ms": round(d_avg * 1e3, 3), "transforms_ms": round(t_avg * 1e3, 3),
           "bits_per_pixel": round(8.0 * sum(sum(len(b) for b in p["blobs"]) for p in parts) / px, 3), "parity_ok": bool(ok),
           "parity_check": "lossless pictures == source pixels (every image); lossy ones: all replicas identical and MSE vs source < 40; status 0; picture 0 of every kind: "
                           "every output plane == the REAL reference's decode of the same stream (reference_decode_of_stream0)",
           "reference_decode_of_stream0": ref_compared if ref_compared else ("oracle/_ref not available on this box" if ref_lib is None else
                                                                                   "not run: pictures above 40 M samples (lossless: every picture is compared with its source pixels instead)"),
           "roofline": {"bound": "hbm", "kernel": "k_maniac_decode", "achieved": round(alg / max(d_avg, 1e-9) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg / max(d_avg, 1e-9) / 1e9 / HBM_PEAK_GBS, 6), "kernel_ms": round(d_avg * 1e3, 3), "algorithmic_bytes_per_launch": int(alg),
                        "transforms": {"ms": round(t_avg * 1e3, 3), "achieved": round(alg_tr / max(t_avg, 1e-9) / 1e9, 1), "unit": "GB/s",
                                       "frac": round(alg_tr / max(t_avg, 1e-9) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": int(alg_tr)}}}
    for p in parts:
        p["batch"].close()
    del parts
    # the reference decoder, one thread, on the same streams (bounded: at least one of each part)
    t_cpu, px_cpu, kind = 0.0, 0, None
    for pi, part in enumerate(spec["parts"]):
        cb = cpu_baseline([b for _, b in streams[pi]], part["w"], part["h"], budget_s=1.5)
        n_dec = int(cb["sample"].split()[0])
        t_cpu += n_dec * part["w"] * part["h"] / 1e6 / cb["value"]
        px_cpu += n_dec * part["w"] * part["h"]
        kind = cb["kind"]
    res["cpu_baseline"] = {"value": round(px_cpu / 1e6 / t_cpu, 4), "unit": "Mpixels/s", "cores": 1, "kind": kind,
                           "sample": "%.1f s of full decodes (entropy + inverse transforms) of this leg's streams, at least one per kind, 1 thread" % t_cpu}
    res["speedup_vs_cpu_1thread"] = round(res["value"] / res["cpu_baseline"]["value"], 2)
    return res, ok


def cpu_baseline(blobs, w, h, budget_s=25.0, source=None):
    """single-thread CPU decode (entropy + inverse transforms) of the same streams on this host.
    source = (seed, channels, bits) of blobs[0] for lossless workloads: the checker's decode of that stream is compared
    with the generator's pixels -- the same pixels every GPU-decoded image is compared with, so at full size
    reference decode == source == GPU decode for that stream (not only decoder(writer(x)) == x)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle_py import Port, Ref
    if Ref.available():
        lib, kind = Ref(), "reference"
    else:
        lib, kind = Port(), "port"
    t_total, n = 0.0, 0
    for blob in blobs:
        dt, ok = lib.time_decode(blob)
        if not ok:
            raise RuntimeError("CPU baseline failed to decode a bench stream")
        t_total += dt
        n += 1
        if t_total > budget_s:
            break
    out = {"value": round(n * w * h / 1e6 / t_total, 4), "unit": "Mpixels/s", "cores": 1, "kind": kind,
           "sample": "%d of the bench's %dx%d streams, full decode (entropy + inverse transforms), 1 thread, %.1f s" % (n, w, h, t_total),
           "host": host_description()}
    if source is not None:
        import numpy as np
        from fuif_amd.synth import photographic
        seed, channels, bits = source
        img = photographic(w, h, channels, bits, seed=seed)
        d = lib.decode(blobs[0])
        out["checker_decodes_stream0_to_source_pixels"] = bool(d.ok and len(d.channels) >= channels and all(
            np.array_equal(np.asarray(d.channels[c]["data"]).reshape(h, w), img[c]) for c in range(channels)))
    return out


def host_description():
    """the GPU box's host CPU: model, hardware threads, physical cores (lscpu-free: /proc/cpuinfo)"""
    model, cores, threads = "?", set(), 0
    try:
        phys = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "?":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("processor"):
                threads += 1
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cores.add((phys, line.split(":", 1)[1].strip()))
    except OSError:
        pass
    # what this process may actually use: the affinity mask and the container's CPU quota (cgroup v2 cpu.max / v1 cfs quota)
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = None
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = round(int(q) / int(per), 2)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = round(q / per, 2)
        except (OSError, ValueError):
            pass
    return {"model": model, "hardware_threads": threads or (os.cpu_count() or 1), "physical_cores": len(cores) or None,
            "affinity_cpus": affinity, "cgroup_cpu_quota": quota}


def cpu_baseline_all_cores(paths, w, h, seconds=6.0):
    """the same reference decoder on every host core at once, one process per image; separate
    processes started with subprocess (this process already holds HIP / RCCL state: no fork)"""
    import subprocess
    host = host_description()
    cores = host["physical_cores"] or os.cpu_count() or 1     # one process per physical core (SMT siblings only add contention here)
    if host.get("affinity_cpus"):
        cores = min(cores, host["affinity_cpus"])
    if host.get("cgroup_cpu_quota"):
        cores = max(1, min(cores, int(host["cgroup_cpu_quota"])))  # a container quota below the core count is the real limit
    worker = os.path.join(ROOT, "oracle", "cpu_worker.py")
    start_at = time.time() + 6.0 + cores * 0.01          # let every worker load before the clock starts
    procs = [subprocess.Popen([sys.executable, worker, str(seconds), str(start_at), str(i)] + paths, stdout=subprocess.PIPE, text=True)
             for i in range(cores)]
    images, wall, failed, busy = 0, 0.0, False, 0.0
    deadline = start_at + seconds + 120.0
    for p in procs:
        try:
            out, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.communicate()
            failed = True
            continue
        f = out.split()
        if p.returncode != 0 or len(f) < 3:
            failed = True
            continue
        images += int(f[0])
        busy += float(f[1])
        wall = max(wall, float(f[2]))
    if failed:
        return None
    if not images:
        return None
    from oracle_py import Ref
    return {"value": round(images * w * h / 1e6 / wall, 2), "unit": "Mpixels/s", "cores": cores, "kind": "reference" if Ref.available() else "port",
            "sample": "%d full decodes of the bench's %dx%d streams by %d concurrent processes (one per usable physical core) in %.1f s; wall = the slowest worker" % (images, w, h, cores, wall),
            "decode_s_under_load": round(busy / images, 2),
            "host": host}


def pmc_traffic(batch, mode):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r*_pmc_traffic.json, collected with tools/experiments/collect_profiles.sh); None if no profile
    of this batch size exists."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("kernel") == "k_maniac_decode" and d.get("batch") == batch and d.get("mode", "images") == mode:
            best = dict(d, _file=os.path.relpath(f, ROOT))
    if best is None:
        return None, None
    # where the figure comes from, so that a stale profile is visible next to this run's own kernel time
    src = {"profile": best["_file"], "round": best.get("round"), "kernel_ms_when_profiled": best.get("kernel_ms_under_pmc"),
           "read_factor_leaf_pattern": best.get("read_factor_leaf_pattern", (best.get("calibration", {}).get("read") or {}).get("factor")),
           "read_factor_supernode_pattern": best.get("read_factor_supernode_pattern"), "read_factor_used": best.get("read_factor_used"),
           "write_factor_used": best.get("write_factor_used"),
           "range_bytes": [best.get("traffic_bytes_per_launch_all_leaf_factor"), best.get("traffic_bytes_per_launch_all_supernode_factor")]}
    return int(best["traffic_bytes_per_launch"]), src


def run_c5(args):
    """--workload c5 = BASELINE config 5: `--batch` (default 8192) mixed images, 1920x1080 8-bit RGB, alternately YCoCg+Squeeze
    lossless and JPEG-transcode-like (YCbCr + 4:2:0 + DCT + Quantize), sharded over the ranks (STRONG scaling: the total is
    fixed, rank r takes fuif_amd.dist.shard_range).  A rank groups its streams by plan signature (two groups), runs each group
    through its own Batch in chunks of at most --chunk images (default 1024), and packs every decoded picture on the GPU into
    one device buffer of interleaved 8-bit samples (k_pack_samples).  A step = decode + inverse transforms + packing of the
    rank's whole shard.  After the timed steps the packed pictures are gathered on rank 0 over RCCL in chunks
    (dist.gather_packed); the gather is timed separately."""
    import torch
    import fuif_amd
    from fuif_amd import dist as fd
    from fuif_amd.synth import photographic
    W, H = (args.width, args.height) if (args.width, args.height) != (3840, 2160) else (1920, 1080)
    total = args.batch if args.batch != 1024 else 8192
    K = max(1, args.distinct // 2)
    rank = int(os.environ.get("RANK", "0"))
    t0 = time.time()
    sq = make_inputs(K, W, H, 3, 8, 3000, args.cache, "squeeze")      # the same K distinct images of each kind on every rank
    dc = make_inputs(K, W, H, 3, 8, 4000, args.cache, "dct420")
    t_gen = time.time() - t0
    rank, local_rank, world = fd.env_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the FUIF decode path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = fd.init(device=dev)
    if dist is None and world == 1 and not args.no_rccl_selfcheck:
        try:   # one GPU: the gather below still goes through RCCL (a process group of one rank), see main()
            dist = fd.init(device=dev, world1=True)
        except Exception:   # noqa: BLE001
            dist = None
    lo, hi = fd.shard_range(total, rank, world)
    mine = list(range(lo, hi))
    kinds = {"squeeze": [g for g in mine if g % 2 == 0], "dct420": [g for g in mine if g % 2 == 1]}
    src = {"squeeze": sq, "dct420": dc}
    chunk = args.chunk or 1024
    groups = {}
    for kind, idx in kinds.items():
        if not idx:
            continue
        blobs = [src[kind][(g // 2) % K][1] for g in idx]
        plan = fuif_amd.Plan(blobs[0])
        n = min(chunk, len(idx))
        batch = fuif_amd.Batch(plan, n, max(sum(len(b) for b in blobs[c0:c0 + n]) for c0 in range(0, len(idx), n)))
        batch.set_group_parallel(not args.no_index)
        groups[kind] = dict(idx=idx, blobs=blobs, plan=plan, batch=batch, n=n, pb=batch.packed_bytes())
    pb = next(iter(groups.values()))["pb"]
    assert all(g["pb"] == pb for g in groups.values())
    packed = torch.empty(len(mine) * pb, dtype=torch.uint8, device=dev)
    slot = {g: i for i, g in enumerate(mine)}     # picture of global image g sits at slot[g] * pb

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        dec = tr = 0.0
        ok = True
        for kind, g in groups.items():
            for c0 in range(0, len(g["idx"]), g["n"]):
                sub = g["blobs"][c0:c0 + g["n"]]
                g["batch"].upload(sub)
                g["batch"].decode()
                g["batch"].undo_transforms()
                # consecutive images of one kind are 2 apart in the global numbering: pack one by one into their slots
                for i in range(len(sub)):
                    g["batch"].pack_out(packed.data_ptr() + slot[g["idx"][c0 + i]] * pb, i, 1)
                g["batch"].sync()
                d, t = g["batch"].timing()
                dec += d; tr += t
                st, _ = g["batch"].status()
                ok = ok and not st.any()
        return ok, dec, tr

    ok = True
    for _ in range(args.warmup):
        o, _, _ = step()
        ok = ok and o
    fence()
    dec_ms, tr_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        o, d, t = step()
        ok = ok and o
        dec_ms.append(d); tr_ms.append(t)
    fence()
    elapsed = fd.max_over_ranks(time.perf_counter() - t0, dist, dev)
    # parity at size: the lossless half must be the generator's pixels; replicas of one source must be identical pictures
    pics = packed.view(len(mine), H, W, 3)
    for k in range(K):
        ref_px = torch.from_numpy(np.moveaxis(photographic(W, H, 3, 8, seed=sq[k][0]), 0, -1).astype(np.uint8)).to(dev)
        for g in kinds["squeeze"]:
            if (g // 2) % K == k:
                ok = ok and bool(torch.equal(pics[slot[g]], ref_px))
        first = None
        for g in kinds["dct420"]:
            if (g // 2) % K == k:
                if first is None:
                    first = pics[slot[g]]
                else:
                    ok = ok and bool(torch.equal(pics[slot[g]], first))
    # final gather: every rank's packed pictures to rank 0, chunked, byte sums checked
    mine_sum = torch.tensor([fd.byte_sum(packed)], dtype=torch.int64, device=dev)   # (in pieces: the int64 temporary of one torch.sum over 12 GB would be 95 GB)
    fence(); t0 = time.perf_counter()
    got = fd.gather_packed(packed, dist, root=0, keep=False)
    fence(); t_gather = time.perf_counter() - t0
    if dist is not None:
        every = [torch.zeros_like(mine_sum) for _ in range(world)]
        dist.all_gather(every, mine_sum)
        if rank == 0:
            ok = ok and got == [int(e.item()) for e in every]
    ok = fd.all_ok(ok, dist, dev)
    if rank == 0:
        value = total * W * H * args.steps / 1e6 / elapsed
        moved = (total - len(mine)) * pb
        res = {"metric": "Mpixels/s decode (c5: mixed Squeeze/DCT 1920x1080)", "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
               "config": {"workload": "C5: batch of %d %dx%d mixed Squeeze/DCT images sharded across %d GPU(s)" % (total, W, H, world),
                          "images_total": total, "images_this_rank": len(mine), "chunk": chunk, "distinct_images_per_kind": K,
                          "parity_roundtrip_ok": ok, "parity_check": "lossless half == source pixels; DCT replicas identical; status 0; gathered byte sums",
                          "entropy_kernel_ms": round(float(np.mean(dec_ms)), 3), "transform_ms": round(float(np.mean(tr_ms)), 3), "input_gen_s": round(t_gen, 1)},
               "final_gather": {"payload": "packed 8-bit RGB pictures, %d bytes each" % pb, "bytes_into_root": int(moved),
                                "gather_ms": round(t_gather * 1e3, 3), "gather_GBps": round(moved / max(t_gather, 1e-9) / 1e9, 1) if world > 1 else None}}
        # dominant kernel: k_maniac_decode over the rank's shard (all chunk launches of a step): stream bytes read once + every
        # coefficient written once as an int16 sample, per kind
        alg = sum(sum(len(b) for b in g["blobs"]) + 2.0 * g["plan"].info.coef_elems * len(g["idx"]) for g in groups.values())
        d_avg = float(np.mean(dec_ms)) / 1e3
        res["roofline"] = {"bound": "hbm", "kernel": "k_maniac_decode", "achieved": round(alg / d_avg / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(alg / d_avg / 1e9 / HBM_PEAK_GBS, 6), "traffic": None, "kernel_ms": round(d_avg * 1e3, 3),
                           "algorithmic_bytes_per_launch": int(alg), "note": "kernel_ms / bytes = the sum over the step's launches (two kinds, chunks of %d) on rank 0" % chunk}
        if world == 1 and not args.no_cpu_baseline:
            mixed = [b for pair in zip([b for _, b in sq], [b for _, b in dc]) for b in pair]
            res["cpu_baseline"] = cpu_baseline(mixed, W, H)
            res["speedup_vs_cpu_1thread"] = round(value / res["cpu_baseline"]["value"], 2)
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("PARITY FAILURE (c5)")


def run_streamed(args, wl, inputs, blobs, dev, dist, rank, world, W, H, C, BITS, K, t_gen):
    """--chunk: the batch goes through one chunk-sized Batch (coefficient, output, tmp slabs for `chunk` images), chunk after
    chunk; a step = one pass over the WHOLE batch.  Upload of a chunk = K distinct streams over PCIe + device-side replicas,
    inside the timed region (it is part of running a batch that does not fit).  Every image of the first pass is compared
    with the generator's pixels (lossless workloads)."""
    import torch
    import fuif_amd
    from fuif_amd import dist as fd
    from fuif_amd.synth import photographic
    chunk = args.chunk
    plan = fuif_amd.Plan(blobs[0])
    info = plan.info
    # A chunk is entropy-decoded in ONE launch into the int16 coefficient slab of a streaming Batch (no output slab); its inverse
    # transforms then run slice by slice into one slice-sized output tensor, each slice checked / consumed before the next
    # (fuifgpu_batch_undo_transforms_to).  Outputs are 4 bytes per sample, coefficients 2: the slice keeps the outputs small.
    n_slice = args.slice if args.slice > 0 else int(max(1, min(chunk, (16 << 30) // (4 * max(info.out_elems, 1)))))
    out = torch.empty(n_slice * info.out_elems, dtype=torch.int32, device=dev)
    cap = max(sum(len(b) for b in blobs[c0:c0 + chunk]) for c0 in range(0, args.batch, chunk))
    batch = fuif_amd.Batch(plan, chunk, cap, streaming=True)
    batch.set_group_parallel(not args.no_index)
    outs = plan.output_channels
    srcs = None
    if wl["lossless"]:
        srcs = [torch.from_numpy(photographic(W, H, C, BITS, seed=inputs[k][0])).to(dev) for k in range(K)]

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def one_pass(check):
        ok, dec, tr, tiles = True, 0.0, 0.0, 0
        for c0 in range(0, args.batch, chunk):
            sub = blobs[c0:c0 + chunk]
            batch.upload(sub)
            batch.decode()
            view = out.view(n_slice, info.out_elems)
            for s0 in range(0, len(sub), n_slice):
                cnt = min(n_slice, len(sub) - s0)
                batch.undo_transforms_to(s0, cnt, out.data_ptr())
                batch.sync()
                d, t = batch.timing()
                tr += t
                if s0 == 0:
                    dec += d
                if check and srcs is not None:
                    for i in range(cnt):
                        for c, oc in enumerate(outs):
                            got = view[i, oc["offset"]: oc["offset"] + oc["w"] * oc["h"]].view(oc["h"], oc["w"])
                            ok = ok and bool(torch.equal(got, srcs[(c0 + s0 + i) % K][c]))
            if check:
                st, _ = batch.status()
                ok = ok and not st.any()
        return ok, dec, tr

    ok = True
    checked = False
    for _ in range(args.warmup):
        o, _, _ = one_pass(not checked)
        ok, checked = ok and o, True
    fence()
    dec_ms, tr_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, d, t = one_pass(False)
        dec_ms.append(d); tr_ms.append(t)
    fence()
    elapsed = fd.max_over_ranks(time.perf_counter() - t0, dist, dev)
    if not checked:
        o, _, _ = one_pass(True)
        ok = ok and o
    ok = fd.all_ok(ok, dist, dev)
    if rank == 0:
        S = sum(len(b) for b in blobs) / args.batch
        alg = args.batch * (S + 2.0 * info.coef_elems)   # stream read once, every coefficient written once as an int16 sample
        d_avg = float(np.mean(dec_ms)) / 1e3
        value = world * args.batch * W * H * args.steps / 1e6 / elapsed
        res = {"metric": "Mpixels/s decode (%s)" % args.workload, "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "int32", "data": "synthetic",
               "config": {"workload": wl["desc"] % (args.batch, W, H), "images_per_gpu": args.batch, "chunk": chunk, "distinct_images": K,
                          "bytes_per_stream": int(S), "channels": C, "bits": BITS, "parity_roundtrip_ok": ok,
                          "parity_check": "decoded == source pixels for every image of one full pass" if wl["lossless"] else "status only",
                          "streaming": "%d entropy launch(es) of up to %d images per step into one int16 coefficient slab; inverse transforms in slices of %d images into "
                                       "one output tensor (fuifgpu_batch_undo_transforms_to); uploads inside the timed region" % (-(-args.batch // chunk), chunk, n_slice),
                          "input_gen_s": round(t_gen, 1)},
               "roofline": {"bound": "hbm", "kernel": "k_maniac_decode", "achieved": round(alg / d_avg / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(alg / d_avg / 1e9 / HBM_PEAK_GBS, 6), "traffic": None, "kernel_ms": round(d_avg * 1e3, 3),
                            "algorithmic_bytes_per_launch": int(alg), "note": "kernel_ms / bytes are per step = the sum over the step's chunk launches",
                            "transforms_ms": round(float(np.mean(tr_ms)), 3)}}
        if world == 1 and not args.no_cpu_baseline:
            # (lossless workloads: the checker's own decode of stream 0 is compared with the generator's pixels as well -- at C4's real
            # size: real reference == source == every GPU-decoded picture)
            res["cpu_baseline"] = cpu_baseline([b for _, b in inputs], W, H, source=(inputs[0][0], C, BITS) if wl["lossless"] else None)
            res["speedup_vs_cpu_1thread"] = round(value / res["cpu_baseline"]["value"], 2)
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("PARITY FAILURE: decoded planes differ from the source pixels")


def overlapped_steps(args, plan, blobs, dev, dist, W, H, index=True, warmup=None, steps=None):
    """The timed region of the default run: consecutive steps OVERLAP on the device (DESIGN.md 4.1 "overlapped launches").

    A launch is as long as the three long channel groups of a picture, and for its last 45 % they are all that runs: half of every
    SIMD's wavefront slots are empty (profiles/r4_occupancy_profile_timeline.txt).  Two streaming batch objects (each with its own
    coefficient slab, decoder scratch, context arenas and stream buffers; no output slab) on two HIP streams take the steps in turn:
    step i's entropy launch is queued behind step i-2 on its own stream and its wavefronts move into the slots step i-1's retiring
    wavefronts give up.  A step is exactly the work of the resident path: ONE entropy launch over all `--batch` streams of the step
    + their inverse transforms, here in slices into a slice-sized output tensor (fuifgpu_batch_undo_transforms_to) + one
    position-weighted 64-bit checksum per decoded image (fuifgpu_plane_checksums) into row `step` of a device table.  Nothing in the
    timed region waits on the host, except that the SECOND step of a run is queued `--overlap-stagger` seconds after the first
    (two launches that start together split the slots evenly and finish together: no overlap of a tail with a busy phase).

    EVERY step is verified: the caller compares every row of the table -- warm-up and timed steps alike -- with the checksums of the
    resident path's outputs, which it has compared with the generator's pixels (main()).
    Returns {elapsed (s, max over ranks, for args.steps steps between two fences), sums (device int64 [warmup + steps, n]),
    launch_ms (own duration of the last launch of each batch object by its HIP events), status_ok, n_slice}."""
    import torch
    import fuif_amd
    from fuif_amd import dist as fd
    n = args.batch
    info = plan.info
    n_slice = args.slice if args.slice > 0 else int(max(1, min(n, (8 << 30) // (4 * max(info.out_elems, 1)))))
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    # (zeros, not empty: planes are 256-byte aligned inside an image's slice and no kernel writes the gaps between them -- the per-image checksum covers
    # the whole slice, so the gaps must hold the same thing here and in the resident path's slab; a 3840x2160 plane happens to leave none)
    warm = max(args.warmup, 1) if warmup is None else max(warmup, 0)
    steps = args.steps if steps is None else steps
    cap = sum(len(b) for b in blobs) + 4096 * n
    batches, outs, sums = [], None, None
    try:
        # allocation and upload first, then ONE agreement over the ranks before the region's first barrier: a rank that cannot hold two batches must not leave
        # the others waiting in a collective it never reaches (the caller falls back to the resident steps on every rank alike)
        try:
            outs = [torch.zeros(n_slice * info.out_elems, dtype=torch.int32, device=dev) for _ in range(2)]
            sums = torch.zeros((warm + steps, n), dtype=torch.int64, device=dev)
            for _ in range(2):
                batches.append(fuif_amd.Batch(plan, n, cap, streaming=True))
            for b, st in zip(batches, streams):
                b.set_in_flight(2)                # (two batches in flight: launches with few tiles leave room for each other's wavefronts)
                b.set_group_parallel(index)       # (False: the streams' group index is ignored -- one wavefront per picture, what a file without the trailer gets)
                b.upload(blobs, stream=st.cuda_stream)
                b.sync(st.cuda_stream)
        except Exception:
            fd.all_ok(False, dist, dev)
            raise
        if not fd.all_ok(True, dist, dev):
            raise RuntimeError("the overlapped region could not be set up on another rank")
        return _overlapped_steps_on(args, plan, blobs, dev, dist, batches, streams, outs, sums, n, n_slice, warm, steps, index)
    finally:
        for b in batches:      # (also when an allocation or a launch failed: the caller falls back to the resident steps and needs the memory)
            b.close()


def _overlapped_steps_on(args, plan, blobs, dev, dist, batches, streams, outs, sums, n, n_slice, warm, steps, index):
    import torch
    import fuif_amd
    from fuif_amd import dist as fd
    info = plan.info

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def enqueue(i, row):
        b, st, out = batches[i % 2], streams[i % 2], outs[i % 2]
        b.decode(st.cuda_stream)
        for s0 in range(0, n, n_slice):
            cnt = min(n_slice, n - s0)
            b.undo_transforms_to(s0, cnt, out.data_ptr(), st.cuda_stream)
            fuif_amd.plane_checksums(out.data_ptr(), info.out_elems, cnt, sums.data_ptr() + 8 * (row * n + s0), st.cuda_stream)

    def run(first_row, count):
        for i in range(count):
            if i == 1 and args.overlap_stagger > 0:
                time.sleep(args.overlap_stagger)
            enqueue(i, first_row + i)
        for b, st in zip(batches, streams):
            b.sync(st.cuda_stream)

    # A launch or an allocation that fails on ONE rank after the set-up agreement must not leave the others in a barrier it never reaches: the
    # failing rank still joins every fence and collective of the region, then all ranks agree once more and fall back to the resident steps together.
    failure = None

    def guarded(first_row, count):
        nonlocal failure
        if failure is None:
            try:
                run(first_row, count)
            except Exception as exc:   # (a HIP error, an out-of-memory condition of the slice path)
                failure = exc

    torch.cuda.synchronize()
    if warm:
        guarded(0, warm)
    fence()
    t0 = time.perf_counter()
    guarded(warm, steps)
    fence()
    elapsed = fd.max_over_ranks(time.perf_counter() - t0, dist, dev)
    if not fd.all_ok(failure is None, dist, dev):
        raise RuntimeError("the overlapped steps failed on %s" % ("this rank: %r" % (failure,) if failure is not None else "another rank"))
    ok = True
    used = batches[: min(2, max(warm, steps))]
    for b in used:
        st_words, _ = b.status()
        ok = ok and not st_words.any()
    launch_ms = [float(b.timing()[0]) for b in used]
    return {"elapsed": elapsed, "sums": sums, "launch_ms": launch_ms, "status_ok": bool(ok), "n_slice": n_slice, "warm": warm, "steps": steps}


def verify_overlapped(ov, out_ptr, out_elems, n, dev, stagger):
    """every overlapped step against the resident path's outputs at `out_ptr` (n x out_elems int32, which the caller has compared with
    the source pixels): the same checksum kernel over them, then every row of the steps' table must equal it.  -> (ok, info for the line)"""
    import torch
    import fuif_amd
    expect = torch.zeros(n, dtype=torch.int64, device=dev)
    fuif_amd.plane_checksums(out_ptr, out_elems, n, expect.data_ptr())
    if dev.type == "cuda":
        torch.cuda.synchronize()
    rows_ok = (ov["sums"] == expect.unsqueeze(0)).all(dim=1)
    steps_ok = bool(rows_ok.all().item()) and ov["status_ok"] and bool((expect != 0).any().item())
    info = {"steps_verified": int(rows_ok.numel()), "steps_identical_to_resident_outputs": int(rows_ok.sum().item()), "status_ok": ov["status_ok"],
            "check": "per image and step: 64-bit position-weighted checksum of the int32 output planes (fuifgpu_plane_checksums), every warm-up and timed "
                     "step, compared after the loop with the same checksum of the resident path's outputs, which equal the source pixels",
            "stagger_s": stagger, "slice_images": ov["n_slice"], "launch_ms_overlapped": [round(x, 3) for x in ov["launch_ms"]]}
    del ov["sums"]
    return steps_ok, info


def pmc_child(args):
    """what live_pmc_traffic() runs under `rocprofv3 --pmc <one counter>`: the headline launch once, nothing else (inputs from the
    parent's cache, no checks, no CPU legs).  Prints the launch's HIP-event time."""
    import fuif_amd
    wl = WORKLOADS[args.workload]
    K = max(1, min(args.distinct, args.batch))
    inputs = make_inputs(K, args.width, args.height, wl["channels"], wl["bits"], 1000, args.cache, wl["kind"])
    blobs = [inputs[i % K][1] for i in range(args.batch)]
    plan = fuif_amd.Plan(blobs[0])
    batch = fuif_amd.Batch(plan, args.batch, sum(len(b) for b in blobs), streaming=True)   # (no output slab: only the entropy kernel runs)
    batch.set_group_parallel(not args.no_index)
    batch.upload(blobs)
    batch.decode()
    batch.sync()
    print(json.dumps({"pmc_child_kernel_ms": round(batch.timing()[0], 3)}))
    batch.close()


def live_pmc_traffic(args, committed_src):
    """HBM traffic of ONE headline launch measured in THIS run: two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE: one counter per
    pass, no trace domain next to them -- the guide's recipe) over a child process that runs the launch once.  The raw counters are
    live; the calibration factors (one per access pattern: 64-byte leaf records, 512-byte supernodes, 2-byte sample stores) are
    the committed ones of tools/ubench_gather.hip (profiles/r*_pmc_traffic.json), named in the result.  Any failure -- no
    rocprofv3, a timeout, an unexpected CSV -- returns (None, reason): the caller falls back to the committed figure and says so."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    out = {}
    tmp = tempfile.mkdtemp(prefix="fuif_pmc_")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["TMPDIR"] = "/tmp"
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", args.workload, "--batch", str(args.batch), "--width", str(args.width),
             "--height", str(args.height), "--distinct", str(args.distinct), "--cache", args.cache] + (["--no-index"] if args.no_index else [])
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            r = subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s failed (%d): %s" % (counter, r.returncode, (r.stderr or r.stdout)[-200:])
            tot, launches = 0.0, set()
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == counter and "k_maniac_decode" in row.get("Kernel_Name", ""):
                        tot += float(row["Counter_Value"])
                        launches.add(row.get("Dispatch_Id"))
            if not launches:
                return None, "no %s rows for k_maniac_decode in rocprofv3's output" % counter
            out[counter] = tot / len(launches)
            try:
                out[counter + "_kernel_ms"] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])["pmc_child_kernel_ms"]
            except (IndexError, KeyError, ValueError):
                pass
    except subprocess.TimeoutExpired:
        return None, "rocprofv3 --pmc pass timed out"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    # one factor per access pattern (committed calibration): reads are a mix of supernodes and leaves weighted by requested bytes
    rf = (committed_src or {}).get("read_factor_used") or 1.0
    wf = (committed_src or {}).get("write_factor_used") or 1.0
    traffic = int(out["FETCH_SIZE"] * 1024 * rf + out["WRITE_SIZE"] * 1024 * wf)
    return traffic, {"measured": "live: two rocprofv3 --pmc passes in this run (FETCH_SIZE, WRITE_SIZE), one launch each", "FETCH_SIZE_KiB": round(out["FETCH_SIZE"]),
                     "WRITE_SIZE_KiB": round(out["WRITE_SIZE"]), "read_factor_used": rf, "write_factor_used": wf,
                     "calibration": "factors from the committed profile %s (tools/ubench_gather.hip on the kernel's access patterns)" % (committed_src or {}).get("profile"),
                     "kernel_ms_under_pmc": [out.get("FETCH_SIZE_kernel_ms"), out.get("WRITE_SIZE_kernel_ms")]}


def ensure_ranks(args):
    """`--gpus N` means N ranks, one per GPU of this node.  Under an external launcher (the driver's `python -m
    torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) WORLD_SIZE is already set and must agree; started as a
    plain `python bench.py --gpus N` this process becomes that launcher: it re-runs itself under torch.distributed.run with
    a free port on 127.0.0.1 and returns its exit code.  A mismatch is an error, never a silent one-GPU run."""
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    world = os.environ.get("WORLD_SIZE")
    if world is not None:
        if int(world) != args.gpus:
            raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks: they must agree (one rank per GPU)" % (args.gpus, world))
        return
    if args.gpus == 1:
        return
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    raise SystemExit(subprocess.call(cmd, env=env))


def launcher_selftest(args):
    """what tests/test_bench_launcher.py runs on a machine without GPUs: the ranks ensure_ranks() started meet on gloo, agree on the
    world size through a collective, and rank 0 prints it.  Not a measurement (no metric of BASELINE.json in the line)."""
    import torch
    from fuif_amd import dist as fd
    rank, local_rank, world = fd.env_world()
    dist = fd.init(backend="gloo")
    n = torch.ones(1, dtype=torch.int64)
    if dist is not None:
        dist.all_reduce(n)
        dist.barrier()
    t = fd.max_over_ranks(0.001 * (rank + 1), dist, torch.device("cpu"))
    if rank == 0:
        print(json.dumps({"metric": "launcher-selftest", "n_gpus": int(n.item()), "world_size": world, "max_over_ranks_ok": abs(t - 0.001 * world) < 1e-9}))
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024, help="images per GPU (BASELINE config: 1024)")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--distinct", type=int, default=16, help="K distinct images replicated to the batch (SURVEY.md §8(d): 16)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + ["c5"], default="c2",
                    help="c2 = BASELINE headline config (default); c5 = 8192 mixed 1920x1080 images sharded over the ranks (strong scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-all-cores", action="store_true",
                    help="skip the all-host-cores leg of the CPU baseline (one reference process per hardware thread, ~30 s)")
    ap.add_argument("--no-h2d", action="store_true", help="skip the PCIe-inclusive measurement (one upload of the whole batch from distinct host buffers)")
    ap.add_argument("--no-index", action="store_true", help="ignore the streams' group index: one wavefront per image for the timed steps")
    ap.add_argument("--no-seq-compare", action="store_true", help="skip the extra one-wavefront-per-image step reported next to the headline")
    ap.add_argument("--chunk", type=int, default=0,
                    help="images resident at a time (0 = the whole batch, -1 = as many as the device holds): the batch is streamed through ONE chunk-sized set of coefficient / "
                         "output slabs, chunk after chunk -- how C4 (256 x 8192x8192x4: 275 GB of coefficients alone) runs on one GPU")
    ap.add_argument("--slice", type=int, default=0, help="with --chunk: images per inverse-transform slice (0 = what fits 16 GiB of int32 outputs)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="time the steps one after the other on ONE resident batch (rounds 1-4) instead of the default: consecutive steps overlap on the device -- two "
                         "streaming batch objects (own coefficient slabs, scratch and context arenas) on two HIP streams take the steps in turn, so that a step's entropy "
                         "launch fills the wavefront slots the previous step leaves empty while only its long channel groups run; every step decodes all its streams in "
                         "its own launch, runs its inverse transforms and is verified through per-image checksums (overlapped_steps)")
    ap.add_argument("--overlap-stagger", type=float, default=0.0,
                    help="overlapped steps: seconds the host waits before it queues the SECOND step of a run (the first one's busy phase at 1024 x 4K; 0 = both at once)")
    ap.add_argument("--seq-steps", type=int, default=4, help="overlapped steps of the one_wavefront_per_image leg (group index ignored; ~12 s each at 1024 x 4K)")
    ap.add_argument("--alone-steps", type=int, default=2,
                    help="overlapped steps: launches timed ALONE afterwards on the resident batch (HIP events: roofline.launch_ms_alone, the inverse transforms' time)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the small C3 / C4 / C5 legs reported as extra keys of the default line")
    ap.add_argument("--extra-legs", default="c3,c4,c5", help="which of them to run")
    ap.add_argument("--reference-encoded", type=int, default=4,
                    help="K pictures of the workload encoded by the REFERENCE encoder (oracle/_ref/fuif, default flags) on this box, given the group index "
                         "(fuif_amd.add_group_index), replicated to the batch and decoded on the resident batch: the `reference_encoded_streams` leg (0 = skip)")
    ap.add_argument("--cache", default=os.environ.get("FUIF_BENCH_CACHE", "/tmp/fuif_bench_cache"))
    ap.add_argument("--no-rccl-selfcheck", action="store_true",
                    help="one GPU: do not start the one-rank RCCL process group that runs the N>1 collectives on cuda:0 (outside the timed region except for the fence's barrier)")
    ap.add_argument("--pmc-child", action="store_true", help="(internal) run the headline launch once and exit: what the live traffic measurement profiles")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two rocprofv3 --pmc passes over one extra launch each, ~2 min): use the committed profile's figure")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="(tests) start the ranks as --gpus asks, meet on the gloo backend, print {\"n_gpus\": world} and stop: no decoding, no GPU")
    args = ap.parse_args()
    ensure_ranks(args)
    if args.launcher_selftest:
        return launcher_selftest(args)

    import fuif_amd
    if args.pmc_child:
        return pmc_child(args)
    rank = int(os.environ.get("RANK", "0"))
    if args.workload == "c5":
        fuif_amd.lib()
        return run_c5(args)
    # the HIP library is built by __graft_entry__.build() and travels in-tree; only a missing library is
    # built here, by local rank 0 alone (N ranks must not run hipcc on the same output file)
    lib_path = os.path.join(ROOT, "fuif_amd", "libfuifgpu.so")
    if not os.path.exists(lib_path):
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            fuif_amd.build()
        else:
            for _ in range(600):
                if os.path.exists(lib_path):
                    break
                time.sleep(1.0)
    fuif_amd.lib()
    wl = WORKLOADS[args.workload]
    W, H, C, BITS = args.width, args.height, wl["channels"], wl["bits"]
    K = max(1, min(args.distinct, args.batch))
    # Inputs first: the encoder pool forks, so it runs before HIP, torch.distributed or any helper thread
    # exists in this process.  Rank r decodes its own images: distinct seeds per rank.
    ref_jobs = None
    if args.reference_encoded > 0 and args.workload == "c2" and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_index and not args.chunk:
        try:
            ref_jobs = reference_encodes_start(min(args.reference_encoded, K), W, H, C, BITS, 1000, args.cache)
        except Exception:   # noqa: BLE001 -- an extra leg must not cost the bench line
            ref_jobs = None
    # the other BASELINE configs at small sizes (extra keys of the line): their streams come out of the same pool of encoder processes
    legs = {}
    if args.workload == "c2" and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_extra_legs and not args.chunk and not args.no_index:
        legs = {name: spec for name, spec in EXTRA_LEGS.items() if name in args.extra_legs.split(",")}
    specs = [(K, W, H, C, BITS, 1000 + 100 * rank, wl["kind"])]
    for name, spec in legs.items():
        specs += [(p["k"], p["w"], p["h"], p["channels"], p["bits"], p["seed0"], p["kind"]) for p in spec["parts"]]
    t0 = time.time()
    made = make_inputs_many(specs, args.cache)
    inputs = made[0]
    leg_streams, at = {}, 1
    for name, spec in legs.items():
        leg_streams[name] = made[at: at + len(spec["parts"])]
        at += len(spec["parts"])
    del made
    t_gen = time.time() - t0
    blobs = [inputs[i % K][1] for i in range(args.batch)]

    import torch
    from fuif_amd import dist as fd
    rank, local_rank, world = fd.env_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the FUIF decode path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = fd.init(device=dev)
    # One GPU: the collectives of the N>1 path still run, through a process group of ONE rank on RCCL (barrier, max-over-ranks
    # all_reduce, checksum all_gather, the chunked gather of the packed pictures), so that an RCCL / HIP-runtime clash in this
    # process shows on the one GPU the driver always has, not first on an 8-GPU node.  A failure to start RCCL is reported
    # in the line (rccl_selfcheck), it does not stop the measurement; --no-rccl-selfcheck skips it.
    rccl_note = None
    if dist is None and world == 1 and not args.no_rccl_selfcheck:
        try:
            dist = fd.init(device=dev, world1=True)
            probe = torch.ones(1, dtype=torch.int64, device=dev)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            rccl_note = "ok: process group of 1 rank on backend %s (RCCL); barrier / all_reduce / all_gather / gather ran on cuda:%d" % (dist.get_backend(), local_rank)
        except Exception as e:   # noqa: BLE001 -- anything RCCL throws is a finding to report, not a reason to lose the bench line
            rccl_note = "FAILED to start RCCL at world size 1: %s: %s" % (type(e).__name__, str(e)[:300])
            dist = None

    if args.chunk < 0:
        # as many images per chunk as the device holds: coefficient + output slabs (int32) and the stream of every resident image,
        # next to ~45 GB of decoder scratch, context arenas and the transform arena (C4: 86 of the 256 8192x8192x4 images)
        pinfo = fuif_amd.Plan(blobs[0]).info
        per_image = 2 * pinfo.coef_elems + max(len(b) for _, b in inputs) + (32 << 20)   # int16 coefficients; the outputs go through one 16 GiB slice
        free_b, _ = torch.cuda.mem_get_info(dev)
        args.chunk = int(max(1, min(args.batch, (free_b - (45 << 30) - (16 << 30) - (12 << 30)) // per_image)))
    if args.chunk:   # (also when one chunk holds the whole batch: the OUTPUTS of such a batch still only fit slice by slice)
        args.chunk = min(args.chunk, args.batch)
        return run_streamed(args, wl, inputs, blobs, dev, dist, rank, world, W, H, C, BITS, K, t_gen)

    plan = fuif_amd.Plan(blobs[0])
    info = plan.info
    # The timed region: consecutive steps overlapped on the device (default), each verified through per-image checksums that are
    # compared below with the resident path's outputs.  --no-overlap / --no-index: the steps of the resident batch are the timed ones.
    ov = ov_seq = overlap_error = None
    if not args.no_overlap and not args.no_index:
        # (a failure of the overlapped region -- it holds two batches' worth of device memory -- must not cost the line: the steps of the resident batch are then
        # the timed ones, as with --no-overlap, and the line says why.  The ranks agree on the outcome inside overlapped_steps, before its first barrier.)
        overlap_error = None
        try:
            ov = overlapped_steps(args, plan, blobs, dev, dist, W, H)
        except (fuif_amd.FuifGpuError, RuntimeError, MemoryError) as e:
            ov, overlap_error = None, "%s: %s" % (type(e).__name__, str(e)[:300])
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        # the same steps with the group index IGNORED (files as the reference CLI writes them: one wavefront per picture), overlapped the same way:
        # two launches in flight = two wavefronts per SIMD, which is what the wide kernel configuration is built for since round 5
        if ov is not None and world == 1 and not args.no_seq_compare:
            # (no warm-up step: the kernels have just run; an EVEN number of steps, so that every step has a partner beside it -- a step alone is 22 s)
            try:
                ov_seq = overlapped_steps(args, plan, blobs, dev, dist, W, H, index=False, warmup=0, steps=max(2, args.seq_steps + (args.seq_steps & 1)))
            except (fuif_amd.FuifGpuError, RuntimeError, MemoryError):
                ov_seq = None      # (the leg then shows the step alone only)
            gc.collect()
            torch.cuda.empty_cache()
    out = torch.zeros(args.batch * info.out_elems, dtype=torch.int32, device=dev)     # (zeros: see overlapped_steps -- the checksums cover the alignment gaps between planes)
    batch = fuif_amd.Batch(plan, args.batch, int(sum(len(b) for b in blobs) * 1.03) + (1 << 20), out_ptr=out.data_ptr())   # (slack: the reference-encoded leg loads other streams)
    t0 = time.time()
    batch.set_group_parallel(not args.no_index)
    batch.upload(blobs)
    batch.sync()
    t_upload = time.time() - t0
    n_tiles = args.batch if args.no_index else sum(max(1, len(fuif_amd.index_parse(inputs[i % K][1]))) for i in range(args.batch))

    def step():
        batch.decode()
        batch.undo_transforms()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    n_warm, n_timed = (1, max(1, args.alone_steps)) if ov is not None else (args.warmup, args.steps)
    for _ in range(n_warm):
        step()
    fence()
    dec_ms, tr_ms = [], []
    t0 = time.perf_counter()
    for _ in range(n_timed):
        step()
        # per-launch kernel time from the HIP events the library records on the launch stream
        batch.sync()
        d, t = batch.timing()
        dec_ms.append(d)
        tr_ms.append(t)
    fence()
    elapsed_alone = fd.max_over_ranks(time.perf_counter() - t0, dist, dev)
    elapsed = ov["elapsed"] if ov is not None else elapsed_alone
    warmup_done = ov["warm"] if ov is not None else n_warm   # (the overlapped region runs at least one warm-up step: the count that really ran goes into the line)

    # ---- correctness at full size: lossless round trip against the generator's pixels ----------
    from fuif_amd.synth import photographic
    st, used = batch.status()
    ok = not st.any()
    outs = plan.output_channels
    view = out.view(args.batch, info.out_elems)
    for k in range(K):
        if wl["lossless"]:
            src = torch.from_numpy(photographic(W, H, C, BITS, seed=inputs[k][0])).to(dev)
        else:
            src = torch.from_numpy(photographic(W, H, C, BITS, seed=inputs[k][0])).to(dev)
        for i in range(k, args.batch, K):
            for c, oc in enumerate(outs):
                got = view[i, oc["offset"]: oc["offset"] + oc["w"] * oc["h"]].view(oc["h"], oc["w"])
                if wl["lossless"]:
                    ok = ok and bool(torch.equal(got, src[c]))
                elif i == k:
                    # lossy chain: bit-exactness is proven by the -m gpu tests against the oracle; here the decoded
                    # picture must be the source within JPEG-q90 error, and (below) every replica must agree
                    err = (got[:H, :W].to(torch.float32) - src[c].to(torch.float32)).pow(2).mean().item()
                    ok = ok and err < 40.0
    # every overlapped step -- warm-up and timed -- against the resident path's outputs (which were just compared with the source
    # pixels): one 64-bit position-weighted checksum per image and step, by the same kernel (fuifgpu_plane_checksums)
    overlap_info = None
    if ov is not None:
        steps_ok, overlap_info = verify_overlapped(ov, out.data_ptr(), info.out_elems, args.batch, dev, args.overlap_stagger)
        ok = ok and steps_ok
    seq_overlap = None
    if ov_seq is not None:
        seq_ok, seq_info = verify_overlapped(ov_seq, out.data_ptr(), info.out_elems, args.batch, dev, args.overlap_stagger)
        ok = ok and seq_ok
        seq_overlap = {"value": round(args.batch * W * H * ov_seq["steps"] / 1e6 / ov_seq["elapsed"], 3), "unit": "Mpixels/s",
                       "ms_per_step": round(ov_seq["elapsed"] / ov_seq["steps"] * 1e3, 3), "steps": ov_seq["steps"], "warmup": ov_seq["warm"],
                       "steps_verified": seq_info["steps_verified"], "steps_identical_to_resident_outputs": seq_info["steps_identical_to_resident_outputs"],
                       "launch_ms_overlapped": seq_info["launch_ms_overlapped"]}
    # cross-rank exchange 1: per-image output checksums (RCCL all_gather)
    checks = fd.plane_checksums(view)
    gathered = fd.gather_checksums(checks, dist)
    # cross-rank exchange 2, the final gather of SURVEY.md §8(e): the decoded pictures, packed on the GPU to the interleaved
    # 8-bit samples of a PNM file (k_pack_samples), go to rank 0 in chunks over RCCL (7 xGMI links into the root).  Outside
    # the timed region; the root checks every rank's byte sum and does not keep the pictures (at C2 they would be 25 GB per
    # rank).  On one GPU there is nothing to move: a sample is packed to time the kernel.
    gather_info = None
    pb = batch.packed_bytes()
    if pb:
        per_chunk = max(1, min(args.batch, (2 << 30) // pb))
        n_pack = args.batch if world > 1 else min(args.batch, per_chunk)
        packed = torch.empty(per_chunk * pb, dtype=torch.uint8, device=dev)
        t_pack = t_gather = 0.0
        sums_ok = True
        for i0 in range(0, n_pack, per_chunk):
            cnt = min(per_chunk, n_pack - i0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            batch.pack_out(packed.data_ptr(), i0, cnt)
            batch.sync(); t_pack += time.perf_counter() - t0
            local = packed[: cnt * pb]
            mine = torch.sum(local, dtype=torch.int64).reshape(1)
            fence(); t0 = time.perf_counter()
            got = fd.gather_packed(local, dist, root=0, keep=False)
            fence(); t_gather += time.perf_counter() - t0
            if dist is not None:
                every = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(every, mine)
                if rank == 0:
                    sums_ok = sums_ok and got == [int(e.item()) for e in every]
        del packed
        ok = ok and sums_ok
        moved = (world - 1) * n_pack * pb
        gather_info = {"payload": "packed 8-bit interleaved pictures (k_pack_samples), %d bytes per image" % pb,
                       "images_per_rank": n_pack, "pack_ms": round(t_pack * 1e3, 3), "pack_GBps": round(n_pack * (pb + 4.0 * info.out_elems) / max(t_pack, 1e-9) / 1e9, 1),
                       "gather_ms": round(t_gather * 1e3, 3) if dist is not None else None, "bytes_into_root": int(moved),
                       "bytes_through_rccl_per_rank": int(n_pack * pb) if dist is not None else 0,
                       "gather_GBps": round(max(moved, n_pack * pb if world == 1 else 0) / max(t_gather, 1e-9) / 1e9, 1) if dist is not None else None,
                       "byte_sums_ok": bool(sums_ok), "rccl_selfcheck": rccl_note}
    # replicas of one source image must agree on every rank
    for r in range(gathered.shape[0]):
        for k in range(K):
            col = gathered[r, k::K]
            ok = ok and bool((col == col[0]).all().item())
    ok = fd.all_ok(ok, dist, dev)

    total_px = world * args.batch * W * H * args.steps
    value = total_px / 1e6 / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    ms_alone = elapsed_alone / n_timed * 1e3        # a step alone on the device (= ms_per_step with --no-overlap)

    # the same batch once more with the group index ignored (one wavefront per image, what a stream
    # without the trailer gets): reported next to the headline, outside the timed region
    seq = None
    if world == 1 and not args.no_index and not args.no_seq_compare:
        batch.set_group_parallel(False)
        batch.upload(blobs)
        batch.sync()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        t_seq = time.perf_counter() - t0
        d, t = batch.timing()
        st2, _ = batch.status()
        same = fd.plane_checksums(view)
        seq = {"value": round(args.batch * W * H / 1e6 / t_seq, 3), "unit": "Mpixels/s", "ms_per_step": round(t_seq * 1e3, 3),
               "entropy_kernel_ms": round(d, 3), "identical_output": bool(torch.equal(same, checks)) and not st2.any(),
               "note": "the rate of files as the reference CLI writes them (no FGIX trailer): one wavefront per image"}
        ok = ok and seq["identical_output"]
        if seq_overlap is not None:
            # the leg's figure is the overlapped one (two steps in flight, like `value`); the step alone on the device stays beside it
            seq = dict(seq_overlap, single_launch={k: seq[k] for k in ("value", "ms_per_step", "entropy_kernel_ms", "identical_output")},
                       identical_output=seq["identical_output"] and seq_overlap["steps_verified"] == seq_overlap["steps_identical_to_resident_outputs"],
                       note="files as the reference CLI writes them (no FGIX trailer): one wavefront per picture; steps overlapped like the headline's (two batches in flight = two "
                            "wavefronts per SIMD in the wide kernel configuration, 20 LDS supernodes per wavefront), every step verified through per-image checksums; "
                            "single_launch = one step alone on the device")

    # The same pictures as the REFERENCE ENCODER writes them ("the same .fuif inputs"): K streams encoded on this box by the
    # unmodified reference CLI while the timed region ran, given the group index by one one-wavefront-per-picture launch (add_group_index, the
    # time reported), replicated to the batch, decoded by the resident batch (steps alone on the device: compare with single_launch), every
    # output compared with the source pixels.
    refenc = None
    if ref_jobs is not None and world == 1:
        try:
            t0 = time.perf_counter()
            ref_streams = reference_encodes_wait(ref_jobs)
            t_wait = time.perf_counter() - t0
            t0 = time.perf_counter()
            indexed = fuif_amd.add_group_index([b for _, b in ref_streams])
            t_index = time.perf_counter() - t0
            Kr = len(indexed)
            if any(len(a) == len(b) for a, (_, b) in zip(indexed, ref_streams)):
                raise RuntimeError("add_group_index left a reference-encoded stream without index")
            rb = [indexed[i % Kr] for i in range(args.batch)]
            batch.set_group_parallel(True)
            batch.upload(rb)
            batch.sync()
            step()
            torch.cuda.synchronize()
            r_dec, r_tr = [], []
            t0 = time.perf_counter()
            for _ in range(2):
                step()
                batch.sync()
                d, t = batch.timing()
                r_dec.append(d); r_tr.append(t)
            torch.cuda.synchronize()
            t_ref = (time.perf_counter() - t0) / 2
            st4, _ = batch.status()
            same = not st4.any()
            for k in range(Kr):
                src = torch.from_numpy(photographic(W, H, C, BITS, seed=ref_streams[k][0])).to(dev)
                for i in range(k, args.batch, Kr):
                    for c, oc in enumerate(outs):
                        got = view[i, oc["offset"]: oc["offset"] + oc["w"] * oc["h"]].view(oc["h"], oc["w"])
                        same = same and bool(torch.equal(got, src[c]))
            v_ref = args.batch * W * H / 1e6 / t_ref
            refenc = {"value": round(v_ref, 3), "unit": "Mpixels/s", "ms_per_step": round(t_ref * 1e3, 3), "steps": 2, "entropy_kernel_ms": round(float(np.mean(r_dec)), 3),
                      "transforms_ms": round(float(np.mean(r_tr)), 3), "distinct_streams": Kr, "bytes_per_stream": int(sum(len(b) for _, b in ref_streams) / Kr),
                      "decoded_equals_source_pixels": bool(same), "index_s": round(t_index, 3), "encode_wait_s": round(t_wait, 3),
                      "vs_single_launch": round(v_ref / (args.batch * W * H / 1e6 / (ms_alone / 1e3)), 4),
                      "note": "streams written on this box by the unmodified reference CLI (oracle/_ref/fuif, default flags: learned trees of up to ~5 000 nodes), "
                              "indexed by fuif_amd.add_group_index (index_s: one launch, one wavefront per picture, K pictures), replicated to the batch; steps alone on "
                              "the device -- compare with single_launch (the product writer's streams of the same kind of pictures)"}
            ok = ok and same
        except Exception as e:   # noqa: BLE001 -- reported in the line
            refenc = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    # PCIe-inclusive rate: the boundary takes HOST buffers.  (1) serial: one upload of the whole batch from 1024 separate host blobs
    # (no replica shortcut), then one step.  (2) pipelined: a sibling Batch (fuifgpu_batch_create_sibling) owns a second set of stream
    # buffers and tile lists over the SAME slabs, decoder scratch and arenas; a host
    # thread parses and uploads step k+1 into it on a copy stream while step k's kernels run.  Reported next to `value`, never
    # as `value`.
    h2d = None
    if world == 1 and not args.no_h2d:
        import threading
        separate = [bytes(bytearray(b)) for b in blobs]
        batch.set_group_parallel(not args.no_index)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        batch.upload(separate)
        batch.sync()
        t_h2d = time.perf_counter() - t0
        serial = round(args.batch * W * H / 1e6 / (t_h2d + ms_alone / 1e3), 3)
        h2d = {"upload_s": round(t_h2d, 3), "bytes": int(sum(len(b) for b in separate)), "value_serial": serial, "unit": "Mpixels/s"}
        # a sibling Batch: a second set of stream buffers (fuifgpu_batch_create_sibling); it launches with the first one's slabs,
        # decoder scratch, context arenas and transform arena
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        free_b, _ = torch.cuda.mem_get_info(dev)
        h2d["free_device_bytes_before_sibling"] = int(free_b)
        other = None
        try:
            other = batch.sibling(sum(len(b) for b in blobs))
        except fuif_amd.FuifGpuError as e:
            h2d["pipelined_error"] = str(e)
        if other is not None:
            try:
                other.set_group_parallel(not args.no_index)
                copy_stream = torch.cuda.Stream(device=dev)
                pair, up_s, failed = [batch, other], [], []

                def uploader(bt):
                    try:
                        torch.cuda.set_device(dev)
                        t = time.perf_counter()
                        bt.upload(separate, stream=copy_stream.cuda_stream)     # returns after its copies have landed
                        up_s.append(time.perf_counter() - t)
                    except Exception as e:                                       # noqa: BLE001 -- reported in the JSON line
                        failed.append(repr(e))

                n_pipe = min(max(2, args.steps), 4)        # (a few steps show the steady state; 20 of them were 150 s of the driver's run)
                # `batch` holds step 0's streams already (the serial upload above); the first overlapped upload also allocates
                th = threading.Thread(target=uploader, args=(other,))
                th.start()
                step()
                th.join()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(1, n_pipe + 1):
                    th = threading.Thread(target=uploader, args=(pair[(k + 1) % 2],))
                    th.start()
                    pair[k % 2].decode()
                    pair[k % 2].undo_transforms()
                    th.join()
                    pair[k % 2].sync()
                torch.cuda.synchronize()
                t_pipe = (time.perf_counter() - t0) / n_pipe
                st3, _ = pair[n_pipe % 2].status()
                same = bool(torch.equal(fd.plane_checksums(view), checks)) and not st3.any() and not failed
                ok = ok and same
                h2d.update({"value_incl_h2d": round(args.batch * W * H / 1e6 / t_pipe, 3), "ms_per_step_pipelined": round(t_pipe * 1e3, 3),
                            "pipelined_steps": n_pipe, "upload_s_beside_the_kernel": round(sum(up_s[1:]) / max(1, len(up_s) - 1), 3),
                            "identical_output": same, "errors": failed or None,
                            "note": "steady state of a two-deep pipeline: every step's %d streams are parsed on the host and copied from pageable "
                                    "memory on a copy stream into the sibling Batch's stream buffers while the previous step decodes; both decode "
                                    "into one pair of slabs; a job's very first upload (upload_s) is not hidden" % args.batch})
            except (RuntimeError, fuif_amd.FuifGpuError) as e:   # e.g. torch out of memory in the checksum pass: report, keep the serial figure
                h2d["pipelined_error"] = repr(e)[:300]
                h2d["value_incl_h2d"] = serial
            other.close()
        else:
            h2d.update({"value_incl_h2d": serial, "note": "host parse of every stream + H2D copies from pageable memory + one step; not overlapped"})
        del separate

    # The big batch is released here: the small legs below and the child processes of the live traffic measurement need the device's memory.
    want_live = rank == 0 and world == 1 and args.workload == "c2" and not args.no_live_traffic
    if legs or want_live:
        n_tiles_keep = n_tiles
        batch.close()
        del view, out
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        n_tiles = n_tiles_keep
    # the other BASELINE configs at small sizes, on the resident path
    leg_results = {}
    for name, spec in legs.items():
        try:
            leg_results[name], leg_ok = run_extra_leg(name, spec, leg_streams[name], dev)
            ok = ok and leg_ok
        except Exception as e:   # noqa: BLE001 -- reported in the line
            leg_results[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    # roofline.traffic measured in THIS run (it used to be read from a committed profile)
    live_traffic, live_src = None, None
    if want_live:
        committed = pmc_traffic(args.batch, "images" if args.no_index else "groups")[1]
        try:
            live_traffic, live_src = live_pmc_traffic(args, committed)
        except Exception as e:   # noqa: BLE001 -- a measurement aid must not cost the bench line
            live_traffic, live_src = None, "%s: %s" % (type(e).__name__, str(e)[:200])

    if rank == 0:
        S = sum(len(b) for b in blobs) / args.batch
        N = info.coef_elems
        P = info.out_elems
        # dominant kernel: k_maniac_decode reads the stream once and writes every coefficient once -- as an int16 sample since round 4
        # (SURVEY 8(d) counted 4 bytes per coefficient for int32 planes; the kernel's algorithmic bytes are what it has to move now)
        alg_kernel = args.batch * (S + 2.0 * N)
        d_avg = max(float(np.mean(dec_ms)) / 1e3, 1e-9)      # (the emulated library's events read 0 ms: tests/_bench_on_emulator.py)
        t_avg = max(float(np.mean(tr_ms)) / 1e3, 1e-9)
        achieved = alg_kernel / d_avg / 1e9
        traffic, traffic_src = pmc_traffic(args.batch, "images" if args.no_index else "groups") if args.workload == "c2" else (None, None)
        if live_traffic is not None:
            traffic, traffic_src = live_traffic, dict(live_src, committed_profile_figure=traffic)
        elif live_src is not None and traffic_src is not None:
            traffic_src = dict(traffic_src, live_measurement_failed=live_src)
        achieved_overlapped = None
        if ov is not None:
            # roofline.achieved / frac are the LAUNCH ALONE: algorithmic bytes per launch over the kernel's own duration by HIP events on the resident batch
            # (what `rocprofv3 --kernel-trace --stats` shows for a lone launch: profiles/).  With overlapped steps a launch shares the device with its
            # neighbour, so per STEP the device moves the same bytes in ms_per_step -- which also holds the step's inverse transforms and checksums -- :
            # that figure is reported beside it as achieved_overlapped / frac_overlapped (round 5 had it the other way round: VERDICT r5 item 6).
            achieved_overlapped = alg_kernel / (ms_per_step / 1e3) / 1e9
        roofline = {"bound": "hbm", "kernel": "k_maniac_decode", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": int(alg_kernel),
                    "algorithmic_bytes": "n x (S + 2 N): stream bytes read once, N int16 coefficient samples written once (rounds 1-3 wrote int32: S + 4 N, "
                                         "which would read %.3f GB/s here)" % (args.batch * (S + 4.0 * N) / d_avg / 1e9),
                    "tiles_per_launch": n_tiles,
                    "note": "serial range decoders, one wavefront per channel group, 6 per SIMD, suspended while they wait for other groups' rows: "
                            "bound by the latency of two dependent memory round trips per symbol (supernode, leaf: ~980 cycles each when ~3000 long "
                            "groups share HBM, profiles/r3_fetch_latency_and_leaf_experiment.txt) and by what the wavefronts of a SIMD issue together, not by "
                            "HBM bandwidth (DESIGN.md 4.1); traffic = HBM bytes of one launch by the PMC counters, measured as traffic_source says (live in this "
                            "run unless it names the committed profile), with one calibration factor per access pattern of the kernel",
                    "transforms": {"ms": round(t_avg * 1e3, 3), "achieved": round(args.batch * (2.0 * N + 4.0 * P) / t_avg / 1e9, 1),
                                   "unit": "GB/s", "algorithmic_bytes": int(args.batch * (2.0 * N + 4.0 * P)),
                                   "note": "int16 coefficients in, int32 planes out; squeeze residuals are read as int16 straight from the slab, the few other coded planes through a widened copy (Plan::widen)"},
                    "path_bytes_per_image": int(S + 4.0 * N + 4.0 * P)}
        if ov is not None:
            roofline.update({"time_basis": "the launch alone (HIP events on the resident batch: launch_ms_alone); achieved_overlapped = the same bytes over ms_per_step of the overlapped steps",
                             "launch_ms_alone": round(d_avg * 1e3, 3), "achieved_overlapped": round(achieved_overlapped, 3), "frac_overlapped": round(achieved_overlapped / HBM_PEAK_GBS, 6),
                             "launch_ms_overlapped": overlap_info["launch_ms_overlapped"],
                             "traffic_note": "traffic = HBM bytes of ONE launch (PMC passes over a launch alone); a step has one launch"})
        else:
            roofline["kernel_ms"] = round(d_avg * 1e3, 3)
        res = {"metric": "Mpixels/s decode (4K Squeeze+YCoCg)" if args.workload == "c2" else "Mpixels/s decode (%s)" % args.workload, "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world,
               "steps": args.steps, "warmup": warmup_done, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
               "value_mode": "overlapped" if ov is not None else "sequential",
               "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
               "config": {"workload": wl["desc"] % (args.batch, W, H),
                          "images_per_gpu": args.batch, "distinct_images": K, "bytes_per_stream": int(S), "bits_per_pixel": round(8.0 * S / (W * H), 3),
                          "writer": "fuif_amd/csrc/writer.cpp learned trees (default rule: 9.7 tree steps per symbol and 12.05 MB per 4K picture; "
                                    "the reference encoder's own streams of the same pictures: 9.8 and 12.03 MB)", "parity_roundtrip_ok": ok,
                          "group_index": "ignored (--no-index): one wavefront per image" if args.no_index else
                                         "FGIX trailer behind each stream (csrc/index.cpp): one wavefront per channel group; the unmodified reference decodes the same files",
                          "parity_check": "decoded == source pixels for all images" if wl["lossless"] else "MSE vs source < 40 and all replicas identical (bit-exactness: tests -m gpu)",
                          "gather": "final gather of the packed pictures to rank 0 (chunked RCCL gather) + all_gather of per-image checksums", "input_gen_s": round(t_gen, 1), "upload_s": round(t_upload, 3),
                          "value_basis": "compressed streams resident in HBM when the timed region starts, at every N; the PCIe-inclusive rate is value_incl_h2d (N = 1 only), never value",
                          "timed_from": "compressed streams resident in HBM (uploaded before the first fence)",
                          "outputs_resident": ov is None,
                          "outputs_note": ("overlapped steps (value_mode): every step's int32 planes go through an 8 GiB slice buffer into per-image checksums, they do not stay resident; "
                                           "single_launch is the figure with all final planes resident in HBM (SURVEY 8(d)'s definition)") if ov is not None else
                                          "all final int32 planes of a step are resident in HBM when it ends"},
               "roofline": roofline}
        if overlap_error is not None:
            res["config"]["overlapped_steps"] = "FAILED (%s): the timed steps are the resident batch's, one after the other, as with --no-overlap" % overlap_error
        if overlap_info is not None:
            res["config"]["overlapped_steps"] = ("two streaming batch objects on two HIP streams take the steps in turn; a step = one entropy launch over all %d streams + "
                                                 "its inverse transforms in slices of %d images + per-image checksums; consecutive steps overlap on the device (the second "
                                                 "step of a run queued %.1f s after the first)" % (args.batch, overlap_info["slice_images"], args.overlap_stagger))
            res["overlap"] = overlap_info
            res["single_launch"] = {"value": round(world * args.batch * W * H * n_timed / 1e6 / elapsed_alone, 3), "unit": "Mpixels/s", "ms_per_step": round(ms_alone, 3),
                                    "steps": n_timed, "entropy_kernel_ms": round(d_avg * 1e3, 3), "transforms_ms": round(t_avg * 1e3, 3),
                                    "note": "the same step alone on the device (one resident batch, steps one after the other): what --no-overlap times, and rounds 1-4's `value`"}
        if refenc is not None:
            res["reference_encoded_streams"] = refenc
        if leg_results:
            res["other_configs"] = leg_results
        if seq is not None:
            res["one_wavefront_per_image"] = seq
        if h2d is not None:
            res["value_incl_h2d"] = h2d["value_incl_h2d"]
            res["h2d"] = h2d
        if gather_info is not None:
            res["final_gather"] = gather_info
        if not args.no_cpu_baseline:
            # (rank 0, outside the timed region, also at N > 1: a scaling line carries its own CPU reference; the other ranks wait at the last barrier)
            res["cpu_baseline"] = cpu_baseline([b for _, b in inputs], W, H, source=(inputs[0][0], C, BITS) if wl["lossless"] else None)
            res["speedup_vs_cpu_1thread"] = round(value / world / res["cpu_baseline"]["value"], 2)
            if world > 1:
                res["cpu_baseline"]["note"] = "speedup_vs_cpu_1thread is per GPU (value / n_gpus / cpu value)"
            if world == 1 and not args.no_cpu_all_cores:
                paths = [_stream_path(args.cache, wl["kind"], W, H, C, BITS, seed) for seed, _ in inputs]
                if all(os.path.exists(p) for p in paths):
                    allc = cpu_baseline_all_cores(paths, W, H)
                    if allc:
                        res["cpu_baseline_all_cores"] = allc
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("PARITY FAILURE: decoded planes differ from the source pixels")


if __name__ == "__main__":
    main()
