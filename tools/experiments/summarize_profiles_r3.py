#!/usr/bin/env python3
"""Condense tools/experiments/collect_profiles_r3.sh output: kernel stats, PMC traffic of k_maniac_decode with the calibration factors
measured on the kernel's own access pattern, and the JSON bench.py reads `roofline.traffic` from."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]


def counter_sum(sub, counter, kernel_substr):
    tot, n = 0.0, 0
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter or kernel_substr not in r.get("Kernel_Name", ""):
                continue
            tot += float(r["Counter_Value"])
            seen.add(r.get("Dispatch_Id"))
        n += len(seen)
    return tot, n


print("== kernel stats (rocprofv3 --kernel-trace --stats, bench.py --steps 2 --warmup 1)")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print("  %-70s calls %5s  total %15s ns  avg %14s ns  %6s %%" % (r.get("Name", "")[:70], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))
try:
    b = json.loads([l for l in open(os.path.join(out, "bench_trace.json")) if l.startswith("{")][-1])
    print("  bench.py under the tracer: value %.1f Mpx/s, kernel_ms (HIP events) %.1f" % (b["value"], b["roofline"]["kernel_ms"]))
except Exception as e:  # noqa: BLE001
    print("  (no bench line:", e, ")")

cal = {}
for name, sub, counter, rec in (("read", "cal_read_fetch", "FETCH_SIZE", 64.0), ("snode", "cal_snode_fetch", "FETCH_SIZE", 512.0), ("write", "cal_write_write", "WRITE_SIZE", 64.0)):
    v, n = counter_sum(sub, counter, "k_gather")
    known = 4096 * 20000 * rec
    cal[name] = {"counter": counter, "KiB": v, "known_bytes": known, "factor": known / (v * 1024.0) if v else None}
    print("== calibration %-11s %s = %.0f KiB for %.0f known bytes -> multiply by %s" % (name, counter, v, known, "%.3f" % cal[name]["factor"] if v else "n/a"))

fetch, nf = counter_sum("pmc_fetch", "FETCH_SIZE", "k_maniac_decode")
write, nw = counter_sum("pmc_write", "WRITE_SIZE", "k_maniac_decode")
res = {"kernel": "k_maniac_decode", "batch": 1024, "mode": "groups", "round": 3,
       "FETCH_SIZE_KiB_per_launch": fetch / max(nf, 1), "WRITE_SIZE_KiB_per_launch": write / max(nw, 1), "calibration": cal}
ff = cal["read"]["factor"] or 1.0
sf = cal["snode"]["factor"] or 2.0
wf = cal["write"]["factor"] or 1.0
# The kernel's fetches are a mix of the two read patterns: per symbol ~1.1 supernodes of 512 bytes and ~0.9 leaves of 64 bytes as REQUESTED
# (profiles/r2_walk_locality.txt), i.e. 91 % of the requested read bytes are supernode bytes.  Both factors are measured; the reported
# traffic weights them by that share (the two bounds are given next to it).
share_sn = 1.1 * 512.0 / (1.1 * 512.0 + 0.9 * 64.0)
mix = share_sn * sf + (1.0 - share_sn) * ff
f_kib, w_kib = fetch / max(nf, 1), write / max(nw, 1)
res["read_factor_leaf_pattern"] = ff
res["read_factor_supernode_pattern"] = sf
res["read_factor_used"] = mix
res["traffic_bytes_per_launch"] = int(f_kib * 1024 * mix + w_kib * 1024 * wf)
res["traffic_bytes_per_launch_all_leaf_factor"] = int(f_kib * 1024 * ff + w_kib * 1024 * wf)
res["traffic_bytes_per_launch_all_supernode_factor"] = int(f_kib * 1024 * sf + w_kib * 1024 * wf)
try:
    res["kernel_ms_under_pmc"] = json.loads([l for l in open(os.path.join(out, "bench_fetch.json")) if l.startswith("{")][-1])["roofline"]["kernel_ms"]
except Exception:  # noqa: BLE001
    pass
res["note"] = ("FETCH_SIZE x read factor + WRITE_SIZE x write factor; factors = known / reported bytes of tools/ubench_gather.hip on the kernel's own "
               "patterns (64-byte leaf records as 2 bytes x 32 lanes; 512-byte supernodes as 8 bytes x 64 lanes; 8 GiB footprint); the read factor "
               "used is the mix of the two weighted by requested bytes, both pure-factor figures are given next to it")
print("== k_maniac_decode per launch: FETCH_SIZE %.0f KiB (%d launches), WRITE_SIZE %.0f KiB (%d launches)" % (res["FETCH_SIZE_KiB_per_launch"], nf, res["WRITE_SIZE_KiB_per_launch"], nw))
print("   traffic with the mixed read factor %.3f: %.3f TB (all reads at the leaf factor %.3f: %.3f TB; at the supernode factor %.3f: %.3f TB)" % (
    mix, res["traffic_bytes_per_launch"] / 1e12, ff, res["traffic_bytes_per_launch_all_leaf_factor"] / 1e12, sf, res["traffic_bytes_per_launch_all_supernode_factor"] / 1e12))
json.dump(res, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
