#!/usr/bin/env python3
"""Condense tools/collect_pmc_r6.sh output: the calibration factors of FETCH_SIZE / WRITE_SIZE on the access patterns of round 6's context layout and the HBM traffic
of one k_maniac_decode launch (1024 x 4K, group index) with them -> pmc_traffic.json, the file bench.py reads its read factor from (profiles/r6_pergroup_pmc_traffic.json)."""
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


cal = {}
for name, counter, rec in (("read", "FETCH_SIZE", 64.0), ("snode", "FETCH_SIZE", 512.0), ("snode4", "FETCH_SIZE", 256.0), ("read32", "FETCH_SIZE", 32.0),
                           ("write", "WRITE_SIZE", 64.0), ("write32", "WRITE_SIZE", 32.0)):
    v, n = counter_sum("cal_" + name, counter, "k_gather")
    known = 4096 * 20000 * rec
    cal[name] = {"counter": counter, "KiB": v, "known_bytes": known, "factor": known / (v * 1024.0) if v else None}
    print("== calibration %-8s %s = %.0f KiB for %.0f known bytes -> multiply by %s" % (name, counter, v, known, "%.3f" % cal[name]["factor"] if v else "n/a"))

fetch, nf = counter_sum("pmc_fetch", "FETCH_SIZE", "k_maniac_decode")
write, nw = counter_sum("pmc_write", "WRITE_SIZE", "k_maniac_decode")
res = {"kernel": "k_maniac_decode", "batch": 1024, "mode": "groups", "round": 6,
       "FETCH_SIZE_KiB_per_launch": fetch / max(nf, 1), "WRITE_SIZE_KiB_per_launch": write / max(nw, 1), "calibration": cal}
f_sn = cal["snode4"]["factor"] or 1.0
# a 32-byte leaf costs a whole 64-byte fetch (the counter shows 64 bytes per record: factor 0.5 against the REQUESTED bytes): for HBM traffic the counter is right as it is
f_lf = max(cal["read32"]["factor"] or 1.0, 1.0)
f_w = cal["write32"]["factor"] or 1.0
# Requested read bytes per symbol of the headline streams (narrow supernodes, compact leaves; tools/supernode_packing.py, profiles/r6_phases_by_channel_narrow.txt):
# ~1.35 supernodes of 256 bytes behind the root and ~0.98 leaves of 32 bytes: 92 % of the requested context bytes are supernode bytes.  Both factors are
# measured; the reported traffic weights them by that share, the two pure-factor figures are given next to it.
share_sn = 1.35 * 256.0 / (1.35 * 256.0 + 0.98 * 32.0)
mix = share_sn * f_sn + (1.0 - share_sn) * f_lf
f_kib, w_kib = fetch / max(nf, 1), write / max(nw, 1)
res["read_factor_leaf_pattern"] = f_lf
res["read_factor_supernode_pattern"] = f_sn
res["read_factor_used"] = mix
res["write_factor_used"] = f_w
res["traffic_bytes_per_launch"] = int(f_kib * 1024 * mix + w_kib * 1024 * f_w)
res["traffic_bytes_per_launch_all_leaf_factor"] = int(f_kib * 1024 * f_lf + w_kib * 1024 * f_w)
res["traffic_bytes_per_launch_all_supernode_factor"] = int(f_kib * 1024 * f_sn + w_kib * 1024 * f_w)
res["traffic_bytes_per_launch_with_round4_factor"] = int(f_kib * 1024 * 1.9081166332304873 + w_kib * 1024 * 1.0)
try:
    res["kernel_ms_under_pmc"] = json.loads([l for l in open(os.path.join(out, "bench_fetch.json")) if l.startswith("{")][-1])["roofline"]["kernel_ms"]
except Exception:  # noqa: BLE001
    pass
res["note"] = ("FETCH_SIZE x read factor + WRITE_SIZE x write factor; factors = known / reported bytes of tools/ubench_gather.hip on the kernel's own patterns since round 6 "
               "(256-byte narrow supernodes as 4 bytes x 64 lanes; 32-byte compact leaves as 2 bytes x 16 lanes; 8 GiB footprint); the read factor used is the mix of the two "
               "weighted by requested bytes; traffic_bytes_per_launch_with_round4_factor = the same counters with the factor rounds 3-5 used (512-byte records dominated then)")
print("== k_maniac_decode per launch: FETCH_SIZE %.0f KiB (%d launches), WRITE_SIZE %.0f KiB (%d launches)" % (res["FETCH_SIZE_KiB_per_launch"], nf, res["WRITE_SIZE_KiB_per_launch"], nw))
print("   traffic with the mixed read factor %.3f: %.3f TB (all reads at the leaf factor %.3f: %.3f TB; at the supernode factor %.3f: %.3f TB; with round 4's factor 1.908: %.3f TB)" % (
    mix, res["traffic_bytes_per_launch"] / 1e12, f_lf, res["traffic_bytes_per_launch_all_leaf_factor"] / 1e12, f_sn, res["traffic_bytes_per_launch_all_supernode_factor"] / 1e12,
    res["traffic_bytes_per_launch_with_round4_factor"] / 1e12))
json.dump(res, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
