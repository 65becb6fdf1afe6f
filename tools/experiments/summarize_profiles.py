#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small text/JSON summary."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
res = {}


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


for f in find("trace/**/*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    res["kernel_stats"] = rows
    for r in rows[:20]:
        print("  %-60s calls %6s total %14s ns avg %14s ns  %6s %%" % (r.get("Name", "")[:60], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))
for name, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in find(name + "/**/*counter_collection.csv"):
        agg = {}
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != key:
                continue
            k = r.get("Kernel_Name", "")[:80]
            a = agg.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r.get("Counter_Value", 0))
        print("== %s (sum over dispatches; rocprofv3 unit = KiB for *_SIZE):" % key)
        res[key] = {k: {"dispatches": v[0], "sum": v[1]} for k, v in agg.items()}
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
            print("  %-80s dispatches %5d  sum %16.1f  per-dispatch %14.1f" % (k, v[0], v[1], v[1] / v[0]))
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
