#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSVs (one directory per pass) for kernels whose name contains argv[2]."""
import csv, glob, os, sys, collections
root, pat = sys.argv[1], sys.argv[2]
tot = collections.OrderedDict(); launches = collections.Counter()
for f in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        seen = set()
        for r in csv.DictReader(fh):
            if pat not in r.get("Kernel_Name", ""): continue
            name = r["Counter_Name"]; tot[name] = tot.get(name, 0.0) + float(r["Counter_Value"])
            key = (name, r.get("Dispatch_Id"))
            if key not in seen: seen.add(key); launches[name] += 1
for k, v in tot.items():
    print("%-28s %20.0f   (%d dispatches)" % (k, v, launches[k]))
g = tot.get
if g("SQ_WAVE_CYCLES"):
    wc = g("SQ_WAVE_CYCLES")
    print("--- derived (SQ cycle counters are in quad-cycles) ---")
    for n in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_INST_CYCLES_SALU"):
        if g(n) is not None: print("%-28s / SQ_WAVE_CYCLES = %.3f" % (n, g(n) / wc))
    for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_SMEM", "SQ_INSTS"):
        if g(n) is not None: print("%-28s per wave quad-cycle = %.4f  (x4 = cycles per instr %.1f)" % (n, g(n) / wc, 4 * wc / g(n) if g(n) else 0))
