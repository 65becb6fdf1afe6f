#!/usr/bin/env python3
"""Locality of the context-tree walk, measured on the CPU restatement (oracle/, FO_LEAFSIM instrumentation): how often a
symbol uses the leaf of the previous symbol, hit rates of direct-mapped leaf caches, and which second-level supernodes the
walk enters (static = first K in breadth-first order resident, as the kernel does; LRU = K most recently used).
ANALYSIS TOOLING (sizes the LDS structures of k_maniac_decode); not part of the product path.

  python tools/leaf_locality.py [w h [seed]]        default 1920 1080 1; the stream is written by csrc/writer.cpp"""
import ctypes
import os
import sys

os.environ["FO_LEAFSIM"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fuif_amd  # noqa: E402
import oracle_py as O  # noqa: E402
from fuif_amd.synth import photographic  # noqa: E402

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
blob = fuif_amd.encode_image(photographic(w, h, 3, 8, seed=seed), 8, ycocg=True, tree_mode=1, index=False)
p = O.Port()
d = p.decode(blob, want_data=False)
print("%dx%d seed %d: %d bytes, %.2f tree steps and %.2f decisions per symbol" % (
    w, h, seed, len(blob), d.stats["tree_steps"] / d.stats["symbols"], d.stats["rac_decisions"] / d.stats["symbols"]))
out = (ctypes.c_uint64 * 6)()
p.lib.fo_leafsim_report(out)
acc, same = out[0], out[1]
print("symbols that walk a tree: %d; same leaf as the previous symbol: %.1f %%" % (acc, 100 * same / acc))
for k, n in enumerate((64, 128, 256, 512)):
    print("  direct-mapped leaf cache, %3d entries: %.1f %% of the leaf switches hit; %.3f misses per symbol" % (
        n, 100 * out[2 + k] / (acc - same), (acc - same - out[2 + k]) / acc))
out = (ctypes.c_uint64 * 11)()
p.lib.fo_snsim_report(out)
r = out[0]
print("walk rounds below the root supernode: %.2f per symbol" % (r / acc))
for k, n in enumerate((2, 3, 4, 7, 12)):
    print("  %2d supernodes resident: first-in-breadth-first-order %.1f %% of those rounds, LRU %.1f %%" % (n, 100 * out[1 + k] / r, 100 * out[6 + k] / r))
