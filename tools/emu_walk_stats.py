#!/usr/bin/env python3
"""Where do the tree-walk rounds behind the root supernode get their supernode from -- LDS or scratch memory?
Emulator builds count it (fuifgpu_emu_walk_stats).  Used to judge supernode numbering schemes without a GPU:

  FUIF_AMD_LIB=<emulated library> python tools/emu_walk_stats.py [w h] [--deep]

(Round 3 used it to compare the speculative-walk variants with the CPU simulation; those variants were timed on hardware in round 4,
were slower, and are gone from the kernel: profiles/r4_spec_walk_experiment.txt.  The dense configuration has no LDS slots any more:
every round behind the root is "from scratch memory" there.)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("FUIF_AMD_LIB", os.path.join(ROOT, "tests", "_emu", "libfuifgpu_emu.so"))
sys.path.insert(0, ROOT)
import fuif_amd  # noqa: E402
from fuif_amd.synth import photographic  # noqa: E402

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
w, h = (int(argv[0]), int(argv[1])) if len(argv) > 1 else (640, 360)
split = 2 if "--deep" in sys.argv else None    # --deep: a split only has to save 2 bits, so that small pictures get trees of several supernode levels
img = photographic(w, h, 3, 8, seed=1000)
blob = fuif_amd.encode_image(img, 8, tree_mode=1, index=True, split_bits=split)
L = fuif_amd.lib()
st = (C.c_ulonglong * 6)()
for parallel, name in ((False, "wide  (58 supernodes in LDS, one tile per image)"), (True, "dense (no supernodes in LDS, one tile per group)")):
    plan = fuif_amd.Plan(blob)
    b = fuif_amd.Batch(plan, 1, len(blob))
    b.set_group_parallel(parallel)
    b.upload([blob])
    L.fuifgpu_emu_walk_stats(st, 1)
    b.decode(); b.sync()
    L.fuifgpu_emu_walk_stats(st, 1)
    ok = all(np.array_equal(p, q) for p, q in zip(fuif_amd.decode_batch([blob])[0][0], img))
    b.close()
    sym, lds, glob = st[0], st[1], st[2]
    print("%s: %d symbols, %.3f rounds/symbol behind the root, %.1f %% of them from LDS, %.3f scratch fetches per symbol (lossless %s)" % (
        name, sym, (lds + glob) / max(sym, 1), 100.0 * lds / max(lds + glob, 1), glob / max(sym, 1), ok))
