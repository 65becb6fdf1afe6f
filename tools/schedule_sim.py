#!/usr/bin/env python3
"""Discrete-event model of the tile scheduler of k_maniac_decode -- CPU only.

What it models: persistent wavefront slots taking tiles from a work list in order; a tile advances at the
per-wave symbol rate of its SIMD's occupancy (measured on independent streams, profiles/r1_occupancy_variants.txt:
1 / 2 / 3 waves per SIMD deliver 1.00 / 1.56 / 1.95x one wave; 4 assumed 2.25x); a tile cannot pass the
fraction its reference channels (up to 6 previous ones) have reached; waiting tiles keep their slot.
It reproduced the measured dense launches of round 1 (1024 x 4K: 4096 slots 17.8 s measured / 16.5-17.5 s
model, 3072 slots 21.4 / 21.5) and was used to compare work-list orders before spending GPU time on them.

  python tools/schedule_sim.py [n_images] [slots]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fuif_amd  # noqa: E402  (host planner only: no GPU)
from fuif_amd.synth import photographic  # noqa: E402

THR = {0: 0.0, 1: 1.0, 2: 1.56, 3: 1.95, 4: 2.25}   # work per SIMD vs waves per SIMD
SYMBOL_S = 1.02e-6                                   # one wave alone: seconds per symbol (profiles/r1_perimage_phase_cycles.txt)


def thr(k):
    k = min(k, 4.0)
    lo = int(np.floor(k))
    hi = min(lo + 1, 4)
    return THR[lo] + (THR[hi] - THR[lo]) * (k - lo)


def channel_work(scale=8):
    """symbols per coded channel of a 3840x2160 RGB YCoCg+Squeeze stream (geometry from the host planner)"""
    blob = fuif_amd.encode_image(photographic(3840 // scale, 2160 // scale, 3, 8, seed=1), 8, tree_mode=0)
    ch = fuif_amd.Plan(blob).coded_channels
    return np.array([c["w"] * c["h"] for c in ch], float) * scale * scale


def simulate(order, n_img, work, slots, n_simd=1024, dt=0.05, spin_cost=0.02):
    nch = len(work)
    prog = np.zeros((n_img, nch))
    done = np.zeros((n_img, nch), bool)
    active, head, t, ndone, idle = [], 0, 0.0, 0, 0.0
    while ndone < n_img * nch:
        while len(active) < slots and head < len(order):
            active.append(order[head]); head += 1
        a = np.array(active)
        ai, ac = a[:, 0], a[:, 1]
        lim = np.ones(len(active))
        for k in range(6):
            rc = ac - 1 - k
            ok = rc >= 0
            rp = np.where(ok, prog[ai, np.maximum(rc, 0)], 1.0)
            rd = np.where(ok, done[ai, np.maximum(rc, 0)], True)
            lim = np.minimum(lim, np.where(rd, 1.0, rp))
        own = prog[ai, ac]
        blocked = own >= lim - 1e-12
        nrun, nspin = int((~blocked).sum()), int(blocked.sum())
        keff = (nrun + spin_cost * nspin) / n_simd
        per_wave = min(thr(keff) / max(keff, 1e-9), 1.0)
        adv = per_wave * dt / SYMBOL_S / work[ac]
        newp = np.minimum(np.where(blocked, own, np.minimum(own + adv, lim)), 1.0)
        prog[ai, ac] = newp
        fin = newp >= 1.0 - 1e-12
        for i in np.nonzero(fin)[0]:
            done[ai[i], ac[i]] = True
        ndone += int(fin.sum())
        active = [active[k] for k in np.nonzero(~fin)[0]]
        idle += (n_simd * 4 - min(nrun, n_simd * 4)) * dt
        t += dt
    return t


def main():
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    slots = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    work = channel_work()
    nch = len(work)
    level = {c: (nch - 1 - c) // 3 for c in range(nch)}
    orders = {
        "group-major (shipped)": [(i, c) for c in range(nch) for i in range(n_img)],
        "image-major": [(i, c) for i in range(n_img) for c in range(nch)],
        "level-major": [(i, c) for L in sorted(set(level.values()), reverse=True) for i in range(n_img) for c in range(nch) if level[c] == L],
        "cohorts of 512": [(i, c) for b0 in range(0, n_img, 512) for c in range(nch) for i in range(b0, min(n_img, b0 + 512))],
    }
    total = n_img * work.sum() * SYMBOL_S
    print("%d images x %.1f M symbols, %d slots; one wave per image would take %.1f s; work bound at 2.25x per SIMD: %.1f s" % (
        n_img, work.sum() / 1e6, slots, work.sum() * SYMBOL_S * max(1.0, n_img / 1024.0), total / 1024 / 2.25))
    for name, order in orders.items():
        print("  %-24s %6.2f s" % (name, simulate(order, n_img, work, slots)))
    if slots == 4096:
        print("  %-24s %6.2f s" % ("group-major, 3072 slots", simulate(orders["group-major (shipped)"], n_img, work, 3072)))


if __name__ == "__main__":
    main()
