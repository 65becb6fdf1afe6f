#!/usr/bin/env python3
"""Corrupted streams through the emulated kernels (build container; best under an AddressSanitizer build of the
emulated library: LD_PRELOAD=libasan.so FUIF_AMD_LIB=<asan build>).  Reference-written streams from
tools/emu_fuzz.py's generator get random bit flips / byte overwrites behind the header; every variant is decoded per
image and, with the CLEAN stream's group index appended, per channel group.  The kernels must terminate, stay inside
their buffers, and -- whenever the oracle accepts the damaged stream -- either flag it or produce the oracle's planes.

  python tools/emu_corrupt_fuzz.py [n_streams] [seed] [variants_per_stream]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import emu_fuzz  # noqa: E402  (sets FUIF_AMD_LIB, imports fuif_amd and the oracle)
from emu_fuzz import fuif_amd, Port  # noqa: E402


def decode(blob, parallel):
    plan = fuif_amd.Plan(blob)
    batch = fuif_amd.Batch(plan, 1, len(blob) + 64)
    try:
        batch.set_group_parallel(parallel)
        batch.upload([blob])
        batch.decode()
        batch.sync()
        st, _ = batch.status()
        pre = batch.coef_planes(0)
        groups = batch.group_index(0)
        batch.undo_transforms()
        batch.sync()
        return int(st[0]), pre, groups
    finally:
        batch.close()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    variants = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    rng = np.random.default_rng(seed)
    port = Port()
    tmp = tempfile.mkdtemp()
    decoded = flagged = agree = disagree = rejected = 0
    for k in range(n):
        flags, blob = emu_fuzz.random_case(rng, tmp)
        if blob is None:
            continue
        try:
            st, _, groups = decode(blob, False)
        except fuif_amd.FuifGpuError:
            continue
        hdr = fuif_amd.Plan(blob).info.data_start
        for v in range(variants):
            b = bytearray(blob)
            for _ in range(1 + v % 3):
                pos = int(rng.integers(hdr, len(b)))
                b[pos] = (b[pos] ^ (1 << int(rng.integers(0, 8)))) if v % 2 else int(rng.integers(0, 256))
            b = bytes(b)
            with open(os.path.join(tmp, "current.fuif"), "wb") as f:
                f.write(b)
            for parallel in (False, True):
                stream = fuif_amd.index_append(b, groups) if parallel and len(groups) > 1 else b
                try:
                    st, pre, _ = decode(stream, parallel)
                except fuif_amd.FuifGpuError:
                    rejected += 1
                    continue
                decoded += 1
                if st & 6:
                    flagged += 1
                    continue
                if parallel:
                    continue      # groups behind the damage start from the clean offsets: no sequential counterpart
                d = port.decode(b, undo=False)
                if d.status != 1:
                    disagree += 1
                    print("kernel accepts what the oracle refuses: stream %d variant %d flags %s" % (k, v, flags), flush=True)
                    continue
                same = all(np.array_equal(g, c["data"]) for g, c in zip(pre, d.channels) if c["size"] == c["w"] * c["h"])
                agree += same
                if not same:
                    disagree += 1
                    keep = os.path.join(ROOT, "gpurun_out", "emu_corrupt_%d_%d_%d.fuif" % (seed, k, v))
                    os.makedirs(os.path.dirname(keep), exist_ok=True)
                    open(keep, "wb").write(b)
                    print("planes differ on an accepted damaged stream: %s (flags %s)" % (keep, flags), flush=True)
        if (k + 1) % 10 == 0:
            print("%d streams: %d decodes, %d flagged corrupt, %d equal to the oracle, %d disagreements, %d rejected by the planner" % (
                k + 1, decoded, flagged, agree, disagree, rejected), flush=True)
    print("done: %d decodes, %d flagged, %d equal, %d disagreements, %d rejected" % (decoded, flagged, agree, disagree, rejected))
    return 1 if disagree else 0


if __name__ == "__main__":
    sys.exit(main())
