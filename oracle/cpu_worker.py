#!/usr/bin/env python3
"""One CPU-baseline worker process: decodes the given .fuif files round-robin with the reference decoder
(oracle/_ref when built, else the plain-C port) until the deadline and prints "<images> <busy seconds>".
bench.py starts one of these per host core for the all-cores figure (one process per image).
TEST INFRASTRUCTURE: never imported by the product."""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from oracle_py import Port, Ref  # noqa: E402


def main():
    seconds, start_at, first = float(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
    blobs = [open(p, "rb").read() for p in sys.argv[4:]]
    lib = Ref() if Ref.available() else Port(build=False)
    lib.time_decode(blobs[first % len(blobs)][:4096])   # page the library in (truncated stream: cheap)
    while time.time() < start_at:
        time.sleep(0.01)
    n, busy, k = 0, 0.0, first
    t_end = time.time() + seconds
    while time.time() < t_end:
        dt, ok = lib.time_decode(blobs[k % len(blobs)])
        if not ok:
            print("0 0")
            return 1
        n += 1; busy += dt; k += 1
    print("%d %.6f %.6f" % (n, busy, time.time() - start_at))
    return 0


if __name__ == "__main__":
    sys.exit(main())
