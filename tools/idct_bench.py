#!/usr/bin/env python3
"""Time of k_idct8x8 alone through its C-ABI entry point (fuifgpu_idct8x8) on one large synthetic component, any build of the library:

  [FUIF_AMD_LIB=build/libfuifgpu_x.so] python tools/idct_bench.py [bw bh] [reps]

bw x bh blocks (default 3840 x 2160: 64 coefficient planes of 33 MB each in, 2.1 GB of samples out); prints the mean wall time of a
call (the entry point waits for its stream) and the algorithmic GB/s (64 int32 coefficients in + 64 int32 samples out per block).
ANALYSIS TOOLING; no torch."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fuif_amd  # noqa: E402

bw, bh = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
L = fuif_amd.lib()
L.fuifgpu_dev_alloc.restype = C.c_void_p
n = bw * bh
rng = np.random.default_rng(5)
src = L.fuifgpu_dev_alloc(C.c_size_t(64 * n * 4))
out = L.fuifgpu_dev_alloc(C.c_size_t(64 * n * 4))
assert src and out, "device allocation failed"
plane = rng.integers(-40, 40, n, dtype=np.int32)
for i in range(64):     # the same plane 64 times, rolled: distinct values per position, one upload buffer
    assert L.fuifgpu_dev_upload(C.c_void_p(src + i * n * 4), np.roll(plane, i).ctypes.data_as(C.c_void_p), C.c_size_t(n * 4)) == 0
ptrs = (C.c_void_p * 64)(*[src + i * n * 4 for i in range(64)])
assert L.fuifgpu_idct8x8(ptrs, bw, bh, C.c_void_p(out), 255, None) == 0     # warm-up
t0 = time.perf_counter()
for _ in range(reps):
    assert L.fuifgpu_idct8x8(ptrs, bw, bh, C.c_void_p(out), 255, None) == 0
dt = (time.perf_counter() - t0) / reps
import hashlib
piece = min(64 * n, 4 << 20)             # samples: the first and the last 16 MB of the output, hashed (A/B of two builds: must be equal)
buf = np.zeros(piece, np.int32)
h = hashlib.sha256()
for at in (0, 64 * n - piece):
    L.fuifgpu_dev_download(buf.ctypes.data_as(C.c_void_p), C.c_void_p(out + at * 4), C.c_size_t(piece * 4))
    h.update(buf.tobytes())
print("%s: %d x %d blocks, %.3f ms per call, %.0f GB/s algorithmic; output sha256 %s" % (
    os.path.basename(os.environ.get("FUIF_AMD_LIB", "libfuifgpu.so")), bw, bh, dt * 1e3, 2 * 64 * n * 4 / dt / 1e9, h.hexdigest()[:16]), flush=True)
L.fuifgpu_dev_free(C.c_void_p(src)); L.fuifgpu_dev_free(C.c_void_p(out))
