#!/usr/bin/env python3
"""Round 6, session 13: C4's entropy launch at FULL size (256 x 8192x8192x4, 14 bit) with round 5's library and with this round's, on ONE box, one after the
other (streaming batch: the int16 coefficient slab only; FUIF_AMD_LIB picks the library, one process per library).  ANALYSIS TOOLING."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    from bench import make_inputs
    import fuif_amd
    n = int(sys.argv[2])
    inputs = make_inputs(2, 8192, 8192, 4, 14, 7000, os.environ.get("FUIF_BENCH_CACHE", "/tmp/fuif_bench_cache"), "squeeze_raw")
    blobs = [inputs[i % 2][1] for i in range(n)]
    plan = fuif_amd.Plan(blobs[0])
    batch = fuif_amd.Batch(plan, n, sum(len(b) for b in blobs) + 4096 * n, streaming=True, tmp_images=2)
    batch.upload(blobs)
    for rep in range(int(sys.argv[3])):
        batch.decode(); batch.sync()
        st, _ = batch.status()
        print("%s n=%d 8192x8192x4: entropy launch %.1f ms, status %s" % (os.path.basename(os.environ.get("FUIF_AMD_LIB", "libfuifgpu.so")), n, batch.timing()[0], "ok" if not st.any() else "FLAGGED"), flush=True)
    sys.exit(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
# (C4_LIBS=a.so,b.so,... overrides the pair: the round's last session compared the library before and after the decoder rewrite)
for lib in os.environ.get("C4_LIBS", "build/libfuifgpu_r5.so,fuif_amd/libfuifgpu.so,build/libfuifgpu_r5.so,fuif_amd/libfuifgpu.so").split(","):
    env = dict(os.environ, FUIF_AMD_LIB=os.path.join(ROOT, lib), FUIFGPU_CTX_MB="32")
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n), "1"], env=env)
