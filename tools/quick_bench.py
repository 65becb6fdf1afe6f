# interim timing script (first GPU contact): N replicas of a golden fixture through the C-ABI
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import fuif_amd
name = sys.argv[1] if len(sys.argv) > 1 else "c1_rgb8_512x512"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
blob = open(os.path.join(ROOT, "tests", "golden", name + ".fuif"), "rb").read()
plan = fuif_amd.Plan(blob)
batch = fuif_amd.Batch(plan, n, len(blob) * n)
blobs = [blob] * n
t0 = time.time(); batch.upload(blobs); batch.sync(); t1 = time.time()
for it in range(2):
    batch.decode(); batch.undo_transforms(); batch.sync()
    d, t = batch.timing()
    mpx = plan.info.w * plan.info.h * n / 1e6
    print("iter %d: n=%d decode %.1f ms transforms %.2f ms -> %.1f Mpx/s (upload %.1f ms)" % (it, n, d, t, mpx / ((d + t) / 1e3), (t1 - t0) * 1e3))
st, used = batch.status()
print("status any:", st.any(), "consumed", used[0], "of", len(blob))
