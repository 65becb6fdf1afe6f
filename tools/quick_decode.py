# one entropy-kernel launch over N replicas of cached 4K bench streams (used under rocprofv3 --pmc)
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fuif_amd
from bench import make_inputs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
inputs = make_inputs(min(n, 8), 3840, 2160, 3, 8, 1000, "/tmp/fuif_bench_cache")
blobs = [inputs[i % len(inputs)][1] for i in range(n)]
plan = fuif_amd.Plan(blobs[0])
batch = fuif_amd.Batch(plan, n, sum(len(b) for b in blobs))
batch.upload(blobs)
t0 = time.time(); batch.decode(); batch.sync()
print("decode %d streams: %.2f s; symbols/stream %d" % (n, time.time() - t0, plan.info.coef_elems))
