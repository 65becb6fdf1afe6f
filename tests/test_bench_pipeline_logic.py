"""bench.py's timed region lets consecutive steps overlap on the device (overlapped_steps: two streaming batch objects on two HIP
streams) and needs a GPU for its streams; its LOGIC -- the batch objects taking the steps in turn, inverse transforms
in slices (ragged last slice), one checksum row per step (warm-up and timed), the comparison of every row with the resident path's
outputs (verify_overlapped) and that a wrong picture anywhere is noticed -- runs here on the wavefront emulator with torch.cuda's
stream calls replaced by no-ops (the emulated library treats host memory as device memory)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

DRIVER = r'''
import contextlib, json, os, sys, types
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import bench
import fuif_amd


class FakeStream:
    def __init__(self, device=None):
        self.cuda_stream = None


torch.cuda.Stream = FakeStream
torch.cuda.stream = lambda s: contextlib.nullcontext()
torch.cuda.synchronize = lambda *a, **k: None
args = types.SimpleNamespace(batch=5, slice=2, warmup=1, steps=3, overlap_stagger=0.0)
wl = bench.WORKLOADS["c2"]
W, H, C, BITS, K = 97, 61, 3, 8, 2
inputs = bench.make_inputs(K, W, H, C, BITS, 1000, %(cache)r, wl["kind"])
blobs = [inputs[i %% K][1] for i in range(args.batch)]
plan = fuif_amd.Plan(blobs[0])
dev = torch.device("cpu")
ov = bench.overlapped_steps(args, plan, blobs, dev, None, W, H)
assert ov["sums"].shape == (4, 5) and ov["status_ok"] and ov["n_slice"] == 2
# the resident path: one batch with an output slab, the same streams (damage: two of them swapped -> pictures 1 and 2 differ)
resident = list(blobs)
if %(damage)d:
    resident[1], resident[2] = resident[2], resident[1]
out = torch.zeros(args.batch * plan.info.out_elems, dtype=torch.int32)
b = fuif_amd.Batch(plan, args.batch, sum(len(x) for x in resident), out_ptr=out.data_ptr())
b.upload(resident); b.decode(); b.undo_transforms(); b.sync()
ok, info = bench.verify_overlapped(ov, out.data_ptr(), plan.info.out_elems, args.batch, dev, 0.0)
print(json.dumps({"ok": ok, "info": info}))
'''


@pytest.mark.parametrize("damage", [1])      # (the undamaged case is part of the end-to-end runs of tests/test_bench_launcher.py)
def test_overlapped_steps_decode_and_verify_every_step(tmp_path, damage):
    if sys.platform != "linux" or os.uname().machine != "x86_64":
        pytest.skip("the emulator's context switch is x86-64 SysV assembly")
    from test_emulated_kernels import build_emulated_library
    env = dict(os.environ, FUIF_AMD_LIB=build_emulated_library(), EMU_ALARM="600")
    r = subprocess.run([sys.executable, "-c", DRIVER % dict(root=ROOT, cache=str(tmp_path / "cache"), damage=damage)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["info"]["steps_verified"] == 4 and line["info"]["slice_images"] == 2
    if damage:
        assert line["ok"] is False and line["info"]["steps_identical_to_resident_outputs"] == 0
    else:
        assert line["ok"] is True and line["info"]["steps_identical_to_resident_outputs"] == 4, line


LEG_DRIVER = r'''
import json, sys
sys.path.insert(0, %(root)r)
import torch
import bench

torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.empty_cache = lambda *a, **k: None
spec = dict(desc="test leg: %%d mixed pictures", n=6, steps=2, mse_max=150.0,     # (tiny pictures: the generator's sinusoids are steep there)
            parts=[dict(kind="squeeze", w=97, h=61, channels=3, bits=8, k=2, seed0=3000), dict(kind="dct420", w=112, h=64, channels=3, bits=8, k=2, seed0=4000)])
streams = bench.make_inputs_many([(p["k"], p["w"], p["h"], p["channels"], p["bits"], p["seed0"], p["kind"]) for p in spec["parts"]], %(cache)r)
if %(damage)d:
    streams[0][1] = (streams[0][1][0] + 5, streams[0][1][1])      # the wrong source picture for one lossless stream: the leg must notice
res, ok = bench.run_extra_leg("t", spec, streams, torch.device("cpu"))
print(json.dumps({"ok": ok, "res": res}))
'''


@pytest.mark.parametrize("damage", [0, 1])
def test_extra_leg_of_the_default_line(tmp_path, damage):
    """run_extra_leg (the small C3 / C4 / C5 legs of the default bench line) on the emulator: two kinds in one leg, a rate, roofline,
    parity against the generator's pixels and a CPU baseline from the reference / the oracle on the same streams"""
    if sys.platform != "linux" or os.uname().machine != "x86_64":
        pytest.skip("the emulator's context switch is x86-64 SysV assembly")
    from test_emulated_kernels import build_emulated_library
    env = dict(os.environ, FUIF_AMD_LIB=build_emulated_library(), EMU_ALARM="600")
    r = subprocess.run([sys.executable, "-c", LEG_DRIVER % dict(root=ROOT, cache=str(tmp_path / "cache"), damage=damage)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["ok"] is (not damage) and line["res"]["parity_ok"] is (not damage)
    res = line["res"]
    assert res["steps"] == 2 and res["value"] > 0 and res["roofline"]["bound"] == "hbm" and res["roofline"]["algorithmic_bytes_per_launch"] > 0
    assert res["cpu_baseline"]["cores"] == 1 and res["cpu_baseline"]["value"] > 0 and res["cpu_baseline"]["kind"] in ("reference", "port")
