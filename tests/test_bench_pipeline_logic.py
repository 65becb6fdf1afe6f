"""bench.py --pipeline (the opt-in mode in which consecutive steps overlap on the device) needs a GPU for its streams; its
LOGIC -- two streaming batch objects taking the steps in turn, inverse transforms in slices (ragged last slice), every image of the checked
pass compared with its source picture on the device, the JSON line -- runs here on the wavefront emulator with torch.cuda's stream calls
replaced by no-ops (the emulated library treats host memory as device memory)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

DRIVER = r'''
import contextlib, os, sys, types
sys.path.insert(0, %(root)r)
import torch
import bench


class FakeStream:
    def __init__(self, device=None):
        self.cuda_stream = None


torch.cuda.Stream = FakeStream
torch.cuda.stream = lambda s: contextlib.nullcontext()
torch.cuda.synchronize = lambda *a, **k: None
args = types.SimpleNamespace(batch=3, slice=2, no_index=False, warmup=1, steps=2, pipeline_stagger=0.0, workload="c2", no_cpu_baseline=True)
wl = bench.WORKLOADS["c2"]
W, H, C, BITS, K = 97, 61, 3, 8, 2
inputs = bench.make_inputs(K, W, H, C, BITS, 1000, %(cache)r, wl["kind"])
if %(damage)d:
    inputs[1] = (inputs[1][0] + 7, inputs[1][1])      # the wrong source picture for every second image: the check must notice
blobs = [inputs[i %% K][1] for i in range(args.batch)]
bench.run_pipelined(args, wl, inputs, blobs, torch.device("cpu"), None, 0, 1, W, H, C, BITS, K, 0.0)
'''


@pytest.mark.parametrize("damage", [0, 1])
def test_pipelined_steps_decode_and_verify_every_image(tmp_path, damage):
    if sys.platform != "linux" or os.uname().machine != "x86_64":
        pytest.skip("the emulator's context switch is x86-64 SysV assembly")
    from test_emulated_kernels import build_emulated_library
    env = dict(os.environ, FUIF_AMD_LIB=build_emulated_library(), EMU_ALARM="600")
    r = subprocess.run([sys.executable, "-c", DRIVER % dict(root=ROOT, cache=str(tmp_path / "cache"), damage=damage)], env=env, capture_output=True, text=True, timeout=900)
    if damage:
        assert r.returncode != 0 and "differs from its source picture" in r.stderr, r.stderr[-600:]
        return
    assert r.returncode == 0, r.stderr[-1500:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["config"]["parity_roundtrip_ok"] is True and line["steps"] == 2 and line["n_gpus"] == 1
    assert "pipelined" in line["config"] and line["roofline"]["bound"] == "hbm"
