"""CPU: `python bench.py --gpus N` must start N ranks itself (`--gpus` was parsed and ignored, a plain invocation
ran one rank and printed "n_gpus": 1), and must refuse a launcher environment that disagrees with --gpus."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def run(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_plain_invocation_with_gpus_2_starts_two_ranks():
    r = run(["--gpus", "2", "--launcher-selftest"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["max_over_ranks_ok"]


def test_one_gpu_needs_no_launcher():
    r = run(["--gpus", "1", "--launcher-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1


def test_world_size_that_disagrees_with_gpus_is_refused():
    r = run(["--gpus", "4", "--launcher-selftest"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)
    r = run(["--gpus", "1", "--launcher-selftest"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)
