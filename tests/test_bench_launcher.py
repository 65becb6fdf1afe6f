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


def test_bench_main_runs_end_to_end_on_two_gloo_ranks_over_the_emulated_library(tmp_path):
    """`bench.py --gpus 2` must print a line with "n_gpus": 2 that carries its own cpu_baseline.  No GPU here: the ranks run
    bench.py's real main() over the wavefront-emulator build of the library (tests/_bench_on_emulator.py replaces torch.cuda's entry points by
    no-ops), meet on gloo, decode their own shards in overlapped steps, verify every step, gather checksums and packed pictures on rank 0."""
    import socket
    if sys.platform != "linux" or os.uname().machine != "x86_64":
        import pytest
        pytest.skip("the emulator's context switch is x86-64 SysV assembly")
    from test_emulated_kernels import build_emulated_library
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, FUIF_AMD_LIB=build_emulated_library(), EMU_ALARM="900", FUIF_BENCH_CACHE=str(tmp_path / "cache"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_bench_on_emulator.py"), "--gpus", "2", "--batch", "4", "--width", "97", "--height", "61", "--distinct", "2", "--steps", "2", "--warmup", "1",
           "--slice", "3", "--no-live-traffic", "--no-cpu-all-cores"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["parity_roundtrip_ok"] is True
    assert d["overlap"]["steps_verified"] == 3 and d["overlap"]["steps_identical_to_resident_outputs"] == 3
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0 and "roofline" in d
    assert d["final_gather"]["byte_sums_ok"] is True and d["final_gather"]["bytes_into_root"] > 0
    assert abs(d["value"] - 2 * 4 * 97 * 61 * 2 / 1e6 / (d["ms_per_step"] * 2 / 1e3)) < 6e-4      # whole-job rate over both ranks (the line rounds to 3 decimals)


def test_bench_main_default_path_on_one_emulated_rank(tmp_path):
    """the default line's world-size-1 path end to end over the emulated library: overlapped steps + the resident path's parity check, the overlapped
    one-wavefront-per-picture leg, the reference-encoded leg (when oracle/_ref/fuif is built), the upload pipeline through a sibling batch, the packed
    gather, the CPU baseline -- every key a driver record of the bench is expected to carry (the small C3 / C4 / C5 legs and the live traffic passes are
    hardware-sized and stay off here; run_extra_leg has its own emulated test)"""
    if sys.platform != "linux" or os.uname().machine != "x86_64":
        import pytest
        pytest.skip("the emulator's context switch is x86-64 SysV assembly")
    from test_emulated_kernels import build_emulated_library
    env = dict(os.environ, FUIF_AMD_LIB=build_emulated_library(), EMU_ALARM="900", FUIF_BENCH_CACHE=str(tmp_path / "cache"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "tests", "_bench_on_emulator.py"), "--gpus", "1", "--batch", "4", "--width", "97", "--height", "61", "--distinct", "2",
           "--steps", "1", "--warmup", "1", "--seq-steps", "2", "--alone-steps", "1", "--slice", "3", "--no-live-traffic", "--no-cpu-all-cores", "--no-extra-legs",
           "--no-rccl-selfcheck", "--reference-encoded", "2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["parity_roundtrip_ok"] is True
    assert d["overlap"]["steps_verified"] == 2 == d["overlap"]["steps_identical_to_resident_outputs"]
    assert d["single_launch"]["steps"] == 1 and "launch_ms_alone" in d["roofline"] and "kernel_ms" not in d["roofline"]
    seq = d["one_wavefront_per_image"]
    assert seq["steps"] == 2 and seq["steps_verified"] == 2 == seq["steps_identical_to_resident_outputs"] and seq["identical_output"] is True and seq["single_launch"]["identical_output"] is True
    assert d["h2d"]["identical_output"] is True and d["value_incl_h2d"] > 0 and d["final_gather"]["byte_sums_ok"] is True
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["checker_decodes_stream0_to_source_pixels"] is True
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "fuif")):
        ref = d["reference_encoded_streams"]
        assert "error" not in ref and ref["decoded_equals_source_pixels"] is True and ref["distinct_streams"] == 2


def test_bench_line_survives_an_overlapped_region_that_cannot_be_set_up(tmp_path):
    """two streaming batches do not fit (here: refused by a test hook in tests/_bench_on_emulator.py): both gloo ranks agree on it before the region's first
    barrier, the timed steps are the resident batch's, the line says so and still carries value / roofline / cpu_baseline"""
    import socket
    if sys.platform != "linux" or os.uname().machine != "x86_64":
        import pytest
        pytest.skip("the emulator's context switch is x86-64 SysV assembly")
    from test_emulated_kernels import build_emulated_library
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, FUIF_AMD_LIB=build_emulated_library(), EMU_ALARM="900", FUIF_BENCH_CACHE=str(tmp_path / "cache"), FUIF_TEST_BREAK_OVERLAP="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_bench_on_emulator.py"), "--gpus", "2", "--batch", "3", "--width", "97", "--height", "61", "--distinct", "2", "--steps", "2", "--warmup", "1",
           "--no-live-traffic", "--no-cpu-all-cores"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["config"]["parity_roundtrip_ok"] is True and d["value"] > 0
    assert d["config"]["overlapped_steps"].startswith("FAILED") and "no memory for a streaming batch" in d["config"]["overlapped_steps"]
    assert "overlap" not in d and "kernel_ms" in d["roofline"] and d["cpu_baseline"]["value"] > 0
