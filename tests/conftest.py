import hashlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The writer's default rule for learned trees (description length of the extra leaf) leaves small pictures with
    # single-leaf trees.  The tests want the opposite: deep trees on small pictures, so that the context-tree walk, the
    # supernode levels and the leaf switches are exercised at sizes the oracle finishes in seconds.
    import fuif_amd
    fuif_amd.DEFAULT_SPLIT_BITS = 16


def plane_hash(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<i4").tobytes()).hexdigest()


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        m = json.load(f)
    # FUIF_TEST_MAX_PIXELS: keep only small fixtures (set by tests/test_emulated_kernels.py, where the -m gpu tests
    # run against the CPU wavefront emulator at ~1 us per cross-lane operation)
    limit = int(os.environ.get("FUIF_TEST_MAX_PIXELS", "0"))
    if limit:
        m["fixtures"] = [e for e in m["fixtures"] if e["cases"][0]["info"]["w"] * e["cases"][0]["info"]["h"] <= limit]
    return m


@pytest.fixture(scope="session")
def port():
    from oracle_py import Port
    return Port()


@pytest.fixture(scope="session")
def ref():
    from oracle_py import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref/libfuifref.so not built (needs /root/reference)")
    return Ref()


@pytest.fixture(scope="session")
def gpulib():
    import fuif_amd
    fuif_amd.build()
    return fuif_amd


def golden_blob(entry, case):
    with open(os.path.join(GOLDEN, entry["file"]), "rb") as f:
        blob = f.read()
    return blob[: case["nbytes"]]


def all_cases(manifest):
    for e in manifest["fixtures"]:
        for c in e["cases"]:
            yield e, c
