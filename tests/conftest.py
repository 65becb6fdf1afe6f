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


def plane_hash(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<i4").tobytes()).hexdigest()


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


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
