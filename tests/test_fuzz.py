"""Corrupted streams: the oracle must terminate on bit-flipped payloads (CPU), and the GPU kernel must
terminate and agree with the oracle on them (-m gpu).  The real reference is not used here (it may
crash on corrupt input); this is a robustness/consistency check of the two implementations in the repo."""
import numpy as np
import pytest

from conftest import golden_blob


def corrupted(manifest, n_per_fixture=6):
    rng = np.random.default_rng(2024)
    for name in ("rgb8_97x61", "rgb8_128x128_E0", "jpeg420_256x192_q90", "rgb8_64x64_U", "rgb8_96x96_nosqueeze"):
        e = next(x for x in manifest["fixtures"] if x["name"] == name)
        blob = bytearray(golden_blob(e, e["cases"][0]))
        for k in range(n_per_fixture):
            b = bytearray(blob)
            lo = 40 if len(b) > 200 else 24          # keep the header: geometry stays plausible
            for _ in range(1 + k % 3):
                pos = int(rng.integers(lo, len(b)))
                b[pos] ^= 1 << int(rng.integers(0, 8))
            yield name, k, bytes(b)


def test_oracle_terminates_on_corrupt_payload(manifest, port):
    n = 0
    for name, k, blob in corrupted(manifest):
        d = port.decode(blob)       # must return (ok or not), never hang or crash
        assert d.info["w"] > 0
        n += 1
    assert n == 30


@pytest.mark.gpu
def test_gpu_agrees_with_oracle_on_corrupt_payload(manifest, port, gpulib):
    for name, k, blob in corrupted(manifest):
        try:
            plan = gpulib.Plan(blob)
        except gpulib.FuifGpuError:
            continue
        batch = gpulib.Batch(plan, 1, len(blob))
        try:
            batch.upload([blob])
            batch.decode()
            batch.sync()
            st, used = batch.status()
            pre = batch.coef_planes(0)
        finally:
            batch.close()
        d = port.decode(blob, undo=False)
        if d.status != 1:
            assert st[0] & 6, (name, k)          # the oracle calls it corrupt: so must the kernel
            continue
        assert (st[0] & 2) == 0, (name, k)
        for g, ch in zip(pre, d.channels):
            if ch["size"] == ch["w"] * ch["h"]:
                assert np.array_equal(g, ch["data"]), (name, k)


def test_planner_survives_corrupt_headers(manifest, gpulib):
    """host code of the product: random damage to the header / transform list (incl. the Palette, Approximate
    and 2D-match parameter lists) is either rejected with an error code or planned to a bounded geometry"""
    rng = np.random.default_rng(7)
    n = planned = 0
    for e in manifest["fixtures"]:
        blob = golden_blob(e, e["cases"][0])
        hdr = min(len(blob), 120)
        for k in range(60):
            b = bytearray(blob)
            for _ in range(1 + k % 4):
                pos = int(rng.integers(4, hdr))
                b[pos] = (b[pos] ^ (1 << int(rng.integers(0, 8)))) if k % 2 else int(rng.integers(0, 256))
            n += 1
            try:
                p = gpulib.Plan(bytes(b))
            except gpulib.FuifGpuError as err:
                assert err.code in (1, 2, 3), err      # NOT_FUIF / CORRUPT / UNSUPPORTED
                continue
            planned += 1
            assert p.info.coef_elems < (1 << 34) and p.info.out_elems < (1 << 34)
            assert len(p.coded_channels) == p.info.nb_coded_channels
            gpulib.index_parse(bytes(b))               # trailer parser on the same damage
    assert n > 1000 and planned > 100
