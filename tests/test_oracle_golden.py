"""CPU: the plain-C oracle restatement vs the golden vectors produced by the REAL reference.

This is what pins the oracle (task brief ③): every fixture x {full, -R k previews, byte-truncated}
must match the reference decoder's planes (SHA-256 of int32 samples) and channel metadata both
before and after Image::undo_transforms()."""
import os

import pytest

from conftest import GOLDEN, all_cases, golden_blob, plane_hash

META = ("w", "h", "minval", "maxval", "q", "hshift", "vshift", "hcshift", "vcshift", "component", "size")


def check(dec, expected, what):
    assert len(dec.channels) == len(expected), what
    for i, (c, e) in enumerate(zip(dec.channels, expected)):
        got = {k: c[k] for k in META}
        exp = {k: e[k] for k in META}
        assert got == exp, "%s channel %d meta" % (what, i)
        assert plane_hash(c["data"]) == e["sha256"], "%s channel %d samples" % (what, i)


def test_fixture_files_intact(manifest):
    import hashlib
    for e in manifest["fixtures"]:
        with open(os.path.join(GOLDEN, e["file"]), "rb") as f:
            assert hashlib.sha256(f.read()).hexdigest() == e["file_sha256"]


def test_oracle_matches_reference_golden(manifest, port):
    n = 0
    for e, c in all_cases(manifest):
        blob = golden_blob(e, c)
        pre, post = port.decode_both(blob, preview=c["preview"], io_kind=0)
        what = "%s/%s" % (e["name"], c["case"])
        assert pre.ok == c["ok"], what
        assert pre.transforms == [tuple([t[0], t[1]]) for t in c["transforms"]] or \
            [list(t) for t in pre.transforms] == [[t[0], t[1]] for t in c["transforms"]], what
        check(pre, c["pre"], what + " pre")
        check(post, c["post"], what + " post")
        n += 1
    assert n >= 30
