"""CPU: the group index trailer (csrc/index.cpp) -- host logic only, no GPU.

* the writer's trailer lists exactly the group starts the oracle walks through when it decodes
  the same stream sequentially;
* an indexed stream is still the same picture for the oracle and for the REAL reference
  (the trailer sits behind the last group and is never read);
* fuifgpu_index_append / fuifgpu_index_parse round-trip, replace an old trailer, and refuse
  indices that do not fit the stream."""
import os

import numpy as np
import pytest
from conftest import GOLDEN

import fuif_amd
from fuif_amd.synth import photographic


@pytest.mark.parametrize("w,h,c,bits,seed", [(97, 61, 3, 8, 2), (64, 48, 1, 8, 3), (80, 72, 4, 14, 4), (200, 9, 3, 8, 13)])
def test_writer_trailer_matches_oracle_group_starts(gpulib, port, w, h, c, bits, seed):
    img = photographic(w, h, c, bits, seed=seed)
    plain = gpulib.encode_image(img, bits, tree_mode=1)
    indexed = gpulib.encode_image(img, bits, tree_mode=1, index=True)
    assert indexed[: len(plain)] == plain and indexed[-4:] == b"FGIX"
    d = port.decode(indexed)
    assert d.ok and d.stats["bytes"] == len(plain)                 # the decoder stops where the stream ends
    assert all(np.array_equal(d.channels[i]["data"], img[i]) for i in range(c))
    groups = gpulib.index_parse(indexed)
    assert groups == d.groups and len(groups) > 3
    assert gpulib.index_parse(plain) == []


def test_reference_ignores_the_trailer(gpulib, ref):
    img = photographic(97, 61, 3, 8, seed=2)
    plain = gpulib.encode_image(img, 8, tree_mode=1)
    indexed = gpulib.encode_image(img, 8, tree_mode=1, index=True)
    a, b = ref.decode(plain), ref.decode(indexed)
    assert a.ok and b.ok
    assert all(np.array_equal(x["data"], y["data"]) for x, y in zip(a.channels, b.channels))


def test_jpeg_like_trailer(gpulib, port):
    from fuif_amd.jpeglike import encode_jpeg_like
    img = photographic(72, 56, 3, 8, seed=972, sigma=1.0)
    plain = encode_jpeg_like(img, 90, True)
    indexed = encode_jpeg_like(img, 90, True, index=True)
    assert indexed[: len(plain)] == plain
    d = port.decode(indexed)
    assert d.ok and gpulib.index_parse(indexed) == d.groups


def load(name):
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()


def test_append_parse_roundtrip_on_reference_written_files(gpulib, port, manifest):
    for fx in manifest["fixtures"]:
        blob = load(fx["file"])
        d = port.decode(blob)
        if not d.ok or len(d.groups) < 1:
            continue
        out = gpulib.index_append(blob, d.groups)
        assert out[: len(blob)] == blob
        assert gpulib.index_parse(out) == d.groups
        assert gpulib.index_append(out, d.groups) == out                       # replaces, does not stack
        if len(d.groups) > 2:
            coarse = d.groups[::2]
            assert gpulib.index_parse(gpulib.index_append(out, coarse)) == coarse  # any subset starting at group 0 is a valid index


def test_bad_indices_are_refused(gpulib, port):
    blob = load("rgb8_97x61.fuif")
    g = port.decode(blob).groups
    with pytest.raises(fuif_amd.FuifGpuError):
        gpulib.index_append(blob, g[1:])                                        # must start at the first group
    with pytest.raises(fuif_amd.FuifGpuError):
        gpulib.index_append(blob, [g[0], g[2], g[1]])                           # not ascending
    with pytest.raises(fuif_amd.FuifGpuError):
        gpulib.index_append(blob, g[:-1] + [(g[-1][0], len(blob) + 5)])         # outside the stream
    # a damaged trailer is not an index (and not an error): the stream decodes sequentially
    good = gpulib.index_append(blob, g)
    assert gpulib.index_parse(good[:-1] + b"Y") == []
    bad_len = good[:-8] + (10 ** 6).to_bytes(4, "little") + b"FGIX"
    assert gpulib.index_parse(bad_len) == []


def test_trailer_varints_cannot_wrap(gpulib, port):
    """a channel or offset delta wider than 64 bits (ten 7-bit groups) must not wrap the running sums back into range"""
    blob = load("rgb8_97x61.fuif")
    g = port.decode(blob).groups
    good = gpulib.index_append(blob, g[:3])
    n = len(good)
    plen = int.from_bytes(good[n - 8:n - 4], "little")
    stream = good[: n - 8 - plen]

    def trailer(entries):
        def vi(v):
            out = [v & 127]; v >>= 7
            while v:
                out.append(128 | (v & 127)); v >>= 7
            return bytes(reversed(out))
        payload = vi(1) + vi(len(entries)) + b"".join(vi(c) + vi(s) for c, s in entries)
        return payload + len(payload).to_bytes(4, "little") + b"FGIX"

    d0 = (g[0][0], g[0][1])
    ok = stream + trailer([d0, (g[1][0] - g[0][0], g[1][1] - g[0][1])])
    assert gpulib.index_parse(ok) == g[:2]
    big = (1 << 64) + (g[1][1] - g[0][1])            # 2^64 + a valid delta: wraps to the valid delta in a uint64 sum
    assert gpulib.index_parse(stream + trailer([d0, (g[1][0] - g[0][0], big)])) == []
    assert gpulib.index_parse(stream + trailer([d0, ((1 << 64) + 1, g[1][1] - g[0][1])])) == []
