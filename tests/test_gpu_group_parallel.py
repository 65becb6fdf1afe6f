"""GPU parity (-m gpu) of the one-wavefront-per-group path (group index, csrc/index.cpp).

The same stream is decoded (a) one wavefront per image and (b) one wavefront per channel group with
the tiles of an image chasing each other row by row; (b) must be bit-identical to (a), to the
golden vectors of the real reference and to the oracle -- planes, channel ranges, status and bytes
consumed -- for reference-written files indexed after the fact, for files the product's writer
indexed itself, for previews, for mixed batches and when there are far more tiles than resident
wavefronts."""
import os

import numpy as np
import pytest

from conftest import all_cases, golden_blob, plane_hash
from fuif_amd.synth import photographic

pytestmark = pytest.mark.gpu


def _run(gpulib, blobs, preview=-1, parallel=True):
    plan = gpulib.Plan(blobs[0])
    batch = gpulib.Batch(plan, len(blobs), sum(len(b) for b in blobs))
    try:
        batch.set_group_parallel(parallel)
        batch.upload(blobs, preview)
        batch.decode()
        batch.sync()
        st, used = batch.status()
        pre = [batch.coef_planes(i) for i in range(len(blobs))]
        meta = [batch.channel_meta(i) for i in range(len(blobs))]
        groups = [batch.group_index(i) for i in range(len(blobs))]
        batch.undo_transforms()
        batch.sync()
        post = [batch.out_planes(i) for i in range(len(blobs))]
        return dict(pre=pre, meta=meta, post=post, st=[int(x) for x in st], used=[int(x) for x in used], groups=groups)
    finally:
        batch.close()


def _same(a, b, i=0, j=0):
    return (all(np.array_equal(x, y) for x, y in zip(a["pre"][i], b["pre"][j])) and
            all(np.array_equal(x, y) for x, y in zip(a["post"][i], b["post"][j])) and
            np.array_equal(a["meta"][i], b["meta"][j]) and a["st"][i] == b["st"][j])


@pytest.mark.parametrize("shard", range(4))      # (four tests: the emulator run of the CPU suite spreads them over its workers)
def test_reference_written_files_indexed_after_the_fact(gpulib, manifest, port, shard):
    """decode once sequentially, keep the group starts the kernel reports, append them as a trailer,
    decode again group-parallel: identical to the first decode and to the golden hashes"""
    failures = []
    for k, e in enumerate(manifest["fixtures"]):
        if k % 4 != shard:
            continue
        c = e["cases"][0]
        blob = golden_blob(e, c)
        seq = _run(gpulib, [blob], parallel=False)
        d = port.decode(blob, undo=False)
        assert seq["groups"][0] == d.groups, e["name"]
        indexed = gpulib.index_append(blob, seq["groups"][0])
        par = _run(gpulib, [indexed])
        if not _same(seq, par) or par["st"][0] != 0 or par["used"][0] != seq["used"][0]:
            failures.append(e["name"])
            continue
        if [plane_hash(p) for p in par["post"][0]] != [x["sha256"] for x in c["post"]]:
            failures.append(e["name"] + " (golden)")
        if par["groups"][0] != seq["groups"][0]:
            failures.append(e["name"] + " (groups)")
    assert not failures, failures


def test_add_group_index_in_one_launch_per_geometry(gpulib, manifest, port):
    """fuif_amd.add_group_index: a mixed list of reference-written streams (three geometries, one of them three times, one already
    indexed, one truncated) comes back in order with the trailer the oracle's group starts spell; the truncated one unchanged"""
    pick = ["rgb8_97x61", "rgba14_80x72", "rgb8_97x61", "pal_rgb_graphic_120x90", "rgb8_97x61"]
    blobs = []
    for n in pick:
        e = next(x for x in manifest["fixtures"] if x["name"] == n)
        blobs.append(golden_blob(e, e["cases"][0]))
    groups0 = port.decode(blobs[0], undo=False).groups
    blobs[2] = gpulib.index_append(blobs[2], groups0)              # already indexed
    blobs[4] = blobs[4][: len(blobs[4]) * 2 // 3]                   # truncated: decodes, but gets no index
    out = gpulib.add_group_index(blobs)
    assert out[2] == blobs[2] and out[4] == blobs[4]
    for i in (0, 1, 3):
        want = port.decode(blobs[i], undo=False).groups
        assert out[i][: len(blobs[i])] == blobs[i] and gpulib.index_parse(out[i]) == want, pick[i]
        assert out[i] == gpulib.index_append(blobs[i], want)
    seq, par = _run(gpulib, [blobs[1]], parallel=False), _run(gpulib, [out[1]])
    assert _same(seq, par) and par["st"][0] == 0


def test_add_group_index_copies_refused_streams_through(gpulib, manifest, port):
    """a stream the planner refuses (`fuif -E 64`: more reference properties than the GPU path takes) or cannot parse (garbage) in the
    list must not abort the call: it comes back unchanged, like fuif_index_main.cpp copies such files through"""
    import os
    from conftest import GOLDEN
    e = next(x for x in manifest["fixtures"] if x["name"] == "rgb8_97x61")
    good = golden_blob(e, e["cases"][0])
    with open(os.path.join(GOLDEN, "outside_gpu_scope_rgb8_64x48_E64.fuif"), "rb") as f:
        outside = f.read()
    garbage = b"FUIF" + bytes(range(200))
    out = gpulib.add_group_index([outside, good, garbage])
    assert out[0] == outside and out[2] == garbage
    assert gpulib.index_parse(out[1]) == port.decode(good, undo=False).groups


def test_previews_of_indexed_streams(gpulib, manifest):
    for e, c in all_cases(manifest):
        if not c["case"].startswith("preview"):
            continue
        full = golden_blob(e, e["cases"][0])
        seq_full = _run(gpulib, [full], parallel=False)
        indexed = gpulib.index_append(full, seq_full["groups"][0])
        par = _run(gpulib, [indexed], preview=c["preview"])
        assert [plane_hash(p) for p in par["post"][0]] == [x["sha256"] for x in c["post"]], (e["name"], c["case"])
        seq = _run(gpulib, [full], preview=c["preview"], parallel=False)
        assert _same(seq, par), (e["name"], c["case"])


@pytest.mark.parametrize("w,h,c,bits,seed", [(512, 384, 3, 8, 40), (97, 61, 3, 8, 2), (160, 200, 4, 14, 41), (301, 47, 1, 8, 42)])
def test_writer_indexed_streams_vs_oracle(gpulib, port, w, h, c, bits, seed):
    img = photographic(w, h, c, bits, seed=seed)
    blob = gpulib.encode_image(img, bits, tree_mode=1, index=True)
    par = _run(gpulib, [blob])
    assert par["st"][0] == 0
    pre, post = port.decode_both(blob)
    assert all(np.array_equal(g, ch["data"]) for g, ch in zip(par["pre"][0], pre.channels) if ch["size"])
    assert all(np.array_equal(par["post"][0][i], img[i]) for i in range(c))
    assert par["groups"][0] == gpulib.index_parse(blob)
    assert par["used"][0] == port.decode(blob, undo=False).stats["bytes"]


def test_mixed_batch_with_more_tiles_than_wavefronts(gpulib, port):
    """indexed and plain streams of one geometry share a launch; 192 images x ~30 groups is more tiles
    than the device holds wavefronts (4 per SIMD), so the persistent wavefronts walk the work list"""
    imgs = [photographic(160, 120, 3, 8, seed=500 + k) for k in range(6)]
    indexed = [gpulib.encode_image(im, 8, tree_mode=1, index=True) for im in imgs]
    plain = [gpulib.encode_image(im, 8, tree_mode=1) for im in imgs]
    blobs, want = [], []
    for k in range(int(os.environ.get("FUIF_TEST_BATCH", "192"))):
        src = (indexed if (k % 3) else plain)[k % 6]
        blobs.append(src)
        want.append(imgs[k % 6])
    par = _run(gpulib, blobs)
    assert par["st"] == [0] * len(blobs)
    for k in range(len(blobs)):
        assert all(np.array_equal(par["post"][k][i], want[k][i]) for i in range(3)), k
    ngroups = {len(g) for g in par["groups"]}
    assert len(ngroups) == 1 and ngroups.pop() > 20      # plain streams report their groups too


def test_jpeg_like_indexed(gpulib, port):
    from fuif_amd.jpeglike import encode_jpeg_like
    img = photographic(136, 120, 3, 8, seed=77, sigma=1.0)
    blob = encode_jpeg_like(img, 90, True, index=True)
    seq = _run(gpulib, [blob], parallel=False)
    par = _run(gpulib, [blob, blob, blob])
    assert par["st"] == [0, 0, 0] and all(_same(seq, par, 0, k) for k in range(3))
    pre, post = port.decode_both(blob)
    assert all(np.array_equal(g, ch["data"]) for g, ch in zip(par["post"][0], post.channels))


def test_truncated_indexed_stream_falls_back_and_side_index_on_truncated_blob(gpulib, port):
    """cutting an indexed file removes the trailer: the stream decodes sequentially, as the reference
    would; an index that points past the cut still gives the sequential result"""
    img = photographic(128, 96, 3, 8, seed=9)
    blob = gpulib.encode_image(img, 8, tree_mode=1, index=True)
    groups = gpulib.index_parse(blob)
    cut = blob[: groups[len(groups) // 2][1] + 7]
    seq = _run(gpulib, [cut], parallel=False)
    assert seq["st"][0] & 1
    pre, post = port.decode_both(cut)
    assert all(np.array_equal(g, ch["data"]) for g, ch in zip(seq["post"][0], post.channels))
    part = [g for g in groups if g[1] < len(cut)]
    par = _run(gpulib, [gpulib.index_append(cut, part)])
    # the appended trailer makes the file longer: what the group that hits the cut reads next differs,
    # so only the planes before it are comparable
    k = len(part) - 1
    first_cut_channel = part[k][0]
    assert all(np.array_equal(a, b) for a, b in list(zip(seq["pre"][0], par["pre"][0]))[:first_cut_channel])


def test_stale_index_is_flagged_not_silently_wrong(gpulib, port):
    """the trailer is untrusted: an index whose offsets are ascending and inside the stream but do not belong to it (here:
    one group start moved by a byte) must not yield a different picture with status 0 -- the tile before the moved start
    does not stop where the next one begins and the image is flagged corrupt (the reference ignores the trailer)"""
    img = photographic(128, 96, 3, 8, seed=11)
    blob = gpulib.encode_image(img, 8, tree_mode=1, index=True)
    groups = gpulib.index_parse(blob)
    good = _run(gpulib, [blob])
    assert good["st"][0] == 0
    k = len(groups) // 2
    moved = list(groups)
    moved[k] = (moved[k][0], moved[k][1] + 1)
    stale = gpulib.index_append(blob, moved)
    assert gpulib.index_parse(stale) == moved        # structurally a valid index
    res = _run(gpulib, [stale, blob])
    assert res["st"][0] & 2                          # corrupt
    assert res["st"][1] == 0 and _same(res, good, 1, 0)   # the intact stream next to it in the batch is untouched


@pytest.mark.parametrize("ctx_kb", ["0", "64"])
def test_pinned_tiles_when_the_context_arena_runs_out(gpulib, port, monkeypatch, ctx_kb):
    """FUIFGPU_CTX_KB (capi.hip, read at upload) sizes the per-image context arena: 0 = no arena at all, every suspendable tile keeps
    its tree and leaves in its wavefront's scratch area and is PINNED to that wavefront (suspended and resumed by it alone); 64 KB
    = the first few tiles of an image get an area, the rest are pinned.  Either way the planes, ranges, status and bytes consumed
    equal the one-wavefront-per-image decode and the oracle (an exhausted arena once meant spinning tiles)."""
    monkeypatch.setenv("FUIFGPU_CTX_KB", ctx_kb)
    img = photographic(512, 384, 3, 8, seed=40)
    blob = gpulib.encode_image(img, 8, tree_mode=1, index=True)
    n = int(os.environ.get("FUIF_TEST_PINNED_BATCH", "24"))
    par = _run(gpulib, [blob] * n)
    assert par["st"] == [0] * n
    pre, post = port.decode_both(blob)
    for k in (0, n // 2, n - 1):
        assert all(np.array_equal(g, ch["data"]) for g, ch in zip(par["pre"][k], pre.channels) if ch["size"])
        assert all(np.array_equal(par["post"][k][i], img[i]) for i in range(3))
    assert par["used"][0] == port.decode(blob, undo=False).stats["bytes"]
    # a mixed batch (indexed + plain streams, more tiles than wavefronts) under the same arena
    imgs = [photographic(160, 120, 3, 8, seed=500 + k) for k in range(3)]
    indexed = [gpulib.encode_image(im, 8, tree_mode=1, index=True) for im in imgs]
    plain = [gpulib.encode_image(im, 8, tree_mode=1) for im in imgs]
    blobs = [(indexed if (k % 3) else plain)[k % 3] for k in range(int(os.environ.get("FUIF_TEST_BATCH", "96")))]
    mixed = _run(gpulib, blobs)
    assert mixed["st"] == [0] * len(blobs)
    for k in range(len(blobs)):
        assert all(np.array_equal(mixed["post"][k][i], imgs[k % 3][i]) for i in range(3)), k
