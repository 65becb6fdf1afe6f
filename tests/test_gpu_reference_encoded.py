"""GPU parity (-m gpu) on LARGE streams written by the reference ENCODER ("Next round" item 1).

Every parity test at 1080p and above used streams from the product's own writer (fuif_amd/csrc/writer.cpp), whose context trees
stop at 4095 nodes; the reference encoder (encoding/encoding.cpp:455-573, maniac/compound_enc.h) learns up to 65 535 nodes per
channel group (maniac/compound.h:277-320, childID is uint16: compound.h:46) and reaches 5 225 on a 4K photographic picture with
default flags, 10 135 on a 1080p one with `-I 2`.  Here the unmodified reference CLI (oracle/_ref/fuif, which travels to the GPU
box prebuilt) encodes one 1920x1080 and one 3840x2160 `photographic(sigma=3)` picture ON THE BOX, and the HIP path decodes each

  (a) as written -- no group index: one wavefront per picture (the wide configuration), and
  (b) with the group index appended (fuif_amd.add_group_index / index_append): one wavefront per channel group,

and every coded plane before the inverse transforms, every output plane after them, the channel metadata and the bytes consumed
are compared with the REAL reference decoding the same file (oracle/_ref/libfuifref.so, `Ref().decode_both`); the byte count comes
from the plain-C restatement (the reference's API does not expose it), which the CPU suite pins to the reference.  The test asserts
through the oracle's statistics that a tree of more than 4095 nodes was in play.  A third case runs the product's writer with the
cap lifted to 40 000 nodes and a split threshold of one bit (39 999 nodes on an 800x600 picture).
"""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from fuif_amd.synth import photographic, write_pnm

pytestmark = pytest.mark.gpu


def _decode(gpulib, blob, parallel):
    plan = gpulib.Plan(blob)
    batch = gpulib.Batch(plan, 1, len(blob) + 4096)
    try:
        batch.set_group_parallel(parallel)
        batch.upload([blob])
        batch.decode()
        batch.sync()
        st, used = batch.status()
        pre = batch.coef_planes(0)
        meta = batch.channel_meta(0)
        groups = batch.group_index(0)
        batch.undo_transforms()
        batch.sync()
        post = batch.out_planes(0)
        return dict(pre=pre, post=post, meta=meta, st=int(st[0]), used=int(used[0]), groups=groups)
    finally:
        batch.close()


def _compare(got, r_pre, r_post, what):
    assert got["st"] == 0, what
    assert len(got["pre"]) == len(r_pre.channels), what
    for i, (g, e) in enumerate(zip(got["pre"], r_pre.channels)):
        if e["size"]:
            assert np.array_equal(g, e["data"]), "%s: coded plane %d differs from the reference's" % (what, i)
            assert (int(got["meta"][i][0]), int(got["meta"][i][1]), int(got["meta"][i][2])) == (e["minval"], e["maxval"], e["q"]), (what, i)
    assert len(got["post"]) == len(r_post.channels), what
    for i, (g, e) in enumerate(zip(got["post"], r_post.channels)):
        assert np.array_equal(g, e["data"]), "%s: output plane %d differs from the reference's" % (what, i)


def _reference_cli_encode(jobs):
    """[(w, h, seed, flags)] -> [(pixels, stream bytes)]: the reference CLI on PPM files, the encodes running side by side"""
    from oracle_py import ref_cli
    cli = ref_cli()
    if cli is None:
        pytest.skip("oracle/_ref/fuif (the reference CLI) was not built")
    env = dict(os.environ)
    if os.path.exists("/opt/conda/lib/libjpeg.so.9"):
        env["LD_PRELOAD"] = "/opt/conda/lib/libjpeg.so.9"
    tmp = tempfile.mkdtemp()
    procs = []
    for k, (w, h, seed, flags) in enumerate(jobs):
        img = photographic(w, h, 3, 8, seed=seed)            # sigma = 3: SURVEY 8(d)'s generator
        src, out = os.path.join(tmp, "%d.ppm" % k), os.path.join(tmp, "%d.fuif" % k)
        write_pnm(src, img, 255)
        procs.append((img, out, subprocess.Popen([cli] + list(flags) + [src, out], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)))
    res = []
    for img, out, p in procs:
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0 and os.path.exists(out), err[-300:]
        with open(out, "rb") as f:
            res.append((img, f.read()))
    return res


# (1080p with two tree-learning passes: 10 135 nodes; 4K with default flags: 5 225 nodes -- measured in the build container, asserted below)
CASES = [(1920, 1080, 77, ["-I", "2"]), (3840, 2160, 1000, [])]


@pytest.fixture(scope="module")
def reference_encoded():
    return _reference_cli_encode(CASES)


@pytest.mark.parametrize("k", range(len(CASES)))
def test_reference_encoded_streams_at_size(gpulib, ref, port, reference_encoded, k):
    w, h = CASES[k][:2]
    img, blob = reference_encoded[k]
    stats = port.decode(blob, want_data=False).stats
    assert stats["max_tree_nodes"] > 4095, stats          # beyond anything the product's writer makes
    r_pre, r_post = ref.decode_both(blob)                  # the real reference, FileIO semantics
    assert r_pre.ok
    for c in range(3):
        assert np.array_equal(r_post.channels[c]["data"], img[c])        # (lossless: reference == source)
    # (a) as the reference wrote it
    seq = _decode(gpulib, blob, parallel=False)
    _compare(seq, r_pre, r_post, "%dx%d as written" % (w, h))
    assert seq["used"] == stats["bytes"]
    assert seq["groups"] == port.decode(blob, undo=False, want_data=False).groups
    # (b) with the group index: the trailer from the group starts the kernel went through (what add_group_index appends)
    indexed = gpulib.index_append(blob, seq["groups"])
    if k == 0:
        assert gpulib.add_group_index([blob])[0] == indexed
    par = _decode(gpulib, indexed, parallel=True)
    _compare(par, r_pre, r_post, "%dx%d with group index" % (w, h))
    assert par["used"] == stats["bytes"] and par["groups"] == seq["groups"]


def test_writer_stream_with_the_node_cap_lifted(gpulib, port):
    """the product's writer with the node cap lifted (the format's limit is 65 535, compound.h:46) and a split that only has to save
    one bit: 39 999 nodes in the long groups' trees -- more supernodes than a wavefront's scratch area holds, so part of every walk
    goes node by node through the parse-order array -- as written (one wavefront) and with the group index (context areas)"""
    img = photographic(800, 600, 3, 8, seed=4242)
    blob = gpulib.encode_image(img, 8, tree_mode=1, split_bits=1, max_tree_nodes=40000)
    stats = port.decode(blob, want_data=False).stats
    assert stats["max_tree_nodes"] > 32768, stats
    d_pre, d_post = port.decode_both(blob)
    seq = _decode(gpulib, blob, parallel=False)
    _compare(seq, d_pre, d_post, "writer stream, 40000-node cap, as written")
    par = _decode(gpulib, gpulib.index_append(blob, seq["groups"]), parallel=True)
    _compare(par, d_pre, d_post, "writer stream, 40000-node cap, with group index")
    assert seq["used"] == par["used"] == stats["bytes"]
    for c in range(3):
        assert np.array_equal(par["post"][c], img[c])
