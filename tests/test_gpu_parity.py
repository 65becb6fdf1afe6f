"""GPU parity (-m gpu): the HIP path through the C-ABI vs the golden vectors of the real reference
and vs the oracle restatement, bit-exact, before and after the inverse transform chain."""
import numpy as np
import pytest

from conftest import all_cases, golden_blob, plane_hash

pytestmark = pytest.mark.gpu


def _decode_case(gpulib, blob, preview):
    plan = gpulib.Plan(blob)
    batch = gpulib.Batch(plan, 1, len(blob))
    try:
        batch.upload([blob], preview)
        batch.decode()
        batch.sync()
        st, used = batch.status()
        pre = batch.coef_planes(0)
        meta = batch.channel_meta(0)
        batch.undo_transforms()
        batch.sync()
        post = batch.out_planes(0)
        return pre, meta, post, int(st[0]), int(used[0])
    finally:
        batch.close()


@pytest.mark.parametrize("shard", range(8))      # (eight tests so that the emulator run of the CPU suite spreads them over its workers)
def test_golden_fixtures_bit_exact(gpulib, manifest, port, shard):
    failures = []
    for k, (e, c) in enumerate(all_cases(manifest)):
        if k % 8 != shard:
            continue
        blob = golden_blob(e, c)
        what = "%s/%s" % (e["name"], c["case"])
        pre, meta, post, st, used = _decode_case(gpulib, blob, c["preview"])
        assert (st & 2) == 0, what
        for i, (g, exp) in enumerate(zip(pre, c["pre"])):
            if exp["size"] == 0:
                ok = not g.any()          # undecoded channel reads as zeros
            elif exp["size"] != exp["w"] * exp["h"]:
                continue                  # constructor-sized plane the stream never reached
            else:
                ok = plane_hash(g) == exp["sha256"]
            if not ok:
                failures.append("%s pre channel %d" % (what, i))
                break
        assert len(post) == len(c["post"]), what
        for i, (g, exp) in enumerate(zip(post, c["post"])):
            if plane_hash(g) != exp["sha256"]:
                failures.append("%s post channel %d" % (what, i))
                break
    assert not failures, failures[:10]


def test_channel_meta_matches_oracle(gpulib, manifest, port):
    for e in manifest["fixtures"]:
        c = e["cases"][0]
        blob = golden_blob(e, c)
        pre, meta, post, st, used = _decode_case(gpulib, blob, -1)
        d = port.decode(blob, undo=False)
        assert used == d.stats["bytes"], e["name"]
        for i, ch in enumerate(d.channels):
            if ch["size"]:
                assert (meta[i][0], meta[i][1], meta[i][2]) == (ch["minval"], ch["maxval"], ch["q"]), (e["name"], i)


def test_batch_of_replicas_and_distinct_streams(gpulib, manifest):
    """several streams of one geometry in one launch: every stream decodes independently
    (full stream, byte-truncated stream and replicas of both share one batch)"""
    e = next(x for x in manifest["fixtures"] if x["name"] == "rgb8_128x128_I0")
    full = next(c for c in e["cases"] if c["case"] == "full")
    trunc = next(c for c in e["cases"] if c["case"] == "trunc50")
    b1, b2 = golden_blob(e, full), golden_blob(e, trunc)
    blobs = [b1, b2, b1, b2, b2, b1, b1]
    outs, st = gpulib.decode_batch(blobs)
    exp = {id(b1): [c["sha256"] for c in full["post"]], id(b2): [c["sha256"] for c in trunc["post"]]}
    for b, planes, s in zip(blobs, outs, st):
        assert [plane_hash(p) for p in planes] == exp[id(b)]
        assert (s & 1) == (1 if b is b2 else 0)


def test_mixed_streams_scheduler(gpulib, manifest):
    """config C5 shape: Squeeze and DCT streams of different sizes in one call, results in caller order"""
    names = ["rgb8_128x128_I0", "jpeg420_256x192_q90", "rgb8_97x61", "rgb8_128x128_E0", "jpeg444_136x120_q85", "gray8_64x48",
             "rgb8_128x128_I0", "jpeg420_256x192_q90"]
    by = {e["name"]: e for e in manifest["fixtures"]}
    blobs = [golden_blob(by[n], by[n]["cases"][0]) for n in names]
    outs, st = gpulib.decode_mixed(blobs, hbm_budget_bytes=64 << 20)   # tiny budget: exercises chunking
    assert not (st & 2).any()
    for n, planes in zip(names, outs):
        assert [plane_hash(p) for p in planes] == [c["sha256"] for c in by[n]["cases"][0]["post"]], n


@pytest.mark.parametrize("shard", range(4))
def test_packed_output_is_the_pam_payload(gpulib, manifest, port, shard):
    """fuifgpu_batch_download_packed: the bytes export/write_pam.h:136-150 would put behind the PNM/PAM header
    (interleaved, clamped, 8 bit or 16 bit big-endian), produced by the packing kernel from the final planes"""
    checked = 0
    for k, e in enumerate(manifest["fixtures"]):
        if k % 4 != shard:
            continue
        c = e["cases"][0]
        blob = golden_blob(e, c)
        info = c["info"]
        post = port.decode(blob)
        w, h = info["w"], info["h"]
        if len(post.channels) > 4 or any(ch["w"] < w or ch["h"] < h for ch in post.channels):
            continue
        plan = gpulib.Plan(blob)
        batch = gpulib.Batch(plan, 2, 2 * len(blob))
        try:
            batch.upload([blob, blob])
            batch.decode()
            batch.undo_transforms()
            batch.sync()
            got = batch.packed(1)
        finally:
            batch.close()
        want = np.stack([ch["data"][:h, :w] for ch in post.channels], axis=-1)
        assert got.shape == want.shape and np.array_equal(got.astype(np.int64), np.clip(want, 0, info["maxval"])), e["name"]
        checked += 1
    assert checked >= 2


def test_undo_transforms_is_once_per_decode(gpulib, manifest):
    """the inverse schedule rewrites channel metadata (Approximate): a second pass over the same decode is refused instead of
    returning twice-transformed data.  The coefficient slab (int16 samples) is NOT consumed: the kernels work on a widened copy,
    so the coefficients read the same before and after"""
    by = {e["name"]: e for e in manifest["fixtures"]}
    e = by["jpeg420_256x192_q90"] if "jpeg420_256x192_q90" in by else manifest["fixtures"][0]
    blob = golden_blob(e, e["cases"][0])
    plan = gpulib.Plan(blob)
    batch = gpulib.Batch(plan, 1, len(blob))
    try:
        batch.upload([blob])
        batch.decode()
        before = batch.coef_planes(0)
        batch.undo_transforms()
        batch.sync()
        first = batch.out_planes(0)
        with pytest.raises(gpulib.FuifGpuError) as ei:
            batch.undo_transforms()
        assert ei.value.code == 4   # FUIFGPU_E_ARG
        after = batch.coef_planes(0)
        assert all(np.array_equal(a, b) for a, b in zip(before, after))
        batch.decode()                 # a new decode makes both legal again
        batch.undo_transforms()
        batch.sync()
        again = batch.out_planes(0)
        assert all(np.array_equal(a, b) for a, b in zip(first, again))
    finally:
        batch.close()


def test_invalid_permutation_flags_the_image_and_zero_fills(gpulib, manifest):
    """a Permute whose permutation is stream data and comes out invalid (here: the stream ends right behind its header, the
    meta-channel reads 0,0,0 -- a repeated channel number): the image is flagged corrupt and its output planes are ZERO,
    not what the previous decode left in the output slab (k_permute_plane used to leave the slab as it was)"""
    e = next(x for x in manifest["fixtures"] if x["name"] == "permute_channel_rgb8_48x40")
    blob = golden_blob(e, e["cases"][0])
    plan = gpulib.Plan(blob)
    batch = gpulib.Batch(plan, 1, len(blob))
    try:
        for stream, good in ((blob, True), (blob[: plan.info.data_start], False), (blob, True)):
            batch.upload([stream])
            batch.decode()
            batch.undo_transforms()
            batch.sync()
            st, _ = batch.status()
            planes = batch.out_planes(0)
            if good:
                assert st[0] == 0
                for g, exp in zip(planes, e["cases"][0]["post"]):
                    assert plane_hash(g) == exp["sha256"]
            else:
                assert st[0] & 2
                for g in planes:
                    assert not g.any()
    finally:
        batch.close()


def test_wide_configuration_for_two_batches_in_flight(gpulib, manifest):
    """fuifgpu_batch_set_in_flight(2): launches with few tiles (streams without index) run the wide kernel instantiation with 20 instead of 58
    supernodes in LDS (two wavefronts per SIMD, room for a second batch's launch) -- same planes, same bytes consumed; two such batches decode
    side by side on two streams (where the runtime has streams: one GPU box; the emulator runs them one after the other)"""
    picks = [e for e in manifest["fixtures"] if e["name"] in ("c1_rgb8_512x512", "rgb8_97x61", "jpeg420_256x192_q90", "rgb8_512x384_I16_bigtrees", "rgba14_80x72")]
    assert picks
    for e in picks:
        c = e["cases"][0]
        blob = golden_blob(e, c)
        plan = gpulib.Plan(blob)
        a, b = gpulib.Batch(plan, 3, 3 * len(blob)), gpulib.Batch(plan, 1, len(blob))     # (one picture: the wide configuration also on the one-wavefront emulator)
        try:
            for bt, n in ((a, 3), (b, 1)):
                bt.set_in_flight(2)
                bt.set_group_parallel(False)
                bt.upload([blob] * n)
            a.decode(); b.decode()
            a.undo_transforms(); b.undo_transforms()
            a.sync(); b.sync()
            for bt, n in ((a, 3), (b, 1)):
                st, used = bt.status()
                assert not (st & 2).any() and (used == used[0]).all(), e["name"]
                for i in range(n):
                    assert [plane_hash(p) for p in bt.out_planes(i)] == [x["sha256"] for x in c["post"]], e["name"]
        finally:
            a.close(); b.close()
