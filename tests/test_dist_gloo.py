"""CPU, world_size 2 over gloo: the N>1 glue of the decode path (sharding, max-over-ranks timing,
final gather of packed pictures and of checksums) without GPUs.  The decode itself has no CPU path: the
checksum test fabricates the planes a shard would produce (a pure function of the global image index), the
packed-gather test takes the pictures from the CPU oracle (two golden fixtures of different geometry) in
the byte layout k_pack_samples writes on the GPU (tests/test_gpu_parity.py checks that layout there)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from fuif_amd import dist as fd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_planes(global_index, elems):
    g = torch.Generator().manual_seed(1234 + global_index)
    return torch.randint(0, 256, (elems,), dtype=torch.int32, generator=g)


def _worker(rank, world, port, n_items, elems, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cpu")
    dist = fd.init(device=dev)
    assert dist is not None and dist.get_backend() == "gloo"
    lo, hi = fd.shard_range(n_items, rank, world)
    # strong-scaling split: pad shards to the same length for all_gather
    per = -(-n_items // world)
    view = torch.zeros(per, elems, dtype=torch.int32)
    for k, gi in enumerate(range(lo, hi)):
        view[k] = _fake_planes(gi, elems)
    checks = fd.plane_checksums(view)
    gathered = fd.gather_checksums(checks, dist)
    t = fd.max_over_ranks(0.5 + rank, dist, dev)
    ok = fd.all_ok(rank == 0 or True, dist, dev)
    bad = fd.all_ok(rank != 1, dist, dev)
    dist.barrier()
    q.put((rank, lo, hi, gathered.numpy(), t, ok, bad))
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    for n in (1, 7, 8, 1024, 8192, 8191):
        for world in (1, 2, 3, 8):
            spans = [fd.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gather_over_gloo():
    world, n_items, elems = 2, 7, 1000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, elems, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expected = fd.plane_checksums(torch.stack([_fake_planes(i, elems) for i in range(n_items)])).numpy()
    for rank, lo, hi, gathered, t, ok, bad in res:
        assert gathered.shape == (world, 4)
        assert t == 1.5            # max over ranks of (0.5, 1.5)
        assert ok and not bad      # a failure on one rank is seen by all
        flat = []
        for r in range(world):
            a, b = fd.shard_range(n_items, r, world)
            flat.extend(gathered[r][: b - a])
        assert np.array_equal(np.array(flat), expected)


def _packed_payload(name):
    """interleaved, clamped 8-bit samples of a golden fixture as the reference CLI would write them (export/write_pam.h:136-150)"""
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle_py import Port
    blob = open(os.path.join(ROOT, "tests", "golden", name + ".fuif"), "rb").read()
    d = Port().decode(blob)
    planes = np.stack([np.clip(c["data"], 0, 255) for c in d.channels[:3]], axis=-1).astype(np.uint8)
    return planes.tobytes()


def _gather_worker(rank, world, port, payloads, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist = fd.init(device=torch.device("cpu"))
    local = torch.frombuffer(bytearray(payloads[rank]), dtype=torch.uint8)
    got = fd.gather_packed(local, dist, root=0, chunk_bytes=100_000)   # several chunks, the last one ragged, rank 1 ends early
    dist.barrier()
    q.put((rank, None if got is None else [g.numpy().tobytes() for g in got]))
    dist.destroy_process_group()


def test_two_rank_gather_of_packed_pictures_of_two_geometries():
    # rank 0: three 97x61 pictures (53 KB), rank 1: one 512x512 picture (786 KB): lengths differ, neither a chunk multiple
    payloads = [_packed_payload("rgb8_97x61") * 3, _packed_payload("c1_rgb8_512x512")]
    assert len(payloads[0]) == 3 * 97 * 61 * 3 and len(payloads[1]) == 512 * 512 * 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, payloads, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1] is None                        # only the root holds the gathered pictures
    assert res[0] == payloads                    # byte for byte, in rank order


def _world1_worker(q):
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        os.environ.pop(k, None)
    dev = torch.device("cpu")
    assert fd.init(device=dev) is None                       # the plain one-process case: no process group
    dist = fd.init(backend="gloo", device=dev, world1=True)   # the self-check form: a group of ONE rank on the real backend
    local = torch.arange(1000, dtype=torch.int64).to(torch.uint8)
    kept = fd.gather_packed(local, dist, chunk_bytes=300, keep=True)
    sums = fd.gather_packed(local, dist, chunk_bytes=300, keep=False)
    q.put((dist.get_world_size(), torch.equal(kept[0], local), sums == [int(local.sum(dtype=torch.int64))], fd.max_over_ranks(0.5, dist, dev), fd.all_ok(True, dist, dev)))
    dist.destroy_process_group()


def test_world_size_one_process_group_runs_the_collectives():
    """fd.init(world1=True): what bench.py's RCCL self-check and tests/test_gpu_rccl_world1.py use on one GPU, here on gloo"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_world1_worker, args=(q,))
    p.start()
    res = q.get(timeout=120)
    p.join(60)
    assert res == (1, True, True, 0.5, True)
