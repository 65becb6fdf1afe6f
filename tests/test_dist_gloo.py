"""CPU, world_size 2 over gloo: the N>1 glue of the decode path (sharding, max-over-ranks timing,
final checksum gather) without GPUs.  The decode itself has no CPU path, so each rank fabricates
the planes its shard would produce (a pure function of the global image index)."""
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
