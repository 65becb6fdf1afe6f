"""Multi-GPU glue of the FUIF decode path (one process per GPU, torch.distributed; RCCL on GPUs).

Images are independent units: a batch shards across ranks with NO data-path collective
(SURVEY.md §8e).  The only exchange is the FINAL GATHER: every rank's decoded pictures, packed on the
GPU to the interleaved 8 / 16-bit samples a PNM/PAM file holds (k_pack_samples, 6.2 MB per 1920x1080
RGB image instead of 24.9 MB of int32 planes), are collected on the root rank in chunks over RCCL
(gather_packed), next to the small per-image checksums and the barrier / max-over-ranks timing of the
bench contract.  Everything here works on CPU tensors with the gloo backend as well, which is how
tests/test_dist_gloo.py covers the N>1 path without GPUs.
"""
import os

import torch


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None, device=None, world1=False):
    """init_process_group from the torchrun environment; returns the dist module or None (world 1).
    world1=True: a single process still gets a process group of ONE rank (rendezvous on 127.0.0.1, a free port), so that
    every collective of the N>1 path -- barrier, all_reduce, all_gather, the chunked gather of the packed pictures -- runs
    through the real backend (RCCL on a GPU) on the one device there is: the RCCL / HIP-runtime self-check of bench.py and
    tests/test_gpu_rccl_world1.py."""
    rank, local_rank, world = env_world()
    if world == 1 and not world1:
        return None
    import torch.distributed as dist
    if world == 1 and "MASTER_PORT" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("LOCAL_RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, **kw)
    return dist


def shard_range(n_items, rank, world):
    """contiguous block of a globally numbered batch owned by `rank` (strong-scaling split, e.g. C5:
    8192 images over 8 GPUs).  Sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def max_over_ranks(seconds, dist, device):
    if dist is None:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def plane_checksums(view, block=8):
    """per-image position-weighted int64 checksum of an (images, elems) int32 slab"""
    n, elems = view.shape
    weights = (torch.arange(elems, device=view.device, dtype=torch.int64) % 65521) + 1
    out = torch.zeros(n, dtype=torch.int64, device=view.device)
    for i0 in range(0, n, block):
        out[i0:i0 + block] = (view[i0:i0 + block].to(torch.int64) * weights).sum(dim=1)
    return out


def gather_checksums(local, dist):
    """all_gather of equally sized per-rank checksum vectors -> (world, n_local)"""
    if dist is None:
        return local.unsqueeze(0)
    parts = [torch.zeros_like(local) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, local)
    return torch.stack(parts)


def byte_sum(t, piece=1 << 28):
    """int64 sum of a uint8 tensor in pieces: torch widens the whole operand for one sum (8 bytes per byte of input)"""
    total = 0
    for o in range(0, t.numel(), piece):
        total += int(torch.sum(t[o:o + piece], dtype=torch.int64).item())
    return total


def gather_packed(local, dist, root=0, chunk_bytes=256 << 20, keep=True):
    """Final gather: `local` is this rank's packed output, a 1-D uint8 tensor (any length, lengths may
    differ between ranks: mixed geometries, uneven shards).  Returns on the root a list with every rank's bytes in rank
    order (the root's own entry is `local` itself), elsewhere None.  The payload moves in chunks of at most chunk_bytes
    per rank and step, so the root's receive buffers are bounded and a chunk of one peer is in flight on every xGMI link
    at a time (RCCL gather = one send / recv pair per peer: the root's 7 links work in parallel).
    keep=False: the root does not store what it receives (a consumer would write it out chunk by chunk) but returns
    one int64 byte sum per rank instead -- for payloads that would not fit next to the root's own decode state."""
    if dist is None:
        return [local] if keep else [byte_sum(local)]
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local.numel()], dtype=torch.int64, device=local.device))
    sizes = [int(t.item()) for t in sizes]
    out = None
    if rank == root:
        if keep:
            out = [local if r == root else torch.empty(sizes[r], dtype=torch.uint8, device=local.device) for r in range(world)]
        else:
            out = [0] * world
            out[root] = byte_sum(local)
    longest = max(sizes)
    # equal-sized slots per step (gather needs them): a rank past its end sends an empty tail padded in the slot.  The send buffer and the
    # root's receive slots are allocated ONCE for the largest step and reused (round 4 allocated `world` fresh 256 MB slots per chunk)
    first = min(chunk_bytes, longest)
    mine_buf = torch.zeros(first, dtype=torch.uint8, device=local.device) if longest else None
    slot_bufs = [torch.empty(first, dtype=torch.uint8, device=local.device) for _ in range(world)] if (rank == root and longest) else None
    for off in range(0, longest, chunk_bytes):
        step = min(chunk_bytes, longest - off)
        mine = mine_buf[:step]
        n_mine = max(0, min(step, local.numel() - off))
        if n_mine:
            mine[:n_mine] = local[off:off + n_mine]
        if n_mine < step:
            mine[n_mine:] = 0
        slots = [b[:step] for b in slot_bufs] if rank == root else None
        dist.gather(mine, gather_list=slots, dst=root)
        if rank == root:
            for r in range(world):
                n_r = max(0, min(step, sizes[r] - off))
                if r != root and n_r:
                    if keep:
                        out[r][off:off + n_r] = slots[r][:n_r]
                    else:
                        out[r] += byte_sum(slots[r][:n_r])
    return out


def all_ok(ok, dist, device):
    if dist is None:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())
