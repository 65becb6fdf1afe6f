"""Multi-GPU glue of the FUIF decode path (one process per GPU, torch.distributed; RCCL on GPUs).

Images are independent units: a batch shards across ranks with NO data-path collective
(SURVEY.md §8e).  The only exchange is the final gather of per-image output checksums, plus the
barrier / max-over-ranks timing of the bench contract.  Everything here works on CPU tensors with
the gloo backend as well, which is how tests/test_dist_gloo.py covers the N>1 path without GPUs.
"""
import os

import torch


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None, device=None):
    """init_process_group from the torchrun environment; returns the dist module or None (world 1)."""
    rank, local_rank, world = env_world()
    if world == 1:
        return None
    import torch.distributed as dist
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


def all_ok(ok, dist, device):
    if dist is None:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())
