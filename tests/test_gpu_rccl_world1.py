"""RCCL on the one GPU a test box has: a process group of ONE rank on the `nccl` backend (= RCCL on ROCm) in
the same process as libfuifgpu.so.  The collectives of the N>1 path -- barrier, all_reduce (max-over-ranks timing, all_ok),
all_gather (checksums), the chunked gather of the packed pictures -- run on cuda:0 on the output of a real decode, and the
gathered bytes are checked against the CPU oracle's planes.  What this catches without an 8-GPU node: an RCCL / HIP-runtime
load-order clash (fuif_amd preloads libamdhip64 for ctypes), a missing HSA_ENABLE_IPC_MODE_LEGACY=0, a stream mix-up between
torch's current stream and the library's launch stream.  Runs in a subprocess: a process group is process-wide state."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "oracle"))
import torch
import fuif_amd
from fuif_amd import dist as fd
from oracle_py import Port

assert "WORLD_SIZE" not in os.environ or os.environ["WORLD_SIZE"] == "1"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist = fd.init(device=dev, world1=True)
assert dist is not None and dist.get_backend() == "nccl" and dist.get_world_size() == 1
port = Port()
golden = os.path.join(sys.argv[1], "tests", "golden")
total = 0
for name, copies in (("c1_rgb8_512x512", 5), ("rgb8_97x61", 3)):
    blob = open(os.path.join(golden, name + ".fuif"), "rb").read()
    exp = port.decode(blob)
    w, h = exp.info["w"], exp.info["h"]
    want = np.stack([c["data"][:h, :w] for c in exp.channels[:3]], axis=-1).astype(np.uint8).tobytes()
    plan = fuif_amd.Plan(blob)
    batch = fuif_amd.Batch(plan, copies, copies * len(blob))
    batch.upload([blob] * copies)
    batch.decode(); batch.undo_transforms(); batch.sync()
    st, _ = batch.status()
    assert not st.any(), st
    pb = batch.packed_bytes()
    assert pb == len(want), (pb, len(want))
    packed = torch.empty(copies * pb, dtype=torch.uint8, device=dev)
    batch.pack_out(packed.data_ptr(), 0, copies)
    batch.sync()
    dist.barrier()
    # the chunked gather with chunks smaller than the payload (several RCCL gather calls), kept and summed
    got = fd.gather_packed(packed, dist, root=0, chunk_bytes=max(4096, pb // 3), keep=True)
    assert len(got) == 1 and got[0].cpu().numpy().tobytes() == want * copies, name
    sums = fd.gather_packed(packed, dist, root=0, chunk_bytes=max(4096, pb // 3), keep=False)
    assert sums == [int(np.frombuffer(want, np.uint8).astype(np.int64).sum()) * copies], name
    # checksums of the int32 output slab through all_gather; max-over-ranks and all_ok through all_reduce
    out = torch.from_numpy(np.concatenate([np.concatenate([p.ravel() for p in batch.out_planes(i)]) for i in range(copies)])).to(dev).view(copies, -1)
    checks = fd.plane_checksums(out)
    gathered = fd.gather_checksums(checks, dist)
    assert gathered.shape == (1, copies) and torch.equal(gathered[0], checks) and len(set(gathered[0].tolist())) == 1
    assert abs(fd.max_over_ranks(0.25, dist, dev) - 0.25) < 1e-12
    assert fd.all_ok(True, dist, dev) and not fd.all_ok(False, dist, dev)
    total += copies
    del batch
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl world-1 ok: %d pictures gathered through %s" % (total, "nccl"))
"""


@pytest.mark.gpu
def test_rccl_collectives_at_world_size_one_on_a_real_decode(gpulib, tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU (RCCL)")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "rccl world-1 ok: 8 pictures" in r.stdout
