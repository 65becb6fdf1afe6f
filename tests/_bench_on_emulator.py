"""TEST INFRASTRUCTURE (tests/test_bench_launcher.py): bench.py's main() end to end WITHOUT a GPU -- the library is the wavefront-emulator build
(FUIF_AMD_LIB, where "device" memory is host memory), the ranks meet on the gloo backend, and torch.cuda's entry points are replaced by no-ops so that
every line of the N > 1 path runs: sharded inputs, the overlapped timed region with its per-step checksums, the resident path's parity check against the
source pixels, the all_gather of checksums, the chunked final gather of the packed pictures, the max-over-ranks timing and rank 0's CPU baseline.
Started as   python -m torch.distributed.run --nproc-per-node N ... tests/_bench_on_emulator.py --gpus N <bench.py flags>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert "_emu" in os.environ.get("FUIF_AMD_LIB", ""), "this wrapper is for the emulated library only"
import torch  # noqa: E402


class _FakeStream:
    def __init__(self, device=None):
        self.cuda_stream = None


_real_device = torch.device
torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda *a, **k: None
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.empty_cache = lambda *a, **k: None
torch.cuda.mem_get_info = lambda *a, **k: (1 << 34, 1 << 35)
torch.cuda.Stream = _FakeStream
torch.device = lambda *a, **k: _real_device("cpu")      # bench.py's torch.device("cuda", local_rank)
import bench  # noqa: E402

if os.environ.get("FUIF_TEST_BREAK_OVERLAP"):
    # the overlapped region cannot be set up (as when two batches do not fit the device): the line must still come, from the resident steps
    import fuif_amd
    _real_batch = fuif_amd.Batch

    def _refusing(plan, n_images, cap, *a, **k):
        if k.get("streaming"):
            raise fuif_amd.FuifGpuError(7, "test: no memory for a streaming batch")
        return _real_batch(plan, n_images, cap, *a, **k)
    fuif_amd.Batch = _refusing
bench.main()
