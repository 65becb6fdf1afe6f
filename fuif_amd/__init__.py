"""fuif_amd -- MI355X (gfx950) native FUIF decode path.

Python host mirror of the reference's decode interface on top of the C-ABI library
``libfuifgpu.so`` (include/fuifgpu.h):

    reference (C++)                               here
    --------------------------------------------  ---------------------------------------------
    fuif_decode<IO>(io, image, options)           Batch.upload(blobs, preview) + Batch.decode()
      (encoding/encoding.cpp:599-720)
    Image::undo_transforms()                      Batch.undo_transforms()
      (image/image.cpp:94-115)
    Image / Channel geometry after meta_apply     Plan.coded_channels / Plan.output_channels

There is NO CPU fallback: everything that touches pixels runs in HIP kernels, and importing a
Batch without the built extension or without a GPU raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("FUIF_AMD_LIB") or os.path.join(_HERE, "libfuifgpu.so")  # FUIF_AMD_LIB: diagnostic (-DFUIF_PROF) build
_SOURCES = ["plan.cpp", "index.cpp", "writer.cpp", "maniac_decode.hip", "maniac_encode.hip", "transforms.hip", "capi.hip"]
_lib = None


class FuifGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("fuifgpu error %d: %s" % (code, msg))
        self.code = code


# -ffp-contract=off: the FP64 paths (iDCT, YCbCr) must round every product and every sum on their own like the reference's
# x86-64 build; hipcc's default fuses them into v_fma_f64 in the BACKEND, which no source pragma switches off
# (tests/test_abi_and_plan.py::test_fp64_kernels_are_not_contracted looks at the ISA)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value"]


def build(force=False, verbose=False):
    """Compile libfuifgpu.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(_HERE, "csrc", s) for s in _SOURCES]
    deps = srcs + [os.path.join(_HERE, "csrc", h) for h in ("fuifgpu_internal.h", "maniac_decode.h", "maniac_encode.h", "transforms.h", "squeeze_arith.h")]
    deps.append(os.path.join(_HERE, "..", "include", "fuifgpu.h"))
    if os.environ.get("FUIF_AMD_LIB"):
        return _LIB_PATH
    if not force and os.path.exists(_LIB_PATH) and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return _LIB_PATH
    cmd = ["hipcc"] + HIPCC_FLAGS + ["-fPIC", "-shared", "-o", _LIB_PATH] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return _LIB_PATH


def build_unit_test(force=False):
    """tests/_bin/test_fast_symbol: the hand-written symbol decoder (inline GCN asm, what ships) against fast_symbol, its C++
    specification (what the wavefront emulator of the CPU suite runs), on the MI355X -- tools/test_fast_symbol.hip includes the
    kernel source itself.  Built here (hipcc cross-compiles) so that the binary travels to the GPU box; tests/test_gpu_fast_symbol.py
    builds it there when it is missing or older than the kernel."""
    root = os.path.dirname(_HERE)
    src = os.path.join(root, "tools", "test_fast_symbol.hip")
    out = os.path.join(root, "tests", "_bin", "test_fast_symbol")
    deps = [src] + [os.path.join(_HERE, "csrc", f) for f in ("maniac_decode.hip", "maniac_decode.h", "fuifgpu_internal.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["hipcc"] + HIPCC_FLAGS + ["-I", os.path.join(_HERE, "csrc"), src, "-o", out])
    return out


class ImageInfo(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("bit_depth", C.c_int32), ("maxval", C.c_int32),
                ("nb_channels", C.c_int32), ("colormodel", C.c_int32), ("max_properties", C.c_int32),
                ("nb_frames", C.c_int32), ("nb_transforms", C.c_int32), ("nb_coded_channels", C.c_int32),
                ("nb_output_channels", C.c_int32), ("nb_ops", C.c_int32), ("responsive_offsets", C.c_int32 * 5),
                ("data_start", C.c_int32), ("coef_elems", C.c_int64), ("out_elems", C.c_int64),
                ("tmp_elems", C.c_int64), ("signature", C.c_uint64)]


class ChannelDesc(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("hshift", C.c_int32), ("vshift", C.c_int32),
                ("hcshift", C.c_int32), ("vcshift", C.c_int32), ("component", C.c_int32), ("reserved", C.c_int32),
                ("offset", C.c_int64)]


class EncodeOptions(C.Structure):
    """fuifgpu_encode_options (include/fuifgpu.h): versioned by its first field; make_encode_options() fills it in"""
    _fields_ = [("struct_size", C.c_uint32), ("ycocg", C.c_int32), ("squeeze", C.c_int32), ("max_properties", C.c_int32), ("tree_mode", C.c_int32),
                ("max_tree_nodes", C.c_int32), ("emit_index", C.c_int32), ("split_bits", C.c_int32), ("gpu_forward", C.c_int32), ("gpu_entropy", C.c_int32)]


def make_encode_options(*fields):
    """EncodeOptions with struct_size set, the fields behind it in header order"""
    return EncodeOptions(C.sizeof(EncodeOptions), *fields)


# every symbol include/fuifgpu.h declares (tests check that the library exports all of them)
ABI_SYMBOLS = [
    "fuifgpu_strerror", "fuifgpu_last_error", "fuifgpu_abi_version", "fuifgpu_plan_create", "fuifgpu_plan_destroy",
    "fuifgpu_plan_info", "fuifgpu_plan_coded_channel", "fuifgpu_plan_output_channel", "fuifgpu_plan_transform",
    "fuifgpu_build_chance_table", "fuifgpu_batch_create", "fuifgpu_batch_create_sibling", "fuifgpu_dev_mem_info", "fuifgpu_encode_images", "fuifgpu_batch_destroy", "fuifgpu_batch_upload",
    "fuifgpu_batch_decode", "fuifgpu_batch_undo_transforms", "fuifgpu_batch_sync", "fuifgpu_batch_status",
    "fuifgpu_batch_create_streaming", "fuifgpu_batch_undo_transforms_to", "fuifgpu_batch_channel_meta", "fuifgpu_batch_coef_ptr", "fuifgpu_batch_out_ptr", "fuifgpu_batch_download_coef",
    "fuifgpu_batch_download_out", "fuifgpu_batch_last_timing", "fuifgpu_batch_profile", "fuifgpu_batch_tile_log", "fuifgpu_batch_sched_stats", "fuifgpu_inv_hsqueeze", "fuifgpu_inv_vsqueeze",
    "fuifgpu_inv_ycocg", "fuifgpu_inv_ycbcr", "fuifgpu_inv_quantize", "fuifgpu_idct8x8", "fuifgpu_upsample", "fuifgpu_inv_palette", "fuifgpu_inv_approximate", "fuifgpu_inv_match", "fuifgpu_fwd_ycocg", "fuifgpu_fwd_hsqueeze", "fuifgpu_fwd_vsqueeze", "fuifgpu_encode_image", "fuifgpu_encode_channels", "fuifgpu_free_blob",
    "fuifgpu_index_parse", "fuifgpu_index_append", "fuifgpu_batch_group_index", "fuifgpu_batch_set_group_parallel",
    "fuifgpu_plan_packed_bytes", "fuifgpu_batch_pack_out", "fuifgpu_batch_download_packed",
    "fuifgpu_dev_alloc", "fuifgpu_dev_free", "fuifgpu_dev_upload", "fuifgpu_dev_download",
    "fuifgpu_plane_checksums", "fuifgpu_device_count", "fuifgpu_set_device", "fuifgpu_get_device", "fuifgpu_batch_device", "fuifgpu_peer_copy", "fuifgpu_batch_set_in_flight",
]


def _preload_hip_runtime():
    """A process must hold ONE HIP runtime.  PyTorch-ROCm ships its own libamdhip64 with the SONAME of
    /opt/rocm's; whichever is loaded first serves both, and with /opt/rocm's first torch's other bundled
    libraries no longer match it (device discovery then fails in this library).  So when torch is
    installed its copy is loaded first -- without importing torch."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            global _hip_runtime
            _hip_runtime = C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


_hip_runtime = None


def hip_runtime():
    """the HIP runtime this process uses (ctypes handle), for hosts that need a stream or an event of their own without importing torch"""
    lib()
    return _hip_runtime if _hip_runtime is not None else C.CDLL("libamdhip64.so")


def lib():
    """Load libfuifgpu.so; raises if the HIP extension was not built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError("fuif_amd/libfuifgpu.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    _preload_hip_runtime()
    L = C.CDLL(_LIB_PATH)
    vp, i32p, u8p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    L.fuifgpu_strerror.restype = C.c_char_p; L.fuifgpu_strerror.argtypes = [C.c_int]
    L.fuifgpu_last_error.restype = C.c_char_p; L.fuifgpu_last_error.argtypes = []
    L.fuifgpu_abi_version.restype = C.c_int
    L.fuifgpu_plan_create.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(vp)]
    L.fuifgpu_plan_destroy.argtypes = [vp]; L.fuifgpu_plan_destroy.restype = None
    L.fuifgpu_plan_info.argtypes = [vp, C.POINTER(ImageInfo)]
    L.fuifgpu_plan_coded_channel.argtypes = [vp, C.c_int, C.POINTER(ChannelDesc)]
    L.fuifgpu_plan_output_channel.argtypes = [vp, C.c_int, C.POINTER(ChannelDesc)]
    L.fuifgpu_plan_transform.argtypes = [vp, C.c_int, i32p, vp, C.c_int, i32p]
    L.fuifgpu_build_chance_table.argtypes = [vp, C.c_uint32, C.c_int]; L.fuifgpu_build_chance_table.restype = None
    L.fuifgpu_batch_create.argtypes = [vp, C.c_int, C.c_size_t, vp, vp, C.c_int, C.POINTER(vp)]
    L.fuifgpu_batch_create_sibling.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.fuifgpu_batch_destroy.argtypes = [vp]; L.fuifgpu_batch_destroy.restype = None
    L.fuifgpu_batch_upload.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, vp]
    L.fuifgpu_batch_decode.argtypes = [vp, vp]
    L.fuifgpu_batch_undo_transforms.argtypes = [vp, vp]
    L.fuifgpu_batch_create_streaming.argtypes = [vp, C.c_int, C.c_size_t, C.c_int, C.POINTER(vp)]
    L.fuifgpu_batch_undo_transforms_to.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    L.fuifgpu_batch_sync.argtypes = [vp, vp]
    L.fuifgpu_batch_status.argtypes = [vp, vp, vp]
    L.fuifgpu_batch_channel_meta.argtypes = [vp, C.c_int, vp]
    L.fuifgpu_batch_coef_ptr.argtypes = [vp, C.c_int]; L.fuifgpu_batch_coef_ptr.restype = vp
    L.fuifgpu_batch_out_ptr.argtypes = [vp, C.c_int]; L.fuifgpu_batch_out_ptr.restype = vp
    L.fuifgpu_batch_download_coef.argtypes = [vp, C.c_int, vp, vp]
    L.fuifgpu_batch_download_out.argtypes = [vp, C.c_int, vp, vp]
    L.fuifgpu_device_count.argtypes = [C.POINTER(C.c_int)]
    L.fuifgpu_set_device.argtypes = [C.c_int]
    L.fuifgpu_get_device.argtypes = [C.POINTER(C.c_int)]
    L.fuifgpu_batch_device.argtypes = [vp, C.POINTER(C.c_int)]
    L.fuifgpu_peer_copy.argtypes = [vp, C.c_int, vp, C.c_int, C.c_size_t, vp]
    L.fuifgpu_plane_checksums.argtypes = [vp, C.c_int64, C.c_int64, C.c_int, vp, vp]
    L.fuifgpu_batch_last_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.fuifgpu_batch_profile.argtypes = [vp, vp]
    L.fuifgpu_batch_tile_log.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int)]
    L.fuifgpu_batch_sched_stats.argtypes = [vp, vp]
    L.fuifgpu_inv_hsqueeze.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int64, C.c_int64, C.c_int64, vp]
    L.fuifgpu_inv_vsqueeze.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int64, C.c_int64, C.c_int64, vp]
    L.fuifgpu_inv_ycocg.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.fuifgpu_inv_ycbcr.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.fuifgpu_inv_quantize.argtypes = [vp, C.c_int64, C.c_int, vp]
    L.fuifgpu_idct8x8.argtypes = [C.POINTER(vp), C.c_int, C.c_int, vp, C.c_int, vp]
    L.fuifgpu_upsample.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    L.fuifgpu_inv_palette.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp, vp]
    L.fuifgpu_inv_approximate.argtypes = [vp, vp, C.c_int64, C.c_int, vp]
    L.fuifgpu_inv_match.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.fuifgpu_fwd_ycocg.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp]
    L.fuifgpu_fwd_hsqueeze.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    L.fuifgpu_fwd_vsqueeze.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    L.fuifgpu_encode_image.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(EncodeOptions), C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.fuifgpu_encode_images.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(EncodeOptions), C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.fuifgpu_free_blob.argtypes = [vp]; L.fuifgpu_free_blob.restype = None
    L.fuifgpu_index_parse.argtypes = [C.c_char_p, C.c_size_t, vp, vp, C.c_int, C.POINTER(C.c_int)]
    L.fuifgpu_index_append.argtypes = [C.c_char_p, C.c_size_t, vp, vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.fuifgpu_batch_group_index.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.POINTER(C.c_int)]
    L.fuifgpu_batch_set_group_parallel.argtypes = [vp, C.c_int]
    L.fuifgpu_batch_set_in_flight.argtypes = [vp, C.c_int]
    L.fuifgpu_plan_packed_bytes.argtypes = [vp, C.c_int]; L.fuifgpu_plan_packed_bytes.restype = C.c_size_t
    L.fuifgpu_batch_pack_out.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
    L.fuifgpu_batch_download_packed.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    _lib = L
    return L


def _check(code):
    if code != 0:
        L = lib()
        msg = L.fuifgpu_strerror(code).decode()
        last = L.fuifgpu_last_error().decode()
        raise FuifGpuError(code, msg + (" (" + last + ")" if last else ""))


class Plan:
    """Parsed header + channel table + inverse schedule of one stream (host only)."""

    def __init__(self, blob):
        L = lib()
        self._h = C.c_void_p()
        _check(L.fuifgpu_plan_create(bytes(blob[:65536]) if len(blob) > 65536 else bytes(blob), min(len(blob), 65536), C.byref(self._h)))
        self.info = ImageInfo()
        _check(L.fuifgpu_plan_info(self._h, C.byref(self.info)))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.fuifgpu_plan_destroy(self._h)
            self._h = None

    def _channels(self, fn, n):
        out = []
        for i in range(n):
            d = ChannelDesc()
            _check(fn(self._h, i, C.byref(d)))
            out.append({k: int(getattr(d, k)) for k, _ in ChannelDesc._fields_ if k != "reserved"})
        return out

    @property
    def coded_channels(self):
        return self._channels(lib().fuifgpu_plan_coded_channel, self.info.nb_coded_channels)

    @property
    def output_channels(self):
        return self._channels(lib().fuifgpu_plan_output_channel, self.info.nb_output_channels)

    @property
    def transforms(self):
        out = []
        for i in range(self.info.nb_transforms):
            tid, n = C.c_int32(), C.c_int32()
            buf = np.zeros(1024, np.int32)
            _check(lib().fuifgpu_plan_transform(self._h, i, C.byref(tid), buf.ctypes.data, 1024, C.byref(n)))
            out.append((int(tid.value), [int(v) for v in buf[: n.value]]))
        return out


class Batch:
    """Device state for ``n_images`` streams sharing one plan (geometry + transform chain)."""

    def __init__(self, plan, n_images, blob_capacity, coef_ptr=None, out_ptr=None, tmp_images=0, streaming=False):
        L = lib()
        self.plan = plan
        self.n = n_images
        self._h = C.c_void_p()
        if streaming:
            # no output slab: the inverse transforms run range by range into caller memory (undo_transforms_to)
            _check(L.fuifgpu_batch_create_streaming(plan._h, n_images, blob_capacity, tmp_images, C.byref(self._h)))
        else:
            _check(L.fuifgpu_batch_create(plan._h, n_images, blob_capacity, coef_ptr, out_ptr, tmp_images, C.byref(self._h)))
        self._keep = None

    def undo_transforms_to(self, first_image, n_images, out_device_ptr, stream=None):
        """inverse transforms of images [first, first + n) of the current decode into DEVICE memory (n * out_elems int32)"""
        _check(lib().fuifgpu_batch_undo_transforms_to(self._h, first_image, n_images, out_device_ptr, stream))

    def sibling(self, blob_capacity):
        """a second set of stream buffers over this Batch's slabs, scratch and arenas (fuifgpu_batch_create_sibling): upload
        into one while the other decodes.  Creating a sibling freezes the launch resources the two share (ABI 2), so either may be
        loaded first; close the sibling before the primary."""
        other = Batch.__new__(Batch)
        other.plan, other.n, other._keep = self.plan, self.n, None
        other._h = C.c_void_p()
        other._primary = self          # keeps the primary alive as long as the sibling
        _check(lib().fuifgpu_batch_create_sibling(self._h, blob_capacity, C.byref(other._h)))
        return other

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.fuifgpu_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def upload(self, blobs, preview=-1, stream=None):
        n = len(blobs)
        arr = (C.c_char_p * n)(*[C.c_char_p(b) for b in blobs])
        sizes = (C.c_size_t * n)(*[len(b) for b in blobs])
        self._keep = (blobs, arr, sizes)
        _check(lib().fuifgpu_batch_upload(self._h, arr, sizes, n, preview, stream))
        self.n_loaded = n

    def decode(self, stream=None):
        _check(lib().fuifgpu_batch_decode(self._h, stream))

    def undo_transforms(self, stream=None):
        _check(lib().fuifgpu_batch_undo_transforms(self._h, stream))

    def sync(self, stream=None):
        _check(lib().fuifgpu_batch_sync(self._h, stream))

    def status(self):
        st = np.zeros(self.n_loaded, np.int32)
        used = np.zeros(self.n_loaded, np.uint32)
        _check(lib().fuifgpu_batch_status(self._h, st.ctypes.data, used.ctypes.data))
        return st, used

    @property
    def device(self):
        d = C.c_int(0)
        _check(lib().fuifgpu_batch_device(self._h, C.byref(d)))
        return d.value

    def set_group_parallel(self, enable):
        """False: ignore group indices (one wavefront per image, as for streams that carry none); applies to the next upload"""
        _check(lib().fuifgpu_batch_set_group_parallel(self._h, int(bool(enable))))

    def set_in_flight(self, n_batches):
        """how many batches the host keeps in flight on this device (fuifgpu_batch_set_in_flight); applies to the next upload"""
        _check(lib().fuifgpu_batch_set_in_flight(self._h, int(n_batches)))

    def group_index(self, image):
        """[(first_channel, byte_offset)] of every channel group the last decode of `image` went through"""
        nch = self.plan.info.nb_coded_channels
        fc, st, n = np.zeros(max(nch, 1), np.int32), np.zeros(max(nch, 1), np.uint32), C.c_int(0)
        _check(lib().fuifgpu_batch_group_index(self._h, image, fc.ctypes.data, st.ctypes.data, nch, C.byref(n)))
        return [(int(fc[i]), int(st[i])) for i in range(n.value)]

    def channel_meta(self, image):
        m = np.zeros((self.plan.info.nb_coded_channels, 4), np.int32)
        _check(lib().fuifgpu_batch_channel_meta(self._h, image, m.ctypes.data))
        return m

    def coef_ptr(self, image=0):
        return lib().fuifgpu_batch_coef_ptr(self._h, image)

    def out_ptr(self, image=0):
        return lib().fuifgpu_batch_out_ptr(self._h, image)

    def timing(self):
        a, b = C.c_float(), C.c_float()
        _check(lib().fuifgpu_batch_last_timing(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def profile(self):
        """(n_loaded, 8) uint64 phase cycle counters of the last decode (diagnostic builds only)"""
        out = np.zeros((self.n_loaded, 8), np.uint64)
        _check(lib().fuifgpu_batch_profile(self._h, out.ctypes.data))
        return out

    def tile_log(self):
        """(n_tiles, 4) uint64 schedule of the last decode launch (see fuifgpu_batch_tile_log); the first call arms logging"""
        n = C.c_int(0)
        _check(lib().fuifgpu_batch_tile_log(self._h, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 4), np.uint64)
        if n.value:
            _check(lib().fuifgpu_batch_tile_log(self._h, out.ctypes.data, n.value, C.byref(n)))
        return out[: n.value]

    def sched_stats(self):
        out = np.zeros(8, np.uint64)
        _check(lib().fuifgpu_batch_sched_stats(self._h, out.ctypes.data))
        return out

    def coef_planes(self, image):
        """coded channel planes of one image as a list of (h,w) int32 arrays (device -> host)"""
        slab = np.zeros(max(self.plan.info.coef_elems, 1), np.int32)
        _check(lib().fuifgpu_batch_download_coef(self._h, image, slab.ctypes.data, None))
        return [slab[c["offset"]: c["offset"] + c["w"] * c["h"]].reshape(c["h"], c["w"]).copy() for c in self.plan.coded_channels]

    def packed(self, image, components=0):
        """(h, w, components) uint8 / uint16 samples of one decoded image, interleaved and clamped on the GPU
        (the payload export/write_pam.h would write); call after undo_transforms()"""
        info = self.plan.info
        n = lib().fuifgpu_plan_packed_bytes(self.plan._h, components)
        if not n:
            raise FuifGpuError(4, "these output channels cannot be packed")
        buf = np.zeros(n, np.uint8)
        _check(lib().fuifgpu_batch_download_packed(self._h, image, components, buf.ctypes.data, None))
        bps = 2 if info.maxval > 255 else 1
        comps = n // (info.w * info.h * bps)
        if bps == 2:
            return buf.view(">u2").astype(np.uint16).reshape(info.h, info.w, comps)
        return buf.reshape(info.h, info.w, comps)

    def packed_bytes(self, components=0):
        return int(lib().fuifgpu_plan_packed_bytes(self.plan._h, components))

    def pack_out(self, dst_device_ptr, first_image=0, n_images=None, components=0, stream=None):
        """interleaved clamped 8 / 16-bit samples of images [first, first+n) into DEVICE memory (n * packed_bytes bytes):
        the payload of the final gather (dist.gather_packed) and of a PNM/PAM file; call after undo_transforms()"""
        n = self.n_loaded - first_image if n_images is None else n_images
        _check(lib().fuifgpu_batch_pack_out(self._h, first_image, n, components, dst_device_ptr, stream))

    def out_planes(self, image):
        slab = np.zeros(max(self.plan.info.out_elems, 1), np.int32)
        _check(lib().fuifgpu_batch_download_out(self._h, image, slab.ctypes.data, None))
        return [slab[c["offset"]: c["offset"] + c["w"] * c["h"]].reshape(c["h"], c["w"]).copy() for c in self.plan.output_channels]


def device_count():
    n = C.c_int(0)
    _check(lib().fuifgpu_device_count(C.byref(n)))
    return n.value


def set_device(device):
    """the calling thread's current device (fuifgpu_set_device): batches created afterwards live there"""
    _check(lib().fuifgpu_set_device(int(device)))


def get_device():
    d = C.c_int(0)
    _check(lib().fuifgpu_get_device(C.byref(d)))
    return d.value


def plane_checksums(planes_device_ptr, elems_per_image, n_images, sums_device_ptr, stream=None, image_stride=None):
    """fuifgpu_plane_checksums: one position-weighted 64-bit sum per image of a device slab of int32 planes into DEVICE memory
    (n_images x uint64), asynchronous on `stream`"""
    _check(lib().fuifgpu_plane_checksums(planes_device_ptr, elems_per_image, elems_per_image if image_stride is None else image_stride,
                                        n_images, sums_device_ptr, stream))


def decode_batch(blobs, preview=-1, undo=True):
    """Decode same-geometry streams on the GPU; returns (list of per-image output planes, status)."""
    plan = Plan(blobs[0])
    cap = sum(len(b) for b in blobs)
    batch = Batch(plan, len(blobs), cap)
    try:
        batch.upload(blobs, preview)
        batch.decode()
        if undo:
            batch.undo_transforms()
        batch.sync()
        st, used = batch.status()
        outs = [batch.out_planes(i) if undo else batch.coef_planes(i) for i in range(len(blobs))]
        return outs, st
    finally:
        batch.close()


# learned trees: 0 = the writer's default rule (description length of the extra leaf), > 0 = flat bits a split must save.
# The test suite sets 16 (bushy trees on small pictures: coverage of the context-tree walk).
DEFAULT_SPLIT_BITS = 0


def encode_image(planes, bit_depth=8, ycocg=True, squeeze=True, max_properties=12, tree_mode=1, max_tree_nodes=4095, index=False,
                 split_bits=None, gpu_forward=False, gpu_entropy=False):
    """(C,H,W) int32 planes -> lossless .fuif bytes (host C++ writer, csrc/writer.cpp).
    index=True appends the group index trailer (csrc/index.cpp) that unlocks one-wavefront-per-group decoding.
    gpu_forward=True runs the forward YCoCg and Squeeze on the GPU (fuifgpu_fwd_*), gpu_entropy=True the MANIAC pixel loop of
    every compressed group (csrc/maniac_encode.hip): same bytes either way."""
    split_bits = DEFAULT_SPLIT_BITS if split_bits is None else split_bits
    planes = np.ascontiguousarray(planes, dtype=np.int32)
    c, h, w = planes.shape
    opt = make_encode_options(int(ycocg), int(squeeze), max_properties, tree_mode, max_tree_nodes, int(index), int(split_bits), int(gpu_forward), int(gpu_entropy))
    out = C.c_void_p()
    n = C.c_size_t(0)
    _check(lib().fuifgpu_encode_image(planes.ctypes.data, w, h, c, bit_depth, C.byref(opt), C.byref(out), C.byref(n)))
    blob = C.string_at(out.value, n.value)
    lib().fuifgpu_free_blob(out)
    return blob


def encode_images(images, bit_depth=8, ycocg=True, squeeze=True, max_properties=12, tree_mode=1, max_tree_nodes=4095, index=False,
                  split_bits=None, gpu_forward=False):
    """a batch of equally sized (C,H,W) pictures -> list of .fuif byte strings, what encode_image writes for each of them; the
    MANIAC pixel loops of ALL their channel groups run in one launch pair on the GPU (fuifgpu_encode_images)"""
    split_bits = DEFAULT_SPLIT_BITS if split_bits is None else split_bits
    arrs = [np.ascontiguousarray(im, dtype=np.int32) for im in images]
    c, h, w = arrs[0].shape
    if any(a.shape != (c, h, w) for a in arrs):
        raise FuifGpuError(4, "encode_images: all pictures of a batch must have one shape")
    n = len(arrs)
    opt = make_encode_options(int(ycocg), int(squeeze), max_properties, tree_mode, max_tree_nodes, int(index), int(split_bits), int(gpu_forward), 1)
    ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
    outs = (C.c_void_p * n)()
    sizes = (C.c_size_t * n)()
    _check(lib().fuifgpu_encode_images(ptrs, n, w, h, c, bit_depth, C.byref(opt), outs, sizes))
    blobs = []
    for k in range(n):
        blobs.append(C.string_at(outs[k], sizes[k]))
        lib().fuifgpu_free_blob(C.c_void_p(outs[k]))
    return blobs


def index_parse(blob):
    """[(first_channel, byte_offset)] from the stream's group index trailer (csrc/index.cpp); [] when it has none"""
    fc, st, n = np.zeros(4096, np.int32), np.zeros(4096, np.uint32), C.c_int(0)
    _check(lib().fuifgpu_index_parse(blob, len(blob), fc.ctypes.data, st.ctypes.data, 4096, C.byref(n)))
    return [(int(fc[i]), int(st[i])) for i in range(min(n.value, 4096))]


def index_append(blob, groups):
    """the stream with a group index trailer built from [(first_channel, byte_offset)] (an existing trailer is replaced)"""
    fc = np.array([g[0] for g in groups], np.int32)
    st = np.array([g[1] for g in groups], np.uint32)
    out, n = C.c_void_p(), C.c_size_t(0)
    _check(lib().fuifgpu_index_append(blob, len(blob), fc.ctypes.data, st.ctypes.data, len(groups), C.byref(out), C.byref(n)))
    res = C.string_at(out.value, n.value)
    lib().fuifgpu_free_blob(out)
    return res


def add_group_index(blobs, hbm_budget_bytes=200 << 30):
    """Existing streams (as the reference encoder writes them) -> the same bytes + the group index trailer, in the caller's order:
    one entropy-decode launch per geometry (a streaming batch: no output slab, no inverse transforms), the group starts the kernel
    went through (fuifgpu_batch_group_index) appended with index_append.  A stream that already has a valid trailer, or whose decode
    is flagged (truncated / corrupt: its group starts are not those of the whole file), or which the planner refuses (out of scope,
    corrupt header), comes back unchanged.  The Python form of
    fuif_amd/boundary/fuif_index_main.cpp."""
    out = list(blobs)
    todo = []
    for i, b in enumerate(blobs):
        try:
            if not index_parse(b):
                Plan(b)                     # a stream the planner refuses (out of scope, corrupt header) is copied through, like a flagged decode
                todo.append(i)
        except FuifGpuError:
            pass
    for sig, (plan, idx) in group_by_signature([blobs[i] for i in todo]).items():
        idx = [todo[k] for k in idx]
        per = 2 * plan.info.coef_elems + 19 * (1 << 20) + max(len(blobs[i]) for i in idx)
        chunk = max(1, min(len(idx), int(hbm_budget_bytes // per), 65535))
        for c0 in range(0, len(idx), chunk):
            part = idx[c0:c0 + chunk]
            sub = [blobs[i] for i in part]
            # (tmp_images = 1: an entropy-only run needs no transform arena; the default would reserve one for a whole chunk of images)
            batch = Batch(plan, len(sub), sum(len(b) for b in sub) + 4096 * len(sub), streaming=True, tmp_images=1)
            try:
                batch.upload(sub)
                batch.decode()
                batch.sync()
                st, _ = batch.status()
                for k, i in enumerate(part):
                    if st[k] == 0:
                        out[i] = index_append(blobs[i], batch.group_index(k))
            finally:
                batch.close()
    return out


def group_by_signature(blobs):
    """host-side batch scheduler, step 1: streams that share geometry + transform chain (equal
    fuifgpu_image_info::signature) can share a launch.  Returns {signature: (Plan, [indices])} in
    first-seen order.  Pure host code (no GPU)."""
    groups = {}
    for i, b in enumerate(blobs):
        p = Plan(b)
        sig = int(p.info.signature)
        if sig not in groups:
            groups[sig] = (p, [])
        groups[sig][1].append(i)
    return groups


def plan_bytes_per_image(plan, avg_blob_bytes=0):
    """HBM bytes one in-flight image needs: coefficient + output slabs + decoder scratch + its stream"""
    return 2 * plan.info.coef_elems + 4 * plan.info.out_elems + 19 * (1 << 20) + int(avg_blob_bytes)   # int16 coefficients, int32 outputs


def decode_mixed(blobs, preview=-1, hbm_budget_bytes=200 << 30):
    """Decode an arbitrary list of streams (mixed sizes / Squeeze and DCT chains, BASELINE config C5's
    shape): group by signature, run one batch per group in chunks that fit `hbm_budget_bytes`, and
    return the per-image output planes in the caller's order plus the status words."""
    outs = [None] * len(blobs)
    status = np.zeros(len(blobs), np.int32)
    for sig, (plan, idx) in group_by_signature(blobs).items():
        per = plan_bytes_per_image(plan, sum(len(blobs[i]) for i in idx) / len(idx))
        chunk = max(1, min(len(idx), int(hbm_budget_bytes // per), 65535))
        for c0 in range(0, len(idx), chunk):
            part = idx[c0:c0 + chunk]
            sub = [blobs[i] for i in part]
            batch = Batch(plan, len(sub), sum(len(b) for b in sub))
            try:
                batch.upload(sub, preview)
                batch.decode()
                batch.undo_transforms()
                batch.sync()
                st, _ = batch.status()
                for k, i in enumerate(part):
                    outs[i] = batch.out_planes(k)
                    status[i] = st[k]
            finally:
                batch.close()
    return outs, status
