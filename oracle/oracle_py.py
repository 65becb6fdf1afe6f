"""ctypes front-ends for the two CPU checkers -- TEST INFRASTRUCTURE ONLY.

* ``Port``  : oracle/libfuiforacle.so, the repo's plain-C restatement (oracle/fuif_oracle.c)
* ``Ref``   : oracle/_ref/libfuifref.so, the REAL cloudinary/fuif sources compiled by oracle/Makefile

Both expose the same ``decode(blob, preview=-1, undo=True, io_kind=0)`` returning a ``Decoded``.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class Decoded:
    def __init__(self):
        self.ok = False
        self.info = {}
        self.channels = []   # list of dict(meta..., data=np.int32[h,w] or flat)
        self.transforms = []
        self.stats = None

    def planes(self):
        return [c["data"] for c in self.channels]


_INFO_KEYS = ["w", "h", "minval", "maxval", "nb_channels", "real_nb_channels", "nb_meta_channels", "nch", "ntr", "error"]
_CH_KEYS = ["w", "h", "minval", "maxval", "q", "hshift", "vshift", "hcshift", "vcshift", "component", "zero", "size"]


class _Lib:
    prefix = None

    def __init__(self, path):
        self.path = path
        self.lib = C.CDLL(path)
        p = self.prefix
        f = getattr(self.lib, p + "decode")
        f.restype = C.c_void_p
        f.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_int)]
        f = getattr(self.lib, p + "undo_transforms"); f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int]
        f = getattr(self.lib, p + "free"); f.restype = None; f.argtypes = [C.c_void_p]
        f = getattr(self.lib, p + "image_info"); f.restype = None; f.argtypes = [C.c_void_p, C.c_void_p]
        f = getattr(self.lib, p + "channel_info"); f.restype = None; f.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        f = getattr(self.lib, p + "channel_data"); f.restype = None; f.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        f = getattr(self.lib, p + "transform_info"); f.restype = None; f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]

    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def decode_raw(self, blob, preview=-1, io_kind=0):
        ok = C.c_int(0)
        h = self._fn("decode")(bytes(blob), len(blob), preview, io_kind, C.byref(ok))
        return h, ok.value

    def collect(self, h, ok, want_data=True):
        d = Decoded()
        d.ok = ok == 1
        d.status = ok
        info = np.zeros(10, np.int32)
        self._fn("image_info")(h, info.ctypes.data)
        d.info = dict(zip(_INFO_KEYS, [int(v) for v in info]))
        for c in range(d.info["nch"]):
            ci = np.zeros(12, np.int32)
            self._fn("channel_info")(h, c, ci.ctypes.data)
            m = dict(zip(_CH_KEYS, [int(v) for v in ci]))
            if want_data:
                data = np.zeros(max(m["size"], 1), np.int32)
                if m["size"]:
                    self._fn("channel_data")(h, c, data.ctypes.data)
                data = data[: m["size"]]
                if m["size"] == m["w"] * m["h"]:
                    data = data.reshape(m["h"], m["w"])
                m["data"] = data
            d.channels.append(m)
        for t in range(d.info["ntr"]):
            ti = np.zeros(512, np.int32)
            self._fn("transform_info")(h, t, ti.ctypes.data, 512)
            d.transforms.append((int(ti[0]), [int(v) for v in ti[2: 2 + min(int(ti[1]), 510)]]))
        return d

    def decode(self, blob, preview=-1, undo=True, io_kind=0, keep=0, want_data=True):
        h, ok = self.decode_raw(blob, preview, io_kind)
        try:
            if ok == 1 and undo:
                if not self._fn("undo_transforms")(h, keep):
                    ok = 0
            d = self.collect(h, ok, want_data)
            if isinstance(self, Port):
                st = np.zeros(4, np.uint64)
                self.lib.fo_stats(C.c_void_p(h), st.ctypes.data)
                d.stats = dict(symbols=int(st[0]), rac_decisions=int(st[1]), tree_steps=int(st[2]), bytes=int(st[3]),
                               max_tree_nodes=int(self.lib.fo_max_tree_nodes(C.c_void_p(h))))
                gc, gs = np.zeros(4096, np.int32), np.zeros(4096, np.uint32)
                ng = self.lib.fo_groups(C.c_void_p(h), gc.ctypes.data, gs.ctypes.data, 4096)
                d.groups = [(int(gc[i]), int(gs[i])) for i in range(min(ng, 4096))]
            return d
        finally:
            self._fn("free")(h)

    def decode_both(self, blob, preview=-1, io_kind=0):
        """(pre-transform Decoded, post-transform Decoded) from ONE entropy decode."""
        h, ok = self.decode_raw(blob, preview, io_kind)
        try:
            pre = self.collect(h, ok)
            if ok == 1 and not self._fn("undo_transforms")(h, 0):
                ok = 0
            post = self.collect(h, ok)
            return pre, post
        finally:
            self._fn("free")(h)

    def time_decode(self, blob, undo=True, io_kind=0):
        """seconds for one full decode (+undo) without copying planes out"""
        import time
        t0 = time.perf_counter()
        h, ok = self.decode_raw(blob, -1, io_kind)
        if ok == 1 and undo:
            self._fn("undo_transforms")(h, 0)
        t1 = time.perf_counter()
        self._fn("free")(h)
        return t1 - t0, ok == 1


class Port(_Lib):
    prefix = "fo_"

    def __init__(self, build=True):
        path = os.path.join(HERE, "libfuiforacle.so")
        src = os.path.join(HERE, "fuif_oracle.c")
        if build and (not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src)):
            subprocess.check_call(["make", "-C", HERE, "port"], stdout=subprocess.DEVNULL)
        super().__init__(path)
        self.lib.fo_stats.restype = None
        self.lib.fo_stats.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.fo_groups.restype = C.c_int
        self.lib.fo_groups.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]


class Ref(_Lib):
    prefix = "fuifref_"

    @staticmethod
    def available():
        return os.path.exists(os.path.join(HERE, "_ref", "libfuifref.so"))

    def __init__(self):
        super().__init__(os.path.join(HERE, "_ref", "libfuifref.so"))
        self.lib.fuifref_encode.restype = C.c_size_t
        self.lib.fuifref_encode.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        self.lib.fuifref_free_blob.restype = None
        self.lib.fuifref_free_blob.argtypes = [C.c_void_p]

    def encode(self, planes, maxval=255, colorspace=-1, squeeze=1, max_group=-1, nb_repeats=0.5,
               max_properties=12, compress=1, predictor=-1, permute=0, permutation=(), softmatch=0, match_distance=0, frames=1, quant=0):
        """permute: 0 none, 1 explicit form, 2 channel form (oracle/ref_driver.cpp); permutation = new order of the channels;
        match_distance != 0: a 2D-match transform with explicit parameters (soft if softmatch; negative = previous frames, then frames >= 2);
        quant > 1: one quantization constant for every non-meta channel"""
        planes = np.ascontiguousarray(planes, dtype=np.int32)
        c, h, w = planes.shape
        perm = list(permutation) + [0] * (4 - len(permutation))
        opts = np.array([colorspace, squeeze, max_group, int(round(nb_repeats * 1000)), max_properties, compress, predictor, permute] + perm[:4] + [softmatch, match_distance, frames, quant], np.int32)
        out = C.c_void_p()
        n = self.lib.fuifref_encode(w, h, c, maxval, planes.ctypes.data, opts.ctypes.data, C.byref(out))
        if not n:
            raise RuntimeError("reference encoder failed")
        blob = C.string_at(out.value, n)
        self.lib.fuifref_free_blob(out)
        return blob


def ref_cli():
    p = os.path.join(HERE, "_ref", "fuif")
    return p if os.path.exists(p) else None


def run_ref_cli(args, **kw):
    env = dict(os.environ)
    if os.path.exists("/opt/conda/lib/libjpeg.so.9"):
        env["LD_PRELOAD"] = "/opt/conda/lib/libjpeg.so.9"
    return subprocess.run([ref_cli()] + list(args), env=env, capture_output=True, text=True, **kw)
