"""JPEG-transcode-like FUIF inputs (BASELINE config C3) without libjpeg.

Builds what the reference's import/read_jpeg.h:56-184 builds from a JPEG file -- quantised 8x8 DCT
coefficient planes in the reference's scan order with the transform list [YCbCr, (ChromaSubsample),
DCT, Quantize] -- directly from RGB pixels (forward YCbCr, 2x2 chroma averaging, orthonormal 8x8 DCT,
IJG quality-scaled quantisation tables) and hands them to the product's stream writer, which adds the
default Squeeze of the DC planes like the CLI does (fuif.cpp:576-578).  The decoded result is defined
by the decoder under test and the oracle; this module only has to produce VALID streams of that shape.
"""
import ctypes as C

import numpy as np

# position of natural (row-major) coefficient index bi in the reference's coefficient order
# (transform/dct.h:120-129, `jpeg_zigzag`; a property of the format)
ZIGZAG = np.array([0, 1, 4, 15, 16, 35, 36, 63, 2, 3, 5, 14, 17, 34, 37, 62, 8, 7, 6, 13, 18, 33, 38, 61,
                   9, 10, 11, 12, 19, 32, 39, 60, 24, 23, 22, 21, 20, 31, 40, 59, 25, 26, 27, 28, 29, 30, 41, 58,
                   48, 47, 46, 45, 44, 43, 42, 57, 49, 50, 51, 52, 53, 54, 55, 56])
DCT_CSHIFTS = np.array([3, 2, 2, 2] + [1] * 12 + [0] * 48)   # transform/dct.h:159-171

_LUMA_Q = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                    14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99])
_CHROMA_Q = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
                      47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32)


def _qtable(base, quality):
    s = 5000 // quality if quality < 50 else 200 - 2 * quality
    return np.clip((base * s + 50) // 100, 1, 255).astype(np.int64)


def _dct_matrix():
    k = np.zeros((8, 8))
    for u in range(8):
        a = np.sqrt(0.5) if u == 0 else 1.0
        for x in range(8):
            k[u, x] = 0.5 * a * np.cos((2 * x + 1) * u * np.pi / 16)
    return k


def dct_planes(rgb, quality=90, subsample420=True, factors=None):
    """(3,H,W) or (1,H,W) int -> list of per-component quantised coefficient arrays [bh][bw][64] + q tables.
    factors = (srh, srv) chroma subsampling factors, e.g. (2, 2) 4:2:0, (2, 1) 4:2:2, (4, 1) 4:1:1 (overrides subsample420)"""
    if factors is None:
        factors = (2, 2) if subsample420 else (1, 1)
    srh, srv = factors
    c, h, w = rgb.shape
    if c == 3:
        r, g, b = [rgb[i].astype(np.float64) for i in range(3)]
        comps = [0.299 * r + 0.587 * g + 0.114 * b,
                 -0.168736 * r - 0.331264 * g + 0.5 * b + 128.0,
                 0.5 * r - 0.418688 * g - 0.081312 * b + 128.0]
        comps = [np.clip(np.rint(x), 0, 255) for x in comps]
    else:
        comps = [rgb[0].astype(np.float64)]
    K = _dct_matrix()
    out, qts, sub = [], [], []
    for ci, p in enumerate(comps):
        s = 1 if (ci > 0 and (srh > 1 or srv > 1)) else 0
        if s:
            ph, pw = (h + srv - 1) // srv * srv, (w + srh - 1) // srh * srh
            pp = np.pad(p, ((0, ph - h), (0, pw - w)), mode="edge")
            p = np.rint(pp.reshape(ph // srv, srv, pw // srh, srh).mean(axis=(1, 3)))
        ch, cw = p.shape
        bh, bw = (ch + 7) // 8, (cw + 7) // 8
        p = np.pad(p, ((0, bh * 8 - ch), (0, bw * 8 - cw)), mode="edge") - 128.0
        blocks = p.reshape(bh, 8, bw, 8).transpose(0, 2, 1, 3)
        coef = np.einsum("ux,abxy,vy->abuv", K, blocks, K)
        qt = _qtable(_LUMA_Q if ci == 0 else _CHROMA_Q, quality)
        q = np.rint(coef.reshape(bh, bw, 64) / qt).astype(np.int32)
        out.append(q)
        qts.append(qt)
        sub.append((1 if srh > 1 else 0, 1 if srv > 1 else 0) if s else (0, 0))   # what read_jpeg.h:156-157 adds to hshift / vshift
    return out, qts, sub


_ABBREV = {(2, 2): 0, (2, 1): 1, (1, 2): 2, (4, 1): 3}   # transform/subsample.h:33-60


def encode_jpeg_like(rgb, quality=90, subsample420=True, tree_mode=1, max_properties=12, index=False, factors=None):
    """RGB (3,H,W) / gray (1,H,W) 8-bit -> .fuif bytes with the JPEG-transcode transform chain; factors = (srh, srv) of the chroma
    planes ((2, 2) = 4:2:0 when subsample420, (2, 1) = 4:2:2, (4, 1) = 4:1:1)"""
    import fuif_amd
    L = fuif_amd.lib()

    class RawChannel(C.Structure):
        _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("hshift", C.c_int32), ("vshift", C.c_int32), ("hcshift", C.c_int32),
                    ("vcshift", C.c_int32), ("component", C.c_int32), ("q", C.c_int32), ("data", C.c_void_p)]

    c, h, w = rgb.shape
    if factors is None:
        factors = (2, 2) if subsample420 else (1, 1)
    if c != 3:
        factors = (1, 1)
    coefs, qts, sub = dct_planes(rgb, quality, factors=factors)
    nb = len(coefs)
    nat_of_pos = np.argsort(ZIGZAG)          # natural index bi of coefficient position k
    chans, keep = [], []
    for i in range(64 * nb):                 # scan script: transform/dct.h:173-207
        comp, pos = i % nb, i // nb
        bi = int(nat_of_pos[pos])
        plane = np.ascontiguousarray(coefs[comp][:, :, bi], dtype=np.int32)
        keep.append(plane)
        bh, bw = plane.shape
        chans.append(RawChannel(bw, bh, 3 + sub[comp][0], 3 + sub[comp][1], int(DCT_CSHIFTS[pos]), int(DCT_CSHIFTS[pos]), comp,
                                int(qts[comp][bi]), plane.ctypes.data))
    words = []
    if nb == 3:
        words += [0, 0]                      # YCbCr
        if factors != (1, 1):
            words += [3, 1, _ABBREV[tuple(factors)]]   # ChromaSubsample, abbreviated parameter (subsample.h:33-60)
    words += [4, 0, 5, 0]                    # DCT (default parameters), Quantize
    arr = (RawChannel * len(chans))(*chans)
    tw = np.array(words, np.int32)
    opt = fuif_amd.make_encode_options(0, 1, max_properties, tree_mode, 4095, int(index), int(fuif_amd.DEFAULT_SPLIT_BITS), 0, 0)
    out, n = C.c_void_p(), C.c_size_t(0)
    L.fuifgpu_encode_channels.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                          C.POINTER(fuif_amd.EncodeOptions), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    fuif_amd._check(L.fuifgpu_encode_channels(arr, len(chans), w, h, nb, 8, tw.ctypes.data, len(words), C.byref(opt), C.byref(out), C.byref(n)))
    blob = C.string_at(out.value, n.value)
    L.fuifgpu_free_blob(out)
    return blob
