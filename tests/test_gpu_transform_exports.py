"""GPU parity (-m gpu) of the single-transform entry points of the C-ABI (include/fuifgpu.h: fuifgpu_inv_hsqueeze,
fuifgpu_inv_vsqueeze, fuifgpu_inv_ycocg, fuifgpu_inv_ycbcr, fuifgpu_inv_quantize, fuifgpu_idct8x8, fuifgpu_upsample, fuifgpu_inv_palette,
fuifgpu_inv_approximate, fuifgpu_inv_match) -- what
Transform::apply(image, true) (transform/transform.cpp:48-63) dispatches to in the boundary layer.

Each is run on raw device planes (random, incl. negative values, odd and tiny sizes, several planes per launch) and
compared bit for bit with the oracle's restatement of the same reference function; the known answers of SURVEY.md
Appendix E.4-E.6 (printed by the real reference) are pinned as well.  On a machine without a GPU the same tests run
against the wavefront emulator build (tests/test_emulated_kernels.py), where "device" memory is host memory."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

EMULATED = "_emu" in os.environ.get("FUIF_AMD_LIB", "")


class Dev:
    """int32 planes in device memory (torch on the GPU box; plain host memory under the emulator)"""

    def __init__(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.int32)
        if EMULATED:
            self.a = arr.copy()
            self.ptr = self.a.ctypes.data
        else:
            import torch
            self.t = torch.from_numpy(arr.copy()).cuda()
            self.ptr = self.t.data_ptr()

    def get(self):
        if EMULATED:
            return self.a.copy()
        import torch
        torch.cuda.synchronize()
        return self.t.cpu().numpy()


@pytest.fixture(scope="module")
def olib():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle_py import Port
    return Port().lib


@pytest.fixture(scope="module")
def glib(gpulib):
    L = gpulib.lib()
    if not EMULATED:
        import torch
        torch.zeros(1).cuda()          # one HIP context for torch and the library
    return L


def _sync():
    if not EMULATED:
        import torch
        torch.cuda.synchronize()


def ref_squeeze(olib, horizontal, avg, res):
    ah, aw = avg.shape
    rh, rw = res.shape
    out = np.zeros((ah, aw + rw) if horizontal else (ah + rh, aw), np.int32)
    assert olib.fo_kat_inv_squeeze(int(horizontal), avg.ctypes.data_as(C.c_void_p), aw, ah, res.ctypes.data_as(C.c_void_p), rw, rh,
                                   out.ctypes.data_as(C.c_void_p))
    return out


# (k_inv_hsqueeze_rows takes 32 pairs per step while 32 more averages exist: widths on both sides of every boundary of that loop)
@pytest.mark.parametrize("w1,w2,h", [(4, 3, 1), (4, 4, 1), (1, 0, 5), (1, 1, 3), (33, 32, 7), (64, 64, 65), (129, 128, 130), (960, 960, 9),
                                     (32, 32, 3), (32, 31, 2), (34, 33, 5), (65, 64, 257), (65, 65, 4), (96, 96, 2), (97, 96, 300), (161, 160, 1)])
def test_inv_hsqueeze_export(glib, olib, w1, w2, h):
    rng = np.random.default_rng(w1 * 1000 + h)
    n_planes = 3
    avg = rng.integers(-3000, 3000, (n_planes, h, w1), dtype=np.int32)
    res = rng.integers(-700, 700, (n_planes, h, max(w2, 1)), dtype=np.int32)[:, :, :w2].copy()
    d_avg, d_res, d_out = Dev(avg), Dev(res if w2 else np.zeros(1, np.int32)), Dev(np.zeros((n_planes, h, w1 + w2), np.int32))
    rc = glib.fuifgpu_inv_hsqueeze(d_avg.ptr, w1, d_res.ptr if w2 else None, w2, h, d_out.ptr, n_planes, h * w1, h * w2, h * (w1 + w2), None)
    assert rc == 0
    got = d_out.get().reshape(n_planes, h, w1 + w2)
    for p in range(n_planes):
        assert np.array_equal(got[p], ref_squeeze(olib, True, avg[p], res[p].reshape(h, w2)))


@pytest.mark.parametrize("h1,h2,w", [(4, 3, 1), (4, 4, 2), (1, 0, 5), (1, 1, 3), (33, 32, 7), (65, 64, 300), (540, 540, 17)])
def test_inv_vsqueeze_export(glib, olib, h1, h2, w):
    rng = np.random.default_rng(h1 * 1000 + w)
    n_planes = 2
    avg = rng.integers(-3000, 3000, (n_planes, h1, w), dtype=np.int32)
    res = rng.integers(-700, 700, (n_planes, max(h2, 1), w), dtype=np.int32)[:, :h2].copy()
    d_avg, d_res, d_out = Dev(avg), Dev(res if h2 else np.zeros(1, np.int32)), Dev(np.zeros((n_planes, h1 + h2, w), np.int32))
    rc = glib.fuifgpu_inv_vsqueeze(d_avg.ptr, h1, d_res.ptr if h2 else None, h2, w, d_out.ptr, n_planes, h1 * w, h2 * w, (h1 + h2) * w, None)
    assert rc == 0
    got = d_out.get().reshape(n_planes, h1 + h2, w)
    for p in range(n_planes):
        assert np.array_equal(got[p], ref_squeeze(olib, False, avg[p], res[p].reshape(h2, w)))


def test_squeeze_known_answers(glib, olib):
    """SURVEY.md Appendix E.4 (real reference): averages [100,104,90,91] + residuals [3,-2,5] -> 101 99 103 105 92 87 91,
    as a row (horizontal) and as a column (vertical)"""
    want = np.array([101, 99, 103, 105, 92, 87, 91], np.int32)
    avg, res = np.array([[100, 104, 90, 91]], np.int32), np.array([[3, -2, 5]], np.int32)
    d_avg, d_res, d_out = Dev(avg), Dev(res), Dev(np.zeros((1, 7), np.int32))
    assert glib.fuifgpu_inv_hsqueeze(d_avg.ptr, 4, d_res.ptr, 3, 1, d_out.ptr, 1, 4, 3, 7, None) == 0
    assert np.array_equal(d_out.get().ravel(), want)
    d_avg, d_res, d_out = Dev(avg.T.copy()), Dev(res.T.copy()), Dev(np.zeros((7, 1), np.int32))
    assert glib.fuifgpu_inv_vsqueeze(d_avg.ptr, 4, d_res.ptr, 3, 1, d_out.ptr, 1, 4, 3, 7, None) == 0
    assert np.array_equal(d_out.get().ravel(), want)
    assert np.array_equal(ref_squeeze(olib, True, avg, res).ravel(), want)


@pytest.mark.parametrize("ycbcr", [0, 1])
@pytest.mark.parametrize("w,h,maxval", [(1, 1, 255), (7, 5, 255), (130, 33, 255), (64, 64, 16383)])
def test_inv_color_exports(glib, olib, ycbcr, w, h, maxval):
    rng = np.random.default_rng(w + 7 * h + ycbcr)
    lo, hi = (-maxval - 10, 2 * maxval) if not ycbcr else (-20, maxval + 20)
    planes = [rng.integers(lo, hi, (h, w), dtype=np.int32) for _ in range(3)]
    want = [p.copy() for p in planes]
    assert olib.fo_kat_inv_color(ycbcr, *[p.ctypes.data_as(C.c_void_p) for p in want], w, h, 0, maxval)
    devs = [Dev(p) for p in planes]
    if ycbcr:
        rc = glib.fuifgpu_inv_ycbcr(devs[0].ptr, devs[1].ptr, devs[2].ptr, w, h, w, w, w, 0, maxval, None)
    else:
        rc = glib.fuifgpu_inv_ycocg(devs[0].ptr, devs[1].ptr, devs[2].ptr, w, h, w, w, w, maxval, None)
    assert rc == 0
    for d, e in zip(devs, want):
        assert np.array_equal(d.get().reshape(h, w), e)


def test_ycocg_known_answers(glib):
    """Appendix E.5: (Y,Co,Cg) -> (R,G,B) at maxval 255"""
    src = np.array([[120, -30, 15], [0, 255, -255], [255, -255, 255], [77, 13, -8]], np.int32)
    want = np.array([[98, 128, 128], [255, 0, 1], [0, 255, 255], [88, 73, 75]], np.int32)
    devs = [Dev(src[:, k].copy()) for k in range(3)]
    assert glib.fuifgpu_inv_ycocg(devs[0].ptr, devs[1].ptr, devs[2].ptr, 4, 1, 4, 4, 4, 255, None) == 0
    got = np.stack([d.get() for d in devs], axis=1)
    assert np.array_equal(got, want)


def _zigzag(olib):
    z = np.zeros(64, np.int32)
    olib.fo_kat_zigzag(z.ctypes.data_as(C.c_void_p))
    return z


@pytest.mark.parametrize("bw,bh,maxval", [(1, 1, 255), (3, 2, 255), (17, 9, 255), (40, 23, 1023)])
def test_idct_export(glib, olib, bw, bh, maxval):
    rng = np.random.default_rng(bw * 64 + bh)
    planes = rng.integers(-60, 60, (64, bh, bw), dtype=np.int32)
    planes[0] = rng.integers(-4 * (maxval + 1), 4 * (maxval + 1), (bh, bw), dtype=np.int32)
    want = np.zeros((bh * 8, bw * 8), np.int32)
    assert olib.fo_kat_inv_dct(planes.ctypes.data_as(C.c_void_p), bw, bh, maxval, want.ctypes.data_as(C.c_void_p))
    z = _zigzag(olib)
    dev = Dev(planes)
    # src64[i] = the plane feeding position i of the 8x8 block = channel z[i] (dct.h:286)
    ptrs = (C.c_void_p * 64)(*[dev.ptr + int(z[i]) * bh * bw * 4 for i in range(64)])
    d_out = Dev(np.zeros((bh * 8, bw * 8), np.int32))
    assert glib.fuifgpu_idct8x8(ptrs, bw, bh, d_out.ptr, maxval, None) == 0
    assert np.array_equal(d_out.get().reshape(bh * 8, bw * 8), want)


def test_idct_known_answer(glib, olib):
    """Appendix E.6: block with b[0]=919 (DC incl. offset), b[1]=24, b[8]=-18, b[9]=7, b[18]=-5, b[63]=2 ->
    row 0 = 117 116 116 114 112 109 106 105, row 7 = 119 120 120 120 118 117 115 114"""
    maxval = 255
    blk = np.zeros(64, np.int32)
    blk[[1, 8, 9, 18, 63]] = [24, -18, 7, -5, 2]
    blk[0] = 919 - (maxval + 1) * 4            # the entry point adds the DC offset (maxval+1)*4 itself (dct.h:281,285)
    dev = Dev(blk)                             # 64 planes of one sample each, in block-position order
    ptrs = (C.c_void_p * 64)(*[dev.ptr + i * 4 for i in range(64)])
    d_out = Dev(np.zeros((8, 8), np.int32))
    assert glib.fuifgpu_idct8x8(ptrs, 1, 1, d_out.ptr, maxval, None) == 0
    got = d_out.get().reshape(8, 8)
    assert got[0].tolist() == [117, 116, 116, 114, 112, 109, 106, 105]
    assert got[7].tolist() == [119, 120, 120, 120, 118, 117, 115, 114]


@pytest.mark.parametrize("n,q", [(1, 7), (1000, 3), (257 * 33, 16), (5, 1)])
def test_inv_quantize_export(glib, n, q):
    """transform/quantize.h:32-49: every sample of the plane times Channel::q (the reference's loop is a plain multiply)"""
    rng = np.random.default_rng(n + q)
    a = rng.integers(-2000, 2000, n).astype(np.int32)
    d = Dev(a)
    assert glib.fuifgpu_inv_quantize(d.ptr, n, q, None) == 0
    _sync()
    assert np.array_equal(d.get(), a * q)


@pytest.mark.parametrize("w,h,srh,srv", [(1, 1, 2, 2), (5, 3, 2, 2), (64, 17, 2, 1), (33, 40, 1, 2), (240, 135, 2, 2)])
def test_upsample_export(glib, olib, w, h, srh, srv):
    rng = np.random.default_rng(w * 3 + h + srh)
    src = rng.integers(-50, 300, (h, w), dtype=np.int32)
    want = np.zeros((h * srv, w * srh), np.int32)
    assert olib.fo_kat_upsample(src.ctypes.data_as(C.c_void_p), w, h, srh, srv, want.ctypes.data_as(C.c_void_p))
    d_in, d_out = Dev(src), Dev(np.zeros((h * srv, w * srh), np.int32))
    assert glib.fuifgpu_upsample(d_in.ptr, w, h, srh, srv, d_out.ptr, None) == 0
    assert np.array_equal(d_out.get().reshape(h * srv, w * srh), want)


@pytest.mark.parametrize("w,h,colours", [(1, 1, 1), (7, 5, 3), (64, 33, 200), (33, 2, 0), (300, 7, 19)])
def test_inv_palette_export(glib, w, h, colours):
    """transform/palette.h:57-64, one component: out = palette_row[CLAMP(index, 0, colours-1)]; an empty palette reads Channel::zero"""
    rng = np.random.default_rng(w * 7 + h + colours)
    idx = rng.integers(-3, colours + 4, (h, w), dtype=np.int32)       # indices on both sides of the palette are clamped
    row = rng.integers(-500, 1500, max(colours, 1), dtype=np.int32)
    want = row[np.clip(idx, 0, max(colours - 1, 0))] if colours else np.zeros((h, w), np.int32)
    d_idx, d_row, d_out = Dev(idx), Dev(row), Dev(np.full((h, w), -7, np.int32))
    assert glib.fuifgpu_inv_palette(d_idx.ptr, w, h, d_row.ptr, colours, d_out.ptr, None) == 0
    assert np.array_equal(d_out.get().reshape(h, w), want)
    assert glib.fuifgpu_inv_palette(d_idx.ptr, w, h, d_row.ptr, colours, d_idx.ptr, None) != 0    # in place is refused


@pytest.mark.parametrize("n,q,have", [(1, 2, True), (1000, 4, True), (257 * 33, 10, False), (5, 1, True)])
def test_inv_approximate_export(glib, n, q, have):
    """transform/approximate.h:44-57: value * q + remainder; a remainder channel that is not available adds nothing (:49,54)"""
    rng = np.random.default_rng(n + q)
    a = rng.integers(-700, 700, n).astype(np.int32)
    r = rng.integers(0, q, n).astype(np.int32)
    d, dr = Dev(a), Dev(r)
    assert glib.fuifgpu_inv_approximate(d.ptr, dr.ptr if have else None, n, q, None) == 0
    _sync()
    assert np.array_equal(d.get(), a * q + (r if have else 0))


def _match_case(rng, w, h, n_planes, maxz, density):
    z = rng.integers(1, maxz + 1, (h, w), dtype=np.int32)
    z[rng.random((h, w)) >= density] = 0
    planes = rng.integers(-40, 300, (n_planes, h, w), dtype=np.int32)
    return z, planes


@pytest.mark.parametrize("soft", [0, 1])
@pytest.mark.parametrize("w,h,n_planes,maxz,density", [(40, 30, 3, 24, 0.5), (64, 64, 1, 60, 0.9), (97, 13, 4, 12, 0.2), (33, 50, 2, 1, 1.0)])
def test_inv_match_free_offsets_export(glib, olib, w, h, n_planes, maxz, density, soft):
    """transform/2dmatch.h:136-146 (match channel q == 1): every matched sample copies (soft: adds) an EARLIER sample, which may
    itself be matched -- chains up to the whole row long at density 1.0 -- incl. sources before the first sample (Channel::zero)"""
    rng = np.random.default_rng(w * 31 + h + maxz + soft)
    z, planes = _match_case(rng, w, h, n_planes, maxz, density)
    want = planes.copy()
    assert olib.fo_kat_inv_match(z.ctypes.data_as(C.c_void_p), w, h, want.ctypes.data_as(C.c_void_p), n_planes, soft, 1, maxz, 1)
    d_z, d_p = Dev(z), Dev(planes)
    ptrs = (C.c_void_p * n_planes)(*[d_p.ptr + k * w * h * 4 for k in range(n_planes)])
    assert glib.fuifgpu_inv_match(d_z.ptr, w, h, ptrs, n_planes, soft, 1, maxz, 1, None) == 0
    assert np.array_equal(d_p.get().reshape(n_planes, h, w), want)


@pytest.mark.parametrize("soft", [0, 1])
@pytest.mark.parametrize("w,fh,frames,n_planes", [(24, 10, 4, 3), (70, 7, 3, 1), (5, 3, 6, 2)])
def test_inv_match_previous_frames_export(glib, olib, w, fh, frames, n_planes, soft):
    """transform/2dmatch.h:147-171 (match channel q == 2*fh*fh + (fh&1)): z frames up in the vertical film strip; a source above the
    first frame reads Channel::zero"""
    rng = np.random.default_rng(w + fh * 100 + soft)
    h = fh * frames
    z, planes = _match_case(rng, w, h, n_planes, frames, 0.6)
    q = 2 * fh * fh + (fh & 1)
    want = planes.copy()
    assert olib.fo_kat_inv_match(z.ctypes.data_as(C.c_void_p), w, h, want.ctypes.data_as(C.c_void_p), n_planes, soft, q, 1, frames)
    d_z, d_p = Dev(z), Dev(planes)
    ptrs = (C.c_void_p * n_planes)(*[d_p.ptr + k * w * h * 4 for k in range(n_planes)])
    assert glib.fuifgpu_inv_match(d_z.ptr, w, h, ptrs, n_planes, soft, q, 1, frames, None) == 0
    assert np.array_equal(d_p.get().reshape(n_planes, h, w), want)


def test_inv_match_export_refusals(glib):
    """a match channel whose q names neither mode is the reference's `return false` (2dmatch.h:172-175); a forward reference (an image
    narrower than the offset spiral) and an offset code beyond the table are refused with the planes untouched"""
    w, h = 3, 8
    planes = np.arange(w * h, dtype=np.int32).reshape(1, h, w)
    z = np.zeros((h, w), np.int32)
    z[4, 0] = 4             # layer 0, code 4: (x+1, y-1), an earlier sample
    d_z, d_p = Dev(z), Dev(planes)
    ptrs = (C.c_void_p * 1)(d_p.ptr)
    assert glib.fuifgpu_inv_match(d_z.ptr, w, h, ptrs, 1, 0, 5, 100, 1, None) == 2          # FUIFGPU_E_CORRUPT: q = 5 is no mode
    z2 = z.copy(); z2[4, 0] = 200                                                        # beyond maxval = the offsets table
    assert glib.fuifgpu_inv_match(Dev(z2).ptr, w, h, ptrs, 1, 0, 1, 100, 1, None) == 3      # FUIFGPU_E_UNSUPPORTED
    # a forward reference needs yo*w + xo > 0: layer 5 (odd), code 1: (x+6, y-1) in a 3-wide image = +3
    code = 4 + 8 + 12 + 16 + 20 + 1
    z4 = np.zeros((h, w), np.int32); z4[2, 0] = code
    assert glib.fuifgpu_inv_match(Dev(z4).ptr, w, h, ptrs, 1, 0, 1, 100, 1, None) == 3
    assert np.array_equal(d_p.get().reshape(1, h, w), planes)


# ---- forward transforms (the writer's GPU path) -----------------------------------------------------
# The inverse of Squeeze is a function of (avg, residual) and a given plane has exactly one pre-image, so
# "reference inverse (oracle restatement) of the GPU's forward == the plane" pins the forward kernels bit for bit.
@pytest.mark.parametrize("w,h", [(1, 1), (2, 1), (1, 2), (3, 3), (8, 5), (33, 17), (64, 64), (255, 2), (2, 255), (640, 37)])
@pytest.mark.parametrize("horizontal", [True, False])
def test_fwd_squeeze_exports_are_the_inverse_of_the_reference_inverse(glib, olib, w, h, horizontal):
    rng = np.random.default_rng(w * 1000 + h * 2 + int(horizontal))
    plane = rng.integers(-300, 1024, size=(h, w)).astype(np.int32)
    plane[: h // 2] = rng.integers(0, 4, size=(h // 2, w))       # flat areas: the tendency term is exercised on both signs
    aw, ah = ((w + 1) // 2, h) if horizontal else (w, (h + 1) // 2)
    rw, rh = (w - aw, h) if horizontal else (w, h - ah)
    d_in, d_avg, d_res = Dev(plane), Dev(np.zeros((ah, aw), np.int32)), Dev(np.zeros((max(rh, 1), max(rw, 1)), np.int32))
    fn = glib.fuifgpu_fwd_hsqueeze if horizontal else glib.fuifgpu_fwd_vsqueeze
    assert fn(d_in.ptr, w, h, d_avg.ptr, d_res.ptr, None) == 0
    _sync()
    avg = d_avg.get()
    res = d_res.get()[:rh, :rw] if rw * rh else np.zeros((rh, rw), np.int32)
    res = np.ascontiguousarray(res.reshape(rh, rw))
    if rw * rh == 0:
        assert np.array_equal(avg, plane)      # nothing to pair: the plane is its own average
        return
    back = ref_squeeze(olib, horizontal, avg, res)
    assert np.array_equal(back, plane)


@pytest.mark.parametrize("w,h,maxval", [(1, 1, 255), (7, 5, 255), (130, 33, 255), (64, 64, 16383)])
def test_fwd_ycocg_export_round_trip(glib, w, h, maxval):
    rng = np.random.default_rng(w + h + maxval)
    rgb = [rng.integers(0, maxval + 1, size=(h, w)).astype(np.int32) for _ in range(3)]
    d = [Dev(p) for p in rgb]
    assert glib.fuifgpu_fwd_ycocg(d[0].ptr, d[1].ptr, d[2].ptr, w, h, None) == 0
    _sync()
    y, co, cg = [x.get() for x in d]
    # transform/ycocg.h:65-95 restated
    assert np.array_equal(y, (((rgb[0] + rgb[2]) >> 1) + rgb[1]) >> 1)
    assert np.array_equal(co, rgb[0] - rgb[2])
    assert np.array_equal(cg, rgb[1] - ((rgb[0] + rgb[2]) >> 1))
    assert glib.fuifgpu_inv_ycocg(d[0].ptr, d[1].ptr, d[2].ptr, w, h, w, w, w, maxval, None) == 0
    _sync()
    for k in range(3):
        assert np.array_equal(d[k].get(), rgb[k])


@pytest.mark.parametrize("w,h,c,bits", [(97, 61, 3, 8), (64, 48, 1, 8), (80, 72, 4, 14), (200, 9, 3, 8), (321, 255, 3, 10)])
def test_writer_with_gpu_forward_transforms_writes_the_same_bytes(gpulib, w, h, c, bits):
    """fuifgpu_encode_image with gpu_forward = 1 (YCoCg + every Squeeze step on the GPU) against the host forward path,
    which tests/test_writer.py pins byte for byte to what the reference CLI writes"""
    from fuif_amd.synth import photographic
    img = photographic(w, h, c, bits, seed=w + h)
    for tree_mode in (0, 1):
        host = gpulib.encode_image(img, bits, tree_mode=tree_mode, index=True)
        dev = gpulib.encode_image(img, bits, tree_mode=tree_mode, index=True, gpu_forward=True)
        assert dev == host


@pytest.mark.parametrize("elems,stride,n", [(64 * 3 + 2, 196, 3), (131075, 131076, 5), (4, 8, 1), (0, 4, 2), (65521 * 2 + 7, 65521 * 2 + 10, 2)])
def test_plane_checksums_export(gpulib, glib, elems, stride, n):
    """fuifgpu_plane_checksums (verification aid of the overlapped-steps bench): sum of sample * (index mod 65521 + 1) per image as a
    wrapping 64-bit integer, against numpy -- tails that are no multiple of 4, the weight's wrap at 65521, negative samples, gaps
    between the images"""
    rng = np.random.default_rng(elems + n)
    slab = rng.integers(-(1 << 31), 1 << 31, size=(n, stride), dtype=np.int64).astype(np.int32)
    d = Dev(slab)
    sums = Dev(np.full(2 * n, -1, np.int32))            # n x uint64, pre-filled with garbage: the call zeroes it on the stream
    gpulib.plane_checksums(d.ptr, elems, n, sums.ptr, None, image_stride=stride)
    got = sums.get().view(np.uint64)
    w = (np.arange(elems, dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
    with np.errstate(over="ignore"):
        want = np.array([(slab[k, :elems].astype(np.int64).view(np.uint64) * w).sum(dtype=np.uint64) for k in range(n)], np.uint64)
    assert np.array_equal(got, want)


def test_device_selection_and_peer_copy(gpulib, glib, manifest):
    """fuifgpu_device_count / set_device / get_device / batch_device / peer_copy (include/fuifgpu.h, round 5): a batch lives on the
    device that was current at its creation and its calls run there; a device that does not exist is refused; a peer copy moves the
    bytes (one GPU per box: device 0 to device 0, the same-device branch; two devices on the emulated node of the CPU suite)"""
    from conftest import golden_blob
    n = gpulib.device_count()
    assert n >= 1
    with pytest.raises(gpulib.FuifGpuError):
        gpulib.set_device(n)
    with pytest.raises(gpulib.FuifGpuError):
        gpulib.set_device(-1)
    last = n - 1
    gpulib.set_device(last)
    assert gpulib.get_device() == last
    e = next(x for x in manifest["fixtures"] if x["name"] == "rgb8_97x61")
    blob = golden_blob(e, e["cases"][0])
    plan = gpulib.Plan(blob)
    batch = gpulib.Batch(plan, 1, len(blob))
    try:
        assert batch.device == last
        gpulib.set_device(0)                      # the batch keeps decoding on ITS device whatever the caller's current device is
        batch.upload([blob]); batch.decode(); batch.undo_transforms(); batch.sync()
        assert gpulib.get_device() == 0           # ... and the caller's device is what it was
        st, _ = batch.status()
        assert not st.any()
        from conftest import plane_hash
        assert [plane_hash(p) for p in batch.out_planes(0)] == [c["sha256"] for c in e["cases"][0]["post"]]
    finally:
        batch.close()
    src = Dev(np.arange(1000, dtype=np.int32))
    gpulib.set_device(last)
    dst = Dev(np.zeros(1000, np.int32))
    gpulib.set_device(0)
    assert glib.fuifgpu_peer_copy(dst.ptr, last, src.ptr, 0, 4000, None) == 0
    if not EMULATED:
        import torch
        for d in range(n):
            torch.cuda.synchronize(d)
    assert np.array_equal(dst.get(), np.arange(1000, dtype=np.int32))
    assert glib.fuifgpu_peer_copy(dst.ptr, n, src.ptr, 0, 4000, None) != 0
