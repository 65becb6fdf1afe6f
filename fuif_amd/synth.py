"""Seeded synthetic "photographic" images.

Per channel: sum of 6 sinusoids A*sin(2*pi*fx*x/w+p1)*cos(2*pi*fy*y/h+p2), A~U(10,40), f~U(0.5,8),
+128 + N(0, sigma), clipped to [0,255]; the 14-bit variant uses A~U(300,1500), +8192 + N(0,40),
clip [0,16383].  numpy.random.default_rng(seed) makes the image a pure function of
(seed, w, h, channels, bits, sigma) so the same pixels exist here and on the GPU box.
"""
import numpy as np


def photographic(w, h, channels=3, bits=8, seed=1, sigma=None):
    rng = np.random.default_rng(seed)
    if bits <= 8:
        amp, mid, sig, maxval = (10.0, 40.0), 128.0, 3.0, 255
    else:
        amp, mid, sig, maxval = (300.0, 1500.0), 8192.0, 40.0, (1 << bits) - 1
        scale = maxval / 16383.0
        amp, mid, sig = (amp[0] * scale, amp[1] * scale), mid * scale, sig * scale
    if sigma is not None:
        sig = sigma
    x = np.arange(w, dtype=np.float64)[None, :] / w
    y = np.arange(h, dtype=np.float64)[:, None] / h
    out = np.empty((channels, h, w), dtype=np.int32)
    for c in range(channels):
        acc = np.full((h, w), mid, dtype=np.float64)
        for _ in range(6):
            a = rng.uniform(*amp)
            fx, fy = rng.uniform(0.5, 8.0, size=2)
            p1, p2 = rng.uniform(0, 2 * np.pi, size=2)
            acc += a * np.sin(2 * np.pi * fx * x + p1) * np.cos(2 * np.pi * fy * y + p2)
        if sig > 0:
            acc += rng.normal(0.0, sig, size=(h, w))
        out[c] = np.clip(np.rint(acc), 0, maxval).astype(np.int32)
    return out


def graphic(w, h, channels=3, bits=8, seed=1, colors=32, step=1):
    """Seeded "screen content": a few dozen flat colours in overlapping rectangles and diagonal bands
    (what makes the reference CLI choose its Palette transform, transform/palette.h).  colors = size of the
    colour set; step > 1 additionally restricts every sample to multiples of `step` (sparse channel histograms:
    the per-channel palette heuristic of fuif.cpp:413-427)."""
    rng = np.random.default_rng(seed)
    maxval = (1 << bits) - 1
    table = rng.integers(0, maxval // step + 1, size=(colors, channels)) * step
    idx = np.zeros((h, w), dtype=np.int64)
    for _ in range(3 * colors):
        x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
        x1, y1 = min(w, x0 + int(rng.integers(2, max(3, w // 2)))), min(h, y0 + int(rng.integers(2, max(3, h // 2))))
        idx[y0:y1, x0:x1] = int(rng.integers(0, colors))
    yy, xx = np.mgrid[0:h, 0:w]
    band = ((xx + 2 * yy) // 7) % 11 == 0
    idx[band] = (idx[band] + 1) % colors
    out = np.empty((channels, h, w), dtype=np.int32)
    for c in range(channels):
        out[c] = table[idx, c]
    return out


def write_pnm(path, planes, maxval=255):
    """planes: (C,H,W) int32 -> P5/P6/P7 file readable by the reference's import/read_pam.h."""
    c, h, w = planes.shape
    dt = np.uint8 if maxval < 256 else np.dtype(">u2")
    inter = np.ascontiguousarray(np.moveaxis(planes, 0, -1)).astype(dt)
    with open(path, "wb") as f:
        if c == 1:
            f.write(b"P5\n%d %d\n%d\n" % (w, h, maxval))
        elif c == 3:
            f.write(b"P6\n%d %d\n%d\n" % (w, h, maxval))
        else:
            tt = {2: b"GRAYSCALE_ALPHA", 4: b"RGB_ALPHA"}[c]
            f.write(b"P7\nWIDTH %d\nHEIGHT %d\nDEPTH %d\nMAXVAL %d\nTUPLTYPE %s\nENDHDR\n" % (w, h, c, maxval, tt))
        f.write(inter.tobytes())
