#!/usr/bin/env python3
"""Differential fuzz of the kernels (on the CPU wavefront emulator) against the oracle, over streams written by
the REAL reference CLI with random encoder options -- build container only (needs oracle/_ref and
tests/_emu/libfuifgpu_emu.so, see tests/test_emulated_kernels.py).

  python tools/emu_fuzz.py [n_cases] [seed]

Every case: random small image (photographic / posterised / screen content; 1, 3 or 4 channels; 8 or 12-14 bit),
random flags from {-P predictors, -E max properties, -G group size, -U, -R 0, -Q quality, -C colourspace, -K/-X/-Y
palettes, -A approximate, -J DCT} -- plus, through the reference's library calls, 2D matches with explicit parameters (soft ones included);
full decode, a random preview and a random truncation; coefficient planes,
final planes and status must equal the oracle's.  Streams the planner does not take (FUIFGPU_E_UNSUPPORTED) are counted."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("FUIF_AMD_LIB", os.path.join(ROOT, "tests", "_emu", "libfuifgpu_emu.so"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fuif_amd  # noqa: E402
from fuif_amd.synth import graphic, photographic, write_pnm  # noqa: E402
from oracle_py import Port, run_ref_cli  # noqa: E402


def random_jpeg_case(rng, tmp):
    """JPEG input: the CLI transcodes the DCT coefficients (YCbCr / Subsample / DCT / Quantize [/ Squeeze of DC])"""
    from PIL import Image
    ch = int(rng.choice([1, 3, 3]))
    w, h = int(rng.integers(9, 120)), int(rng.integers(9, 100))
    img = photographic(w, h, ch, 8, seed=int(rng.integers(1 << 30)), sigma=float(rng.choice([0.0, 1.0, 3.0])))
    arr = np.moveaxis(img, 0, -1).astype(np.uint8)
    src = os.path.join(tmp, "in.jpg")
    kw = dict(quality=int(rng.integers(25, 97)))
    if ch == 3:
        kw["subsampling"] = int(rng.choice([0, 1, 2]))
    Image.fromarray(arr[..., 0] if ch == 1 else arr).save(src, **kw)
    flags = []
    if rng.random() < 0.3:
        flags += ["-R", "0"]
    if rng.random() < 0.3:
        flags += ["-E", str(int(rng.choice([0, 4, 12])))]
    if rng.random() < 0.3:
        flags += ["-G", str(int(rng.integers(1, 40)))]
    flags += ["-I", str(rng.choice(["0", "0.5"]))]
    out = os.path.join(tmp, "out.fuif")
    if os.path.exists(out):
        os.remove(out)
    r = run_ref_cli(flags + [src, out])
    if r.returncode != 0 or not os.path.exists(out):
        return ["jpeg", str(kw)] + flags, None
    return ["jpeg", str(kw)] + flags, open(out, "rb").read()


def random_animation_case(rng, tmp):
    """a few stacked frames; by default the CLI adds the 2D match against previous frames (fuif.cpp:440)"""
    w, h, n = int(rng.integers(9, 50)), int(rng.integers(9, 40)), int(rng.integers(2, 5))
    base = photographic(w, h, 3, 8, seed=int(rng.integers(1 << 30)))
    for i in range(n):
        fr = photographic(w, h, 3, 8, seed=int(rng.integers(1 << 30)))
        if rng.random() < 0.7:
            keep = rng.random((h, w)) < 0.8
            fr = np.where(keep[None], base, fr)
        write_pnm(os.path.join(tmp, "fr-%02d.ppm" % i), fr, 255)
    flags = [] if rng.random() < 0.7 else ["-M", "0"]
    if rng.random() < 0.3:
        flags += ["-R", "0"]
    flags += ["-I", "0"]
    out = os.path.join(tmp, "out.fuif")
    if os.path.exists(out):
        os.remove(out)
    r = run_ref_cli(flags + [os.path.join(tmp, "fr-%02d.ppm"), out])
    for i in range(n):
        os.remove(os.path.join(tmp, "fr-%02d.ppm" % i))
    if r.returncode != 0 or not os.path.exists(out):
        return ["anim%d" % n] + flags, None
    return ["anim%d" % n] + flags, open(out, "rb").read()


_REF = []


def random_library_match_case(rng):
    """2D matches with explicit parameters -- SOFT ones included, which the CLI never writes (fuif.cpp:445) -- through the reference's
    library calls (oracle/ref_driver.cpp fuifref_encode): stills with free offsets and film strips matched against previous frames,
    lossless or with a quantization constant behind the match (then a soft match adds up non-zero differences)"""
    from oracle_py import Ref
    if not _REF:
        _REF.append(Ref())
    ch = int(rng.choice([1, 3, 4]))
    kw = dict(softmatch=int(rng.integers(0, 2)), quant=int(rng.choice([0, 0, 2, 5])), squeeze=int(rng.integers(0, 2)), nb_repeats=float(rng.choice([0.0, 0.5])))
    if rng.random() < 0.3:
        w, fh, frames = int(rng.integers(12, 50)), int(rng.integers(8, 30)), int(rng.integers(2, 5))
        base = photographic(w, fh, ch, 8, seed=int(rng.integers(1 << 30)))
        strip = [np.where((rng.random((fh, w)) < 0.8)[None], base, photographic(w, fh, ch, 8, seed=int(rng.integers(1 << 30)))) for _ in range(frames)]
        img = np.concatenate(strip, axis=1)
        kw.update(frames=frames, match_distance=-int(rng.integers(1, frames)))
    else:
        img = graphic(int(rng.integers(24, 120)), int(rng.integers(24, 100)), ch, 8, seed=int(rng.integers(1 << 30)), colors=int(rng.integers(50, 500)))
        kw.update(match_distance=int(rng.choice([20, 60, 300])))
    try:
        return ["library-match", str(kw)], _REF[0].encode(img, maxval=255, **kw)
    except RuntimeError:
        return ["library-match", str(kw)], None


def random_case(rng, tmp):
    pick = rng.random()
    if pick < 0.2:
        return random_jpeg_case(rng, tmp)
    if pick < 0.3:
        return random_animation_case(rng, tmp)
    if pick < 0.38:
        return random_library_match_case(rng)
    ch = int(rng.choice([1, 3, 3, 3, 4]))
    bits = int(rng.choice([8, 8, 8, 12, 14]))
    w, h = int(rng.integers(9, 90)), int(rng.integers(9, 80))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        img = graphic(w, h, ch, bits, seed=int(rng.integers(1 << 30)), colors=int(rng.integers(2, 60)), step=int(rng.choice([1, 1, 3, 8])))
    else:
        img = photographic(w, h, ch, bits, seed=int(rng.integers(1 << 30)), sigma=float(rng.choice([0.0, 1.0, 3.0])) if bits == 8 else None)
        if kind == 1:
            img = img // 8 * 8
    flags = []
    if rng.random() < 0.5:
        flags += ["-P", "".join(str(int(rng.integers(0, 7))) for _ in range(int(rng.integers(1, 4))))]
    if rng.random() < 0.5:
        flags += ["-E", str(int(rng.choice([0, 2, 4, 6, 12, 16, 24, 40, 50])))]
    if rng.random() < 0.3:
        flags += ["-G", str(int(rng.integers(1, 6)))]
    if rng.random() < 0.1:
        flags += ["-U"]
    if rng.random() < 0.25:
        flags += ["-R", "0"]
    if rng.random() < 0.3:
        flags += ["-Q", str(int(rng.integers(30, 100)))]
    if ch >= 3 and rng.random() < 0.3:
        flags += ["-C", str(int(rng.choice([0, 1, 2])))]
    if rng.random() < 0.3:
        flags += ["-K", str(int(rng.choice([0, 16, 256, 1024])))]
    if rng.random() < 0.3:
        flags += ["-X", str(int(rng.choice([0, 30, 90]))), "-Y", str(int(rng.choice([0, 30, 90])))]
    if rng.random() < 0.15:
        flags += ["-A", "%d,%d" % (int(rng.integers(1, 4)), int(rng.integers(1, 6)))]
    if bits == 8 and ch == 3 and rng.random() < 0.1:
        flags += ["-J"]
    if rng.random() < 0.2:
        flags += ["-M", str(int(rng.choice([3, 20, 100, 1000])))]
    flags += ["-I", str(rng.choice(["0", "0.5", "1"]))]
    src = os.path.join(tmp, "in." + ("pam" if ch in (2, 4) else "ppm" if ch == 3 else "pgm"))
    out = os.path.join(tmp, "out.fuif")
    write_pnm(src, img, (1 << bits) - 1)
    if os.path.exists(out):
        os.remove(out)
    r = run_ref_cli(flags + [src, out])
    if r.returncode != 0 or not os.path.exists(out):
        return flags, None
    return flags, open(out, "rb").read()


def compare(port, blob, preview, with_index=False):
    plan = fuif_amd.Plan(blob)
    batch = fuif_amd.Batch(plan, 1, len(blob) + 4096)
    try:
        batch.upload([blob], preview)
        batch.decode()
        batch.sync()
        st, used = batch.status()
        pre = batch.coef_planes(0)
        groups = batch.group_index(0)
        batch.undo_transforms()
        batch.sync()
        st_undone = int(batch.status()[0][0])
        post = batch.out_planes(0)
        if with_index and len(groups) > 1:
            # the same stream, one tile per channel group: must give the same planes
            indexed = fuif_amd.index_append(blob, groups)
            batch.upload([indexed], preview)
            batch.decode()
            batch.sync()
            st2, _ = batch.status()
            pre2 = batch.coef_planes(0)
            batch.undo_transforms()
            batch.sync()
            post2 = batch.out_planes(0)
            if st2[0] != st[0] or any(not np.array_equal(x, y) for x, y in zip(pre, pre2)) or any(not np.array_equal(x, y) for x, y in zip(post, post2)):
                return "group-parallel decode differs from the per-image decode"
    finally:
        batch.close()
    a, b = port.decode_both(blob, preview=preview)
    if not a.ok:
        return "oracle refused"
    if st[0] & 2:
        return "kernel says corrupt"
    for i, (g, c) in enumerate(zip(pre, a.channels)):
        if c["size"] == c["w"] * c["h"] and not np.array_equal(g, c["data"]):
            return "coefficient plane %d differs" % i
    if b.info.get("error"):
        # the reference's undo_transforms would fail or leave defined behaviour here (e.g. `-A` over a match channel: its maxval is not restored, so
        # 2dmatch.h:126-127 indexes its offsets table out of range): the inverse kernels must have flagged the image as well; planes are not compared
        return None if st_undone & 6 else "the oracle's undo_transforms fails, the kernels flag nothing"
    if len(post) != len(b.channels):
        return "channel count after undo differs"
    for i, (g, c) in enumerate(zip(post, b.channels)):
        if not np.array_equal(g, c["data"]):
            return "final plane %d differs" % i
    return None


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    port = Port()
    tmp = os.environ.get("EMU_FUZZ_TMP") or tempfile.mkdtemp()
    os.makedirs(tmp, exist_ok=True)
    done = unsupported = failed_encode = 0
    bad = []
    for k in range(n):
        flags, blob = random_case(rng, tmp)
        if blob is None or len(blob) == 0:     # (the CLI occasionally exits 0 leaving an empty file behind)
            failed_encode += 1
            continue
        try:
            cases = [(blob, -1), (blob, int(rng.integers(0, 5))), (blob[: int(len(blob) * rng.uniform(0.15, 0.95))], -1)]
            for bl, pv in cases:
                with open(os.path.join(tmp, "current.fuif"), "wb") as f:   # a crash inside the library leaves the culprit behind
                    f.write(bl)
                with open(os.path.join(tmp, "current.txt"), "w") as f:
                    f.write("case %d flags %s preview %d bytes %d\n" % (k, flags, pv, len(bl)))
                err = compare(port, bl, pv, with_index=len(bl) == len(blob))
                if err:
                    bad.append((k, flags, len(bl), pv, err))
                    keep = os.path.join(ROOT, "gpurun_out", "emu_fuzz_case_%d_%d.fuif" % (seed, k))
                    os.makedirs(os.path.dirname(keep), exist_ok=True)
                    open(keep, "wb").write(bl)
                    print("MISMATCH case %d flags %s bytes %d preview %d: %s -> %s" % (k, flags, len(bl), pv, err, keep), flush=True)
            done += 1
        except fuif_amd.FuifGpuError as e:
            if e.code == 3:
                unsupported += 1
            elif e.code in (1, 2) and len(bl) < len(blob) and (len(bl) < 8 or not port.decode(bl, undo=False).ok):
                done += 1          # a cut inside the header: the oracle refuses it as well
            else:
                bad.append((k, flags, len(blob), -1, str(e)))
                print("ERROR case %d flags %s: %s" % (k, flags, e), flush=True)
        if (k + 1) % 20 == 0:
            print("%d cases: %d compared, %d unsupported, %d encoder refusals, %d mismatches" % (k + 1, done, unsupported, failed_encode, len(bad)), flush=True)
    print("done: %d compared, %d unsupported, %d encoder refusals, %d mismatches" % (done, unsupported, failed_encode, len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
