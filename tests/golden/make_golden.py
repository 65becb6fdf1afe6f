#!/usr/bin/env python3
"""Generate the committed golden fixtures with the REAL reference (run in the build container only).

  inputs  : seeded synthetic images (fuif_amd/synth.py) written as PNM/PAM (or JPEG via Pillow)
  encoder : the unmodified reference CLI  oracle/_ref/fuif   (built by oracle/Makefile from /root/reference)
  expected: per-plane SHA-256 (int32 little-endian bytes) + geometry of every channel BEFORE and AFTER
            Image::undo_transforms(), produced by the real reference decoder (oracle/_ref/libfuifref.so,
            FileIO semantics = what `fuif -d` uses), for full decodes, responsive previews (-R k) and
            byte-truncated files.

Nothing here is reference source; the .fuif files and hashes are data.  Re-run: python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from fuif_amd.synth import graphic, photographic, write_pnm  # noqa: E402
from oracle_py import Ref, run_ref_cli  # noqa: E402


def plane_hash(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<i4").tobytes()).hexdigest()


def describe(dec):
    out = []
    for c in dec.channels:
        m = {k: c[k] for k in ("w", "h", "minval", "maxval", "q", "hshift", "vshift", "hcshift", "vcshift", "component", "size")}
        m["sha256"] = plane_hash(c["data"])
        out.append(m)
    return out


# name, generator args, CLI flags
SPECS = [
    ("c1_rgb8_512x512", dict(w=512, h=512, channels=3, bits=8, seed=1), []),
    ("rgb8_97x61", dict(w=97, h=61, channels=3, bits=8, seed=2), []),
    ("gray8_64x48", dict(w=64, h=48, channels=1, bits=8, seed=3), []),
    ("rgba14_80x72", dict(w=80, h=72, channels=4, bits=14, seed=4), ["-K", "0", "-X", "0", "-Y", "0"]),
    ("raw14x4_64x64_squeezeonly", dict(w=64, h=64, channels=4, bits=14, seed=5), ["-C", "0", "-K", "0", "-X", "0", "-Y", "0"]),
    ("rgb8_128x128_I0", dict(w=128, h=128, channels=3, bits=8, seed=6), ["-I", "0"]),
    ("rgb8_128x128_E0", dict(w=128, h=128, channels=3, bits=8, seed=6), ["-E", "0"]),
    # 18 reference properties = 9 earlier channels in every context tree: the most the GPU path's 32 property lanes hold (2*9+13 = 31)
    ("rgb8_112x96_E18", dict(w=112, h=96, channels=3, bits=8, seed=61), ["-E", "18"]),
    # round 4: more than 31 properties -- 65-word property rows, half-length chunks, references beyond the unrolled nine (2*25 + 13 = 63 of 64 lanes at -E 50)
    ("rgb8_112x96_E24", dict(w=112, h=96, channels=3, bits=8, seed=62), ["-E", "24"]),
    ("rgb8_120x88_E50", dict(w=120, h=88, channels=3, bits=8, seed=63), ["-E", "50"]),
    ("rgb8_64x64_U", dict(w=64, h=64, channels=3, bits=8, seed=7), ["-U"]),
    ("rgb8_160x120_Q80", dict(w=160, h=120, channels=3, bits=8, seed=8), ["-Q", "80"]),
    ("rgb8_96x96_nosqueeze", dict(w=96, h=96, channels=3, bits=8, seed=9), ["-R", "0"]),
    ("rgb8_tall_40x200", dict(w=40, h=200, channels=3, bits=8, seed=10), []),
    ("rgb8_smooth_256x256", dict(w=256, h=256, channels=3, bits=8, seed=11, sigma=0.0), []),
    # (the reference ENCODER is not deterministic on this one: its bytes differ from run to run -- every version is a valid
    # stream that decodes to the same planes; the committed file is one of them and the manifest pins what IT decodes to)
    # no Squeeze: the planes are the Image constructor's (already hold w*h zeros), and the channel range excludes 0 --
    # the rows a truncated stream never reaches stay 0, not Channel::zero (encoding.cpp:368, image.h:64-65,73-75)
    ("gray8_nosqueeze_60x40", dict(w=60, h=40, channels=1, bits=8, seed=14), ["-R", "0"]),
    # round 5: context trees far beyond anything the product's own writer makes (its cap is 4095 nodes): -I 16 = sixteen tree-learning
    # passes (fuif.cpp:110, encoding.cpp:455-573) give the largest group a tree of 19 327 nodes (compound.h:277-320 allows 65 535), i.e. more
    # supernodes than a wavefront's scratch area holds (the node-by-node walk behind kSlowFlag) and 9 664 leaves
    ("rgb8_512x384_I16_bigtrees", dict(w=512, h=384, channels=3, bits=8, seed=78), ["-I", "16"]),
]
# screen content / sparse histograms with DEFAULT CLI flags: the reference picks Palette (transform/palette.h)
# by itself (fuif.cpp:399-427); -A k,q adds Approximate (transform/approximate.h) on the last k channels
GRAPHIC_SPECS = [
    ("pal_rgb_graphic_120x90", dict(w=120, h=90, channels=3, bits=8, seed=31, colors=40), []),
    ("pal_rgba_graphic_72x64", dict(w=72, h=64, channels=4, bits=8, seed=32, colors=20), []),
    ("pal_gray_sparse_100x70", dict(w=100, h=70, channels=1, bits=8, seed=33, colors=24, step=5), []),
    ("pal_rgb_sparse_128x96", dict(w=128, h=96, channels=3, bits=8, seed=34, colors=700, step=8), []),
    # -M n on a still: 2D match with free offsets into the already decoded neighbourhood (2dmatch.h:136-146), after a palette
    ("match_rgb_graphic_96x80", dict(w=96, h=80, channels=3, bits=8, seed=51, colors=400), ["-M", "40"]),
    ("match_rgb_graphic_nosqueeze_72x60", dict(w=72, h=60, channels=3, bits=8, seed=53, colors=300), ["-M", "200", "-K", "0", "-R", "0"]),
    ("pal_rgb_graphic_nosqueeze_64x48", dict(w=64, h=48, channels=3, bits=8, seed=35, colors=12), ["-R", "0"]),
    ("pal_rgb_channelwise_96x72", dict(w=96, h=72, channels=3, bits=8, seed=37, poster=8), []),
    # Quantize + Approximate where an approximated channel becomes all zeroes (no q in the stream) while its remainder is
    # not: Channel::q has to come back from the remainder before the Quantize inverse (approximate.h:49-50)
    ("approx_quant_rgb8_40x30", dict(w=40, h=30, channels=3, bits=8, seed=202, photographic=True), ["-Q", "65", "-A", "3,9"]),
    # 12-bit gray: the CLI compacts the channel into a palette by itself, -A 2 then approximates the palette META-channel too,
    # whose remainder copy (hshift -1) is coded AFTER an ordinary channel: context_predict.h:253-262 with a negative shift
    ("approx_on_palette_gray12_24x50", dict(w=24, h=50, channels=1, bits=12, seed=15, photographic=True), ["-R", "0", "-A", "2,4"]),
    ("approx_rgb8_96x80_A3", dict(w=96, h=80, channels=3, bits=8, seed=36, photographic=True), ["-A", "3,3"]),
]
JPEG_SPECS = [
    ("jpeg420_256x192_q90", dict(w=256, h=192, channels=3, bits=8, seed=20), dict(quality=90, subsampling=2)),
    ("jpeg444_136x120_q85", dict(w=136, h=120, channels=3, bits=8, seed=21), dict(quality=85, subsampling=0)),
    ("jpeggray_120x88_q80", dict(w=120, h=88, channels=1, bits=8, seed=22), dict(quality=80)),
    # 4:2:2: chroma subsampled horizontally only (the generic upsampling kernel, factors 2 x 1), odd block counts
    ("jpeg422_200x104_q88", dict(w=200, h=104, channels=3, bits=8, seed=23), dict(quality=88, subsampling=1)),
    # 4:1:1 (chroma subsampled 4 x 1; subsample.h:54-59): no JPEG writer at hand produces it, so the STREAM comes from the product's
    # own writer (fuif_amd/jpeglike.py: the transform list and coefficient planes read_jpeg.h would build from such a JPEG) and the
    # expected planes from the real reference decoding it -- the inverse is the replication branch of subsample.h:116-126
    ("jpeg411_176x72_q85", dict(w=176, h=72, channels=3, bits=8, seed=24), dict(quality=85, writer_factors=(4, 1))),
]
# raw YUV 4:2:0 input (fuif -y WxH: import/read_yuv... builds Y + two half-size chroma planes and records [YCbCr, ChromaSubsampling], no DCT): with odd sizes
# the upsampled chroma planes (98x62) are LARGER than the Y plane (97x61) -- their samples outside the colour transform's region keep the upsampled value
# and only get the final clamp (image.cpp:107-113); -Q drives values out of range.  Round 5: the fused upsampling + YCbCr kernel's edge cases.
YUV_SPECS = [
    ("yuv420p_97x61", dict(w=97, h=61, channels=3, bits=8, seed=321, sigma=6.0), dict(yuv420p=True)),
    ("yuv420p_75x49_Q60", dict(w=75, h=49, channels=3, bits=8, seed=322, sigma=25.0), dict(yuv420p=True, quality=60)),
]
# animation (FUAF): frames are stacked vertically; -M 0 keeps the 2D-match transform (out of scope) off
ANIM_SPECS = [
    ("anim3_48x32", dict(w=48, h=32, channels=3, bits=8, seed=700), dict(frames=3)),
    # default flags for an animation: 2D match against the previous frames (fuif.cpp:440-448, 2dmatch.h:147-171)
    ("anim4_match_40x28", dict(w=40, h=28, channels=3, bits=8, seed=710, static=True), dict(frames=4, match=True)),
]
# Permute (transform/permute.h): the CLI never applies it (fuif.cpp:362-368 is commented out), so these go through the
# reference's own library calls -- Image::do_transform(Transform(TRANSFORM_PERMUTE)), fuif_prepare_encode, fuif_encode -- in
# oracle/ref_driver.cpp (fuifref_encode).  "explicit": the permutation is a transform parameter; "channel": it is the content
# of a 1-row meta-channel; with colorspace/squeeze off Permute is the LAST transform, which also exercises the decode-time
# metadata permutation of encoding.cpp:576-596,712.
PERMUTE_SPECS = [
    ("permute_explicit_rgb8_48x40", dict(w=48, h=40, channels=3, bits=8, seed=81), dict(permute=1, permutation=(2, 0, 1))),
    ("permute_channel_rgb8_48x40", dict(w=48, h=40, channels=3, bits=8, seed=82), dict(permute=2, permutation=(2, 0, 1))),
    ("permute_channel_last_rgb8_44x36", dict(w=44, h=36, channels=3, bits=8, seed=83), dict(permute=2, permutation=(1, 2, 0), colorspace=0, squeeze=0)),
    ("permute_explicit_rgba14_40x36", dict(w=40, h=36, channels=4, bits=14, seed=84), dict(permute=1, permutation=(3, 1, 0, 2))),
    ("permute_channel_rgba14_40x36", dict(w=40, h=36, channels=4, bits=14, seed=85), dict(permute=2, permutation=(2, 3, 1, 0), colorspace=0)),
]
# Soft 2D matches (transform/2dmatch.h:136-140,150-158: the match channel marks a sample as a DIFFERENCE to an earlier sample or frame):
# the CLI never sets the flag (fuif.cpp:445), so these too come from the reference's library calls in oracle/ref_driver.cpp -- Transform
# (TRANSFORM_2DMATCH) with parameters {0, n-1, 1, distance}.  A lossless stream's differences are all zero; with a quantization constant
# on every channel behind the match (or a truncated file) the decoder adds up non-zero differences along the chains.
SOFTMATCH_SPECS = [
    ("softmatch_rgb_graphic_96x80_q3", dict(w=96, h=80, channels=3, bits=8, seed=51, colors=400), dict(softmatch=1, match_distance=40, quant=3)),
    ("softmatch_rgb_graphic_nosqueeze_72x60", dict(w=72, h=60, channels=3, bits=8, seed=53, colors=300), dict(softmatch=1, match_distance=200, squeeze=0)),
    ("softmatch_anim4_40x28_q2", dict(w=40, h=28, channels=3, bits=8, seed=710, static=True), dict(softmatch=1, match_distance=-2, frames=4, quant=2)),
]
PREVIEWS = {"yuv420p_97x61": [1, 3], "softmatch_rgb_graphic_96x80_q3": [2], "c1_rgb8_512x512": [0, 1, 2, 3, 4], "rgb8_97x61": [0, 2, 4], "jpeg420_256x192_q90": [0, 1, 2, 3, 4],
            "pal_rgb_graphic_120x90": [1, 3], "approx_rgb8_96x80_A3": [2], "approx_quant_rgb8_40x30": [3], "match_rgb_graphic_96x80": [3]}
TRUNCATE_EXTRA = {"softmatch_rgb_graphic_nosqueeze_72x60": [0.7], "softmatch_anim4_40x28_q2": [0.6], "approx_on_palette_gray12_24x50": [0.8], "match_rgb_graphic_96x80": [0.6]}
TRUNCATE = {"yuv420p_75x49_Q60": [0.6], "rgb8_512x384_I16_bigtrees": [0.8], "permute_channel_rgb8_48x40": [0.5], "permute_explicit_rgb8_48x40": [0.6], "rgb8_97x61": [0.2, 0.55, 0.93], "rgb8_128x128_I0": [0.5], "rgb8_112x96_E18": [0.7], "rgb8_120x88_E50": [0.6], "jpeg420_256x192_q90": [0.4], "rgb8_64x64_U": [0.6],
            "pal_rgba_graphic_72x64": [0.5], "pal_rgb_sparse_128x96": [0.7],
            "gray8_nosqueeze_60x40": [0.6], "rgb8_96x96_nosqueeze": [0.45]}


def main():
    ref = Ref()
    manifest = {"generator": "tests/golden/make_golden.py", "reference": "cloudinary/fuif @ /root/reference (unmodified)", "fixtures": []}
    # GOLDEN_ONLY=name[,name]: (re)generate just these fixtures and merge them into the committed manifest -- the reference
    # ENCODER is not deterministic on every input (gray8_nosqueeze_60x40), so a full regeneration would churn files
    only = [n for n in os.environ.get("GOLDEN_ONLY", "").split(",") if n]
    if only:
        with open(os.path.join(HERE, "manifest.json")) as f:
            manifest = json.load(f)
        manifest["fixtures"] = [e for e in manifest["fixtures"] if e["name"] not in only]
    tmp = tempfile.mkdtemp()
    for name, gen, flags in SPECS + GRAPHIC_SPECS + [(n, g, j) for n, g, j in JPEG_SPECS] + YUV_SPECS + ANIM_SPECS + PERMUTE_SPECS + SOFTMATCH_SPECS:
        if only and name not in only:
            continue
        gen = dict(gen)
        static = gen.pop("static", False)
        if "colors" in gen:
            img = graphic(**gen)
        else:
            gen.pop("photographic", None)
            poster = gen.pop("poster", 0)
            img = photographic(**gen)
            if poster:
                img = img // poster * poster   # more than 256 colours, sparse per-channel histograms
        maxval = (1 << gen["bits"]) - 1
        out = os.path.join(HERE, name + ".fuif")
        if isinstance(flags, dict) and ("permute" in flags or "softmatch" in flags):
            if flags.get("frames", 1) > 1:     # the vertical film strip of an animation whose frames share most pixels with the first one
                strip = []
                for i in range(flags["frames"]):
                    g2 = dict(gen); g2["seed"] = gen["seed"] + i
                    fr = photographic(**g2)
                    keep = np.ones(fr.shape[1:], bool); keep[4 + 3 * i: 14 + 3 * i, 6 + 5 * i: 20 + 5 * i] = False
                    strip.append(np.where(keep[None], img, fr))
                img = np.concatenate(strip, axis=1)
            blob = ref.encode(img, maxval=maxval, **flags)
            with open(out, "wb") as f:
                f.write(blob)
            src, cli_flags = None, []
        elif isinstance(flags, dict) and flags.get("yuv420p"):
            w, h = gen["w"], gen["h"]
            cw, ch2 = (w + 1) // 2, (h + 1) // 2
            src = os.path.join(tmp, name + ".yuv")
            with open(src, "wb") as f:      # planar Y, then the two chroma planes at half size (what a 4:2:0 raw video frame holds)
                f.write(img[0].astype(np.uint8).tobytes())
                f.write(np.ascontiguousarray(img[1][::2, ::2][:ch2, :cw]).astype(np.uint8).tobytes())
                f.write(np.ascontiguousarray(img[2][::2, ::2][:ch2, :cw]).astype(np.uint8).tobytes())
            cli_flags = ["-y", "%dx%d" % (w, h)] + (["-Q", str(flags["quality"])] if "quality" in flags else [])
        elif isinstance(flags, dict) and "frames" in flags:
            base = photographic(**gen)
            for i in range(flags["frames"]):
                g2 = dict(gen); g2["seed"] = gen["seed"] + i
                fr = photographic(**g2)
                if static:   # frames share most pixels with the first one: something for the match transform to find
                    keep = np.ones(fr.shape[1:], bool); keep[4 + 3 * i: 14 + 3 * i, 6 + 5 * i: 20 + 5 * i] = False
                    fr = np.where(keep[None], base, fr)
                write_pnm(os.path.join(tmp, name + "-%02d.ppm" % i), fr, maxval)
            src = os.path.join(tmp, name + "-%02d.ppm")
            cli_flags = [] if flags.get("match") else ["-M", "0"]
        elif isinstance(flags, dict) and "writer_factors" in flags:
            sys.path.insert(0, os.path.join(HERE, "..", ".."))
            from fuif_amd.jpeglike import encode_jpeg_like
            with open(out, "wb") as f:
                f.write(encode_jpeg_like(img, flags["quality"], factors=tuple(flags["writer_factors"])))
            src, cli_flags = None, []
        elif isinstance(flags, dict):
            from PIL import Image
            arr = np.moveaxis(img, 0, -1).astype(np.uint8)
            pil = Image.fromarray(arr[..., 0] if gen["channels"] == 1 else arr)
            src = os.path.join(tmp, name + ".jpg")
            pil.save(src, **flags)
            cli_flags = []
        else:
            src = os.path.join(tmp, name + (".pam" if gen["channels"] in (2, 4) else ".ppm" if gen["channels"] == 3 else ".pgm"))
            write_pnm(src, img, maxval)
            cli_flags = flags
        if src is not None:
            r = run_ref_cli(cli_flags + [src, out])
            if r.returncode != 0 or not os.path.exists(out):
                raise SystemExit("reference CLI failed for %s: %s %s" % (name, r.stdout[-400:], r.stderr[-400:]))
        blob = open(out, "rb").read()
        entry = {"name": name, "file": name + ".fuif", "bytes": len(blob), "file_sha256": hashlib.sha256(blob).hexdigest(),
                 "source": gen, "cli_flags": cli_flags if not isinstance(flags, dict) else (["<library: oracle/ref_driver.cpp fuifref_encode>", json.dumps(flags)] if ("permute" in flags or "softmatch" in flags) else ["<stream written by fuif_amd/jpeglike.py; expected planes = the reference decoding it>", json.dumps(flags)] if "writer_factors" in flags else ["<jpeg/anim/yuv>", json.dumps(flags)] + cli_flags), "cases": []}
        cases = [("full", -1, len(blob))]
        cases += [("preview%d" % k, k, len(blob)) for k in PREVIEWS.get(name, [])]
        cases += [("trunc%02d" % int(f * 100), -1, int(len(blob) * f)) for f in TRUNCATE.get(name, []) + TRUNCATE_EXTRA.get(name, [])]
        for cname, preview, nbytes in cases:
            pre, post = ref.decode_both(blob[:nbytes], preview=preview, io_kind=0)
            entry["cases"].append({"case": cname, "preview": preview, "nbytes": nbytes, "ok": bool(pre.ok),
                                   "info": pre.info, "transforms": pre.transforms, "pre": describe(pre), "post": describe(post)})
        # lossless fixtures must reproduce the source pixels
        if (not isinstance(flags, dict) and "-Q" not in flags) or (isinstance(flags, dict) and "permute" in flags):
            full = entry["cases"][0]
            _, post = ref.decode_both(blob)
            assert all(np.array_equal(post.channels[c]["data"], img[c]) for c in range(gen["channels"])), name
            full["lossless_roundtrip"] = True
        manifest["fixtures"].append(entry)
        print("%-32s %7d bytes  %d cases  channels=%d" % (name, len(blob), len(entry["cases"]), entry["cases"][0]["info"]["nch"]))
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0, separators=(",", ":"))
    print("total fixture bytes:", sum(e["bytes"] for e in manifest["fixtures"]))


if __name__ == "__main__":
    main()
