"""GPU parity (-m gpu) of the writer's MANIAC pixel loop on the GPU (csrc/maniac_encode.hip, fuifgpu_encode_options::gpu_entropy):
the context model of every pixel in parallel + one wavefront per group for the symbol coder and the range coder must write the
very bytes the host writer writes -- which tests/test_writer.py pins to the reference CLI (`fuif -I 0`) and to the reference
decoder.  (Named to run last: it covers a "next" row of SURVEY 8(f), not the decode path.)"""
import os

import numpy as np
import pytest

from fuif_amd.synth import photographic

pytestmark = pytest.mark.gpu

EMULATED = ("_emu" in os.path.basename(os.environ.get("FUIF_AMD_LIB", "")))
SHAPES = [(97, 61, 3, 8, True), (64, 48, 1, 8, True), (40, 30, 4, 14, False)] if EMULATED else \
         [(97, 61, 3, 8, True), (640, 480, 3, 8, True), (333, 200, 1, 12, True), (256, 256, 4, 14, False)]


@pytest.mark.parametrize("w,h,c,bits,ycocg", SHAPES)
@pytest.mark.parametrize("tree_mode", [0, 1])
def test_gpu_pixel_loop_writes_the_host_writers_bytes(gpulib, w, h, c, bits, ycocg, tree_mode):
    img = photographic(w, h, c, bits, seed=7000 + w + tree_mode)
    split = 2 if (tree_mode and w * h < 20000) else None       # small pictures: let the learner split at all
    host = gpulib.encode_image(img, bits, ycocg=ycocg, tree_mode=tree_mode, index=True, split_bits=split)
    dev = gpulib.encode_image(img, bits, ycocg=ycocg, tree_mode=tree_mode, index=True, split_bits=split, gpu_entropy=True)
    assert dev == host
    both = gpulib.encode_image(img, bits, ycocg=ycocg, tree_mode=tree_mode, index=True, split_bits=split, gpu_entropy=True, gpu_forward=True)
    assert both == host


def test_flat_and_tiny_planes(gpulib):
    """constant channels (no coder at all), planes the writer stores uncompressed (the roll-back of encoding.cpp:545-551 runs
    after the GPU attempt) and a picture too small for Squeeze"""
    flat = np.full((3, 40, 56), 77, np.int32)
    for img, kw in ((flat, {}), (photographic(5, 4, 3, 8, seed=3), {}), (photographic(31, 17, 1, 8, seed=4), dict(squeeze=False))):
        host = gpulib.encode_image(img, 8, tree_mode=1, **kw)
        dev = gpulib.encode_image(img, 8, tree_mode=1, gpu_entropy=True, **kw)
        assert dev == host


def test_batch_of_pictures_in_one_launch_pair(gpulib):
    """fuifgpu_encode_images: every channel group of every picture of the batch gets its own wavefront in ONE coder launch; each
    stream must be the bytes the one-picture writer produces (learned trees and the single-leaf mode)"""
    w, h = (72, 56) if EMULATED else (480, 360)
    imgs = [photographic(w, h, 3, 8, seed=7100 + i) for i in range(5)]
    imgs.append(np.full((3, h, w), 9, np.int32))            # a flat picture: only trivial channels, no job at all
    for tree_mode in (1, 0):
        split = 2 if (tree_mode and w * h < 20000) else None
        want = [gpulib.encode_image(im, 8, tree_mode=tree_mode, index=True, split_bits=split) for im in imgs]
        got = gpulib.encode_images(imgs, 8, tree_mode=tree_mode, index=True, split_bits=split)
        assert got == want


def test_gpu_pixel_loop_writes_the_reference_clis_bytes(gpulib):
    """closes the chain on the GPU in one step: tests/golden/rgb8_128x128_I0.fuif was written by the unmodified reference CLI
    (`fuif -I 0`, tests/golden/make_golden.py) from photographic(128, 128, 3, 8, seed=6); the GPU pixel loop (one picture, the
    batch form, and with the forward transforms on the GPU as well) must write those bytes -- up to the one stray byte the
    reference appends (BlobIO::bytes_used = seek_pos + 1, fileio.h:252-254)"""
    from conftest import GOLDEN
    ref_bytes = open(os.path.join(GOLDEN, "rgb8_128x128_I0.fuif"), "rb").read()
    img = photographic(128, 128, 3, 8, seed=6)
    for kw in (dict(gpu_entropy=True), dict(gpu_entropy=True, gpu_forward=True)):
        mine = gpulib.encode_image(img, 8, tree_mode=0, **kw)
        assert mine == ref_bytes[: len(mine)] and 0 <= len(ref_bytes) - len(mine) <= 1, kw
    batch = gpulib.encode_images([img, img], 8, tree_mode=0)
    assert all(b == ref_bytes[: len(b)] and len(ref_bytes) - len(b) <= 1 for b in batch)
