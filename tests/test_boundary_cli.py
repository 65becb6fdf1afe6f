"""The drop-in boundary end to end: the UNCHANGED reference CLI (fuif.cpp and its import/export code,
compiled from /root/reference in the build container) linked against libfuifgpu.so through
fuif_amd/boundary/fuif_gpu_boundary.cpp.  `fuif_gpu -d x.fuif out.ppm` must write exactly the file the
reference CLI writes (checked against the oracle's planes; the golden manifest pins those to the real
reference)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

CLI = os.path.join(ROOT, "fuif_amd", "boundary", "_build", "fuif_gpu")


def run_cli(args, fallback=False):
    env = dict(os.environ)
    if os.path.exists("/opt/conda/lib/libjpeg.so.9"):
        env["LD_PRELOAD"] = "/opt/conda/lib/libjpeg.so.9"
    # no environment switch: by DEFAULT a stream the GPU planner rejects makes the CLI fail instead of quietly running the
    # reference's decoder (which would keep these tests green without the GPU path); FUIFGPU_ALLOW_CPU_FALLBACK=1 is the opt-in;
    # FUIFGPU_VERBOSE makes the path taken visible
    env["FUIFGPU_VERBOSE"] = "1"
    env.pop("FUIFGPU_ALLOW_CPU_FALLBACK", None)
    if fallback:
        env["FUIFGPU_ALLOW_CPU_FALLBACK"] = "1"
    return subprocess.run([CLI] + args, env=env, capture_output=True, text=True, timeout=300)


def need_cli():
    if not os.path.exists(CLI):
        pytest.skip("fuif_amd/boundary/_build/fuif_gpu not built (needs /root/reference + png/jpeg headers)")


def test_shipped_binding_holds_no_call_of_the_reference_decoder():
    """the binding object of the shipped build has no relocation against fuif_decode_cpu / fuif_decode_file_cpu /
    cpu_decode_whole -- the reference's decoder is linked (encoder and decoder are one translation unit) but unreachable from it"""
    obj = os.path.join(ROOT, "fuif_amd", "boundary", "_build", "boundary.o")
    if not os.path.exists(obj):
        pytest.skip("fuif_amd/boundary/_build not built (needs /root/reference + png/jpeg headers)")
    dis = subprocess.run(["objdump", "-dr", obj], capture_output=True, text=True, timeout=120).stdout
    assert dis and dis.count("decode_cpu") == 0 and "cpu_decode_whole" not in dis
    # the source: every call of the reference's decoder sits inside an #ifdef FUIFGPU_WITH_CPU_FALLBACK region (before its #else)
    inside, depth = False, 0
    for line in open(os.path.join(ROOT, "fuif_amd", "boundary", "fuif_gpu_boundary.cpp")):
        t = line.strip()
        if t.startswith("#ifdef FUIFGPU_WITH_CPU_FALLBACK"):
            inside, depth = True, 1
        elif inside and t.startswith(("#if", "#ifdef", "#ifndef")):
            depth += 1
        elif inside and t.startswith("#else") and depth == 1:
            inside = False
        elif inside and t.startswith("#endif"):
            depth -= 1
            inside = depth > 0
        elif not t.startswith("//") and ("fuif_decode_cpu(" in t or "cpu_decode_whole(" in t or "fuif_decode_file_cpu(" in t):
            assert inside, line


def test_cli_identify_and_loud_failure_without_gpu(tmp_path):
    need_cli()
    r = run_cli(["-i", os.path.join(GOLDEN, "rgb8_97x61.fuif")])
    assert r.returncode == 0 and "97x61" in r.stdout      # header-only path: answered by the binding from the header bytes (identify_header)
    # ... and byte for byte what the unmodified reference CLI prints, at every verbosity level, stills / animations / palettes / DCT chains
    ref_cli = os.path.join(ROOT, "oracle", "_ref", "fuif")
    if os.path.exists(ref_cli):
        env = dict(os.environ)
        if os.path.exists("/opt/conda/lib/libjpeg.so.9"):
            env["LD_PRELOAD"] = "/opt/conda/lib/libjpeg.so.9"
        for name in ("rgb8_97x61", "anim3_48x32", "pal_rgb_graphic_120x90", "pal_rgb_channelwise_96x72", "jpeg420_256x192_q90", "rgba14_80x72",
                     "outside_gpu_scope_rgb8_64x48_E64"):
            for verbosity in ([], ["-v"], ["-vv"], ["-vvvvvv"]):
                args = ["-i"] + verbosity + [os.path.join(GOLDEN, name + ".fuif")]
                a = subprocess.run([CLI] + args, env=env, capture_output=True, text=True, timeout=60)
                b = subprocess.run([ref_cli] + args, env=env, capture_output=True, text=True, timeout=60)
                assert (a.returncode, a.stdout, a.stderr) == (b.returncode, b.stdout, b.stderr), (name, verbosity)
        bad = tmp_path / "bad.fuif"
        bad.write_bytes(b"FUI")
        a = subprocess.run([CLI, "-i", str(bad)], env=env, capture_output=True, text=True, timeout=60)
        b = subprocess.run([ref_cli, "-i", str(bad)], env=env, capture_output=True, text=True, timeout=60)
        assert (a.returncode, a.stdout, a.stderr) == (b.returncode, b.stdout, b.stderr)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = run_cli(["-d", os.path.join(GOLDEN, "rgb8_97x61.fuif"), str(tmp_path / "o.ppm")])
    assert r.returncode != 0 and "no CPU fallback" in (r.stdout + r.stderr)


def expected_pnm_payload(planes, maxval):
    inter = np.stack(planes, axis=-1)
    if maxval > 255:
        return inter.astype(">u2").tobytes()
    return inter.astype(np.uint8).tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["rgb8_97x61", "c1_rgb8_512x512", "gray8_64x48", "rgba14_80x72", "rgb8_160x120_Q80", "jpeg444_136x120_q85",
                                  "pal_rgb_graphic_120x90", "pal_rgb_channelwise_96x72", "approx_rgb8_96x80_A3"])
def test_reference_cli_decodes_through_gpu(name, port, tmp_path):
    need_cli()
    src = os.path.join(GOLDEN, name + ".fuif")
    out = str(tmp_path / "out.pam")
    r = run_cli(["-d", src, out])
    assert r.returncode == 0, r.stdout + r.stderr
    assert "entropy-decoded on the GPU" in r.stderr
    d = port.decode(open(src, "rb").read())
    w, h = d.info["w"], d.info["h"]
    planes = [c["data"][:h, :w] for c in d.channels]
    data = open(out, "rb").read()
    payload = expected_pnm_payload(planes, d.info["maxval"])
    assert data.endswith(payload) and len(data) - len(payload) < 100


@pytest.mark.gpu
def test_reference_cli_partial_decode_through_gpu(port, tmp_path):
    """`fuif -d -R 2` (responsive 1:8 preview) through the GPU path"""
    need_cli()
    src = os.path.join(GOLDEN, "c1_rgb8_512x512.fuif")
    out = str(tmp_path / "out.ppm")
    r = run_cli(["-d", "-R", "2", src, out])
    assert r.returncode == 0, r.stdout + r.stderr
    d = port.decode(open(src, "rb").read(), preview=2)
    planes = [c["data"] for c in d.channels]
    assert open(out, "rb").read().endswith(expected_pnm_payload(planes, 255))


@pytest.mark.gpu
def test_reference_cli_decodes_an_animation_through_gpu(ref, tmp_path):
    """FUAF (3 frames, and 4 frames with the CLI's default 2D match against previous frames): the GPU path, not the CPU
    route, and the same output file as the unmodified reference CLI"""
    need_cli()
    ref_cli = os.path.join(ROOT, "oracle", "_ref", "fuif")
    if not os.path.exists(ref_cli):
        pytest.skip("oracle/_ref/fuif not built")
    env = dict(os.environ)
    if os.path.exists("/opt/conda/lib/libjpeg.so.9"):
        env["LD_PRELOAD"] = "/opt/conda/lib/libjpeg.so.9"
    for name in ("anim3_48x32", "anim4_match_40x28"):
        src = os.path.join(GOLDEN, name + ".fuif")
        a, b = str(tmp_path / (name + "_gpu.pam")), str(tmp_path / (name + "_ref.pam"))
        r = run_cli(["-d", src, a])
        assert r.returncode == 0, r.stdout + r.stderr
        assert "entropy-decoded on the GPU" in r.stderr
        rb = subprocess.run([ref_cli, "-d", src, b], env=env, capture_output=True, text=True, timeout=300)
        assert rb.returncode == 0, rb.stderr
        assert open(a, "rb").read() == open(b, "rb").read(), name


@pytest.mark.gpu
def test_reference_cli_yuv_output_uses_the_per_transform_binding(tmp_path):
    """`fuif -d x.fuif out.yuv` = Image::undo_transforms(2): Transform::apply(image, true) transform by transform; the
    Squeeze inverse must come from the GPU entry points and the file must equal the unmodified reference CLI's"""
    need_cli()
    ref_cli = os.path.join(ROOT, "oracle", "_ref", "fuif")
    if not os.path.exists(ref_cli):
        pytest.skip("oracle/_ref/fuif not built")
    env = dict(os.environ)
    if os.path.exists("/opt/conda/lib/libjpeg.so.9"):
        env["LD_PRELOAD"] = "/opt/conda/lib/libjpeg.so.9"
    src = os.path.join(GOLDEN, "jpeg420_256x192_q90.fuif")
    a, b = str(tmp_path / "gpu.yuv"), str(tmp_path / "ref.yuv")
    r = run_cli(["-d", src, a])
    assert r.returncode == 0, r.stdout + r.stderr
    # undo_transforms(2) keeps the colour transform and the chroma subsampling: Squeeze, Quantization and DCT are undone, each
    # through its GPU entry point (the CLI runs with no environment switch: by default a transform this layer binds may not
    # fall back to the reference's loop)
    for name in ("Squeeze", "Quantization", "DCT"):
        assert "inverse %s on the GPU (Transform::apply)" % name in r.stderr, r.stderr
    assert "with the reference's CPU code" not in r.stderr
    rb = subprocess.run([ref_cli, "-d", src, b], env=env, capture_output=True, text=True, timeout=300)
    assert rb.returncode == 0
    assert open(a, "rb").read() == open(b, "rb").read()


@pytest.mark.gpu
def test_batch_entry_decodes_many_files_in_one_launch(tmp_path):
    """fuif_decode_files (fuif_amd/boundary/fuifgpu_boundary.h), the batch form of fuif_decode_file + undo_transforms, through its
    many-files front end: 40 files of one geometry and three others in ONE invocation; the files of one geometry share a
    launch (reported on stderr) and every output equals what the unmodified reference CLI writes for that file"""
    need_cli()
    import shutil
    batch_cli = os.path.join(ROOT, "fuif_amd", "boundary", "_build", "fuif_gpu_batch")
    ref_cli = os.path.join(ROOT, "oracle", "_ref", "fuif")
    if not (os.path.exists(batch_cli) and os.path.exists(ref_cli)):
        pytest.skip("fuif_gpu_batch / oracle/_ref/fuif not built")
    env = dict(os.environ)
    if os.path.exists("/opt/conda/lib/libjpeg.so.9"):
        env["LD_PRELOAD"] = "/opt/conda/lib/libjpeg.so.9"
    outdir = tmp_path / "out"
    outdir.mkdir()
    files = []
    for k in range(40):
        p = tmp_path / ("c1_%02d.fuif" % k)
        shutil.copy(os.path.join(GOLDEN, "c1_rgb8_512x512.fuif"), p)
        files.append(str(p))
    others = ["jpeg420_256x192_q90", "pal_rgb_graphic_120x90", "rgba14_80x72"]
    files += [os.path.join(GOLDEN, n + ".fuif") for n in others]
    r = subprocess.run([batch_cli, str(outdir)] + files, env=dict(env, FUIFGPU_VERBOSE="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    assert "40 file(s) of 512x512 decoded in one batch on the GPU" in r.stderr
    expect = {}
    for src in [files[0]] + files[40:]:
        b = str(tmp_path / "ref.pam")
        rb = subprocess.run([ref_cli, "-d", src, b], env=env, capture_output=True, text=True, timeout=300)
        assert rb.returncode == 0
        expect[os.path.basename(src)[:-5]] = open(b, "rb").read()
    for k in range(40):
        assert open(str(outdir / ("c1_%02d.pam" % k)), "rb").read() == expect["c1_00"]
    for n in others:
        assert open(str(outdir / (n + ".pam")), "rb").read() == expect[n], n


def check_batch_entry_on_several_devices(run_batch, run_ref, tmp_path, devices, env_devices=False):
    """fuif_decode_files_on (fuifgpu_boundary.h; `fuif_gpu_batch --devices a,b`, or FUIFGPU_DEVICES in the environment for plain
    fuif_decode_files): the files of every geometry are dealt round-robin to the listed GPUs, one host thread and one batch object per
    GPU; every output file equals what the unmodified reference CLI writes; a device that does not exist fails the call before
    anything is decoded.  `devices` = "0,0" on the one GPU of a box (two threads, two batches on it), "0,1" on the emulated node."""
    import shutil
    outdir = tmp_path / "multi"
    outdir.mkdir()
    files = []
    for k in range(5):
        q = tmp_path / ("m_%02d.fuif" % k)
        shutil.copy(os.path.join(GOLDEN, "rgb8_97x61.fuif"), q)
        files.append(str(q))
    others = ["jpeg420_256x192_q90", "rgba14_80x72"]
    files += [os.path.join(GOLDEN, n + ".fuif") for n in others]
    args = [str(outdir)] + files
    r = run_batch(args if env_devices else ["--devices", devices] + args, {"FUIFGPU_DEVICES": devices} if env_devices else {})
    assert r.returncode == 0, r.stderr[-800:]
    a, b = devices.split(",")
    assert "3 file(s) of 97x61 decoded in one batch on the GPU (device %s)" % a in r.stderr, r.stderr      # files 0, 2, 4
    assert "2 file(s) of 97x61 decoded in one batch on the GPU (device %s)" % b in r.stderr, r.stderr      # files 1, 3
    expect = {}
    for src in [files[0]] + files[5:]:
        ref_out = str(tmp_path / "ref_multi.pam")
        rb = run_ref(["-d", src, ref_out])
        assert rb.returncode == 0
        expect[os.path.basename(src)[:-5]] = open(ref_out, "rb").read()
    for k in range(5):
        assert open(str(outdir / ("m_%02d.pam" % k)), "rb").read() == expect["m_00"]
    for n in others:
        assert open(str(outdir / (n + ".pam")), "rb").read() == expect[n], n
    if not env_devices:
        bad = tmp_path / "multi_bad"
        bad.mkdir()
        r = run_batch(["--devices", a + ",63", str(bad)] + files, {})
        assert r.returncode != 0 and "no GPU 63 on this node" in r.stderr and not os.listdir(str(bad)), r.stderr[-400:]


@pytest.mark.gpu
@pytest.mark.parametrize("env_devices", [False, True])
def test_batch_entry_spreads_files_over_a_device_list(tmp_path, env_devices):
    """one GPU per box here: the list names it twice -- two host threads, each with its own batch object on device 0"""
    need_cli()
    batch_cli = os.path.join(ROOT, "fuif_amd", "boundary", "_build", "fuif_gpu_batch")
    ref_cli = os.path.join(ROOT, "oracle", "_ref", "fuif")
    if not (os.path.exists(batch_cli) and os.path.exists(ref_cli)):
        pytest.skip("fuif_gpu_batch / oracle/_ref/fuif not built")
    env = dict(os.environ)
    if os.path.exists("/opt/conda/lib/libjpeg.so.9"):
        env["LD_PRELOAD"] = "/opt/conda/lib/libjpeg.so.9"
    env.pop("FUIFGPU_DEVICES", None)
    check_batch_entry_on_several_devices(
        lambda args, extra: subprocess.run([batch_cli] + args, env=dict(env, FUIFGPU_VERBOSE="1", **extra), capture_output=True, text=True, timeout=600),
        lambda args: subprocess.run([ref_cli] + args, env=env, capture_output=True, text=True, timeout=300), tmp_path, "0,0", env_devices)


OUTSIDE = os.path.join(GOLDEN, "outside_gpu_scope_rgb8_64x48_E64.fuif")   # written by `fuif -E 64`: more reference properties than the GPU path takes
OUTSIDE_PPM_SHA256 = "5200e0a07cf1683c449896c13553d91b9f1881553aba0d98cd7aaee9ecedc274"   # what the unmodified reference CLI decodes it to (= the source picture)


# fixture -> the inverses that must be reported as run on the GPU through Transform::apply when the chain is undone one transform at a
# time (fuif_amd/boundary/fuif_stepwise_main.cpp): every transform but the first goes through the per-transform binding; the inverse
# of Permute moves no sample and stays the reference's statement
STEPWISE = {
    "match_rgb_graphic_nosqueeze_72x60": ["Matching", "Palette", "YCoCg"],
    "match_rgb_graphic_96x80": ["Squeeze", "Matching", "Palette", "YCoCg"],
    "anim4_match_40x28": ["Squeeze", "Matching", "YCoCg"],
    "pal_rgb_channelwise_96x72": ["Squeeze", "Palette", "YCoCg"],
    "pal_rgba_graphic_72x64": ["Squeeze", "Palette", "YCoCg"],
    "approx_quant_rgb8_40x30": ["Approximation", "Quantization", "Squeeze", "YCoCg"],
    "approx_on_palette_gray12_24x50": ["Approximation", "Palette"],
    "jpeg420_256x192_q90": ["Squeeze", "Quantization", "DCT", "ChromaSubsampling", "YCbCr"],
    "softmatch_rgb_graphic_96x80_q3": ["Quantization", "Squeeze", "Matching", "YCoCg"],
    "softmatch_rgb_graphic_nosqueeze_72x60": ["Matching"],
    "softmatch_anim4_40x28_q2": ["Quantization", "Squeeze", "Matching", "YCoCg"],
    "permute_channel_rgb8_48x40": ["YCoCg"],
    "permute_explicit_rgba14_40x36": ["Squeeze", "YCoCg"],
}


def check_stepwise_undo(run_stepwise, run_ref, tmp_path, previews=([], ["-R", "2"], ["-R", "0"])):
    """shared with tests/test_emulated_kernels.py (CPU, emulated library).  Image::undo_transforms(k) for k = n-1 .. 0: every call undoes
    one transform through Transform::apply(image, true) (image/image.cpp:94-115), which the binding sends to the single-transform
    entry points of the C-ABI -- Palette, Approximate and 2D-match included.  The PAM file must be the unmodified CLI's, also for a
    responsive decode (undecoded channels read as zeros in every inverse)."""
    for name, expected in STEPWISE.items():
        src = os.path.join(GOLDEN, name + ".fuif")
        for extra in previews:
            a, b = str(tmp_path / "step.pam"), str(tmp_path / "ref.pam")
            for f in (a, b):
                if os.path.exists(f):
                    os.remove(f)
            ra = run_stepwise(extra + [src, a])
            assert ra.returncode == 0, (name, extra, ra.stderr[-600:])
            rb = run_ref(["-d"] + extra + [src, b])
            assert rb.returncode == 0, (name, extra, rb.stderr[-300:])
            assert open(a, "rb").read() == open(b, "rb").read(), (name, extra)
            assert "with the reference's CPU code" not in ra.stderr, (name, ra.stderr[-600:])
            for t in expected:
                assert "inverse %s on the GPU (Transform::apply)" % t in ra.stderr, (name, extra, t, ra.stderr[-600:])


@pytest.mark.gpu
def test_undoing_the_chain_one_transform_at_a_time_runs_every_inverse_on_the_gpu(tmp_path):
    need_cli()
    stepwise = os.path.join(ROOT, "fuif_amd", "boundary", "_build", "fuif_gpu_stepwise")
    ref_cli = os.path.join(ROOT, "oracle", "_ref", "fuif")
    if not (os.path.exists(stepwise) and os.path.exists(ref_cli)):
        pytest.skip("fuif_gpu_stepwise / oracle/_ref/fuif not built")
    env = dict(os.environ, FUIFGPU_VERBOSE="1")
    if os.path.exists("/opt/conda/lib/libjpeg.so.9"):
        env["LD_PRELOAD"] = "/opt/conda/lib/libjpeg.so.9"
    env.pop("FUIFGPU_ALLOW_CPU_FALLBACK", None)
    check_stepwise_undo(lambda args: subprocess.run([stepwise] + args, env=env, capture_output=True, text=True, timeout=300),
                        lambda args: subprocess.run([ref_cli] + args, env=env, capture_output=True, text=True, timeout=300), tmp_path,
                        previews=([], ["-R", "2"]))   # (every process start pays the HIP runtime's second; -R 0 runs in the CPU suite on the emulator)


def check_index_tool(run_index, run_gpu_cli, run_ref, tmp_path):
    """shared with tests/test_emulated_kernels.py.  fuif_gpu_index (fuif_amd/boundary/fuif_index_main.cpp) gives files written by the
    reference encoder the group index: output = input bytes + FGIX trailer; the unmodified reference CLI decodes it like the input;
    the GPU path decodes it group by group to the same file; an indexed file is recognised and copied; a stream outside the GPU
    scope is copied without index."""
    import fuif_amd
    names = ["rgb8_97x61", "rgba14_80x72", "jpeg420_256x192_q90", "pal_rgb_graphic_120x90", "rgb8_120x88_E50"]
    outdir = tmp_path / "indexed"
    outdir.mkdir()
    files = [os.path.join(GOLDEN, n + ".fuif") for n in names] + [OUTSIDE]
    r = run_index([str(outdir)] + files)
    assert r.returncode == 0, r.stderr[-800:]
    assert "%d file(s) indexed, 1 copied unchanged, 0 failed" % len(names) in r.stderr, r.stderr[-400:]
    assert open(str(outdir / os.path.basename(OUTSIDE)), "rb").read() == open(OUTSIDE, "rb").read()
    for n in names:
        src = open(os.path.join(GOLDEN, n + ".fuif"), "rb").read()
        out = open(str(outdir / (n + ".fuif")), "rb").read()
        assert out[:len(src)] == src and out.endswith(b"FGIX") and len(out) > len(src) + 8, n
        assert len(fuif_amd.index_parse(out)) > 1, n
        a, b, c = str(tmp_path / "ref_plain.pam"), str(tmp_path / "ref_indexed.pam"), str(tmp_path / "gpu_indexed.pam")
        assert run_ref(["-d", os.path.join(GOLDEN, n + ".fuif"), a]).returncode == 0
        assert run_ref(["-d", str(outdir / (n + ".fuif")), b]).returncode == 0          # the reference never reads the trailer
        rg = run_gpu_cli(["-d", str(outdir / (n + ".fuif")), c])
        assert rg.returncode == 0, rg.stderr[-400:]
        assert open(a, "rb").read() == open(b, "rb").read() == open(c, "rb").read(), n
        tiles = [ln for ln in rg.stderr.splitlines() if " tiles, " in ln]
        assert tiles and " 1 tiles," not in tiles[0], rg.stderr[-400:]                   # one tile per channel group, not one per image
    again = tmp_path / "again"
    again.mkdir()
    r = run_index([str(again)] + [str(outdir / (n + ".fuif")) for n in names])
    assert r.returncode == 0 and "0 file(s) indexed, %d copied unchanged" % len(names) in r.stderr, r.stderr[-400:]
    for n in names:
        assert open(str(again / (n + ".fuif")), "rb").read() == open(str(outdir / (n + ".fuif")), "rb").read()
    # two inputs of one basename would land on one output file: refused before anything is written
    clash = tmp_path / "clash"
    clash.mkdir()
    r = run_index([str(clash), os.path.join(GOLDEN, names[0] + ".fuif"), str(outdir / (names[0] + ".fuif"))])
    assert r.returncode == 2 and "would both be written" in r.stderr and not os.listdir(str(clash)), r.stderr[-400:]


def check_encoder_writes_index(run_gpu_cli_env, run_ref, tmp_path):
    """shared with tests/test_emulated_kernels.py.  `fuif in.ppm out.fuif` through the binding: by default the reference's file exactly;
    with FUIFGPU_WRITE_INDEX=1 the same bytes plus the group index trailer (fuif_encode_file bound in fuif_gpu_boundary.cpp), which the
    unmodified CLI decodes to the source picture"""
    import fuif_amd
    from fuif_amd.synth import photographic, write_pnm
    src = str(tmp_path / "in.ppm")
    write_pnm(src, photographic(w=97, h=61, channels=3, bits=8, seed=2), 255)
    plain, indexed, ref = str(tmp_path / "plain.fuif"), str(tmp_path / "indexed.fuif"), str(tmp_path / "ref.fuif")
    assert run_gpu_cli_env([src, plain], {}).returncode == 0
    r = run_gpu_cli_env([src, indexed], {"FUIFGPU_WRITE_INDEX": "1"})
    assert r.returncode == 0 and "group index of" in r.stderr, r.stderr[-400:]
    assert run_ref([src, ref]).returncode == 0
    a, b, c = (open(f, "rb").read() for f in (plain, indexed, ref))
    assert a == c                                            # the binding leaves the encoder alone
    assert b[: len(a)] == a and b.endswith(b"FGIX") and len(fuif_amd.index_parse(b)) > 1
    back = str(tmp_path / "back.ppm")
    assert run_ref(["-d", indexed, back]).returncode == 0
    assert open(back, "rb").read() == open(src, "rb").read()


@pytest.mark.gpu
def test_encoding_through_the_binding_can_write_the_group_index(tmp_path):
    need_cli()
    ref_cli = os.path.join(ROOT, "oracle", "_ref", "fuif")
    if not os.path.exists(ref_cli):
        pytest.skip("oracle/_ref/fuif not built")
    env = dict(os.environ, FUIFGPU_VERBOSE="1")
    if os.path.exists("/opt/conda/lib/libjpeg.so.9"):
        env["LD_PRELOAD"] = "/opt/conda/lib/libjpeg.so.9"
    env.pop("FUIFGPU_WRITE_INDEX", None)
    check_encoder_writes_index(lambda args, extra: subprocess.run([CLI] + args, env=dict(env, **extra), capture_output=True, text=True, timeout=300),
                               lambda args: subprocess.run([ref_cli] + args, env=env, capture_output=True, text=True, timeout=300), tmp_path)


@pytest.mark.gpu
def test_index_tool_gives_reference_written_files_the_group_index(tmp_path):
    need_cli()
    tool = os.path.join(ROOT, "fuif_amd", "boundary", "_build", "fuif_gpu_index")
    ref_cli = os.path.join(ROOT, "oracle", "_ref", "fuif")
    if not (os.path.exists(tool) and os.path.exists(ref_cli)):
        pytest.skip("fuif_gpu_index / oracle/_ref/fuif not built")
    env = dict(os.environ, FUIFGPU_VERBOSE="1")
    if os.path.exists("/opt/conda/lib/libjpeg.so.9"):
        env["LD_PRELOAD"] = "/opt/conda/lib/libjpeg.so.9"
    check_index_tool(lambda args: subprocess.run([tool] + args, env=env, capture_output=True, text=True, timeout=300),
                     lambda args: run_cli(args), lambda args: subprocess.run([ref_cli] + args, env=env, capture_output=True, text=True, timeout=300), tmp_path)


def check_cpu_route_is_opt_in(run, tmp_path):
    """shared with tests/test_emulated_kernels.py (CPU): a stream outside the GPU scope is a loud error and no output file -- and in the
    SHIPPED binding (round 6: built without -DFUIFGPU_WITH_CPU_FALLBACK) no environment switch changes that: the reference's decoder
    is not reachable from it"""
    out = str(tmp_path / "outside.ppm")
    for switch in (False, True):
        r = run(["-d", OUTSIDE, out], switch)
        assert r.returncode != 0 and "outside the GPU path" in (r.stdout + r.stderr), r.stdout + r.stderr
        assert "built without the reference's CPU decoder" in (r.stdout + r.stderr), r.stdout + r.stderr
        assert "decoding with the reference's CPU code" not in r.stderr
        assert not os.path.exists(out) or os.path.getsize(out) == 0


@pytest.mark.gpu
def test_cpu_route_is_opt_in(tmp_path):
    need_cli()
    check_cpu_route_is_opt_in(lambda args, fb: run_cli(args, fallback=fb), tmp_path)
