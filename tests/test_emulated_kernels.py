"""CPU: the GPU parity tests, run against the kernels' own sources compiled for a wavefront emulator.

tools/emu/ compiles fuif_amd/csrc/*.hip UNCHANGED (plus -DFUIF_EMU for three inline-asm helpers) with g++ against
a stand-in hip_runtime.h in which a 64-lane wavefront is 64 cooperative fibers and every cross-lane operation
(readlane, readfirstlane, ballot, ds_bpermute, __syncthreads) is a rendezvous.  That checks the LOGIC of the
entropy kernel, the tile scheduler's bookkeeping and every inverse-transform kernel against the golden vectors
on a machine without a GPU -- each round only has minutes of GPU time, kernel edits should not need them to find
out that a refactor broke bit-exactness.  It is test infrastructure: the emulated library is built under
tests/_emu/, is only ever loaded through FUIF_AMD_LIB by this file, and says nothing about speed or about the
memory-model side of the tile hand-off (those are the -m gpu tests' job on the MI355X)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

EMU_DIR = os.path.join(ROOT, "tests", "_emu")
EMU_LIB = os.path.join(EMU_DIR, "libfuifgpu_emu.so")
CSRC = os.path.join(ROOT, "fuif_amd", "csrc")
SOURCES = ["plan.cpp", "index.cpp", "writer.cpp", "maniac_decode.hip", "maniac_encode.hip", "transforms.hip", "capi.hip"]


def build_emulated_library(extra=(), name="libfuifgpu_emu.so"):
    lib = os.path.join(EMU_DIR, name)
    os.makedirs(EMU_DIR, exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "tools", "emu", "emu_runtime.cpp"),
                                                               os.path.join(ROOT, "tools", "emu", "hip", "hip_runtime.h"),
                                                               os.path.join(ROOT, "include", "fuifgpu.h")]
    if os.path.exists(lib) and all(os.path.getmtime(lib) >= os.path.getmtime(d) for d in deps):
        return lib
    cmd = ["g++", "-std=c++17", "-O2", "-pthread", "-fPIC", "-shared", "-DFUIF_EMU", "-ffp-contract=off", "-Wno-attributes"] + list(extra) + \
          ["-I", os.path.join(ROOT, "tools", "emu"), "-x", "c++"] + [os.path.join(CSRC, s) for s in SOURCES] + \
          [os.path.join(ROOT, "tools", "emu", "emu_runtime.cpp"), "-o", lib]
    subprocess.check_call(cmd)
    return lib


# (node id, rough seconds on the emulator): the run below deals them to concurrent pytest processes longest first -- xdist hands every worker
# a run of CONSECUTIVE tests to start with, which put the parametrised shards of one slow test on one worker
SELECTED = [
    ("tests/test_gpu_synthetic.py::test_deep_trees_walk_through_chained_supernodes[False]", 50),
    ("tests/test_gpu_group_parallel.py::test_mixed_batch_with_more_tiles_than_wavefronts", 48),
    ("tests/test_gpu_synthetic.py::test_deep_trees_walk_through_chained_supernodes[True]", 45),
    ("tests/test_gpu_group_parallel.py::test_previews_of_indexed_streams", 38),
    ("tests/test_zz_gpu_encoder.py", 35),
    ("tests/test_gpu_synthetic.py::test_sibling_batch_pipelines_uploads", 27),
    ("tests/test_fuzz.py::test_gpu_agrees_with_oracle_on_corrupt_payload", 26),
] + [("tests/test_gpu_group_parallel.py::test_reference_written_files_indexed_after_the_fact[%d]" % k, 25) for k in range(4)] + [
    ("tests/test_gpu_parity.py::test_packed_output_is_the_pam_payload[%d]" % k, 23) for k in range(4)] + [
    ("tests/test_gpu_parity.py::test_batch_of_replicas_and_distinct_streams", 22),
    ("tests/test_gpu_synthetic.py::test_unsqueeze_kernels_on_geometries_around_their_tile_edges", 80),
    ("tests/test_gpu_transform_exports.py", 20),
] + [("tests/test_gpu_parity.py::test_golden_fixtures_bit_exact[%d]" % k, 12) for k in range(8)] + [
    ("tests/test_gpu_group_parallel.py::test_jpeg_like_indexed", 10),
    ("tests/test_gpu_group_parallel.py::test_writer_indexed_streams_vs_oracle[97-61-3-8-2]", 8),
    ("tests/test_gpu_group_parallel.py::test_writer_indexed_streams_vs_oracle[301-47-1-8-42]", 8),
    ("tests/test_gpu_group_parallel.py::test_add_group_index_in_one_launch_per_geometry", 8),
    ("tests/test_gpu_group_parallel.py::test_truncated_indexed_stream_falls_back_and_side_index_on_truncated_blob", 6),
    ("tests/test_gpu_group_parallel.py::test_stale_index_is_flagged_not_silently_wrong", 5),
    ("tests/test_gpu_parity.py::test_undo_transforms_is_once_per_decode", 5),
    ("tests/test_gpu_parity.py::test_invalid_permutation_flags_the_image_and_zero_fills", 5),
    ("tests/test_gpu_synthetic.py::test_sibling_outliving_its_primary_is_refused_not_dangling", 3),
    ("tests/test_gpu_synthetic.py::test_yuv420p_fixtures_with_and_without_the_fused_colour_kernel", 6),
    ("tests/test_gpu_parity.py::test_wide_configuration_for_two_batches_in_flight", 25),
    ("tests/test_gpu_synthetic.py::test_jpeg_like_chain_fused_and_unfused_match_oracle", 20),
    ("tests/test_gpu_group_parallel.py::test_add_group_index_copies_refused_streams_through", 2),
] + [("tests/test_gpu_synthetic.py::test_context_formats_of_round_6[%s-%s]" % (ix, case), 30) for ix, case in (
    # (round 6's context formats: two with the group index -- dense configuration, context areas --, one without: the wide configuration's LDS-resident
    # supernodes; narrow + compact with the index runs among the CONCURRENT cases, where its tiles are suspended and resumed)
    ("True", "narrow_full_leaves"), ("True", "tall_wide_format"), ("False", "deep_bits_wide_format"))]


def run_dealt(weighted, env, workers, timeout=1700):
    """the -m gpu tests `weighted` = [(node id, seconds)] in `workers` concurrent pytest processes, dealt longest first to the least
    loaded process; every process must pass"""
    bins = [[0, []] for _ in range(max(1, workers))]
    for node, w in sorted(weighted, key=lambda t: -t[1]):
        b = min(bins, key=lambda x: x[0])
        b[0] += w
        b[1].append(node)
    procs = [subprocess.Popen([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"] + b[1], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for b in bins if b[1]]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += "\n[timed out]"
        outs.append((p.returncode, out))
    for rc, out in outs:
        assert rc == 0 and " passed" in out and "failed" not in out, out[-3000:]
    return outs


CONCURRENT = [
    "tests/test_gpu_group_parallel.py::test_reference_written_files_indexed_after_the_fact",
    "tests/test_gpu_group_parallel.py::test_mixed_batch_with_more_tiles_than_wavefronts",
    "tests/test_gpu_group_parallel.py::test_jpeg_like_indexed",
    # round 6: suspended and resumed tiles whose supernodes are narrow / whose leaves are compact (the format flags travel in the tile record)
    "tests/test_gpu_synthetic.py::test_context_formats_of_round_6[True-narrow_compact]",
]


def _workers(n):
    try:
        import xdist  # noqa: F401
        return ["-n", str(n)]
    except ImportError:
        return []


def test_tile_hand_off_with_concurrent_emulated_wavefronts():
    """the same group-parallel tests with FOUR persistent wavefronts running at once (one OS thread each, EMU_WAVES /
    EMU_THREADS): tiles really wait for each other's headers and rows here.  x86 is more strongly ordered than the
    GPU, so this checks the LOGIC of the protocol (who publishes what, every exit path, no deadlock), not its fences."""
    if sys.platform != "linux" or os.uname().machine != "x86_64":
        pytest.skip("the emulator's context switch is x86-64 SysV assembly")
    lib = build_emulated_library()
    env = dict(os.environ)
    env.update(FUIF_AMD_LIB=lib, FUIF_TEST_MAX_PIXELS="50000", FUIF_TEST_BATCH="12", EMU_ALARM="1500", EMU_WAVES="4", EMU_THREADS="4")
    cmd = [sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"] + CONCURRENT + _workers(2)   # (x 4 emulator threads each)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1700)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.parametrize("ctx_kb", ["0", "64"])
def test_pinned_tiles_with_concurrent_emulated_wavefronts(ctx_kb):
    """FUIFGPU_CTX_KB=0 / 64: the context arenas are empty / run out after the first tiles, so suspendable tiles are PINNED to the
    wavefront that started them (maniac_decode.hip, pinned_tix).  Four persistent emulated wavefronts really suspend and resume
    them; the dense group-parallel cases must decode as with full arenas (ADVICE r2's deadlock)."""
    if sys.platform != "linux" or os.uname().machine != "x86_64":
        pytest.skip("the emulator's context switch is x86-64 SysV assembly")
    lib = build_emulated_library()
    env = dict(os.environ)
    # (the small fixtures only: the hand-off protocol at full fixture size is the test above; this one is about where a tile's context lives)
    env.update(FUIF_AMD_LIB=lib, FUIF_TEST_MAX_PIXELS="12000", FUIF_TEST_BATCH="9", FUIF_TEST_PINNED_BATCH="3", EMU_ALARM="1500", EMU_WAVES="4", EMU_THREADS="4",
               FUIFGPU_CTX_KB=ctx_kb)
    cmd = [sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider",
           "tests/test_gpu_group_parallel.py::test_reference_written_files_indexed_after_the_fact",
           "tests/test_gpu_group_parallel.py::test_writer_indexed_streams_vs_oracle[97-61-3-8-2]"] + _workers(2)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1700)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_gpu_parity_tests_pass_on_the_wavefront_emulator():
    if sys.platform != "linux" or os.uname().machine != "x86_64":
        pytest.skip("the emulator's context switch is x86-64 SysV assembly")
    lib = build_emulated_library()
    env = dict(os.environ)
    env.update(FUIF_AMD_LIB=lib, FUIF_TEST_MAX_PIXELS="50000", FUIF_TEST_BATCH="12", EMU_ALARM="1500")
    run_dealt(SELECTED, env, max(1, min(7, (os.cpu_count() or 2) - 1)))


def test_device_selection_on_an_emulated_node_of_two_devices():
    """EMU_DEVICES=2: fuifgpu_set_device / batch_device / peer_copy between two "devices" (host memory both; the current device is
    per-thread state in the emulator as in HIP) -- a batch created on device 1 decodes there while the caller sits on device 0"""
    if sys.platform != "linux" or os.uname().machine != "x86_64":
        pytest.skip("the emulator's context switch is x86-64 SysV assembly")
    env = dict(os.environ, FUIF_AMD_LIB=build_emulated_library(), EMU_DEVICES="2", EMU_ALARM="600")
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider",
                        "tests/test_gpu_transform_exports.py::test_device_selection_and_peer_copy",
                        "tests/test_gpu_group_parallel.py::test_add_group_index_copies_refused_streams_through"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "2 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]


def test_node_by_node_walk_beyond_the_supernode_cap():
    """the library built with room for three supernodes per tree (-DFUIF_MAX_SUPER=3; the product reserves thousands, so the
    path never runs in the other tests): every subtree beyond them is walked node by node from the parse-order array"""
    if sys.platform != "linux" or os.uname().machine != "x86_64":
        pytest.skip("the emulator's context switch is x86-64 SysV assembly")
    lib = build_emulated_library(extra=["-DFUIF_MAX_SUPER=3"], name="libfuifgpu_emu_cap3.so")
    env = dict(os.environ)
    env.update(FUIF_AMD_LIB=lib, FUIF_TEST_MAX_PIXELS="12000", EMU_ALARM="1500")
    run_dealt([("tests/test_gpu_synthetic.py::test_deep_trees_walk_through_chained_supernodes[False]", 55),
               ("tests/test_gpu_synthetic.py::test_deep_trees_walk_through_chained_supernodes[True]", 40)] +
              [("tests/test_gpu_parity.py::test_golden_fixtures_bit_exact[%d]" % k, 7) for k in range(8)], env, 4)


def test_reference_cli_through_the_boundary_writes_the_reference_files(tmp_path):
    """build container only: the UNCHANGED reference CLI linked through fuif_amd/boundary to the emulated library
    (found via LD_LIBRARY_PATH under the product library's name) must write byte for byte the file the real reference
    CLI writes -- palette, approximate, 2D-match, truncation-sensitive and JPEG-transcoded fixtures included"""
    import shutil
    gpu_cli = os.path.join(ROOT, "fuif_amd", "boundary", "_build", "fuif_gpu")
    ref_cli = os.path.join(ROOT, "oracle", "_ref", "fuif")
    if not (os.path.exists(gpu_cli) and os.path.exists(ref_cli)):
        pytest.skip("needs the boundary binary and oracle/_ref/fuif (built from /root/reference)")
    if sys.platform != "linux" or os.uname().machine != "x86_64":
        pytest.skip("the emulator's context switch is x86-64 SysV assembly")
    libdir = tmp_path / "lib"
    libdir.mkdir()
    shutil.copy(build_emulated_library(), libdir / "libfuifgpu.so")
    env = dict(os.environ)
    if os.path.exists("/opt/conda/lib/libjpeg.so.9"):
        env["LD_PRELOAD"] = "/opt/conda/lib/libjpeg.so.9"
    env.pop("FUIFGPU_ALLOW_CPU_FALLBACK", None)
    env_gpu = dict(env, LD_LIBRARY_PATH=str(libdir), EMU_ALARM="600")   # no switch set: by default it is the GPU path or nothing
    names = ["rgb8_97x61", "pal_rgb_graphic_120x90", "pal_rgba_graphic_72x64", "pal_rgb_channelwise_96x72", "approx_quant_rgb8_40x30",
             "approx_on_palette_gray12_24x50", "match_rgb_graphic_96x80", "gray8_nosqueeze_60x40", "jpeg420_256x192_q90", "rgba14_80x72",
             "anim3_48x32", "anim4_match_40x28", "softmatch_rgb_graphic_96x80_q3", "softmatch_anim4_40x28_q2"]
    # the reference's CPU decoder behind the binding is opt-in: a stream outside the GPU scope is a loud error by default
    from test_boundary_cli import check_cpu_route_is_opt_in
    check_cpu_route_is_opt_in(lambda args, fb: subprocess.run([gpu_cli] + args, env=dict(env_gpu, FUIFGPU_VERBOSE="1", **({"FUIFGPU_ALLOW_CPU_FALLBACK": "1"} if fb else {})),
                                                              capture_output=True, text=True, timeout=600), tmp_path)
    # the chain undone one transform at a time: every inverse through the binding's Transform::apply and the single-transform entry points
    stepwise = os.path.join(ROOT, "fuif_amd", "boundary", "_build", "fuif_gpu_stepwise")
    if os.path.exists(stepwise):
        from test_boundary_cli import check_stepwise_undo
        check_stepwise_undo(lambda args: subprocess.run([stepwise] + args, env=dict(env_gpu, FUIFGPU_VERBOSE="1"), capture_output=True, text=True, timeout=600),
                            lambda args: subprocess.run([ref_cli] + args, env=env, capture_output=True, text=True, timeout=600), tmp_path)
    # the indexing front end: reference-written files + FGIX trailer, decoded group by group afterwards
    index_tool = os.path.join(ROOT, "fuif_amd", "boundary", "_build", "fuif_gpu_index")
    if os.path.exists(index_tool):
        from test_boundary_cli import check_index_tool
        check_index_tool(lambda args: subprocess.run([index_tool] + args, env=dict(env_gpu, FUIFGPU_VERBOSE="1"), capture_output=True, text=True, timeout=600),
                         lambda args: subprocess.run([gpu_cli] + args, env=dict(env_gpu, FUIFGPU_VERBOSE="1"), capture_output=True, text=True, timeout=600),
                         lambda args: subprocess.run([ref_cli] + args, env=env, capture_output=True, text=True, timeout=600), tmp_path)
    from test_boundary_cli import check_encoder_writes_index
    enc_dir = tmp_path / "enc"
    enc_dir.mkdir()
    check_encoder_writes_index(lambda args, extra: subprocess.run([gpu_cli] + args, env=dict(env_gpu, FUIFGPU_VERBOSE="1", **extra), capture_output=True, text=True, timeout=600),
                               lambda args: subprocess.run([ref_cli] + args, env=env, capture_output=True, text=True, timeout=600), enc_dir)
    for name in names:
        src = os.path.join(ROOT, "tests", "golden", name + ".fuif")
        for extra in ([], ["-R", "2"]):
            a, b = str(tmp_path / "gpu.pam"), str(tmp_path / "ref.pam")
            ra = subprocess.run([gpu_cli, "-d"] + extra + [src, a], env=env_gpu, capture_output=True, text=True, timeout=600)
            rb = subprocess.run([ref_cli, "-d"] + extra + [src, b], env=env, capture_output=True, text=True, timeout=600)
            assert ra.returncode == rb.returncode == 0, (name, extra, ra.stderr[-300:], rb.stderr[-300:])
            assert open(a, "rb").read() == open(b, "rb").read(), (name, extra)
    # the batch entry of the binding (fuif_decode_files, fuifgpu_boundary.h) through its many-files front end: files of three
    # geometries, two of them twice, one launch per geometry; every output file equal to the unmodified CLI's
    batch_cli = os.path.join(ROOT, "fuif_amd", "boundary", "_build", "fuif_gpu_batch")
    if os.path.exists(batch_cli):
        outdir = tmp_path / "batch"
        outdir.mkdir()
        picks = ["rgb8_97x61", "jpeg420_256x192_q90", "pal_rgb_graphic_120x90", "rgba14_80x72"]
        files = [os.path.join(ROOT, "tests", "golden", n + ".fuif") for n in picks]
        dup = tmp_path / "rgb8_97x61_again.fuif"
        shutil.copy(files[0], dup)
        ra = subprocess.run([batch_cli, str(outdir)] + files + [str(dup)], env=dict(env_gpu, FUIFGPU_VERBOSE="1"), capture_output=True, text=True, timeout=900)
        assert ra.returncode == 0, ra.stderr[-600:]
        assert "2 file(s) of 97x61 decoded in one batch on the GPU" in ra.stderr
        for n in picks + ["rgb8_97x61_again"]:
            src = str(dup) if n.endswith("_again") else os.path.join(ROOT, "tests", "golden", n + ".fuif")
            b = str(tmp_path / "ref_batch.pam")
            rb = subprocess.run([ref_cli, "-d", src, b], env=env, capture_output=True, text=True, timeout=600)
            assert rb.returncode == 0
            assert open(str(outdir / (n + ".pam")), "rb").read() == open(b, "rb").read(), n
        # a group larger than what the device holds goes through one Batch object chunk by chunk (here forced to one file per chunk)
        outdir2 = tmp_path / "batch_chunked"
        outdir2.mkdir()
        rc = subprocess.run([batch_cli, str(outdir2)] + files + [str(dup)], env=dict(env_gpu, FUIFGPU_VERBOSE="1", FUIFGPU_BOUNDARY_CHUNK="1"),
                            capture_output=True, text=True, timeout=900)
        assert rc.returncode == 0, rc.stderr[-600:]
        assert "2 file(s) of 97x61 decoded in 2 batches of up to 1 on the GPU" in rc.stderr
        for n in picks + ["rgb8_97x61_again"]:
            assert open(str(outdir2 / (n + ".pam")), "rb").read() == open(str(outdir / (n + ".pam")), "rb").read(), n
        # several GPUs behind the boundary (fuif_decode_files_on): the emulated node has two "devices" (EMU_DEVICES), one host thread each
        from test_boundary_cli import check_batch_entry_on_several_devices
        for env_devices in (False, True):
            md = tmp_path / ("multi_dev_%d" % env_devices)
            md.mkdir()
            check_batch_entry_on_several_devices(
                lambda args, extra: subprocess.run([batch_cli] + args, env=dict(env_gpu, FUIFGPU_VERBOSE="1", EMU_DEVICES="2", **extra), capture_output=True, text=True, timeout=900),
                lambda args: subprocess.run([ref_cli] + args, env=env, capture_output=True, text=True, timeout=600), md, "0,1", env_devices)
    # `-d x.fuif out.yuv` keeps the colour transform and the chroma subsampling: Image::undo_transforms(2) (fuif.cpp:230),
    # i.e. Transform::apply(image, true) per transform -- Squeeze, Quantization and DCT inverses through the boundary's binding to
    # fuifgpu_inv_hsqueeze / fuifgpu_inv_vsqueeze / fuifgpu_inv_quantize / fuifgpu_idct8x8
    src = os.path.join(ROOT, "tests", "golden", "jpeg420_256x192_q90.fuif")
    a, b = str(tmp_path / "gpu.yuv"), str(tmp_path / "ref.yuv")
    ra = subprocess.run([gpu_cli, "-d", src, a], env=dict(env_gpu, FUIFGPU_VERBOSE="1"), capture_output=True, text=True, timeout=600)
    rb = subprocess.run([ref_cli, "-d", src, b], env=env, capture_output=True, text=True, timeout=600)
    assert ra.returncode == rb.returncode == 0, (ra.stderr[-300:], rb.stderr[-300:])
    for tname in ("Squeeze", "Quantization", "DCT"):
        assert "inverse %s on the GPU (Transform::apply)" % tname in ra.stderr, ra.stderr
    assert open(a, "rb").read() == open(b, "rb").read()
