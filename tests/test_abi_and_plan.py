"""CPU: the C-ABI library loads, exports every symbol include/fuifgpu.h declares, and its host
planner reproduces the reference's channel geometry (no compute calls: no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, all_cases, golden_blob


def test_header_symbols_exported(gpulib):
    hdr = open(os.path.join(ROOT, "include", "fuifgpu.h")).read()
    declared = set(re.findall(r"\b(fuifgpu_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(gpulib.ABI_SYMBOLS), declared ^ set(gpulib.ABI_SYMBOLS)
    L = ctypes.CDLL(os.path.join(ROOT, "fuif_amd", "libfuifgpu.so"))
    for s in declared:
        assert hasattr(L, s), s
    assert gpulib.lib().fuifgpu_abi_version() == 3


def test_plan_matches_reference_geometry(gpulib, manifest):
    keys = ("w", "h", "hshift", "vshift", "hcshift", "vcshift", "component")
    for e in manifest["fixtures"]:
        c = e["cases"][0]
        blob = golden_blob(e, c)
        p = gpulib.Plan(blob)
        assert (p.info.w, p.info.h, p.info.maxval) == (c["info"]["w"], c["info"]["h"], c["info"]["maxval"])
        assert [[t[0], t[1]] for t in p.transforms] == [[t[0], t[1]] for t in c["transforms"]], e["name"]
        coded = p.coded_channels
        assert len(coded) == len(c["pre"]), e["name"]
        # ... and one that IS the last transform permutes the labels of the CODED table at decode time (encoding.cpp:576-596)
        coded_labels = bool(c["transforms"]) and c["transforms"][-1][0] == 9 and not c["transforms"][-1][1]
        for i, (a, b) in enumerate(zip(coded, c["pre"])):
            want = tuple(b[k] for k in keys)
            if coded_labels and i >= 1:
                want = want[:-1] + (-1,)
            assert tuple(a[k] for k in keys) == want, e["name"]
        outs = p.output_channels
        assert len(outs) == len(c["post"]), e["name"]
        # a Permute whose permutation is stream data (no parameters) and that is not the last transform leaves component
        # LABELS that depend on that data (permute.h:48 moves whole Channel objects): the geometry-only plan reports -1
        data_labels = any(t[0] == 9 and not t[1] for t in c["transforms"][:-1])
        for a, b in zip(outs, c["post"]):
            assert (a["w"], a["h"]) == (b["w"], b["h"]), e["name"]
            assert a["component"] == (b["component"] if not (data_labels and b["component"] >= 0) else -1), e["name"]
        # slabs: planes do not overlap and stay inside the slab
        spans = sorted((ch["offset"], ch["offset"] + ch["w"] * ch["h"]) for ch in coded if ch["w"] * ch["h"])
        assert all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1))
        assert spans[-1][1] <= p.info.coef_elems


def test_plan_rejects_garbage(gpulib):
    with pytest.raises(gpulib.FuifGpuError) as ei:
        gpulib.Plan(b"not a fuif file at all")
    assert ei.value.code == 1
    with pytest.raises(gpulib.FuifGpuError):
        gpulib.Plan(b"FUIF")


def test_chance_tables_known_answers(gpulib):
    """build_table known answers: SHA-256 / spot values of the tables the REAL reference's
    build_table (maniac/chance.cpp:31-65) produced in the build container (oracle/_ref), plus the
    spot values listed in SURVEY.md Appendix E.1."""
    import hashlib
    t = np.zeros(8192, np.uint16)
    gpulib.lib().fuifgpu_build_chance_table(t.ctypes.data, 0x0d000000, 6)
    assert hashlib.sha256(t.astype("<u2").tobytes()).hexdigest() == "acb83c93de3338dd8b92e21122b2be1d428010240f48ce11b2128283cd3f99cc"
    assert [(int(t[2 * i]), int(t[2 * i + 1])) for i in (1, 6, 100, 1024, 2048, 3072, 4000, 4090, 4095)] == \
        [(4096, 0), (6, 214), (95, 303), (972, 1180), (1944, 2152), (2916, 3124), (3797, 4005), (3882, 4090), (4096, 0)]
    assert int(t.astype(np.int64).sum()) == 16773120
    gpulib.lib().fuifgpu_build_chance_table(t.ctypes.data, 0xFFFFFFFF // 19, 2)
    assert hashlib.sha256(t.astype("<u2").tobytes()).hexdigest() == "fb750fe49057a7a8059e8798e2935ce8909007c55ab401747b72096636439528"
    assert [(int(t[2 * i]), int(t[2 * i + 1])) for i in (2, 6, 100, 1024, 2048, 4091)] == \
        [(2, 217), (5, 221), (95, 310), (970, 1186), (1940, 2156), (3876, 4092)]


def test_no_gpu_fails_loudly(gpulib, manifest):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    e = manifest["fixtures"][1]
    with pytest.raises(gpulib.FuifGpuError) as ei:
        gpulib.decode_batch([golden_blob(e, e["cases"][0])])
    assert ei.value.code == 5  # FUIFGPU_E_HIP: no CPU fallback


def test_group_by_signature_is_host_only(gpulib, manifest):
    """mixed-stream scheduling (config C5 shape): grouping needs no GPU and keeps caller order inside a group"""
    names = ["rgb8_128x128_I0", "jpeg420_256x192_q90", "rgb8_97x61", "rgb8_128x128_I0", "jpeg420_256x192_q90", "gray8_64x48"]
    by = {e["name"]: e for e in manifest["fixtures"]}
    blobs = [golden_blob(by[n], by[n]["cases"][0]) for n in names]
    groups = gpulib.group_by_signature(blobs)
    assert sorted(idx for _, idx in groups.values()) == [[0, 3], [1, 4], [2], [5]]
    for plan, idx in groups.values():
        assert gpulib.plan_bytes_per_image(plan) > 2 * plan.info.coef_elems + 4 * plan.info.out_elems


def test_header_is_plain_c(tmp_path):
    """include/fuifgpu.h is the C-ABI: it has to compile as C99 with nothing but <stddef.h>/<stdint.h> behind it"""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "abi.c"
    src.write_text('#include "fuifgpu.h"\nint use(void) { fuifgpu_encode_options o; (void)o; return FUIFGPU_OK + (int)sizeof(fuifgpu_image_info); }\n'
                   'typedef char encode_options_are_40_bytes[sizeof(fuifgpu_encode_options) == 40 ? 1 : -1];\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_encode_options_are_versioned_by_their_size(gpulib):
    """fuifgpu_encode_options::struct_size (the struct once grew without a version and old callers had bytes read past
    their object): 10 x 4 bytes today, pinned; a caller built against a header that ends before gpu_entropy passes a shorter
    struct -- the library must not look behind it (here: a sentinel that would select the GPU coder and fail on this host)"""
    import ctypes as C
    import numpy as np
    from fuif_amd.synth import photographic
    assert C.sizeof(gpulib.EncodeOptions) == 40
    img = np.ascontiguousarray(photographic(48, 40, 3, 8, seed=12), dtype=np.int32)
    full = gpulib.encode_image(img, 8, tree_mode=0)

    def call(raw):
        out, n = C.c_void_p(), C.c_size_t(0)
        rc = gpulib.lib().fuifgpu_encode_image(img.ctypes.data, 48, 40, 3, 8, C.cast(raw, C.POINTER(gpulib.EncodeOptions)), C.byref(out), C.byref(n))
        blob = C.string_at(out.value, n.value) if rc == 0 else None
        if rc == 0:
            gpulib.lib().fuifgpu_free_blob(out)
        return rc, blob
    # struct_size 36 = every field up to gpu_forward; the 4 bytes behind it hold garbage that must not be read as gpu_entropy
    words = (C.c_int32 * 10)(36, 1, 1, 12, 0, 4095, 0, int(gpulib.DEFAULT_SPLIT_BITS), 0, 0x7fffffff)
    rc, blob = call(words)
    assert rc == 0 and blob == full
    for bad in (0, 4, 38):
        words[0] = bad
        assert call(words)[0] == 4   # FUIFGPU_E_ARG
    words[0] = 400                   # a NEWER caller: the fields this library knows are read, the rest ignored
    words[9] = 0
    rc, blob = call((C.c_int32 * 100)(*list(words)))
    assert rc == 0 and blob == full


def test_fp64_kernels_are_not_contracted(tmp_path):
    """the reference build has no fused multiply-add; hipcc fuses by default, in the backend.  Compile transforms.hip with the
    product's flags and look at the ISA of the two FP64 kernels: no v_fma / v_fmac, and the iDCT's 22 + 56 operations per
    8-point pass (352 products + the DC offset; 896 sums + round()'s)."""
    import shutil
    import subprocess
    import fuif_amd
    if not shutil.which("hipcc"):
        pytest.skip("needs hipcc")
    out = str(tmp_path / "transforms.s")
    subprocess.run(["hipcc"] + fuif_amd.HIPCC_FLAGS + ["-S", "--cuda-device-only", os.path.join(ROOT, "fuif_amd", "csrc", "transforms.hip"), "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
    for name, instances in (("k_idct8x8", 2), ("k_inv_ycbcr", 1), ("k_ups2_ycbcr", 1)):      # (the iDCT is built twice: int32 / int16 AC loads)
        bodies = [m.group(1) for m in re.finditer(r"^_ZN7fuifgpu\d+%s\w*:[^\n]*\n(.*?)s_endpgm" % name, text, re.S | re.M)]
        assert len(bodies) == instances, (name, len(bodies))
        for body in bodies:
            assert not re.search(r"\bv_fmac?_f(64|32)", body), name + " holds fused multiply-adds"
            if name == "k_idct8x8":
                assert len(re.findall(r"\bv_mul_f64", body)) == 353
                assert 896 <= len(re.findall(r"\bv_add_f64", body)) <= 896 + 3 * 64 + 1
            if name == "k_ups2_ycbcr":
                assert len(re.findall(r"\bv_mul_f64", body)) == 16      # 4 products per pixel (ycbcr.h:56-58), 4 pixels per lane
