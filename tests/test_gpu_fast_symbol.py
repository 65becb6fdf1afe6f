"""-m gpu: the hand-written symbol decoder that SHIPS (fast_symbol_hw, one inline-asm block of csrc/maniac_decode.hip) against
fast_symbol, its C++ specification -- which is what the wavefront emulator of the CPU suite executes (-DFUIF_EMU) -- on 400 000
random coder states, chance sets and stream bytes (tiny and huge chances, ranges at the renormalisation bound, exhausted
exponents); the leaf commit against its formula; ds_bpermute with address bits above bit 7.  tools/test_fast_symbol.hip includes
the kernel source itself, so the test runs the very code of the library (reader<15>: maniac/symbol.h:154-185; RacInput::get:
maniac/rac.h:82-95; SimpleBitChance::put: maniac/chance.h:77-79).  VERDICT r4 item 8: until round 5 only the session scripts ran it."""
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu


def test_asm_symbol_decoder_equals_its_cpp_specification(gpulib):
    exe = gpulib.build_unit_test()
    r = subprocess.run([exe, "400000"], capture_output=True, text=True, timeout=300)
    tail = (r.stdout + r.stderr)[-1500:]
    m = re.search(r"(\d+) cases: (\d+) decoder mismatches, (\d+) commit mismatches, (\d+) bpermute mismatches \((\d+) zero symbols\)", r.stdout)
    assert r.returncode == 0 and m, tail
    cases, bad, commit_bad, bperm_bad, zero = (int(x) for x in m.groups())
    assert cases == 400000 and bad == 0 and commit_bad == 0 and bperm_bad == 0, tail
    assert 0 < zero < cases      # (both the zero-symbol path and the exponent / mantissa loops were taken)
