"""The unsqueeze kernels run a branch-free form of the reference's two Squeeze formulas (fuif_amd/csrc/squeeze_arith.h).
This compiles that header for the host next to the formulas as the reference writes them (transform/squeeze.h:61-77,103-107)
and compares them case by case: every triple of a small cube (all orderings, ties and parities) and tens of millions of random
triples at scales up to 2^21 (samples have at most 17 significant bits)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

HARNESS = r"""
#include <cstdio>
#include <cstdlib>
#include "squeeze_arith.h"
// transform/squeeze.h:61-77 as written there
static int ref_tendency(int B, int a, int n) {
    int diff = 0;
    if (B >= a && a >= n) {
        diff = (4 * B - 3 * n - a + 6) / 12;
        if (diff - (diff & 1) > 2 * (B - a)) diff = 2 * (B - a) + 1;
        if (diff + (diff & 1) > 2 * (a - n)) diff = 2 * (a - n);
    } else if (B <= a && a <= n) {
        diff = (4 * B - 3 * n - a - 6) / 12;
        if (diff + (diff & 1) < 2 * (B - a)) diff = 2 * (B - a) - 1;
        if (diff - (diff & 1) < 2 * (a - n)) diff = 2 * (a - n);
    }
    return diff;
}
// squeeze.h:103-107
static void ref_pair(int avg, int diff, int &A, int &B) { A = ((avg << 1) + diff + (diff > 0 ? -(diff & 1) : (diff & 1))) >> 1; B = A - diff; }
int main() {
    long bad = 0, n = 0;
    for (int B = -40; B <= 40; B++) for (int a = -40; a <= 40; a++) for (int c = -40; c <= 40; c++) { n++; bad += ref_tendency(B, a, c) != fuifgpu::smooth_tendency(B, a, c); }
    unsigned long long x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (unsigned)(x >> 16); };
    for (long i = 0; i < 30000000; i++) {
        const int sc = 1 << (rnd() % 22);
        const int B = (int)(rnd() % (2u * sc + 1)) - sc, a = B + ((int)(rnd() % (2u * sc + 1)) - sc) / (int)(1 + rnd() % 8), c = a + ((int)(rnd() % (2u * sc + 1)) - sc) / (int)(1 + rnd() % 8);
        n++; bad += ref_tendency(B, a, c) != fuifgpu::smooth_tendency(B, a, c);
    }
    for (int avg = -200; avg <= 200; avg++) for (int d = -500; d <= 500; d++) { int A1, B1, A2, B2; ref_pair(avg, d, A1, B1); fuifgpu::unsqueeze_pair(avg, d, A2, B2); n++; bad += (A1 != A2 || B1 != B2); }
    for (long i = 0; i < 10000000; i++) { const int avg = (int)(rnd() % 4000001u) - 2000000, d = (int)(rnd() % 4000001u) - 2000000; int A1, B1, A2, B2; ref_pair(avg, d, A1, B1); fuifgpu::unsqueeze_pair(avg, d, A2, B2); n++; bad += (A1 != A2 || B1 != B2); }
    printf("%ld cases, %ld mismatches\n", n, bad);
    return bad != 0;
}
"""


def test_branch_free_squeeze_formulas_equal_the_reference_form(tmp_path):
    if not shutil.which("g++"):
        pytest.skip("needs g++")
    src = tmp_path / "harness.cpp"
    src.write_text(HARNESS)
    exe = str(tmp_path / "harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "fuif_amd", "csrc"), str(src), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 mismatches" in r.stdout
