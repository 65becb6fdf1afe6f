// fuif_amd/csrc/squeeze_arith.h -- the two integer formulas of the Squeeze transform (transform/squeeze.h:61-77,103-107) in the
// branch-free form the unsqueeze kernels of transforms.hip run.  A header of its own so that tests/test_squeeze_arith.py can
// compile the very same text for the host and compare it, case by case, with the reference's form.
#pragma once
#include <cstdint>

#if defined(__HIPCC__) && !defined(FUIF_EMU)
#define SQ_DEV __device__ __forceinline__
#else
#define SQ_DEV static inline
#ifndef FUIF_EMU
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
#endif
#endif

namespace fuifgpu {

// transform/squeeze.h:61-77:
//     if (B >= a && a >= n) { d = (4*B - 3*n - a + 6) / 12;  if (d - (d&1) > 2*(B-a)) d = 2*(B-a) + 1;  if (d + (d&1) > 2*(a-n)) d = 2*(a-n); }
//     else if (B <= a && a <= n) { d = (4*B - 3*n - a - 6) / 12;  if (d + (d&1) < 2*(B-a)) d = 2*(B-a) - 1;  if (d - (d&1) < 2*(a-n)) d = 2*(a-n); }
//     else d = 0;
// without branches.  With u = B-a, v = a-n the numerator is 4u + 3v +- 6 and the second case is the first one of the negated
// inputs, negated (C's '/' truncates toward zero, and d&1 is the parity of |d|): the function is odd.  So work on U = |u|,
// V = |v| -- a non-negative numerator, an unsigned division by 12 (one multiply-high) -- then  d - (d&1) > 2U  <=>  d > 2U + 1
// and  d + (d&1) > 2V  <=>  d > 2V  (2V is even) make both clamps plain minima, and the sign goes back on at the end.
// The unsqueeze kernels are bound by these integer instructions, not by memory (round 3, profiles/r3_transforms.txt): the
// branchy form cost ~75 VALU operations per pair with both branches executed under divergence and four quarter-rate multiplies.
SQ_DEV int smooth_tendency(int B, int a, int n) {
    const int u = B - a, v = a - n;
    const int nu = -u, nv = -v;
    const unsigned U = (unsigned)(u > nu ? u : nu), V = (unsigned)(v > nv ? v : nv);
    unsigned d = __umulhi(4u * U + 3u * V + 6u, 0xAAAAAAABu) >> 3;   // / 12
    const unsigned c1 = 2u * U + 1u, c2 = 2u * V;
    d = d < c1 ? d : c1;
    d = d < c2 ? d : c2;
    const int uv = u | v;                        // negative: the falling case (u <= 0, v <= 0, not both 0)
    const int m = uv >> 31;
    const bool monotone = (uv & (nu | nv)) >= 0;   // u, v >= 0  or  u, v <= 0
    return monotone ? ((int)d ^ m) - m : 0;
}
// squeeze.h:103-107: A = ((avg<<1) + diff + (diff > 0 ? -(diff&1) : (diff&1))) >> 1 ; B = A - diff.
// The bracket is diff rounded to an even number toward zero, so A = avg + diff / 2 with C's truncating division.
SQ_DEV void unsqueeze_pair(int avg, int diff, int &A, int &B) {
    A = avg + ((diff + (int)((unsigned)diff >> 31)) >> 1);
    B = A - diff;
}

}  // namespace fuifgpu
