// fuif_amd/csrc/transforms.hip -- inverse-transform kernels of the FUIF decode path for gfx950.
//
// All of these are integer/FP64 streaming kernels bound by HBM bandwidth; none is GEMM-shaped, so
// no MFMA.  Planes are row-major int32.  One launch processes one schedule op for a whole chunk
// of images: grid.z = image, address = base[buf] + z*stride[buf] + plane offset.
//
//   k_inv_vsqueeze   transform/squeeze.h:173-224   one lane per column, serial down the rows
//   k_inv_hsqueeze_rows  transform/squeeze.h:81-132  one lane per row (the recurrence runs along x), a cache line per step
//   k_inv_ycocg      transform/ycocg.h:49-61       elementwise on three planes (+ clamp)
//   k_inv_ycbcr      transform/ycbcr.h:49-60       float operands, double arithmetic, no FMA
//   k_dequant        transform/quantize.h:32-49    elementwise * Channel::q (per image, per plane)
//   k_idct8x8        transform/dct.h:88-107,282-291 FP64, reference summation order, no FMA
//   k_upsample       transform/subsample.h:90-115  "fancy" 2x chroma upsampling
//   k_clamp / k_copy_clamp   image/image.cpp:107-113
//   k_inv_palette    transform/palette.h:57-64     gather through the decoded palette meta-channel
//   k_permute_plane  transform/permute.h:31-54     plane gather by a permutation that is stream data (meta-channel form)
//   k_inv_approx     transform/approximate.h:44-57 quotient * q + remainder, in place
//   k_inv_match_frames  transform/2dmatch.h:147-171  copy / add the co-located sample of an earlier frame
//   k_pack_samples   export/write_pam.h:136-150    interleaved 8/16-bit samples of the final planes (what a PNM/PAM holds)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "fuifgpu_internal.h"
#include "transforms.h"
#include "squeeze_arith.h"

// The reference's x86-64 build has no fused multiply-add: every product and every sum of the FP64 paths (iDCT, YCbCr) is rounded
// on its own.  hipcc fuses a*b+c into v_fma_f64 by default, in the backend, also through __dmul_rn / __dadd_rn (plain operators
// in HIP) and whatever `#pragma clang fp contract(off)` says: this file MUST be compiled with -ffp-contract=off
// (fuif_amd.HIPCC_FLAGS; tests/test_abi_and_plan.py checks the ISA).  Rounds 1-2 shipped contracted code; it passed every
// fixture because a last-bit difference in a double only shows when the value sits within ~1e-13 of a rounding boundary of the
// integer result.

namespace fuifgpu {

namespace {

#define DEV __device__ __forceinline__

DEV int32_t *plane_ptr(const Bases &b, const PlaneRef &p, int z) { return b.base[p.buf] + (int64_t)z * b.stride[p.buf] + p.off; }
// a squeeze residual: TR = int32_t -> like every other plane; TR = coef_t -> the coded plane itself, int16 samples in the coefficient slab (Op::r16)
template <typename TR> DEV const TR *residual_ptr(const Bases &b, const PlaneRef &p, int z);
template <> DEV const int32_t *residual_ptr<int32_t>(const Bases &b, const PlaneRef &p, int z) { return plane_ptr(b, p, z); }
template <> DEV const coef_t *residual_ptr<coef_t>(const Bases &b, const PlaneRef &p, int z) { return b.c16 + (int64_t)z * b.stride[BUF_COEF] + p.off; }
DEV int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

}  // namespace

// ---------------------------------------------------------------------------------------------
// vertical unsqueeze: avg (w x h1) + residual (w x h2) -> out (w x (h1+h2)), h1-h2 in {0,1}
#ifndef FUIF_VS_STEP
#define FUIF_VS_STEP 8
#endif
constexpr int VS_STEP = FUIF_VS_STEP;
template <typename TR>
__global__ __launch_bounds__(256) void k_inv_vsqueeze(Bases b, PlaneRef pa, PlaneRef pr, PlaneRef po, int clamp, int lo, int hi) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int w = pa.w, h1 = pa.h, h2 = pr.h;
    if (x >= w) return;
    const int32_t *a = plane_ptr(b, pa, blockIdx.z) + x;
    const TR *r = residual_ptr<TR>(b, pr, blockIdx.z) + x;
    int32_t *o = plane_ptr(b, po, blockIdx.z) + x;
    int avg = a[0];
    int prevB = avg;  // first pair uses tendency(avg,avg,next): squeeze.h:186
    int y = 0;
    // VS_STEP row pairs per step, their loads issued together (the recurrence down the column is serial: with one pair per
    // step a lane has a single pair of loads in flight)
    for (; y + VS_STEP < h1 && y + VS_STEP <= h2; y += VS_STEP) {   // avg rows y+1 .. y+VS_STEP all exist
        int nv[VS_STEP], rv[VS_STEP];
#pragma unroll
        for (int k = 0; k < VS_STEP; k++) {
            nv[k] = a[(int64_t)(y + 1 + k) * w];
            rv[k] = r[(int64_t)(y + k) * w];
        }
#pragma unroll
        for (int k = 0; k < VS_STEP; k++) {
            const int diff = rv[k] + smooth_tendency(prevB, avg, nv[k]);
            int A, B;
            unsqueeze_pair(avg, diff, A, B);
            o[(int64_t)(2 * (y + k)) * w] = clamp ? clampi(A, lo, hi) : A;
            o[(int64_t)(2 * (y + k) + 1) * w] = clamp ? clampi(B, lo, hi) : B;
            prevB = B;
            avg = nv[k];
        }
    }
    for (; y < h2; y++) {
        const int next_avg = (y + 1 < h1) ? a[(int64_t)(y + 1) * w] : avg;
        const int res = r[(int64_t)y * w];
        const int diff = res + smooth_tendency(prevB, avg, next_avg);
        int A, B;
        unsqueeze_pair(avg, diff, A, B);
        o[(int64_t)(2 * y) * w] = clamp ? clampi(A, lo, hi) : A;
        o[(int64_t)(2 * y + 1) * w] = clamp ? clampi(B, lo, hi) : B;
        prevB = B;
        avg = next_avg;
    }
    if ((h1 + h2) & 1) {  // squeeze.h:217-222
        const int v = a[(int64_t)(h1 - 1) * w];
        o[(int64_t)(2 * (h1 - 1)) * w] = clamp ? clampi(v, lo, hi) : v;
    }
}

// ---------------------------------------------------------------------------------------------
// horizontal unsqueeze: avg (w1 x h) + residual (w2 x h) -> out ((w1+w2) x h), w1-w2 in {0,1}.
// The row recurrence (left = previous B) is serial along x, so lanes own rows: one lane per row, HS_STEP pairs per step.
// A lane walks its own row, so one load instruction touches 64 rows and 64 cache lines; all loads of a step are issued
// before the first use, so every 128-byte line a lane opens is fetched from L2 once and finished from L1 while it is
// still there.  No LDS, no barriers.  Measured on 256 x 4K (profiles/r2_priority_and_balance.txt, r2_transforms.txt): the
// round-1 kernel (64 rows x 32 pairs staged through 33 KB of LDS, 4 wavefronts per CU, three phases between barriers)
// 87.6 ms for the whole inverse schedule; this kernel with 4 pairs per step 68.0 ms (every line re-fetched up to 8
// times); with 32 pairs per step 49.3 ms.
struct __attribute__((packed, aligned(4))) Int4U { int32_t v[4]; };
struct __attribute__((packed, aligned(2))) Short4U { int16_t v[4]; };
// four consecutive residuals (any 4-byte / 2-byte aligned address): one 16-byte or one 8-byte load
DEV Int4U load4(const int32_t *p) { return *reinterpret_cast<const Int4U *>(p); }
DEV Int4U load4(const coef_t *p) {
    const Short4U s = *reinterpret_cast<const Short4U *>(p);
    Int4U v; v.v[0] = s.v[0]; v.v[1] = s.v[1]; v.v[2] = s.v[2]; v.v[3] = s.v[3];
    return v;
}
constexpr int HS_STEP = 32;   // pairs per step = one 128-byte line of each input row, two of the output row
template <typename TR>
__global__ __launch_bounds__(256) void k_inv_hsqueeze_rows(Bases b, PlaneRef pa, PlaneRef pr, PlaneRef po, int clamp, int lo, int hi) {
    const int y = blockIdx.x * 256 + threadIdx.x;
    const int w1 = pa.w, w2 = pr.w, h = pa.h, wo = w1 + w2;
    if (y >= h) return;
    const int32_t *a = plane_ptr(b, pa, blockIdx.z) + (int64_t)y * w1;
    const TR *r = residual_ptr<TR>(b, pr, blockIdx.z) + (int64_t)y * w2;
    int32_t *o = plane_ptr(b, po, blockIdx.z) + (int64_t)y * wo;
    int avg = a[0];
    int left = avg;   // first pair: tendency(avg, avg, next), squeeze.h:89
    int x = 0;
    for (; x + HS_STEP < w1 && x + HS_STEP <= w2; x += HS_STEP) {   // avg[x+1 .. x+HS_STEP] all exist
        Int4U rv[HS_STEP / 4], nv[HS_STEP / 4];
#pragma unroll
        for (int q = 0; q < HS_STEP / 4; q++) {
            rv[q] = load4(r + x + 4 * q);
            nv[q] = *reinterpret_cast<const Int4U *>(a + x + 1 + 4 * q);
        }
#pragma unroll
        for (int q = 0; q < HS_STEP / 4; q++) {
            Int4U o0, o1;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int next_avg = nv[q].v[k];
                const int diff = rv[q].v[k] + smooth_tendency(left, avg, next_avg);
                int A, B;
                unsqueeze_pair(avg, diff, A, B);
                left = B;
                avg = next_avg;
                if (clamp) { A = clampi(A, lo, hi); B = clampi(B, lo, hi); }
                if (k < 2) { o0.v[2 * k] = A; o0.v[2 * k + 1] = B; } else { o1.v[2 * k - 4] = A; o1.v[2 * k - 3] = B; }
            }
            *reinterpret_cast<Int4U *>(o + 2 * (x + 4 * q)) = o0;
            *reinterpret_cast<Int4U *>(o + 2 * (x + 4 * q) + 4) = o1;
        }
    }
    for (; x < w2; x++) {
        const int next_avg = x + 1 < w1 ? a[x + 1] : avg;   // squeeze.h:100
        const int diff = r[x] + smooth_tendency(left, avg, next_avg);
        int A, B;
        unsqueeze_pair(avg, diff, A, B);
        o[2 * x] = clamp ? clampi(A, lo, hi) : A;
        o[2 * x + 1] = clamp ? clampi(B, lo, hi) : B;
        left = B;
        avg = next_avg;
    }
    if (wo & 1) {  // squeeze.h:129
        const int v = a[w1 - 1];
        o[wo - 1] = clamp ? clampi(v, lo, hi) : v;
    }
}

// The same recurrence with COALESCED global accesses: one wavefront (= one workgroup, so __syncthreads() is wave-local) owns 64
// rows and moves them in tiles of 64 rows x HL_P pairs through its own LDS tile.  Load: 8 lanes fetch the 128 bytes of one
// input row (16 bytes each), 8 rows per instruction; each lane then reads ITS row out of LDS, runs the HL_P pairs, puts the
// 2*HL_P outputs back into LDS (the output tile aliases the input tiles: every lane holds its inputs in registers by then)
// and the tile is written out 16 lanes per 256-byte output row.  Row pitches of 36 / 68 words keep the b128 LDS accesses of
// 16 neighbouring lanes on different banks.
#ifndef FUIF_HL_P
#define FUIF_HL_P 32
#endif
constexpr int HL_P = FUIF_HL_P;            // pairs per tile row: 32 (128-byte input segments, 18 KB of LDS) or 16 (64-byte segments, 9 KB)
constexpr int HL_IN_LANES = HL_P / 4;      // lanes that fetch one input row segment (16 bytes each)
constexpr int HL_IN_STEPS = HL_P / 4;      // load instructions per input tile: 64 rows / (64 / HL_IN_LANES) rows per instruction
constexpr int HL_OUT_LANES = HL_P / 2, HL_OUT_STEPS = HL_P / 2;
constexpr int HL_IN_PITCH = HL_P + 4;        // words
constexpr int HL_OUT_PITCH = 2 * HL_P + 4;
template <typename TR>
__global__ __launch_bounds__(64) void k_inv_hsqueeze_tiles(Bases b, PlaneRef pa, PlaneRef pr, PlaneRef po, int clamp, int lo, int hi) {
    __shared__ __attribute__((aligned(16))) int32_t tile[2 * 64 * HL_IN_PITCH];   // HL_P = 32: 18432 bytes; the 64 x 68-word output tile needs 17408
    const int lane = threadIdx.x;
    const int y0 = blockIdx.x * 64;
    const int w1 = pa.w, w2 = pr.w, h = pa.h, wo = w1 + w2;
    const int rows = min(64, h - y0);
    if (rows <= 0) return;
    const int32_t *A = plane_ptr(b, pa, blockIdx.z) + (int64_t)y0 * w1;
    const TR *R = residual_ptr<TR>(b, pr, blockIdx.z) + (int64_t)y0 * w2;
    int32_t *O = plane_ptr(b, po, blockIdx.z) + (int64_t)y0 * wo;
    const bool mine = lane < rows;
    const int32_t *a = A + (int64_t)(mine ? lane : 0) * w1;
    const TR *r = R + (int64_t)(mine ? lane : 0) * w2;
    int32_t *o = O + (int64_t)(mine ? lane : 0) * wo;
    int avg = a[0];
    int left = avg;   // first pair: tendency(avg, avg, next), squeeze.h:89
    int x = 0;
    int32_t *t_res = tile, *t_avg = tile + 64 * HL_IN_PITCH;
    // Software pipeline: the global loads of tile k+1 are issued before tile k is computed and wait in registers (the
    // occupancy is set by the LDS tile, two wavefronts per SIMD; registers are free)
    const int piece8 = lane % HL_IN_LANES, sub8 = lane / HL_IN_LANES;
    Int4U grv[HL_IN_STEPS], gnv[HL_IN_STEPS];
    auto fetch = [&](int xt) {
#pragma unroll
        for (int i = 0; i < HL_IN_STEPS; i++) {
            const int row = min((64 / HL_IN_LANES) * i + sub8, rows - 1);   // rows beyond the plane repeat its last row: no branches around the loads
            grv[i] = load4(R + (int64_t)row * w2 + xt + 4 * piece8);
            gnv[i] = *reinterpret_cast<const Int4U *>(A + (int64_t)row * w1 + xt + 1 + 4 * piece8);
        }
    };
    if (HL_P < w1 && HL_P <= w2) fetch(0);
    for (; x + HL_P < w1 && x + HL_P <= w2; x += HL_P) {   // avg[x+1 .. x+HL_P] all exist
#pragma unroll
        for (int i = 0; i < HL_IN_STEPS; i++) {
            const int row = (64 / HL_IN_LANES) * i + sub8;
            *reinterpret_cast<int4 *>(t_res + row * HL_IN_PITCH + 4 * piece8) = make_int4(grv[i].v[0], grv[i].v[1], grv[i].v[2], grv[i].v[3]);
            *reinterpret_cast<int4 *>(t_avg + row * HL_IN_PITCH + 4 * piece8) = make_int4(gnv[i].v[0], gnv[i].v[1], gnv[i].v[2], gnv[i].v[3]);
        }
        __syncthreads();
        int4 rv[HL_P / 4], nv[HL_P / 4];
#pragma unroll
        for (int q = 0; q < HL_P / 4; q++) {
            rv[q] = *reinterpret_cast<const int4 *>(t_res + lane * HL_IN_PITCH + 4 * q);
            nv[q] = *reinterpret_cast<const int4 *>(t_avg + lane * HL_IN_PITCH + 4 * q);
        }
        __syncthreads();   // every lane has its inputs: the output tile may overwrite them
        if (x + 2 * HL_P < w1 && x + 2 * HL_P <= w2) fetch(x + HL_P);
#pragma unroll
        for (int q = 0; q < HL_P / 4; q++) {
            const int rr[4] = {rv[q].x, rv[q].y, rv[q].z, rv[q].w}, nn[4] = {nv[q].x, nv[q].y, nv[q].z, nv[q].w};
            int ov[8];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int next_avg = nn[k];
                const int diff = rr[k] + smooth_tendency(left, avg, next_avg);
                int A2, B2;
                unsqueeze_pair(avg, diff, A2, B2);
                left = B2;
                avg = next_avg;
                if (clamp) { A2 = clampi(A2, lo, hi); B2 = clampi(B2, lo, hi); }
                ov[2 * k] = A2; ov[2 * k + 1] = B2;
            }
            *reinterpret_cast<int4 *>(tile + lane * HL_OUT_PITCH + 8 * q) = make_int4(ov[0], ov[1], ov[2], ov[3]);
            *reinterpret_cast<int4 *>(tile + lane * HL_OUT_PITCH + 8 * q + 4) = make_int4(ov[4], ov[5], ov[6], ov[7]);
        }
        __syncthreads();
        {
            const int piece = lane % HL_OUT_LANES, sub = lane / HL_OUT_LANES;
#pragma unroll
            for (int i = 0; i < HL_OUT_STEPS; i++) {
                const int row = (64 / HL_OUT_LANES) * i + sub;
                if (row < rows) {
                    const int4 v = *reinterpret_cast<const int4 *>(tile + row * HL_OUT_PITCH + 4 * piece);
                    Int4U u; u.v[0] = v.x; u.v[1] = v.y; u.v[2] = v.z; u.v[3] = v.w;
                    *reinterpret_cast<Int4U *>(O + (int64_t)row * wo + 2 * x + 4 * piece) = u;
                }
            }
        }
        __syncthreads();   // the next tile's loads overwrite what the stores just read
    }
    if (!mine) return;
    for (; x < w2; x++) {
        const int next_avg = x + 1 < w1 ? a[x + 1] : avg;   // squeeze.h:100
        const int diff = r[x] + smooth_tendency(left, avg, next_avg);
        int A2, B2;
        unsqueeze_pair(avg, diff, A2, B2);
        o[2 * x] = clamp ? clampi(A2, lo, hi) : A2;
        o[2 * x + 1] = clamp ? clampi(B2, lo, hi) : B2;
        left = B2;
        avg = next_avg;
    }
    if (wo & 1) {  // squeeze.h:129
        const int v = a[w1 - 1];
        o[wo - 1] = clamp ? clampi(v, lo, hi) : v;
    }
}

// OP_HSQ2_YCOCG: the horizontal unsqueeze of Co and of Cg (squeeze.h:81-132) and the inverse YCoCg (ycocg.h:49-61) in one pass.
// Same tiling as k_inv_hsqueeze_tiles with HF_P = 16 pairs per tile row: a lane runs the recurrences of ITS row for both chroma
// planes and leaves the 2 x 32 outputs in LDS; the colour transform then runs in the STORE layout (8 lanes per 128-byte row
// segment), where the Y samples are read straight from global memory and R, G, B are written straight back: the full-size Co
// and Cg planes never exist in memory.
constexpr int HF_P = 16;
constexpr int HF_IN_PITCH = HF_P + 4;
constexpr int HF_OUT_PITCH = 2 * HF_P + 4;
template <typename TR>
__global__ __launch_bounds__(64) void k_inv_hsq2_ycocg(Bases b, PlaneRef pa0, PlaneRef pr0, PlaneRef pa1, PlaneRef pr1, PlaneRef py, PlaneRef pg, PlaneRef pb,
                                                       int maxval) {
    __shared__ __attribute__((aligned(16))) int32_t tile[4 * 64 * HF_IN_PITCH];   // 20480 bytes; the two 64 x 36-word output tiles need 18432
    const int lane = threadIdx.x;
    const int y0 = blockIdx.x * 64;
    const int w1 = pa0.w, w2 = pr0.w, h = pa0.h, wo = w1 + w2;
    const int rows = min(64, h - y0);
    if (rows <= 0) return;
    const int32_t *A[2] = {plane_ptr(b, pa0, blockIdx.z) + (int64_t)y0 * w1, plane_ptr(b, pa1, blockIdx.z) + (int64_t)y0 * w1};
    const TR *R[2] = {residual_ptr<TR>(b, pr0, blockIdx.z) + (int64_t)y0 * w2, residual_ptr<TR>(b, pr1, blockIdx.z) + (int64_t)y0 * w2};
    int32_t *OY = plane_ptr(b, py, blockIdx.z) + (int64_t)y0 * py.w;
    int32_t *OG = plane_ptr(b, pg, blockIdx.z) + (int64_t)y0 * pg.w;
    int32_t *OB = plane_ptr(b, pb, blockIdx.z) + (int64_t)y0 * pb.w;
    const bool mine = lane < rows;
    const int my = mine ? lane : 0;
    int avg[2] = {A[0][(int64_t)my * w1], A[1][(int64_t)my * w1]};
    int left[2] = {avg[0], avg[1]};   // first pair: tendency(avg, avg, next), squeeze.h:89
    auto rgb = [&](int Yv, int Co, int Cg, int &Rr, int &Gg, int &Bb) {   // ycocg.h:49-61
        const int Y = clampi(Yv, 0, maxval);
        Gg = clampi(Y - ((-Cg) >> 1), 0, maxval);
        Bb = clampi(Y + ((1 - Cg) >> 1) - (Co >> 1), 0, maxval);
        Rr = clampi(Co + Bb, 0, maxval);
    };
    int x = 0;
    for (; x + HF_P < w1 && x + HF_P <= w2; x += HF_P) {   // avg[x+1 .. x+HF_P] all exist
        Int4U yv[8];
        {
            // 4 lanes per 64-byte input row segment, 16 rows per instruction; Y: 8 lanes per 128-byte segment, 8 rows per instruction
            const int piece = lane & 3, sub = lane >> 2;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = 16 * i + sub, src = min(row, rows - 1);   // rows beyond the plane repeat its last row: no branches around the loads
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const Int4U rv = load4(R[c] + (int64_t)src * w2 + x + 4 * piece);
                    const Int4U nv = *reinterpret_cast<const Int4U *>(A[c] + (int64_t)src * w1 + x + 1 + 4 * piece);
                    *reinterpret_cast<int4 *>(tile + (2 * c) * 64 * HF_IN_PITCH + row * HF_IN_PITCH + 4 * piece) = make_int4(rv.v[0], rv.v[1], rv.v[2], rv.v[3]);
                    *reinterpret_cast<int4 *>(tile + (2 * c + 1) * 64 * HF_IN_PITCH + row * HF_IN_PITCH + 4 * piece) = make_int4(nv.v[0], nv.v[1], nv.v[2], nv.v[3]);
                }
            }
            const int ypiece = lane & 7, ysub = lane >> 3;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int row = min(8 * i + ysub, rows - 1);
                yv[i] = *reinterpret_cast<const Int4U *>(OY + (int64_t)row * py.w + 2 * x + 4 * ypiece);
            }
        }
        __syncthreads();
        int4 rv[2][HF_P / 4], nv[2][HF_P / 4];
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int q = 0; q < HF_P / 4; q++) {
                rv[c][q] = *reinterpret_cast<const int4 *>(tile + (2 * c) * 64 * HF_IN_PITCH + lane * HF_IN_PITCH + 4 * q);
                nv[c][q] = *reinterpret_cast<const int4 *>(tile + (2 * c + 1) * 64 * HF_IN_PITCH + lane * HF_IN_PITCH + 4 * q);
            }
        __syncthreads();   // every lane has its inputs: the output tiles may overwrite them
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int q = 0; q < HF_P / 4; q++) {
                const int rr[4] = {rv[c][q].x, rv[c][q].y, rv[c][q].z, rv[c][q].w}, nn[4] = {nv[c][q].x, nv[c][q].y, nv[c][q].z, nv[c][q].w};
                int ov[8];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int next_avg = nn[k];
                    const int diff = rr[k] + smooth_tendency(left[c], avg[c], next_avg);
                    int A2, B2;
                    unsqueeze_pair(avg[c], diff, A2, B2);
                    left[c] = B2;
                    avg[c] = next_avg;
                    ov[2 * k] = A2; ov[2 * k + 1] = B2;
                }
                *reinterpret_cast<int4 *>(tile + c * 64 * HF_OUT_PITCH + lane * HF_OUT_PITCH + 8 * q) = make_int4(ov[0], ov[1], ov[2], ov[3]);
                *reinterpret_cast<int4 *>(tile + c * 64 * HF_OUT_PITCH + lane * HF_OUT_PITCH + 8 * q + 4) = make_int4(ov[4], ov[5], ov[6], ov[7]);
            }
        __syncthreads();
        {
            const int piece = lane & 7, sub = lane >> 3;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int row = 8 * i + sub;
                if (row < rows) {
                    const int4 co = *reinterpret_cast<const int4 *>(tile + row * HF_OUT_PITCH + 4 * piece);
                    const int4 cg = *reinterpret_cast<const int4 *>(tile + 64 * HF_OUT_PITCH + row * HF_OUT_PITCH + 4 * piece);
                    const int cov[4] = {co.x, co.y, co.z, co.w}, cgv[4] = {cg.x, cg.y, cg.z, cg.w};
                    Int4U r4, g4, b4;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        int Rr, Gg, Bb;
                        rgb(yv[i].v[k], cov[k], cgv[k], Rr, Gg, Bb);
                        r4.v[k] = Rr; g4.v[k] = Gg; b4.v[k] = Bb;
                    }
                    *reinterpret_cast<Int4U *>(OY + (int64_t)row * py.w + 2 * x + 4 * piece) = r4;
                    *reinterpret_cast<Int4U *>(OG + (int64_t)row * pg.w + 2 * x + 4 * piece) = g4;
                    *reinterpret_cast<Int4U *>(OB + (int64_t)row * pb.w + 2 * x + 4 * piece) = b4;
                }
            }
        }
        __syncthreads();   // the next tile's loads overwrite what the stores just read
    }
    if (!mine) return;
    const int32_t *a0 = A[0] + (int64_t)lane * w1, *a1 = A[1] + (int64_t)lane * w1;
    const TR *r0 = R[0] + (int64_t)lane * w2, *r1 = R[1] + (int64_t)lane * w2;
    int32_t *oy = OY + (int64_t)lane * py.w, *og = OG + (int64_t)lane * pg.w, *ob = OB + (int64_t)lane * pb.w;
    for (; x < w2; x++) {
        int P[2][2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int32_t *a = c ? a1 : a0;
            const TR *r = c ? r1 : r0;
            const int next_avg = x + 1 < w1 ? a[x + 1] : avg[c];   // squeeze.h:100
            const int diff = r[x] + smooth_tendency(left[c], avg[c], next_avg);
            unsqueeze_pair(avg[c], diff, P[c][0], P[c][1]);
            left[c] = P[c][1];
            avg[c] = next_avg;
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            int Rr, Gg, Bb;
            rgb(oy[2 * x + k], P[0][k], P[1][k], Rr, Gg, Bb);
            oy[2 * x + k] = Rr; og[2 * x + k] = Gg; ob[2 * x + k] = Bb;
        }
    }
    if (wo & 1) {  // squeeze.h:129
        int Rr, Gg, Bb;
        rgb(oy[wo - 1], a0[w1 - 1], a1[w1 - 1], Rr, Gg, Bb);
        oy[wo - 1] = Rr; og[wo - 1] = Gg; ob[wo - 1] = Bb;
    }
}

// ---------------------------------------------------------------------------------------------
// transform/ycocg.h:49-61, in place on three planes (own row pitches), region w x h
__global__ __launch_bounds__(256) void k_inv_ycocg(Bases b, PlaneRef p0, PlaneRef p1, PlaneRef p2, int w, int h, int maxval) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w || y >= h) return;
    int32_t *c0 = plane_ptr(b, p0, blockIdx.z) + (int64_t)y * p0.w + x;
    int32_t *c1 = plane_ptr(b, p1, blockIdx.z) + (int64_t)y * p1.w + x;
    int32_t *c2 = plane_ptr(b, p2, blockIdx.z) + (int64_t)y * p2.w + x;
    const int Y = clampi(*c0, 0, maxval);
    const int Co = *c1, Cg = *c2;
    const int G = clampi(Y - ((-Cg) >> 1), 0, maxval);
    const int B = clampi(Y + ((1 - Cg) >> 1) - (Co >> 1), 0, maxval);
    const int R = clampi(Co + B, 0, maxval);
    *c0 = R; *c1 = G; *c2 = B;
}

// transform/ycbcr.h:49-60: `float` operands, double arithmetic left to right, no contraction,
// CLAMP in double, truncating conversion to the integer sample.
__global__ __launch_bounds__(256) void k_inv_ycbcr(Bases b, PlaneRef p0, PlaneRef p1, PlaneRef p2, int w, int h, int minval, int maxval) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w || y >= h) return;
    int32_t *c0 = plane_ptr(b, p0, blockIdx.z) + (int64_t)y * p0.w + x;
    int32_t *c1 = plane_ptr(b, p1, blockIdx.z) + (int64_t)y * p1.w + x;
    int32_t *c2 = plane_ptr(b, p2, blockIdx.z) + (int64_t)y * p2.w + x;
    const float half = (float)((maxval + 1) / 2);
    const float yy = (float)*c0;
    const float cb = __fsub_rn((float)*c1, half);
    const float cr = __fsub_rn((float)*c2, half);
    const double dy = (double)yy, dcb = (double)cb, dcr = (double)cr;
    double r = __dadd_rn(__dadd_rn(dy, __dmul_rn(1.402, dcr)), 0.5);
    double g = __dadd_rn(__dsub_rn(__dsub_rn(dy, __dmul_rn(0.344136, dcb)), __dmul_rn(0.714136, dcr)), 0.5);
    double bl = __dadd_rn(__dadd_rn(dy, __dmul_rn(1.772, dcb)), 0.5);
    const double mn = (double)minval, mx = (double)maxval;
    r = r < mn ? mn : (r > mx ? mx : r);
    g = g < mn ? mn : (g > mx ? mx : g);
    bl = bl < mn ? mn : (bl > mx ? mx : bl);
    *c0 = (int)r; *c1 = (int)g; *c2 = (int)bl;
}

// ---------------------------------------------------------------------------------------------
// transform/quantize.h:32-49: plane *= q, q = ChannelMeta::q of the plane's source channel.
// grid.y walks the op's plane list.
// TR = coef_t (Op::r16): the planes are coded planes nobody has touched -- read the int16 samples from the coefficient slab and write
// sample * q into the int32 copy (widening and scaling in one pass, also when q == 1); TR = int32_t: in place on the int32 copy.
template <typename TR>
__global__ __launch_bounds__(256) void k_dequant(Bases b, const PlaneRef *list, const ChannelMeta *meta, int n_channels, int img_first) {
    const PlaneRef p = list[blockIdx.y];
    const int q = meta[(int64_t)(img_first + blockIdx.z) * n_channels + p.qsrc].q;
    constexpr bool kFrom16 = !std::is_same<TR, int32_t>::value;
    if (q == 1 && !kFrom16) return;
    const int64_t n = (int64_t)p.w * p.h;
    int32_t *d = plane_ptr(b, p, blockIdx.z);
    const TR *s = residual_ptr<TR>(b, p, blockIdx.z);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = (int)s[i] * q;
}

__global__ __launch_bounds__(256) void k_clamp(Bases b, PlaneRef src, PlaneRef dst, int lo, int hi) {
    const int64_t n = (int64_t)dst.w * dst.h;
    const int32_t *s = plane_ptr(b, src, blockIdx.z);
    int32_t *d = plane_ptr(b, dst, blockIdx.z);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = clampi(s[i], lo, hi);
}

// transform/palette.h:57-64: out(y,x) = palette(component, CLAMP(index(y,x), 0, colours-1)).  The palette is a
// decoded meta-channel (colours x components samples); rows of it stay in L1/L2.
__global__ __launch_bounds__(256) void k_inv_palette(Bases b, PlaneRef pidx, PlaneRef ppal, PlaneRef po, int component, int colours,
                                                     int clamp, int lo, int hi) {
    const int64_t n = (int64_t)po.w * po.h;
    const int32_t *idx = plane_ptr(b, pidx, blockIdx.z);
    const int32_t *pal = plane_ptr(b, ppal, blockIdx.z) + (int64_t)component * ppal.w;
    int32_t *d = plane_ptr(b, po, blockIdx.z);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = colours > 0 ? pal[clampi(idx[i], 0, colours - 1)] : 0;   // an empty palette reads Channel::zero (image.h:82)
        d[i] = clamp ? clampi(v, lo, hi) : v;
    }
}

// transform/permute.h:31-54 with the permutation in a meta-channel: output plane i is a copy of candidate plane perm[i],
// perm = the decoded samples of the 1-row meta plane -- per image.  A value that is no channel number (the reference would
// index outside its channel vector) flags the image corrupt and ZERO-FILLS the output plane, so that a flagged image is
// deterministic and shows nothing of an earlier batch's slab.  A value that repeats an earlier one is refused as well:
// stricter than inv_permute (:39-46 copies blindly, twice) but the same verdict as the decode-time metadata permutation
// of encoding.cpp:576-596 / meta_permute (:68-73) and as oracle/fuif_oracle.c gives.
__global__ __launch_bounds__(256) void k_permute_plane(Bases b, PlaneRef pperm, const PlaneRef *cand, int nb, int which, PlaneRef po, int clamp, int lo,
                                                       int hi, int32_t *status, int img_first) {
    const int c = plane_ptr(b, pperm, blockIdx.z)[which];
    bool bad = c < 0 || c >= nb;
    for (int j = 0; j < which && !bad; j++) bad = plane_ptr(b, pperm, blockIdx.z)[j] == c;
    int32_t *d = plane_ptr(b, po, blockIdx.z);
    const int64_t n = (int64_t)po.w * po.h;
    if (bad) {
        if (status && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&status[img_first + blockIdx.z], (int32_t)ST_CORRUPT);
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = 0;
        return;
    }
    const int32_t *s = plane_ptr(b, cand[c], blockIdx.z);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = clamp ? clampi(s[i], lo, hi) : s[i];
}

// transform/approximate.h:44-57: ch = ch*q + remainder.  A remainder channel the stream never reached has no
// samples in the reference (`chr.data.size() == 0`, :49,54): then nothing is added and Channel::q keeps its
// value; otherwise Channel::q is taken from the remainder (:50).  `decoded` is per image, so the q hand-over is
// done here on the ChannelMeta the later k_dequant reads.
__global__ __launch_bounds__(256) void k_inv_approx(Bases b, PlaneRef pc, PlaneRef pr, int q, int ctor_data, ChannelMeta *meta, int n_channels,
                                                    int img_first) {
    ChannelMeta *m = meta ? meta + (int64_t)(img_first + blockIdx.z) * n_channels : nullptr;
    const bool reached = m && pr.qsrc >= 0 && m[pr.qsrc].decoded != 0;
    const bool have = reached || ctor_data;
    const int64_t n = (int64_t)pc.w * pc.h;
    int32_t *d = plane_ptr(b, pc, blockIdx.z);
    const int32_t *r = plane_ptr(b, pr, blockIdx.z);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        d[i] = d[i] * q + (have ? r[i] : 0);
    if (have && m && blockIdx.x == 0 && threadIdx.x == 0 && pc.qsrc >= 0 && pr.qsrc >= 0) m[pc.qsrc].q = reached ? m[pr.qsrc].q : 1;
}

// transform/2dmatch.h:147-171, the previous-frame mode (match channel q == 2*fh*fh + (fh&1), :147-149): frames are
// stacked vertically, z = m(y,x) != 0 means "equal to (soft: add) the sample z frames up".  One lane per column,
// serial down the rows (a source may itself be a matched sample of an earlier frame).  Channel::value() only
// checks the linear index (image.h:82-85): a source before the first sample reads Channel::zero.
// The free-offset mode (q == 1) is handled by k_match_init / k_match_jump / k_match_apply below.
__global__ __launch_bounds__(256) void k_inv_match_frames(Bases b, PlaneRef pm, const PlaneRef *list, int n_list, int softmatch, int fh,
                                                          const ChannelMeta *meta, int n_channels, int img_first, int32_t *status) {
    const int z_img = blockIdx.z;
    const int q = (meta && pm.qsrc >= 0) ? meta[(int64_t)(img_first + z_img) * n_channels + pm.qsrc].q : 0;
    if (q != 2 * fh * fh + (fh & 1)) {
        // q == 1 is the free-offset mode (next kernels); anything else the reference refuses (2dmatch.h:172-175)
        if (q != 1 && status && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&status[img_first + z_img], ST_UNSUPPORTED | ST_CORRUPT);
        return;
    }
    const int w = pm.w, h = pm.h;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= w) return;
    const int32_t *m = plane_ptr(b, pm, z_img);
    for (int y = 0; y < h; y++) {
        const int zv = m[(int64_t)y * w + x];
        if (!zv) continue;
        const int64_t src = ((int64_t)y - (int64_t)zv * fh) * w + x;
        const bool inside = src >= 0 && src < (int64_t)w * h;
        for (int k = 0; k < n_list; k++) {
            int32_t *p = plane_ptr(b, list[k], z_img);
            const int sv = inside ? p[src] : 0;
            p[(int64_t)y * w + x] = softmatch ? p[(int64_t)y * w + x] + sv : sv;
        }
    }
}

// ---- 2D match with free offsets (transform/2dmatch.h:136-146, match channel q == 1) --------------------------
// Reference semantics: in scan order, z = m(y,x) != 0 means value(y,x) = value(y+oy, x+ox) with (ox,oy) = the z-th
// position of a spiral through the already decoded neighbourhood (:33-78) -- the source may itself be a copy.
// Channel::value() only checks the LINEAR index (image.h:82-85), so the source is sample p + oy*w + ox, and a
// source before the first sample reads Channel::zero.  Parallel form: S[p] = p for unmatched samples, else the
// source index; ceil(log2(n)) rounds of S[p] = S[S[p]] make every S[p] a root; one gather finishes the job.
// Soft matches (value += source, :136-140; the CLI never writes them, fuif.cpp:445) take the same route with one accumulator
// per listed plane: value(p) = residual(p) + value(source(p)), so A[p] = residual(p) for a matched sample and 0 for a root,
// every doubling step adds A[S[p]] to A[p] (two copies of the accumulators, read one / write the other), and the gather adds
// the root's sample.  Forward references (only possible in images narrower than the spiral) are flagged
// FUIFGPU_ST_UNSUPPORTED and the image is left unmatched.
// The list of a soft match is [planes (n), accumulators (n), accumulators (n)]; Op::p1 of a jump / apply op = which copy it reads.
DEV void match_offset(int code, int &xo, int &yo) {   // 2dmatch.h:50-78
    int layer = 0, size = 4;
    while (code > size) { code -= size; layer++; size += 4; }
    if (layer & 1) {
        if (code <= layer) { xo = 1 + layer; yo = -code; }
        else if (code <= 3 + 3 * layer) { xo = 2 + 2 * layer - code; yo = -1 - layer; }
        else { xo = -1 - layer; yo = -4 - 4 * layer + code; }
    } else {
        if (code <= 1 + layer) { xo = -1 - layer; yo = 1 - code; }
        else if (code <= 4 + 3 * layer) { xo = -3 - 2 * layer + code; yo = -1 - layer; }
        else { xo = 1 + layer; yo = -5 - 4 * layer + code; }
    }
}
DEV bool match_free_mode(const PlaneRef &pm, const ChannelMeta *meta, int n_channels, int img) {
    return meta && pm.qsrc >= 0 && meta[(int64_t)img * n_channels + pm.qsrc].q == 1;
}
__global__ __launch_bounds__(256) void k_match_init(Bases b, PlaneRef pm, PlaneRef ps, int softmatch, const PlaneRef *list, int n_list,
                                                    const ChannelMeta *meta, int n_channels, int img_first, int32_t *status) {
    const int img = img_first + blockIdx.z;
    if (!match_free_mode(pm, meta, n_channels, img)) return;
    const int maxz = meta[(int64_t)img * n_channels + pm.qsrc].maxval;
    const int w = pm.w;
    const int64_t n = (int64_t)pm.w * pm.h;
    const int32_t *m = plane_ptr(b, pm, blockIdx.z);
    int32_t *S = plane_ptr(b, ps, blockIdx.z);
    int flag = 0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const int z = m[p];
        int64_t s = p;
        if (z) {
            if (z < 0 || z > maxz) flag |= ST_UNSUPPORTED | ST_CORRUPT;   // offsets_table[z] out of range in the reference
            else {
                int xo, yo;
                match_offset(z, xo, yo);
                const int64_t src = p + (int64_t)yo * w + xo;
                if (src < 0 || src >= n) s = -1;
                else if (src > p) flag |= ST_UNSUPPORTED;                        // would read a sample that is rewritten later
                else s = src;
            }
        }
        S[p] = (int32_t)s;
        if (softmatch)
            for (int k = 0; k < n_list; k++) plane_ptr(b, list[n_list + k], blockIdx.z)[p] = s != p ? plane_ptr(b, list[k], blockIdx.z)[p] : 0;
    }
    if (flag && status) atomicOr(&status[img], flag);
}
__global__ __launch_bounds__(256) void k_match_jump(Bases b, PlaneRef pm, PlaneRef pin, PlaneRef pout, int softmatch, const PlaneRef *list, int n_list,
                                                    int from, const ChannelMeta *meta, int n_channels, int img_first) {
    if (!match_free_mode(pm, meta, n_channels, img_first + blockIdx.z)) return;
    const int64_t n = (int64_t)pin.w * pin.h;
    const int32_t *S = plane_ptr(b, pin, blockIdx.z);
    int32_t *T = plane_ptr(b, pout, blockIdx.z);
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const int s = S[p];
        T[p] = s >= 0 ? S[s] : s;
        if (softmatch)
            for (int k = 0; k < n_list; k++) {
                const int32_t *a = plane_ptr(b, list[n_list * (1 + from) + k], blockIdx.z);
                plane_ptr(b, list[n_list * (2 - from) + k], blockIdx.z)[p] = a[p] + (s >= 0 ? a[s] : 0);
            }
    }
}
__global__ __launch_bounds__(256) void k_match_apply(Bases b, PlaneRef pm, PlaneRef ps, int softmatch, const PlaneRef *list, int n_list, int from,
                                                     const ChannelMeta *meta, int n_channels, int img_first, const int32_t *status) {
    const int img = img_first + blockIdx.z;
    if (!match_free_mode(pm, meta, n_channels, img)) return;
    if (status && (status[img] & ST_UNSUPPORTED)) return;   // flagged by k_match_init: leave the planes alone
    const int64_t n = (int64_t)ps.w * ps.h;
    const int32_t *S = plane_ptr(b, ps, blockIdx.z);
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const int s = S[p];
        if (s == p) continue;                                // roots are never written: the gather below only reads roots
        for (int k = 0; k < n_list; k++) {
            int32_t *pl = plane_ptr(b, list[k], blockIdx.z);
            const int root = s < 0 ? 0 : pl[s];
            pl[p] = softmatch ? plane_ptr(b, list[n_list * (1 + from) + k], blockIdx.z)[p] + root : root;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// transform/dct.h:60-77 -- the constants exactly as the reference prints them
constexpr double kDCT[64] = {
    0.3535533906, 0.3535533906, 0.3535533906, 0.3535533906, 0.3535533906, 0.3535533906, 0.3535533906, 0.3535533906,
    0.4903926402, 0.4157348062, 0.2777851165, 0.0975451610, -0.0975451610, -0.2777851165, -0.4157348062, -0.4903926402,
    0.4619397663, 0.1913417162, -0.1913417162, -0.4619397663, -0.4619397663, -0.1913417162, 0.1913417162, 0.4619397663,
    0.4157348062, -0.0975451610, -0.4903926402, -0.2777851165, 0.2777851165, 0.4903926402, 0.0975451610, -0.4157348062,
    0.3535533906, -0.3535533906, -0.3535533906, 0.3535533906, 0.3535533906, -0.3535533906, -0.3535533906, 0.3535533906,
    0.2777851165, -0.4903926402, 0.0975451610, 0.4157348062, -0.4157348062, -0.0975451610, 0.4903926402, -0.2777851165,
    0.1913417162, -0.4619397663, 0.4619397663, -0.1913417162, -0.1913417162, 0.4619397663, -0.4619397663, 0.1913417162,
    0.0975451610, -0.2777851165, 0.4157348062, -0.4903926402, 0.4903926402, -0.4157348062, 0.2777851165, -0.0975451610,
};

// One lane per 8x8 block.  Column pass then row pass (dct.h:101-106); every output is
// 0.0 + sum_{u=0..7} k[8u+o]*in[u] accumulated left to right in double with separate mul and add
// (the x86-64 reference build has no FMA), DC gets the float DC offset in float arithmetic
// (dct.h:281,285), result rounded half away from zero (dct.h:289).
//
// The 64 constants are 7 magnitudes with signs, and k*x and (-k)*x are the same product with the other sign, exactly.  So one
// 8-point pass needs 22 multiplications, not 64 (row u of the matrix holds 1, 4, 2, 4, 1, 4, 2, 4 different magnitudes), and
// 56 additions / subtractions in the reference's order: the leading `0.0 +` is dropped -- it only matters for a first product
// of -0.0, which cannot occur (row 0 of the matrix is positive and no input is -0.0: the coefficients are integers and a sum
// that starts from +0.0 never yields -0.0).  Seven constants live in 14 SGPRs (round 2 kept all 64 in SGPRs: 252 spilled
// lanes, two v_readlane per multiplication).
constexpr double kDctMag[7] = {0.3535533906, 0.4903926402, 0.4157348062, 0.2777851165, 0.0975451610, 0.4619397663, 0.1913417162};
// sign * (magnitude index + 1) of kDCT[8u + o]
constexpr signed char kDctCode[64] = {
    1, 1, 1, 1, 1, 1, 1, 1,
    2, 3, 4, 5, -5, -4, -3, -2,
    6, 7, -7, -6, -6, -7, 7, 6,
    3, -5, -2, -4, 4, 2, 5, -3,
    1, -1, -1, 1, 1, -1, -1, 1,
    4, -2, 5, 3, -3, -5, 2, -4,
    7, -6, 6, -7, -7, 6, -6, 7,
    5, -4, 3, -2, 2, -3, 4, -5,
};
constexpr bool dct_code_matches_matrix() {
    for (int i = 0; i < 64; i++) {
        const int c = kDctCode[i];
        const double v = c > 0 ? kDctMag[c - 1] : -kDctMag[-c - 1];
        if (v != kDCT[i]) return false;
    }
    return true;
}
static_assert(dct_code_matches_matrix(), "kDctCode / kDctMag must spell out transform/dct.h:60-77");
DEV void idct_1d(const double (&in)[8], double (&out)[8]) {
    double pr[8][7];
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
        for (int m = 0; m < 7; m++) pr[u][m] = __dmul_rn(kDctMag[m], in[u]);   // the products no output uses are dead code
#pragma unroll
    for (int o = 0; o < 8; o++) {
        double acc = pr[0][0];
#pragma unroll
        for (int u = 1; u < 8; u++) {
            const int c = kDctCode[8 * u + o];
            acc = c > 0 ? __dadd_rn(acc, pr[u][c - 1]) : __dadd_rn(acc, -pr[u][-c - 1]);
        }
        out[o] = acc;
    }
}
// A source plane of kind BUF_COEF16Q (planner peephole fuse_dequant_into_idct) is a coded plane as the entropy kernel stored it: the int16 sample is
// read from the coefficient slab and multiplied by the channel's quantisation constant on the way in (quantize.h:32-49 folded into the load).
// kAc16 (Op::pad2): ALL 63 AC planes are of that kind -- known at compile time, so the 8 loads of a column are issued together as before; only the DC plane
// (entry 0: a product of the unsqueeze chain in a default stream) is tested at run time.  With a per-plane test inside the loop every load waited for
// its own round trip and the kernel's time doubled (23 -> 45 ms per C3 step, profiles/r5_c3_kernel_stats_scalar_q_loads.csv).
template <bool kAc16>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_idct8x8(Bases b, const PlaneRef *list, PlaneRef po, int bw, int bh, int maxval, int clamp, int lo, int hi,
                                                const ChannelMeta *meta, int n_channels, int img_first) {
    const int bx = blockIdx.x * blockDim.x + threadIdx.x;
    const int by = blockIdx.y;
    // The quantisation constants of the 64 source planes, one per lane, parked in LDS: the column loop below reads them with constant offsets
    // (no dependent scalar loads -- plane descriptor, then q -- inside the loop).
    __shared__ int qs[64];
    {
        const PlaneRef pl = list[threadIdx.x];
        qs[threadIdx.x] = (pl.buf == BUF_COEF16Q && meta) ? meta[(int64_t)(img_first + blockIdx.z) * n_channels + pl.qsrc].q : 1;
    }
    __syncthreads();
    if (bx >= bw || by >= bh) return;
    const float dcoff = (float)(((double)maxval + 1.0) * 4.0);
    double tmp[64];   // [output row o][column x] after the column pass
#pragma unroll
    for (int x = 0; x < 8; x++) {
        double col[8], res[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const PlaneRef p = list[u * 8 + x];
            int v;
            if ((u != 0 || x != 0) ? kAc16 : p.buf == BUF_COEF16Q)
                v = (int)(b.c16 + (int64_t)blockIdx.z * b.stride[BUF_COEF] + p.off)[(int64_t)by * p.w + bx] * qs[u * 8 + x];
            else
                v = plane_ptr(b, p, blockIdx.z)[(int64_t)by * p.w + bx];
            col[u] = (u == 0 && x == 0) ? (double)__fadd_rn((float)v, dcoff) : (double)v;
        }
        idct_1d(col, res);
#pragma unroll
        for (int o = 0; o < 8; o++) tmp[o * 8 + x] = res[o];
#ifndef FUIF_EMU
        // one column's plane descriptors at a time: with all 64 scalar descriptor loads hoisted to the top the kernel spilled 88 SGPRs; the
        // barrier costs nothing (profiles/r4_idct_variants.txt: 0.554 ms with and without on a 1920 x 2160-block component)
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
    int32_t *o = plane_ptr(b, po, blockIdx.z) + (int64_t)(by * 8) * po.w + bx * 8;
#pragma unroll
    for (int y = 0; y < 8; y++) {
        double row[8], res[8];
#pragma unroll
        for (int u = 0; u < 8; u++) row[u] = tmp[8 * y + u];
        idct_1d(row, res);
        int outv[8];
#pragma unroll
        for (int oo = 0; oo < 8; oo++) {
            int v = (int)round(res[oo]);
            outv[oo] = clamp ? clampi(v, lo, hi) : v;
        }
        int4 *dst = reinterpret_cast<int4 *>(o + (int64_t)y * po.w);
        dst[0] = make_int4(outv[0], outv[1], outv[2], outv[3]);
        dst[1] = make_int4(outv[4], outv[5], outv[6], outv[7]);
    }
}

// ---------------------------------------------------------------------------------------------
// transform/subsample.h:90-115: horizontal pass (3a+b+1)>>2 / (3a+c+2)>>2 with edge replication,
// then the same vertically on the horizontally upsampled rows.  One lane per output sample.
__global__ __launch_bounds__(256) void k_upsample(Bases b, PlaneRef pi, PlaneRef po, int srh, int srv, int clamp, int lo, int hi) {
    const int X = blockIdx.x * blockDim.x + threadIdx.x;
    const int Y = blockIdx.y;
    const int ow = pi.w, oh = pi.h;
    if (X >= po.w || Y >= po.h) return;
    const int32_t *in = plane_ptr(b, pi, blockIdx.z);
    auto hval = [&](int y, int Xo) -> int {  // value of the horizontally upsampled row y at column Xo
        if (srh == 2) {
            const int x = Xo >> 1;
            const int c = in[(int64_t)y * ow + x];
            if (Xo & 1) return (3 * c + in[(int64_t)y * ow + (x + 1 < ow ? x + 1 : x)] + 2) >> 2;
            return (3 * c + in[(int64_t)y * ow + (x ? x - 1 : 0)] + 1) >> 2;
        }
        return in[(int64_t)y * ow + Xo];
    };
    int v;
    if (srh > 2 || srv > 2) {
        v = in[(int64_t)(Y / srv) * ow + X / srh];   // subsample.h:116-126: plain replication for ratios above 2 (4:1:1)
    } else if (srv == 2) {
        const int y = Y >> 1;
        const int c = hval(y, X);
        if (Y & 1) v = (3 * c + hval(y + 1 < oh ? y + 1 : y, X) + 2) >> 2;
        else v = (3 * c + hval(y ? y - 1 : 0, X) + 1) >> 2;
    } else {
        v = hval(Y, X);
    }
    plane_ptr(b, po, blockIdx.z)[(int64_t)Y * po.w + X] = clamp ? clampi(v, lo, hi) : v;
}

// The 4:2:0 case (both factors 2) with one lane per INPUT sample, which owns the 2x2 outputs above it: 9 loads (its 3x3
// neighbourhood, shared with the neighbouring lanes through L1) for 4 samples instead of 4 loads for 1, a quarter of the
// lanes and blocks, 8-byte stores.  Same arithmetic as k_upsample (subsample.h:90-115).  C3, 1024 x 4K: k_upsample was the
// slowest inverse transform of the JPEG-transcode schedule (61.8 ms per step, 1.4 TB/s; profiles/r2_c3_kernel_stats.csv).
struct __attribute__((packed, aligned(4))) Int2U { int32_t v[2]; };
__global__ __launch_bounds__(256) void k_upsample_2x2(Bases b, PlaneRef pi, PlaneRef po, int clamp, int lo, int hi) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int ow = pi.w, oh = pi.h;
    if (x >= ow || 2 * x >= po.w || 2 * y >= po.h) return;
    const int32_t *in = plane_ptr(b, pi, blockIdx.z);
    const int xm = x ? x - 1 : 0, xp = x + 1 < ow ? x + 1 : x;
    const int ym = y ? y - 1 : 0, yp = y + 1 < oh ? y + 1 : y;
    // the horizontally upsampled row yy at columns 2x (even) and 2x + 1 (odd)
    auto hrow = [&](int yy, int &he, int &ho) {
        const int32_t *r = in + (int64_t)yy * ow;
        const int c = r[x];
        he = (3 * c + r[xm] + 1) >> 2;
        ho = (3 * c + r[xp] + 2) >> 2;
    };
    int me, mo, ce, co, pe, po2;
    hrow(ym, me, mo); hrow(y, ce, co); hrow(yp, pe, po2);
    int v00 = (3 * ce + me + 1) >> 2, v01 = (3 * co + mo + 1) >> 2;     // row 2y: with the row above
    int v10 = (3 * ce + pe + 2) >> 2, v11 = (3 * co + po2 + 2) >> 2;    // row 2y + 1: with the row below
    if (clamp) { v00 = clampi(v00, lo, hi); v01 = clampi(v01, lo, hi); v10 = clampi(v10, lo, hi); v11 = clampi(v11, lo, hi); }
    int32_t *o = plane_ptr(b, po, blockIdx.z) + (int64_t)(2 * y) * po.w + 2 * x;
    const bool two_cols = 2 * x + 1 < po.w, two_rows = 2 * y + 1 < po.h;
    if (two_cols) {
        Int2U a; a.v[0] = v00; a.v[1] = v01;
        *reinterpret_cast<Int2U *>(o) = a;
        if (two_rows) { Int2U c2; c2.v[0] = v10; c2.v[1] = v11; *reinterpret_cast<Int2U *>(o + po.w) = c2; }
    } else {
        o[0] = v00;
        if (two_rows) o[po.w] = v10;
    }
}

// ---------------------------------------------------------------------------------------------
// OP_UPS2_YCBCR: 4:2:0 chroma upsampling (k_upsample_2x2's arithmetic, subsample.h:90-115) + inverse YCbCr (k_inv_ycbcr's arithmetic, ycbcr.h:49-60) in
// one pass.  One lane per chroma INPUT sample owning its 2x2 outputs: 9 + 9 L1-shared chroma loads and two 8-byte loads of Y for 4 pixels, six 8-byte
// stores; the full-size Cb / Cr planes are never written and read back.  Output samples of the chroma planes outside the colour transform's w x h
// region (block padding) get the upsampled value, clamped to [lo, hi] like the final clamp of image.cpp:107-113 leaves them.
__global__ __launch_bounds__(256) void k_ups2_ycbcr(Bases b, PlaneRef py, PlaneRef pcb, PlaneRef pcr, PlaneRef o0, PlaneRef o1, PlaneRef o2, int w, int h, int lo, int hi) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int cw = pcb.w, ch = pcb.h;
    if (x >= cw) return;
    const int32_t *cb_in = plane_ptr(b, pcb, blockIdx.z), *cr_in = plane_ptr(b, pcr, blockIdx.z);
    const int xm = x ? x - 1 : 0, xp = x + 1 < cw ? x + 1 : x;
    const int ym = y ? y - 1 : 0, yp = y + 1 < ch ? y + 1 : y;
    int cbv[4], crv[4];     // [row][column] of the 2x2 outputs
    auto up = [&](const int32_t *in, int (&v)[4]) {
        auto hrow = [&](int yy, int &he, int &ho) {
            const int32_t *r = in + (int64_t)yy * cw;
            const int c = r[x];
            he = (3 * c + r[xm] + 1) >> 2;
            ho = (3 * c + r[xp] + 2) >> 2;
        };
        int me, mo, ce, co, pe, po2;
        hrow(ym, me, mo); hrow(y, ce, co); hrow(yp, pe, po2);
        v[0] = (3 * ce + me + 1) >> 2; v[1] = (3 * co + mo + 1) >> 2;      // row 2y: with the row above
        v[2] = (3 * ce + pe + 2) >> 2; v[3] = (3 * co + po2 + 2) >> 2;     // row 2y + 1: with the row below
    };
    up(cb_in, cbv); up(cr_in, crv);
    const float half = (float)((hi + 1) / 2);
    const double mn = (double)lo, mx = (double)hi;
    int32_t *Yp = plane_ptr(b, py, blockIdx.z), *O0 = plane_ptr(b, o0, blockIdx.z), *O1 = plane_ptr(b, o1, blockIdx.z), *O2 = plane_ptr(b, o2, blockIdx.z);
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int Y = 2 * y + r;
        int out0[2], out1[2], out2[2];
        bool in_region[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int X = 2 * x + c;
            in_region[c] = X < w && Y < h;
            const int cb_i = cbv[2 * r + c], cr_i = crv[2 * r + c];
            if (in_region[c]) {
                const float yy = (float)Yp[(int64_t)Y * py.w + X];
                const float cb = __fsub_rn((float)cb_i, half);
                const float cr = __fsub_rn((float)cr_i, half);
                const double dy = (double)yy, dcb = (double)cb, dcr = (double)cr;
                double rr = __dadd_rn(__dadd_rn(dy, __dmul_rn(1.402, dcr)), 0.5);
                double g = __dadd_rn(__dsub_rn(__dsub_rn(dy, __dmul_rn(0.344136, dcb)), __dmul_rn(0.714136, dcr)), 0.5);
                double bl = __dadd_rn(__dadd_rn(dy, __dmul_rn(1.772, dcb)), 0.5);
                rr = rr < mn ? mn : (rr > mx ? mx : rr);
                g = g < mn ? mn : (g > mx ? mx : g);
                bl = bl < mn ? mn : (bl > mx ? mx : bl);
                out0[c] = (int)rr; out1[c] = (int)g; out2[c] = (int)bl;
            } else {
                out0[c] = 0; out1[c] = clampi(cb_i, lo, hi); out2[c] = clampi(cr_i, lo, hi);
            }
        }
        // (the chroma planes are 2cw x 2ch, so both columns and both rows exist there; the Y plane only has the region)
        int32_t *d1 = O1 + (int64_t)Y * o1.w + 2 * x, *d2 = O2 + (int64_t)Y * o2.w + 2 * x;
        Int2U a1; a1.v[0] = out1[0]; a1.v[1] = out1[1];
        Int2U a2; a2.v[0] = out2[0]; a2.v[1] = out2[1];
        *reinterpret_cast<Int2U *>(d1) = a1;
        *reinterpret_cast<Int2U *>(d2) = a2;
        int32_t *d0 = O0 + (int64_t)Y * o0.w + 2 * x;
        if (in_region[0] && in_region[1]) { Int2U a0; a0.v[0] = out0[0]; a0.v[1] = out0[1]; *reinterpret_cast<Int2U *>(d0) = a0; }
        else if (in_region[0]) d0[0] = out0[0];
    }
}

// ---------------------------------------------------------------------------------------------
// Forward YCoCg and Squeeze (the encoder-side subset that is data parallel).  Unlike their inverses
// they have no recurrence -- the residual of a pair reads ORIGINAL neighbours -- so it is one lane per pair.
// transform/ycocg.h:65-95, in place on three contiguous planes of n samples
__global__ __launch_bounds__(256) void k_fwd_ycocg(int32_t *c0, int32_t *c1, int32_t *c2, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int R = c0[i], G = c1[i], B = c2[i];
    c0[i] = (((R + B) >> 1) + G) >> 1;
    c1[i] = R - B;
    c2[i] = G - ((R + B) >> 1);
}
// transform/squeeze.h:135-170: in (w x h) -> avg ((w+1)/2 x h) + residual (w/2 x h)
__global__ __launch_bounds__(256) void k_fwd_hsqueeze(const int32_t *in, int w, int h, int32_t *avg, int32_t *res) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    const int w1 = (w + 1) / 2, w2 = w - w1;
    if (x >= w1 || y >= h) return;
    const int32_t *row = in + (int64_t)y * w;
    if (x >= w2) { avg[(int64_t)y * w1 + x] = row[2 * x]; return; }   // the odd last column is its own average
    const int A = row[2 * x], B = row[2 * x + 1];
    const int a = (A + B + (A > B)) >> 1;
    avg[(int64_t)y * w1 + x] = a;
    int next = a;
    if (x + 1 < w2) { const int C = row[2 * x + 2], D = row[2 * x + 3]; next = (C + D + (C > D)) >> 1; }
    else if (w & 1) next = row[2 * x + 2];
    const int left = x > 0 ? row[2 * x - 1] : a;
    res[(int64_t)y * w2 + x] = (A - B) - smooth_tendency(left, a, next);
}
// transform/squeeze.h:227-263: in (w x h) -> avg (w x (h+1)/2) + residual (w x h/2)
__global__ __launch_bounds__(256) void k_fwd_vsqueeze(const int32_t *in, int w, int h, int32_t *avg, int32_t *res) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    const int h1 = (h + 1) / 2, h2 = h - h1;
    if (x >= w || y >= h1) return;
    if (y >= h2) { avg[(int64_t)y * w + x] = in[(int64_t)(2 * y) * w + x]; return; }   // the odd last row
    const int A = in[(int64_t)(2 * y) * w + x], B = in[(int64_t)(2 * y + 1) * w + x];
    const int a = (A + B + (A > B)) >> 1;
    avg[(int64_t)y * w + x] = a;
    int next = a;
    if (y + 1 < h2) { const int C = in[(int64_t)(2 * y + 2) * w + x], D = in[(int64_t)(2 * y + 3) * w + x]; next = (C + D + (C > D)) >> 1; }
    else if (h & 1) next = in[(int64_t)(2 * y + 2) * w + x];
    const int top = y > 0 ? in[(int64_t)(2 * y - 1) * w + x] : a;
    res[(int64_t)y * w + x] = (A - B) - smooth_tendency(top, a, next);
}
__global__ __launch_bounds__(256) void k_scale(int32_t *d, int64_t n, int q) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] *= q;
}
void launch_scale(int32_t *plane, int64_t n, int q, hipStream_t stream) {
    if (n <= 0 || q == 1) return;
    hipLaunchKernelGGL(k_scale, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 65535)), dim3(256), 0, stream, plane, n, q);
}
void launch_fwd_ycocg(int32_t *c0, int32_t *c1, int32_t *c2, int64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(k_fwd_ycocg, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, c0, c1, c2, n);
}
void launch_fwd_squeeze(bool horizontal, const int32_t *in, int w, int h, int32_t *avg, int32_t *res, hipStream_t stream) {
    if (horizontal) hipLaunchKernelGGL(k_fwd_hsqueeze, dim3(((w + 1) / 2 + 255) / 256, h), dim3(256), 0, stream, in, w, h, avg, res);
    else hipLaunchKernelGGL(k_fwd_vsqueeze, dim3((w + 255) / 256, (h + 1) / 2), dim3(256), 0, stream, in, w, h, avg, res);
}

// export/write_pam.h:136-150 (the RGB / gray / +alpha path): for every pixel of the w x h image the first
// `components` channels, CLAMP(v, minval, maxval), one byte per sample or two bytes big-endian.  Planes may be
// wider than the image (DCT-padded): the row pitch is the plane's own width.  A quarter of the bytes of the
// int32 planes (an eighth for 8-bit images) is what a host that only wants the picture has to pull over PCIe.
__global__ __launch_bounds__(256) void k_pack_samples(Bases b, PackedPlanes pp, int w, int h, int lo, int hi, int bytes_per_sample, uint8_t *dst,
                                                      int64_t dst_stride) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    uint8_t *o = dst + (int64_t)blockIdx.z * dst_stride + ((int64_t)y * w + x) * pp.n * bytes_per_sample;
    for (int c = 0; c < pp.n; c++) {
        const int v = clampi(plane_ptr(b, pp.p[c], blockIdx.z)[(int64_t)y * pp.p[c].w + x], lo, hi);
        if (bytes_per_sample == 2) { o[2 * c] = (uint8_t)(v >> 8); o[2 * c + 1] = (uint8_t)(v & 0xFF); }
        else o[c] = (uint8_t)v;
    }
}

void launch_pack(const Bases &b, const PackedPlanes &pp, int w, int h, int lo, int hi, int bytes_per_sample, uint8_t *dst, int64_t dst_stride,
                 int n_images, hipStream_t stream) {
    if (w <= 0 || h <= 0 || n_images <= 0) return;
    hipLaunchKernelGGL(k_pack_samples, dim3((w + 255) / 256, h, n_images), dim3(256), 0, stream, b, pp, w, h, lo, hi, bytes_per_sample, dst, dst_stride);
}

// ---------------------------------------------------------------------------------------------
// Verification aid (no counterpart in the reference): one position-weighted 64-bit sum per image over a slab of int32 planes,
// sum_i v[i] * (i mod 65521 + 1) (two's-complement wrap-around) -- what fuif_amd.dist.plane_checksums computes with torch.  A host
// that decodes step after step (bench.py's overlapped steps, a service that re-decodes) keeps 8 bytes per image and step instead of
// the planes and compares afterwards; streaming, HBM bound (4 bytes per sample).  16-byte loads; blockIdx.y = image.
__global__ __launch_bounds__(256) void k_plane_checksums(const int32_t *planes, int64_t elems, int64_t stride, unsigned long long *sums) {
    const int32_t *src = planes + (int64_t)blockIdx.y * stride;
    unsigned long long acc = 0;
    const int64_t quads = elems / 4;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += step) {
        const int4 v = *reinterpret_cast<const int4 *>(src + 4 * q);
        const unsigned long long w0 = (unsigned long long)((4 * q) % 65521);      // weights w0+1 .. w0+4, each reduced mod 65521
        const long long vs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            unsigned long long wk = w0 + (unsigned long long)k;
            if (wk >= 65521ull) wk -= 65521ull;
            acc += (unsigned long long)vs[k] * (wk + 1ull);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = quads * 4; i < elems; i++) acc += (unsigned long long)(long long)src[i] * ((unsigned long long)(i % 65521) + 1ull);
    __shared__ unsigned long long part[256];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(&sums[blockIdx.y], part[0]);
}
void launch_plane_checksums(const int32_t *planes, int64_t elems, int64_t stride, int n_images, unsigned long long *sums, hipStream_t stream) {
    if (n_images <= 0) return;
    (void)hipMemsetAsync(sums, 0, sizeof(unsigned long long) * (size_t)n_images, stream);
    if (elems <= 0) return;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((elems / 4 + 2047) / 2048, 256));
    hipLaunchKernelGGL(k_plane_checksums, dim3((unsigned)blocks, (unsigned)n_images), dim3(256), 0, stream, planes, elems, stride, sums);
}

// ---------------------------------------------------------------------------------------------
static inline dim3 grid1d(int64_t n, int block, int z, int y = 1) {
    int64_t g = (n + block - 1) / block;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return dim3((unsigned)g, (unsigned)y, (unsigned)z);
}

// ---------------------------------------------------------------------------------------------
// int16 coefficient samples -> int32 for the coded planes the inverse kernels want as int32 (Plan::widen: the few planes that are not
// squeeze residuals or freshly dequantised DCT coefficients); streaming, HBM bound (6 bytes per sample).
// one launch for the planes of Plan::widen: blockIdx.y = plane, blockIdx.z = image
__global__ __launch_bounds__(256) void k_widen_planes(const coef_t *src, int32_t *dst, int64_t stride, const int64_t *pairs) {
    const int64_t off = pairs[2 * blockIdx.y], n = pairs[2 * blockIdx.y + 1];
    const coef_t *s = src + (int64_t)blockIdx.z * stride + off;
    int32_t *d = dst + (int64_t)blockIdx.z * stride + off;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) d[i] = s[i];
}
void launch_widen_planes(const coef_t *src, int32_t *dst, int64_t stride, const int64_t *dev_pairs, int n_planes, int64_t max_elems, int n_images, hipStream_t stream) {
    if (n_planes <= 0 || n_images <= 0) return;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((max_elems + 1023) / 1024, 512));
    hipLaunchKernelGGL(k_widen_planes, dim3((unsigned)blocks, (unsigned)n_planes, (unsigned)n_images), dim3(256), 0, stream, src, dst, stride, dev_pairs);
}

void launch_op(const Op &op, const Bases &b, const PlaneRef *dev_list, ChannelMeta *meta, int n_channels, int img_first,
               int n_images, hipStream_t stream, int32_t *status) {
    switch (op.kind) {
        case OP_VSQUEEZE: {
            const int w = op.src[0].w;
            if (w <= 0 || op.dst[0].h <= 0) break;
            if (op.r16) hipLaunchKernelGGL(k_inv_vsqueeze<coef_t>, dim3((w + 255) / 256, 1, n_images), dim3(256), 0, stream, b, op.src[0], op.src[1], op.dst[0], op.clamp_out, op.lo, op.hi);
            else hipLaunchKernelGGL(k_inv_vsqueeze<int32_t>, dim3((w + 255) / 256, 1, n_images), dim3(256), 0, stream, b, op.src[0], op.src[1], op.dst[0], op.clamp_out, op.lo, op.hi);
            break;
        }
        case OP_HSQUEEZE: {
            const int h = op.src[0].h;
            if (h <= 0 || op.dst[0].w <= 0) break;
            static const int tiles = [] { const char *e = getenv("FUIFGPU_HSQUEEZE_TILES"); return e ? atoi(e) : 1; }();   // 0: the lane-per-row kernel for every width (A/B measurements)
            if (tiles && op.src[1].w >= 2 * HL_P) {
                if (op.r16) hipLaunchKernelGGL(k_inv_hsqueeze_tiles<coef_t>, dim3((h + 63) / 64, 1, n_images), dim3(64), 0, stream, b, op.src[0], op.src[1], op.dst[0], op.clamp_out, op.lo, op.hi);
                else hipLaunchKernelGGL(k_inv_hsqueeze_tiles<int32_t>, dim3((h + 63) / 64, 1, n_images), dim3(64), 0, stream, b, op.src[0], op.src[1], op.dst[0], op.clamp_out, op.lo, op.hi);
            } else {
                if (op.r16) hipLaunchKernelGGL(k_inv_hsqueeze_rows<coef_t>, dim3((h + 255) / 256, 1, n_images), dim3(256), 0, stream, b, op.src[0], op.src[1], op.dst[0], op.clamp_out, op.lo, op.hi);
                else hipLaunchKernelGGL(k_inv_hsqueeze_rows<int32_t>, dim3((h + 255) / 256, 1, n_images), dim3(256), 0, stream, b, op.src[0], op.src[1], op.dst[0], op.clamp_out, op.lo, op.hi);
            }
            break;
        }
        case OP_HSQ2_YCOCG: {
            const int h = op.src[0].h;
            if (h <= 0 || op.p0 <= 0) break;
            if (op.r16) hipLaunchKernelGGL(k_inv_hsq2_ycocg<coef_t>, dim3((h + 63) / 64, 1, n_images), dim3(64), 0, stream, b, op.src[0], op.src[1], op.src[2], op.ext[0], op.dst[0], op.dst[1], op.dst[2], op.hi);
            else hipLaunchKernelGGL(k_inv_hsq2_ycocg<int32_t>, dim3((h + 63) / 64, 1, n_images), dim3(64), 0, stream, b, op.src[0], op.src[1], op.src[2], op.ext[0], op.dst[0], op.dst[1], op.dst[2], op.hi);
            break;
        }
        case OP_YCOCG:
            hipLaunchKernelGGL(k_inv_ycocg, dim3((op.p0 + 255) / 256, op.p1, n_images), dim3(256), 0, stream, b, op.src[0], op.src[1], op.src[2],
                               op.p0, op.p1, op.hi);
            break;
        case OP_YCBCR:
            hipLaunchKernelGGL(k_inv_ycbcr, dim3((op.p0 + 255) / 256, op.p1, n_images), dim3(256), 0, stream, b, op.src[0], op.src[1], op.src[2],
                               op.p0, op.p1, op.lo, op.hi);
            break;
        case OP_QUANT: {
            // grid.y = plane of the op's list; 64 grid-striding blocks per plane
            if (op.r16) hipLaunchKernelGGL(k_dequant<coef_t>, dim3(64, op.pad, n_images), dim3(256), 0, stream, b, dev_list + op.idct_first, meta, n_channels, img_first);
            else hipLaunchKernelGGL(k_dequant<int32_t>, dim3(64, op.pad, n_images), dim3(256), 0, stream, b, dev_list + op.idct_first, meta, n_channels, img_first);
            break;
        }
        case OP_IDCT:
            if (op.pad2 && meta) hipLaunchKernelGGL(k_idct8x8<true>, dim3((op.p0 + 63) / 64, op.p1, n_images), dim3(64), 0, stream, b, dev_list + op.idct_first, op.dst[0],
                                                    op.p0, op.p1, op.hi, op.clamp_out, op.lo, op.hi, meta, n_channels, img_first);
            else hipLaunchKernelGGL(k_idct8x8<false>, dim3((op.p0 + 63) / 64, op.p1, n_images), dim3(64), 0, stream, b, dev_list + op.idct_first, op.dst[0],
                                    op.p0, op.p1, op.hi, op.clamp_out, op.lo, op.hi, meta, n_channels, img_first);
            break;
        case OP_UPS2_YCBCR:
            if (op.src[1].w <= 0 || op.src[1].h <= 0) break;
            hipLaunchKernelGGL(k_ups2_ycbcr, dim3((op.src[1].w + 255) / 256, op.src[1].h, n_images), dim3(256), 0, stream, b, op.src[0], op.src[1], op.src[2],
                               op.dst[0], op.dst[1], op.dst[2], op.p0, op.p1, op.lo, op.hi);
            break;
        case OP_UPSAMPLE:
            if (op.p0 == 2 && op.p1 == 2)
                hipLaunchKernelGGL(k_upsample_2x2, dim3((op.src[0].w + 255) / 256, op.src[0].h, n_images), dim3(256), 0, stream, b, op.src[0], op.dst[0],
                                   op.clamp_out, op.lo, op.hi);
            else
                hipLaunchKernelGGL(k_upsample, dim3((op.dst[0].w + 255) / 256, op.dst[0].h, n_images), dim3(256), 0, stream, b, op.src[0], op.dst[0],
                                   op.p0, op.p1, op.clamp_out, op.lo, op.hi);
            break;
        case OP_COPY_CLAMP:
        case OP_CLAMP:
            hipLaunchKernelGGL(k_clamp, grid1d((int64_t)op.dst[0].w * op.dst[0].h, 256, n_images), dim3(256), 0, stream, b, op.src[0], op.dst[0],
                               op.lo, op.hi);
            break;
        case OP_PALETTE:
            if ((int64_t)op.dst[0].w * op.dst[0].h <= 0) break;
            hipLaunchKernelGGL(k_inv_palette, grid1d((int64_t)op.dst[0].w * op.dst[0].h, 256, n_images), dim3(256), 0, stream, b, op.src[0],
                               op.src[1], op.dst[0], op.p0, op.p1, op.clamp_out, op.lo, op.hi);
            break;
        case OP_PERMUTE:
            if ((int64_t)op.dst[0].w * op.dst[0].h <= 0 || op.pad < 1) break;
            hipLaunchKernelGGL(k_permute_plane, grid1d((int64_t)op.dst[0].w * op.dst[0].h, 256, n_images), dim3(256), 0, stream, b, op.src[0],
                               dev_list + op.idct_first, op.pad, op.p0, op.dst[0], op.clamp_out, op.lo, op.hi, status, img_first);
            break;
        case OP_MATCH:
            if (!meta || op.src[0].w <= 0) break;
            hipLaunchKernelGGL(k_inv_match_frames, dim3((op.src[0].w + 255) / 256, 1, n_images), dim3(256), 0, stream, b, op.src[0],
                               dev_list + op.idct_first, op.pad, op.p0, op.p1, meta, n_channels, img_first, status);
            break;
        case OP_MATCH_INIT:
            if (!meta || (int64_t)op.dst[0].w * op.dst[0].h <= 0) break;
            hipLaunchKernelGGL(k_match_init, grid1d((int64_t)op.dst[0].w * op.dst[0].h, 256, n_images), dim3(256), 0, stream, b, op.src[0], op.dst[0],
                               op.p0, dev_list + op.idct_first, op.pad / 3, meta, n_channels, img_first, status);
            break;
        case OP_MATCH_JUMP:
            if (!meta || (int64_t)op.dst[0].w * op.dst[0].h <= 0) break;
            hipLaunchKernelGGL(k_match_jump, grid1d((int64_t)op.dst[0].w * op.dst[0].h, 256, n_images), dim3(256), 0, stream, b, op.src[1], op.src[0],
                               op.dst[0], op.p0, dev_list + op.idct_first, op.pad / 3, op.p1, meta, n_channels, img_first);
            break;
        case OP_MATCH_APPLY:
            if (!meta || (int64_t)op.src[0].w * op.src[0].h <= 0) break;
            hipLaunchKernelGGL(k_match_apply, grid1d((int64_t)op.src[0].w * op.src[0].h, 256, n_images), dim3(256), 0, stream, b, op.src[1], op.src[0],
                               op.p0, dev_list + op.idct_first, op.p0 ? op.pad / 3 : op.pad, op.p1, meta, n_channels, img_first, status);
            break;
        case OP_APPROX:
            if ((!meta && (op.src[0].qsrc >= 0 || op.src[1].qsrc >= 0)) || (int64_t)op.dst[0].w * op.dst[0].h <= 0) break;   // raw planes carry no ChannelMeta
            hipLaunchKernelGGL(k_inv_approx, grid1d((int64_t)op.dst[0].w * op.dst[0].h, 256, n_images), dim3(256), 0, stream, b, op.src[0],
                               op.src[1], op.p0, op.p1, meta, n_channels, img_first);
            break;
        default:
            break;
    }
}

}  // namespace fuifgpu
