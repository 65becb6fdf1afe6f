// fuif_amd/csrc/maniac_encode.hip -- the MANIAC pixel loop of the WRITER for gfx950.
//
// The reference encoder (encoding/encoding.cpp:74-207 with maniac/rac_enc.h:28-100, maniac/symbol_enc.h and the write side of
// maniac/compound.h) walks the pixels of a channel group once: properties + prediction, tree walk to a leaf, then the residual
// goes through that leaf's adaptive chances into the range coder.  On the encode side every sample is known in advance, so only
// the last step is serial.  Two kernels per group:
//
//   k_enc_model  one lane per PIXEL: the 2k+13 properties and the prediction (context_predict.h:124-168, 233-289), the walk
//                through the group's context tree (compound.h:142-153) -> per pixel {prediction, leaf}.  No dependency between
//                pixels: this is the part that is serial in the decoder and parallel here.
//   k_enc_rac    one wavefront per group: 64 pixels at a time are staged through LDS (residual, its range, leaf), lane 0 runs
//                the symbol binarisation (symbol.h:154-185, write side), the chance updates (chance.h:77-79) and the 24-bit range
//                coder with its delayed-byte carry handling (rac_enc.h:40-85), and appends the bytes.
//
// The tree itself and the group header are written by the host writer (csrc/writer.cpp) into the same range coder before the
// pixels: the coder's state {range, low, delayed byte, pending 0xFF run} is handed in and handed back.  Output is byte-identical to
// the host writer's (tests/test_zz_gpu_encoder.py), which tests/test_writer.py pins to the reference CLI.
//
// Round 3: first correct path.  A 4K picture's longest group is a chain of 4.1 million symbols on ONE lane, so a single picture
// (maniac_encode_group_gpu: one group per launch pair, synchronous) is slower than on a CPU core; the point of the design is the
// batch: maniac_encode_jobs_gpu runs one wavefront per group over many pictures, as k_maniac_decode does.  Parity of the
// single-group kernels was checked on the MI355X (profiles/r3_gpu_encoder_tests.txt); the job-list kernels share their device code
// and are emulator-verified; nothing was timed in round 3.  The single-group coder keeps leaf chances in global memory (a
// dependent read and a write per binary decision, by lane 0); the batch coder (enc_rac_run_wave) runs the scalar code on the whole
// wavefront and keeps the current leaf in a register with the next one prefetched, as the decoder does.  Its 16 KB chance table in
// LDS limits a CU to 9 such wavefronts: the first thing to look at once it can be timed.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "../../include/fuifgpu.h"
#include "maniac_encode.h"

namespace fuifgpu {

namespace {

#define DEV __device__ __forceinline__

constexpr int CH_ZERO = 0, CH_SIGN = 1, CH_EXP = 2, CH_MANT = 16, CH_N = 31;   // symbol.h:97-113: zero, sign, 14 exponent, 15 mantissa chances

DEV int e_iabs(int x) { return x < 0 ? -x : x; }
DEV int e_ilog2(uint32_t l) { return l == 0 ? 0 : 31 - __builtin_clz(l); }
DEV int e_slog(int x) {   // context_predict.h:52-60
    if (x == 0) return 0;
    if (x > 0) return 32 - __builtin_clz((unsigned)x);
    return -(32 - __builtin_clz((unsigned)(-x)));
}
DEV int e_median3(int a, int b, int c) {
    if (a < b) { if (b < c) return b; return a < c ? c : a; }
    if (a < c) return a;
    return b < c ? c : b;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// context model of every pixel, in parallel
namespace {
DEV void enc_model_pixel(const EncGroup &g, const EncNode *tree, int n_nodes, int64_t i, int32_t *guess_out, int32_t *leaf_out) {
    const int64_t n = (int64_t)g.w * g.h;
    if (i >= n) return;
    const int w = g.w;
    const int y = (int)(i / w), x = (int)(i - (int64_t)y * w);
    int32_t p[kMaxProps];
    int o = 0;
    for (int k = 0; k < g.nrefs; k++) {   // context_predict.h:233-289, one pixel
        const EncRef rc = g.refs[k];
        int ry = (y << g.vshift) >> rc.vshift; if (ry >= rc.h) ry = rc.h - 1;
        int rx = (x << g.hshift) >> rc.hshift; if (rx >= rc.w) rx = rc.w - 1;
        const int v = rc.data[(int64_t)ry * rc.w + rx];
        p[o++] = e_iabs(v); p[o++] = e_slog(v);
    }
    const int32_t *d = g.plane;
    // context_predict.h:126-133
    const int left = x ? d[i - 1] : g.zero;
    const int top = y ? d[i - w] : g.zero;
    const int topleft = (x && y) ? d[i - w - 1] : left;
    const int topright = (x + 1 < w && y) ? d[i - w + 1] : top;
    const int leftleft = x > 1 ? d[i - 2] : left;
    const int toptop = y > 1 ? d[i - 2 * (int64_t)w] : top;
    p[o++] = e_iabs(top); p[o++] = e_iabs(left); p[o++] = e_slog(top); p[o++] = e_slog(left);
    p[o++] = y; p[o++] = x;
    p[o++] = left + top - topleft; p[o++] = topleft + topright - top;
    p[o++] = e_slog(left - topleft); p[o++] = e_slog(topleft - top); p[o++] = e_slog(top - topright);
    p[o++] = e_slog(top - toptop); p[o++] = e_slog(left - leftleft);
    int guess;
    switch (g.predictor) {   // context_predict.h:157-166
        case 0: guess = g.zero; break;
        case 1: guess = (left + top) / 2; break;
        case 3: guess = left; break;
        case 4: guess = top; break;
        case 5: guess = (left + topleft + top + topright) / 4; break;
        case 6: { const int t = left + top - topleft; guess = t < g.minval ? g.minval : (t > g.maxval ? g.maxval : t); break; }
        default: guess = e_median3(left + top - topleft, left, top); break;
    }
    int pos = 0;
    EncNode nd = tree[0];
    for (int depth = 0; depth < n_nodes && nd.prop >= 0; depth++) {   // compound.h:142-153 (a walk visits every node at most once)
        pos = p[nd.prop & (kMaxProps - 1)] > nd.split ? nd.child : nd.child + 1;
        pos = pos < n_nodes ? pos : 0;
        nd = tree[pos];
    }
    guess_out[i] = guess;
    leaf_out[i] = nd.leaf;
}
}  // namespace

__global__ __launch_bounds__(256) void k_enc_model(EncGroup g, const EncNode *tree, int n_nodes, int32_t *guess_out, int32_t *leaf_out) {
    enc_model_pixel(g, tree, n_nodes, (int64_t)blockIdx.x * 256 + threadIdx.x, guess_out, leaf_out);
}
// the same for a list of groups (of many pictures) in one launch: the blocks of all jobs are numbered through, first_block[k] is
// the first block of job k (first_block[n_jobs] = all blocks), a block finds its job by bisection -- no empty blocks, whatever
// the mix of group sizes (a 4K picture has groups of 40 and of 4 million pixels)
__global__ __launch_bounds__(256) void k_enc_model_jobs(const EncJobDev *jobs, const uint32_t *first_block, int n_jobs) {
    const uint32_t blk = blockIdx.x;
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {                       // the last job whose first block is <= blk
        const int mid = (lo + hi + 1) >> 1;
        if (first_block[mid] <= blk) lo = mid; else hi = mid - 1;
    }
    const EncJobDev &j = jobs[lo];
    enc_model_pixel(j.g, j.tree, j.n_nodes, (int64_t)(blk - first_block[lo]) * 256 + threadIdx.x, j.guess, j.leaf);
}

// ---------------------------------------------------------------------------------------------
// the serial part: one wavefront, lane 0 codes
namespace {

struct DevRac {   // rac_enc.h:28-100 (RacOutput24): the decoder is three bytes ahead, carries ripple into the delayed byte
    uint32_t range, low;
    int32_t delayed, pending;
    uint8_t *out;
    uint32_t cap, count;
    DEV void emit(int b) { if (count < cap) out[count] = (uint8_t)b; count++; }
    DEV void shift() {
        const uint32_t byte = low >> 16;   // bit 8 = a carry into the bytes already generated
        if (delayed < 0) delayed = (int32_t)byte;
        else if (byte < 0xFF) { emit(delayed); for (; pending; pending--) emit(0xFF); delayed = (int32_t)byte; }
        else if (byte > 0xFF) { emit(delayed + 1); for (; pending; pending--) emit(0x00); delayed = (int32_t)(byte & 0xFF); }
        else pending++;
        low = (low & 0xFFFF) << 8;
        range <<= 8;
    }
    DEV void put12(uint32_t b12, int bit) {   // rac.h:43-52 chance_12bit_chance + rac_enc.h:60-72
        const uint32_t chance = (((range & 0xFFFu) * b12 + 0x800u) >> 12) + ((range >> 12) * b12);
        if (bit) { low += range - chance; range = chance; }
        else range -= chance;
        for (int k = 0; k < 4 && range <= 0x10000u; k++) shift();   // at most three shifts while range >= 1 (rac_enc.h:66-71 loops; bounded here so that no input can hang a wavefront)
    }
};

DEV void dev_coder_write(DevRac &r, uint16_t *ch, int idx, const uint16_t *table, int bit) {
    const uint32_t c = ch[idx];
    r.put12(c, bit);
    ch[idx] = table[c * 2 + bit];   // chance.h:77-79
}

// write side of symbol.h:154-185
DEV void dev_write_symbol(DevRac &r, uint16_t *ch, const uint16_t *table, int min, int max, int value) {
    if (value == 0) { dev_coder_write(r, ch, CH_ZERO, table, 1); return; }
    dev_coder_write(r, ch, CH_ZERO, table, 0);
    const int sign = value > 0;
    if (min < 0 && max > 0) dev_coder_write(r, ch, CH_SIGN, table, sign);
    const int a = e_iabs(value);
    const int e = e_ilog2((uint32_t)a);
    const int amax = sign ? max : -min;
    const int emax = e_ilog2((uint32_t)amax);
    for (int i = 0; i < emax; i++) {
        dev_coder_write(r, ch, CH_EXP + i, table, i == e);
        if (i == e) break;
    }
    int have = 1 << e;
    for (int pos = e; pos > 0;) {
        pos--;
        const int minabs1 = have | (1 << pos);
        if (minabs1 > amax) continue;
        const int bit = (a >> pos) & 1;
        dev_coder_write(r, ch, CH_MANT + pos, table, bit);
        if (bit) have = minabs1;
    }
}

}  // namespace

namespace {
// state[0..3] = RacEncState in / out, state[4] = bytes emitted (out), state[5] = 1 when `out` was too small
DEV void enc_rac_run(const int32_t *plane, const int32_t *guess, const int32_t *leafidx, int64_t n, int minval, int maxval, uint16_t *leaves,
                     const uint16_t *table_g, uint32_t *state, uint8_t *out, uint32_t out_cap) {
    __shared__ uint16_t table[8192];
    __shared__ int32_t s_diff[64], s_min[64], s_max[64], s_leaf[64];
    const int lane = threadIdx.x;
    for (int k = lane; k < 8192; k += 64) table[k] = table_g[k];
    DevRac r;
    r.range = state[0]; r.low = state[1]; r.delayed = (int32_t)state[2]; r.pending = (int32_t)state[3];
    r.out = out; r.cap = out_cap; r.count = 0;
    __syncthreads();
    for (int64_t x0 = 0; x0 < n; x0 += 64) {
        const int nx = (int)(n - x0 < 64 ? n - x0 : 64);
        if (lane < nx) {
            const int gs = guess[x0 + lane];
            s_diff[lane] = plane[x0 + lane] - gs;
            s_min[lane] = minval - gs;
            s_max[lane] = maxval - gs;
            s_leaf[lane] = leafidx[x0 + lane];
        }
        __syncthreads();
        if (lane == 0) {
            for (int j = 0; j < nx; j++) {
                const int mn = s_min[j], mx = s_max[j];
                if (mn == mx) continue;   // compound.h:228: nothing to code
                dev_write_symbol(r, leaves + (int64_t)s_leaf[j] * CH_N, table, mn, mx, s_diff[j]);   // (s_leaf < n_leaves: k_enc_model only hands out leaf numbers of the tree)
            }
        }
        __syncthreads();
    }
    if (lane == 0) {
        state[0] = r.range; state[1] = r.low; state[2] = (uint32_t)r.delayed; state[3] = (uint32_t)r.pending;
        state[4] = r.count; state[5] = r.count > r.cap ? 1u : 0u;
    }
}
}  // namespace

__global__ __launch_bounds__(64) void k_enc_rac(const int32_t *plane, const int32_t *guess, const int32_t *leafidx, int64_t n, int minval, int maxval,
                                                uint16_t *leaves, const uint16_t *table_g, uint32_t *state, uint8_t *out, uint32_t out_cap) {
    enc_rac_run(plane, guess, leafidx, n, minval, maxval, leaves, table_g, state, out, out_cap);
}
// ---- the coder as the batch runs it: the whole wavefront executes the scalar code (uniform control flow, scalars made uniform with
// readfirstlane), so the lanes can hold state: lane i (< 31) keeps chance i of the CURRENT leaf in a register, as k_maniac_decode
// does.  A binary decision reads its chance with v_readlane and updates one lane; memory is touched only when the leaf changes
// (31 lanes store the old one, load the new one: two 62-byte accesses per switch instead of a dependent global read and a write
// per decision) -- and because the encoder knows the NEXT pixel's leaf already, that load is issued one symbol early.
// A lane's own earlier store to the same leaf is seen by its later load (single-thread order), the current leaf is never the
// prefetched one.
namespace {
DEV int e_rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

struct WaveRac {   // DevRac with uniform state in every lane; only lane 0 stores bytes
    uint32_t range, low;
    int32_t delayed, pending;
    uint8_t *out;
    uint32_t cap, count;
    int lane;
    DEV void emit(int b) { if (lane == 0 && count < cap) out[count] = (uint8_t)b; count++; }
    DEV void shift() {
        const uint32_t byte = low >> 16;
        if (delayed < 0) delayed = (int32_t)byte;
        else if (byte < 0xFF) { emit(delayed); for (; pending; pending--) emit(0xFF); delayed = (int32_t)byte; }
        else if (byte > 0xFF) { emit(delayed + 1); for (; pending; pending--) emit(0x00); delayed = (int32_t)(byte & 0xFF); }
        else pending++;
        low = (low & 0xFFFF) << 8;
        range <<= 8;
    }
    DEV void put12(uint32_t b12, int bit) {
        const uint32_t chance = (((range & 0xFFFu) * b12 + 0x800u) >> 12) + ((range >> 12) * b12);
        if (bit) { low += range - chance; range = chance; }
        else range -= chance;
        for (int k = 0; k < 4 && range <= 0x10000u; k++) shift();
    }
};

DEV void wave_coder_write(WaveRac &r, int &leafv, int idx, const uint16_t *table, int bit) {
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane(leafv, idx) & 0xFFFFu;
    r.put12(c, bit);
    const int nc = (int)table[c * 2 + bit];   // chance.h:77-79 (every lane reads the same LDS word)
    if (r.lane == idx) leafv = nc;
}

// write side of symbol.h:154-185, uniform over the wavefront
DEV void wave_write_symbol(WaveRac &r, int &leafv, const uint16_t *table, int min, int max, int value) {
    if (value == 0) { wave_coder_write(r, leafv, CH_ZERO, table, 1); return; }
    wave_coder_write(r, leafv, CH_ZERO, table, 0);
    const int sign = value > 0;
    if (min < 0 && max > 0) wave_coder_write(r, leafv, CH_SIGN, table, sign);
    const int a = e_iabs(value);
    const int e = e_ilog2((uint32_t)a);
    const int amax = sign ? max : -min;
    const int emax = e_ilog2((uint32_t)amax);
    for (int i = 0; i < emax; i++) {
        wave_coder_write(r, leafv, CH_EXP + i, table, i == e);
        if (i == e) break;
    }
    int have = 1 << e;
    for (int pos = e; pos > 0;) {
        pos--;
        const int minabs1 = have | (1 << pos);
        if (minabs1 > amax) continue;
        const int bit = (a >> pos) & 1;
        wave_coder_write(r, leafv, CH_MANT + pos, table, bit);
        if (bit) have = minabs1;
    }
}

DEV void enc_rac_run_wave(const int32_t *plane, const int32_t *guess, const int32_t *leafidx, int64_t n, int minval, int maxval, uint16_t *leaves,
                          const uint16_t *table_g, uint32_t *state, uint8_t *out, uint32_t out_cap) {
    __shared__ uint16_t table[8192];
    __shared__ int32_t s_diff[64], s_min[64], s_max[64], s_leaf[64];
    const int lane = threadIdx.x;
    for (int k = lane; k < 8192; k += 64) table[k] = table_g[k];
    WaveRac r;
    r.range = (uint32_t)e_rfl((int)state[0]); r.low = (uint32_t)e_rfl((int)state[1]); r.delayed = e_rfl((int)state[2]); r.pending = e_rfl((int)state[3]);
    r.out = out; r.cap = out_cap; r.count = 0; r.lane = lane;
    int cur = -1, leafv = 0;         // the current leaf and, in lane i < 31, its chance i
    int pre = -1, prev = 0;          // a prefetched leaf (its number, its chances)
    __syncthreads();
    for (int64_t x0 = 0; x0 < n; x0 += 64) {
        const int nx = (int)(n - x0 < 64 ? n - x0 : 64);
        if (lane < nx) {
            const int gs = guess[x0 + lane];
            s_diff[lane] = plane[x0 + lane] - gs;
            s_min[lane] = minval - gs;
            s_max[lane] = maxval - gs;
            s_leaf[lane] = leafidx[x0 + lane];
        }
        __syncthreads();
        for (int j = 0; j < nx; j++) {
            const int mn = e_rfl(s_min[j]), mx = e_rfl(s_max[j]);
            if (mn == mx) continue;   // compound.h:228: nothing to code
            const int lf = e_rfl(s_leaf[j]);
            if (lf != cur) {
                if (cur >= 0 && lane < CH_N) leaves[(int64_t)cur * CH_N + lane] = (uint16_t)leafv;
                if (lf == pre) leafv = prev;
                else if (lane < CH_N) leafv = leaves[(int64_t)lf * CH_N + lane];
                cur = lf;
                pre = -1;
            }
            // the next pixel's leaf is known already: fetch its chances while this symbol is coded
            if (j + 1 < nx) {
                const int nl = e_rfl(s_leaf[j + 1]);
                if (nl != cur && nl != pre) {
                    if (lane < CH_N) prev = leaves[(int64_t)nl * CH_N + lane];
                    pre = nl;
                }
            }
            wave_write_symbol(r, leafv, table, mn, mx, e_rfl(s_diff[j]));
        }
        __syncthreads();
    }
    if (cur >= 0 && lane < CH_N) leaves[(int64_t)cur * CH_N + lane] = (uint16_t)leafv;
    if (lane == 0) {
        state[0] = r.range; state[1] = r.low; state[2] = (uint32_t)r.delayed; state[3] = (uint32_t)r.pending;
        state[4] = r.count; state[5] = r.count > r.cap ? 1u : 0u;
    }
}
}  // namespace

// one wavefront per job: the groups of a whole batch of pictures code side by side, as k_maniac_decode's tiles decode
__global__ __launch_bounds__(64) void k_enc_rac_jobs(const EncJobDev *jobs, const uint16_t *table_g) {
    const EncJobDev &j = jobs[blockIdx.x];
    enc_rac_run_wave(j.g.plane, j.guess, j.leaf, j.n, j.g.minval, j.g.maxval, j.leaves, table_g, j.state, j.out, j.out_cap);
}

// ---------------------------------------------------------------------------------------------
void EncScratch::release() {
    hipFree(d_guess); hipFree(d_leaf); hipFree(d_bytes); hipFree(d_tree); hipFree(d_leaves); hipFree(d_table); hipFree(d_state);
    *this = EncScratch();
}

int maniac_encode_group_gpu(const EncGroup &g, const EncNode *tree, int n_nodes, int n_leaves, const uint16_t *leaf_init, const uint16_t *pixel_table,
                            RacEncState *state, std::vector<uint8_t> &out, EncScratch &s) {
    if (!g.plane || !tree || n_nodes < 1 || n_leaves < 1 || !leaf_init || !pixel_table || !state || g.w < 1 || g.h < 1 || g.nrefs < 0 || g.nrefs > kMaxRefs)
        return FUIFGPU_E_ARG;
    const size_t n = (size_t)g.w * g.h;
#define ECHK(call) do { if ((call) != hipSuccess) return FUIFGPU_E_HIP; } while (0)
    if (n > s.pixel_cap) {
        hipFree(s.d_guess); hipFree(s.d_leaf); s.d_guess = s.d_leaf = nullptr; s.pixel_cap = 0;
        ECHK(hipMalloc((void **)&s.d_guess, n * 4)); ECHK(hipMalloc((void **)&s.d_leaf, n * 4));
        s.pixel_cap = n;
    }
    // a symbol is at most 1 + 1 + 14 + 14 binary decisions, each well under a byte after renormalisation: 4 bytes per sample is generous
    const size_t bytes_cap = std::min<size_t>(n * 4 + 1024, 0xFFFFFF00u);
    if (bytes_cap > s.bytes_cap) {
        hipFree(s.d_bytes); s.d_bytes = nullptr; s.bytes_cap = 0;
        ECHK(hipMalloc((void **)&s.d_bytes, bytes_cap));
        s.bytes_cap = bytes_cap;
    }
    if ((size_t)n_nodes > s.tree_cap) {
        hipFree(s.d_tree); s.d_tree = nullptr; s.tree_cap = 0;
        ECHK(hipMalloc((void **)&s.d_tree, sizeof(EncNode) * (size_t)n_nodes));
        s.tree_cap = (size_t)n_nodes;
    }
    if ((size_t)n_leaves > s.leaves_cap) {
        hipFree(s.d_leaves); s.d_leaves = nullptr; s.leaves_cap = 0;
        ECHK(hipMalloc((void **)&s.d_leaves, sizeof(uint16_t) * CH_N * (size_t)n_leaves));
        s.leaves_cap = (size_t)n_leaves;
    }
    if (!s.d_table) ECHK(hipMalloc((void **)&s.d_table, sizeof(uint16_t) * 8192));
    if (!s.d_state) ECHK(hipMalloc((void **)&s.d_state, sizeof(uint32_t) * 8));
    std::vector<uint16_t> leaves((size_t)n_leaves * CH_N);
    for (int l = 0; l < n_leaves; l++) memcpy(&leaves[(size_t)l * CH_N], leaf_init, sizeof(uint16_t) * CH_N);
    uint32_t st[8] = {state->range, state->low, (uint32_t)state->delayed, (uint32_t)state->pending, 0, 0, 0, 0};
    ECHK(hipMemcpy(s.d_tree, tree, sizeof(EncNode) * (size_t)n_nodes, hipMemcpyHostToDevice));
    ECHK(hipMemcpy(s.d_leaves, leaves.data(), sizeof(uint16_t) * leaves.size(), hipMemcpyHostToDevice));
    ECHK(hipMemcpy(s.d_table, pixel_table, sizeof(uint16_t) * 8192, hipMemcpyHostToDevice));
    ECHK(hipMemcpy(s.d_state, st, sizeof(st), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_enc_model, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, g, s.d_tree, n_nodes, s.d_guess, s.d_leaf);
    hipLaunchKernelGGL(k_enc_rac, dim3(1), dim3(64), 0, nullptr, g.plane, s.d_guess, s.d_leaf, (int64_t)n, g.minval, g.maxval, s.d_leaves, s.d_table, s.d_state,
                       s.d_bytes, (uint32_t)s.bytes_cap);
    ECHK(hipGetLastError());
    ECHK(hipMemcpy(st, s.d_state, sizeof(st), hipMemcpyDeviceToHost));   // synchronises with the null stream's kernels
    if (st[5]) return FUIFGPU_E_NOMEM;
    const size_t old = out.size();
    out.resize(old + st[4]);
    if (st[4]) ECHK(hipMemcpy(out.data() + old, s.d_bytes, st[4], hipMemcpyDeviceToHost));
    state->range = st[0]; state->low = st[1]; state->delayed = (int32_t)st[2]; state->pending = (int32_t)st[3];
#undef ECHK
    return FUIFGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// the bodies of all jobs, packed back to back (one block per job copies out[0 .. state[4]) to dst + offset[job])
__global__ __launch_bounds__(256) void k_enc_gather(const EncJobDev *jobs, const uint64_t *offset, uint8_t *dst) {
    const EncJobDev &j = jobs[blockIdx.x];
    const uint32_t n = j.state[4] < j.out_cap ? j.state[4] : j.out_cap;
    uint8_t *d = dst + offset[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < n; i += 256) d[i] = j.out[i];
}

// A batch: one device arena for all jobs.  Its head -- job table, chance table, every job's tree, leaf chances and coder state --
// is assembled on the host and uploaded in ONE copy; the states come back in one copy, the bodies are packed by k_enc_gather
// and come back in one more (a 1024-picture batch has ~62 000 jobs: per-job copies would cost seconds).
int maniac_encode_jobs_gpu(std::vector<EncJob> &jobs, const uint16_t *pixel_table) {
    if (!pixel_table) return FUIFGPU_E_ARG;
    if (jobs.empty()) return FUIFGPU_OK;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t nj = jobs.size();
    // layout of the head: [EncJobDev x nj][table][states: 32 bytes x nj][per job: tree, leaves]; then per job: guess, leaf, out;
    // then the offsets and the packed bodies
    const size_t off_table = up(sizeof(EncJobDev) * nj), off_states = off_table + up(sizeof(uint16_t) * 8192);
    size_t head = off_states + up(32 * nj);
    std::vector<size_t> off_tree(nj), off_leaves(nj);
    for (size_t k = 0; k < nj; k++) {
        const EncJob &j = jobs[k];
        if (!j.g.plane || j.tree.empty() || j.n_leaves < 1 || j.g.w < 1 || j.g.h < 1 || j.g.nrefs < 0 || j.g.nrefs > kMaxRefs) return FUIFGPU_E_ARG;
        off_tree[k] = head; head += up(sizeof(EncNode) * j.tree.size());
        off_leaves[k] = head; head += up(sizeof(uint16_t) * CH_N * (size_t)j.n_leaves);
    }
    size_t total = head;
    std::vector<size_t> off_guess(nj), off_leaf(nj), off_out(nj);
    std::vector<uint32_t> cap(nj);
    for (size_t k = 0; k < nj; k++) {
        const size_t n = (size_t)jobs[k].g.w * jobs[k].g.h;
        // a symbol is at most 1 + 1 + 14 + 14 binary decisions, each well under a byte after renormalisation: 4 bytes per sample is generous
        cap[k] = (uint32_t)std::min<size_t>(n * 4 + 1024, 0xFFFFFF00u);
        off_guess[k] = total; total += up(n * 4);
        off_leaf[k] = total; total += up(n * 4);
        off_out[k] = total; total += up(cap[k]);
    }
    const size_t off_offsets = total; total += up(sizeof(uint64_t) * nj);
    const size_t off_first = total; total += up(sizeof(uint32_t) * (nj + 1));
    uint8_t *arena = nullptr;
    if (hipMalloc((void **)&arena, total) != hipSuccess) return FUIFGPU_E_HIP;
    int rc = FUIFGPU_OK;
#define ECHK(call) do { if (rc == FUIFGPU_OK && (call) != hipSuccess) rc = FUIFGPU_E_HIP; } while (0)
    std::vector<uint8_t> host(head, 0);
    EncJobDev *h_jobs = reinterpret_cast<EncJobDev *>(host.data());
    memcpy(host.data() + off_table, pixel_table, sizeof(uint16_t) * 8192);
    for (size_t k = 0; k < nj; k++) {
        const EncJob &j = jobs[k];
        EncJobDev &d = h_jobs[k];
        d.g = j.g; d.n = (int64_t)j.g.w * j.g.h; d.n_nodes = (int32_t)j.tree.size(); d.pad = 0; d.pad2 = 0;
        d.tree = reinterpret_cast<const EncNode *>(arena + off_tree[k]);
        d.leaves = reinterpret_cast<uint16_t *>(arena + off_leaves[k]);
        d.state = reinterpret_cast<uint32_t *>(arena + off_states + 32 * k);
        d.guess = reinterpret_cast<int32_t *>(arena + off_guess[k]);
        d.leaf = reinterpret_cast<int32_t *>(arena + off_leaf[k]);
        d.out = arena + off_out[k];
        d.out_cap = cap[k];
        memcpy(host.data() + off_tree[k], j.tree.data(), sizeof(EncNode) * j.tree.size());
        for (int l = 0; l < j.n_leaves; l++) memcpy(host.data() + off_leaves[k] + sizeof(uint16_t) * CH_N * (size_t)l, j.leaf_init, sizeof(uint16_t) * CH_N);
        const uint32_t st[8] = {j.state.range, j.state.low, (uint32_t)j.state.delayed, (uint32_t)j.state.pending, 0, 0, 0, 0};
        memcpy(host.data() + off_states + 32 * k, st, sizeof(st));
    }
    ECHK(hipMemcpy(arena, host.data(), head, hipMemcpyHostToDevice));
    const EncJobDev *d_jobs = reinterpret_cast<const EncJobDev *>(arena);
    const uint16_t *d_table = reinterpret_cast<const uint16_t *>(arena + off_table);
    if (rc == FUIFGPU_OK) {
        std::vector<uint32_t> first(nj + 1);
        uint64_t blocks = 0;
        for (size_t k = 0; k < nj; k++) { first[k] = (uint32_t)blocks; blocks += ((uint64_t)jobs[k].g.w * jobs[k].g.h + 255) / 256; }
        first[nj] = (uint32_t)blocks;
        if (blocks > 0x7FFFFFFFull) rc = FUIFGPU_E_ARG;   // > 5 * 10^11 pixels in one batch: split it
        ECHK(hipMemcpy(arena + off_first, first.data(), sizeof(uint32_t) * (nj + 1), hipMemcpyHostToDevice));
        if (rc == FUIFGPU_OK) {   // the coder reads what the model kernel wrote: it only runs behind it (never on uninitialised guesses)
            hipLaunchKernelGGL(k_enc_model_jobs, dim3((unsigned)blocks), dim3(256), 0, nullptr, d_jobs, reinterpret_cast<const uint32_t *>(arena + off_first), (int)nj);
            hipLaunchKernelGGL(k_enc_rac_jobs, dim3((unsigned)nj), dim3(64), 0, nullptr, d_jobs, d_table);
            ECHK(hipGetLastError());
        }
    }
    std::vector<uint32_t> states(8 * nj);
    ECHK(hipMemcpy(states.data(), arena + off_states, 32 * nj, hipMemcpyDeviceToHost));   // synchronises with the null stream's kernels
    std::vector<uint64_t> offsets(nj);
    uint64_t packed = 0;
    for (size_t k = 0; k < nj && rc == FUIFGPU_OK; k++) {
        if (states[8 * k + 5]) rc = FUIFGPU_E_NOMEM;
        offsets[k] = packed;
        packed += states[8 * k + 4];
    }
    if (rc == FUIFGPU_OK && packed) {
        uint8_t *d_packed = nullptr;
        if (hipMalloc((void **)&d_packed, packed) != hipSuccess) rc = FUIFGPU_E_HIP;
        ECHK(hipMemcpy(arena + off_offsets, offsets.data(), sizeof(uint64_t) * nj, hipMemcpyHostToDevice));
        if (rc == FUIFGPU_OK) {
            hipLaunchKernelGGL(k_enc_gather, dim3((unsigned)nj), dim3(256), 0, nullptr, d_jobs, reinterpret_cast<const uint64_t *>(arena + off_offsets), d_packed);
            ECHK(hipGetLastError());
        }
        std::vector<uint8_t> all(packed);
        ECHK(hipMemcpy(all.data(), d_packed, packed, hipMemcpyDeviceToHost));
        if (rc == FUIFGPU_OK)
            for (size_t k = 0; k < nj; k++) jobs[k].body.assign(all.begin() + offsets[k], all.begin() + offsets[k] + states[8 * k + 4]);
        hipFree(d_packed);
    }
    for (size_t k = 0; k < nj && rc == FUIFGPU_OK; k++)
        jobs[k].state = RacEncState{states[8 * k], states[8 * k + 1], (int32_t)states[8 * k + 2], (int32_t)states[8 * k + 3]};
#undef ECHK
    hipFree(arena);
    return rc;
}

}  // namespace fuifgpu
