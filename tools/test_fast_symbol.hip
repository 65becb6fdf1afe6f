// tools/test_fast_symbol.hip -- unit test of the hand-written pieces of k_maniac_decode on the MI355X (test tooling, not product code):
//   * fast_symbol_hw (inline asm) against fast_symbol (its C++ specification) on random coder states, chances and stream bytes;
//   * leaf_commit's EXEC-masked table lookup against the plain formula;
//   * ds_bpermute_b32 with address bits above bit 7 set (the supernode records keep the exit word there).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I fuif_amd/csrc tools/test_fast_symbol.hip -o build/test_fast_symbol && build/test_fast_symbol
#include "../fuif_amd/csrc/maniac_decode.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace fuifgpu {

struct Case {
    uint32_t range, low, pos, amax_pos, amax_neg;
    uint32_t chances[32];
    uint32_t window[64];
};
struct Result {
    int32_t res[2];
    uint32_t range[2], low[2], pos[2], touched[2], bits[2];
    uint32_t commit_bad, bperm_bad;
};

__global__ __launch_bounds__(64) FUIF_OCCUPANCY void k_test(const Case *cases, Result *results, const uint16_t *table, int n) {
    const int lane = threadIdx.x;
    for (int c = blockIdx.x; c < n; c += gridDim.x) {
        const Case &k = cases[c];
        Result out{};
        FastSym F;
        F.amax_pos = rfl((int)k.amax_pos); F.amax_neg = rfl((int)k.amax_neg);
        F.emax_pos = rfl(ilog2u((uint32_t)F.amax_pos)); F.emax_neg = rfl(ilog2u((uint32_t)F.amax_neg));
        F.ilast_pos = rfl(F.emax_pos + 1); F.ilast_neg = rfl(F.emax_neg + 1);
        for (int v = 0; v < 2; v++) {
            Stream s;
            s.p = reinterpret_cast<const uint8_t *>(k.window); s.size = 256; s.pos = rflu(k.pos); s.limit = 0; s.win_base = 0;
            s.win = k.window[lane]; s.eof_flag = 0; s.blob_mode = 1;
            Rac r; r.range = rflu(k.range); r.low = rflu(k.low);
            // every second case with amax <= 255 runs on COMPACT leaves (16 chances, mantissa from slot 9, lanes 16..63 mirror 0..15); the others on the
            // 31-chance form (lanes 32..63 mirror 0..31)
            const bool compact = (c & 1) && k.amax_pos <= 255u && k.amax_neg <= 255u;
            const int slot = lane & (compact ? 15 : 31);
            LeafRegs L; L.leafv = slot < 31 ? (int)k.chances[slot] : 0; L.touched = 0; L.bits = 0;
            L.mb = rfl(compact ? kMantCompact : CH_MANT); L.mirror = rflu(compact ? 0x10001u : 1u);
            const int res = v ? fast_symbol_hw(r, s, L, F) : fast_symbol(r, s, L, F);
            out.res[v] = res; out.range[v] = r.range; out.low[v] = r.low; out.pos[v] = s.pos; out.touched[v] = L.touched; out.bits[v] = L.bits;
            if (v) {
                // commit: the asm against the formula, on the masks the asm decoder produced
                const int before = L.leafv;
                const uint32_t t = L.touched, b = L.bits;
                int want = before;
                if ((t >> slot) & 1u) want = table[before * 2 + ((b >> slot) & 1u)];
                leaf_commit(L, lane, table);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                out.commit_bad = (uint32_t)__popcll(__ballot(L.leafv != want));
            }
        }
        // ds_bpermute with the exit word above the property byte: lane i asks for lane (i * 7 + c) % 64
        {
            const int src = (lane * 7 + c) & 63;
            const int addr = (src << 2) | (int)(0x9ABCDE00u + ((uint32_t)c << 8));
            const int got = __builtin_amdgcn_ds_bpermute(addr, lane * 1000 + 7);
            out.bperm_bad = (uint32_t)__popcll(__ballot(got != src * 1000 + 7));
        }
        if (lane == 0) results[c] = out;
    }
}

}  // namespace fuifgpu

int main(int argc, char **argv) {
    using namespace fuifgpu;
    const int n = argc > 1 ? atoi(argv[1]) : 200000;
    std::vector<Case> cases(n);
    srand(12345);
    auto rnd = []() { return ((uint32_t)rand() << 16) ^ (uint32_t)rand(); };
    for (int c = 0; c < n; c++) {
        Case &k = cases[c];
        const int mode = c % 8;
        k.range = 0x10001u + rnd() % (0x1000000u - 0x10000u);
        if (mode == 1) k.range = 0x10001u + rnd() % 0x300u;          // just above the renormalisation bound
        k.low = rnd() % k.range;
        if (mode == 2) k.low = k.range - 1 - rnd() % 16;
        k.pos = rnd() % 190;
        k.amax_pos = 1 + rnd() % (mode == 3 ? 3 : mode == 4 ? 32767 : (c & 2) ? 255 : 600);
        k.amax_neg = 1 + rnd() % (mode == 3 ? 2 : mode == 4 ? 32767 : (c & 2) ? 255 : 600);
        // the ranges at which the unrolled exponent decisions of fast_symbol_hw end: emax = 3 (chance 4 is the last one), 4 and 5
        if (c % 16 == 11) { k.amax_pos = 8 + rnd() % 8; k.amax_neg = 8 + rnd() % 8; }
        if (c % 16 == 13) { k.amax_pos = 8 + rnd() % 56; k.amax_neg = 16 + rnd() % 16; }
        for (int i = 0; i < 32; i++) {
            k.chances[i] = 1 + rnd() % 4095;
            if (mode == 5) k.chances[i] = 1 + rnd() % 40;              // tiny chances: double renormalisations
            if (mode == 6) k.chances[i] = 4095 - rnd() % 40;
            if (mode == 7 && i >= 2 && i < 16) k.chances[i] = 1 + rnd() % 200;   // long exponents: reach emax
        }
        for (int i = 0; i < 64; i++) k.window[i] = rnd();
    }
    std::vector<uint16_t> table(8192);
    for (auto &t : table) t = (uint16_t)(1 + rnd() % 4095);
    Case *d_cases; Result *d_res; uint16_t *d_table;
    hipMalloc(&d_cases, sizeof(Case) * n); hipMalloc(&d_res, sizeof(Result) * n); hipMalloc(&d_table, 16384);
    hipMemcpy(d_cases, cases.data(), sizeof(Case) * n, hipMemcpyHostToDevice);
    hipMemcpy(d_table, table.data(), 16384, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_test, dim3(1024), dim3(64), 0, 0, d_cases, d_res, d_table, n);
    if (hipDeviceSynchronize() != hipSuccess) { printf("KERNEL FAILED: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
    std::vector<Result> res(n);
    hipMemcpy(res.data(), d_res, sizeof(Result) * n, hipMemcpyDeviceToHost);
    int bad = 0, commit_bad = 0, bperm_bad = 0, zero = 0, exhausted = 0;
    int by_e[16] = {0};   // nonzero symbols by exponent (the decoder has its own exit for e = 0..3 and a loop beyond)
    for (int c = 0; c < n; c++) {
        const Result &r = res[c];
        const bool same = r.res[0] == r.res[1] && r.range[0] == r.range[1] && r.low[0] == r.low[1] && r.pos[0] == r.pos[1] && r.touched[0] == r.touched[1] && r.bits[0] == r.bits[1];
        zero += r.res[0] == 0;
        if (r.res[0]) { int a = abs(r.res[0]), e = 0; while (a >> (e + 1)) e++; by_e[e < 15 ? e : 15]++; }
        if (!same && bad++ < 12)
            printf("case %d (range %x low %x pos %u amax %u/%u): spec res %d R %x L %x pos %u touched %x bits %x | hw res %d R %x L %x pos %u touched %x bits %x\n", c, cases[c].range,
                   cases[c].low, cases[c].pos, cases[c].amax_pos, cases[c].amax_neg, r.res[0], r.range[0], r.low[0], r.pos[0], r.touched[0], r.bits[0], r.res[1], r.range[1], r.low[1],
                   r.pos[1], r.touched[1], r.bits[1]);
        commit_bad += r.commit_bad != 0; bperm_bad += r.bperm_bad != 0;
    }
    printf("%d cases: %d decoder mismatches, %d commit mismatches, %d bpermute mismatches (%d zero symbols)\n", n, bad, commit_bad, bperm_bad, zero);
    printf("nonzero symbols by exponent:");
    for (int e = 0; e < 16; e++) printf(" %d", by_e[e]);
    printf("\n");
    return bad || commit_bad || bperm_bad ? 1 : 0;
}
