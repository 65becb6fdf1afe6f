// tools/ubench_issue.hip -- instruction ISSUE model of a gfx950 SIMD with 1, 2, 4, 8 resident wavefronts.
// k_maniac_decode uses a wavefront as one scalar processor; with several of them per SIMD the question is what
// they compete for (scalar issue, vector issue, branch unit, LDS).  Every test body is 64 instructions (or
// groups) per loop iteration; the kernel is launched with 1024*k single-wave workgroups (k per SIMD on 256 CUs)
// and every wave reports the s_memtime cycles it needed.  Output: cycles per instruction per wave, and the
// aggregate issue rate per SIMD (instructions per cycle) = k / that.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o /tmp/ubench_issue && /tmp/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }

constexpr int NT = 16;

__global__ __launch_bounds__(64) void k(unsigned long long *out, int *buf, int iters, int test) {
    __shared__ int lds[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) lds[i] = (i * 7 + 1) & 1023;
    __syncthreads();
    int s = __builtin_amdgcn_readfirstlane(buf[0]);
    int s1 = s + 1, s2 = s + 2, s3 = s + 3;
    int v = buf[lane], v1 = v + 1, v2 = v + 2, v3 = v + 3;
    unsigned long long w64 = (unsigned)v;
    int r = 0;
    unsigned long long t0 = now();
    switch (test) {
    case 0:  // dependent SALU
        for (int i = 0; i < iters; i++) asm volatile(REP64("s_add_i32 %0, %0, 1\n") : "+s"(s) : : "scc");
        break;
    case 1:  // dependent VALU
        for (int i = 0; i < iters; i++) asm volatile(REP64("v_add_u32 %0, %0, 1\n") : "+v"(v));
        break;
    case 2:  // alternating independent SALU / VALU chains (32 + 32)
        for (int i = 0; i < iters; i++) asm volatile(REP16("s_add_i32 %0, %0, 1\n v_add_u32 %1, %1, 1\n s_add_i32 %0, %0, 1\n v_add_u32 %1, %1, 1\n") : "+s"(s), "+v"(v) : : "scc");
        break;
    case 3:  // 4 independent SALU chains
        for (int i = 0; i < iters; i++) asm volatile(REP16("s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1\n s_add_i32 %2, %2, 1\n s_add_i32 %3, %3, 1\n") : "+s"(s), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        break;
    case 4:  // 4 independent VALU chains
        for (int i = 0; i < iters; i++) asm volatile(REP16("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n") : "+v"(v), "+v"(v1), "+v"(v2), "+v"(v3));
        break;
    case 5:  // v_readlane -> s_add -> v_mov round trip (64 triples = 192 instructions)
        for (int i = 0; i < iters; i++) asm volatile(REP64("v_readlane_b32 %1, %0, 3\n s_add_i32 %1, %1, 1\n v_mov_b32 %0, %1\n") : "+v"(v), "+s"(s) : : "scc");
        break;
    case 6:  // not-taken branches: 32 x (s_cmp, s_cbranch)
        for (int i = 0; i < iters; i++) asm volatile(REP16("s_cmp_eq_u32 0, 1\n s_cbranch_scc1 1f\n s_cmp_eq_u32 0, 1\n s_cbranch_scc1 1f\n") "1:\n" ::: "scc");
        break;
    case 7:  // taken branches: 32 x (s_cmp, s_cbranch over nothing)
        for (int i = 0; i < iters; i++) asm volatile(REP16("s_cmp_eq_u32 0, 0\n s_cbranch_scc1 1f\n 1:\n s_cmp_eq_u32 0, 0\n s_cbranch_scc1 2f\n 2:\n") ::: "scc");
        break;
    case 8:  // dependent v_mad_u64_u32
        for (int i = 0; i < iters; i++) asm volatile(REP64("v_mad_u64_u32 %0, vcc, %1, 3, %0\n") : "+v"(w64) : "v"(v) : "vcc");
        break;
    case 9:  // dependent v_mad_u32_u24
        for (int i = 0; i < iters; i++) asm volatile(REP64("v_mad_u32_u24 %0, %0, 3, %1\n") : "+v"(v) : "v"(v1));
        break;
    case 10: // s_nop 0
        for (int i = 0; i < iters; i++) asm volatile(REP64("s_nop 0\n"));
        break;
    case 11: // dependent ds_bpermute
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k2 = 0; k2 < 64; k2++) v = __builtin_amdgcn_ds_bpermute(((v + 1) & 63) << 2, v);
        }
        break;
    case 12: // dependent uniform ds_read + readfirstlane
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k2 = 0; k2 < 64; k2++) s = __builtin_amdgcn_readfirstlane(lds[s & 1023]);
        }
        break;
    case 13: // v_cmp -> s_cbranch_vccz (not taken) pairs: 32 pairs
        for (int i = 0; i < iters; i++) asm volatile(REP16("v_cmp_gt_u32 vcc, 0, %0\n s_cbranch_vccnz 1f\n v_cmp_gt_u32 vcc, 0, %0\n s_cbranch_vccnz 1f\n") "1:\n" : : "v"(v) : "vcc");
        break;
    case 14: // the shape of one binary decision, scalar flavour: 16 x (readlane, sub, cmp, cselect, cselect, sub, cmp, branch-not-taken) = 128 instr
        for (int i = 0; i < iters; i++) asm volatile(REP16("v_readlane_b32 %2, %0, 5\n s_sub_u32 %3, %1, %2\n s_cmp_ge_u32 %1, %2\n s_cselect_b32 %1, %3, %2\n s_cselect_b32 %3, %2, 0\n s_sub_u32 %1, %1, %3\n s_cmp_le_u32 %1, 0x10000\n s_cbranch_scc1 1f\n") "1:\n"
                     : "+v"(v), "+s"(s), "+s"(s1), "+s"(s2) : : "scc");
        break;
    case 15: // the same decision, vector flavour with uniform values in VGPRs: 16 x (sub, cmp, cndmask, cndmask, sub, cmp, branch vcc) = 7 per group = 112 instr
        for (int i = 0; i < iters; i++) asm volatile(REP16("v_sub_u32 %2, %0, %1\n v_cmp_ge_u32 vcc, %0, %1\n v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %2, 0, %1, vcc\n v_sub_u32 %0, %0, %2\n v_cmp_gt_u32 vcc, 0x10000, %0\n s_cbranch_vccnz 1f\n") "1:\n"
                     : "+v"(v), "+v"(v1), "+v"(v2) : : "vcc");
        break;
    }
    unsigned long long t1 = now();
    r += s + s1 + s2 + s3 + v + v1 + v2 + v3 + (int)w64;
    if (lane == 0) out[blockIdx.x] = t1 - t0;
    if (r == 0x7fffffff) buf[lane] = r;
}

int main() {
    int *buf; unsigned long long *out;
    const int maxblocks = 1024 * 8;
    hipMalloc(&buf, 8192 * 4); hipMalloc(&out, maxblocks * 8);
    std::vector<int> h(8192);
    for (int i = 0; i < 8192; i++) h[i] = (i * 13 + 5) & 1023;
    hipMemcpy(buf, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    const char *names[NT] = {"dependent s_add", "dependent v_add", "alternating s_add/v_add", "4 independent s_add", "4 independent v_add",
                             "readlane+s_add+v_mov (per instr)", "s_cmp+branch not taken (per instr)", "s_cmp+branch taken (per instr)", "dependent v_mad_u64_u32",
                             "dependent v_mad_u32_u24", "s_nop 0", "dependent ds_bpermute", "dependent ds_read+readfirstlane (per hop)", "v_cmp+cbranch_vccnz not taken (per instr)",
                             "scalar decision (8 instr group, per instr)", "vector decision (7 instr group, per instr)"};
    const int per[NT] = {64, 64, 64, 64, 64, 192, 64, 64, 64, 64, 64, 64, 64, 64, 128, 112};
    const int iters = 400;
    std::vector<unsigned long long> o(maxblocks);
    printf("%-46s %10s %10s %10s %10s   (cycles per instruction per wave; aggregate instr/cycle/SIMD in brackets)\n", "waves per SIMD ->", "1", "2", "4", "8");
    for (int t = 0; t < NT; t++) {
        printf("%-46s", names[t]);
        for (int kk : {1, 2, 4, 8}) {
            const int blocks = 1024 * kk;
            hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, buf, iters, t);
            hipDeviceSynchronize();
            hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, buf, iters, t);
            hipDeviceSynchronize();
            hipMemcpy(o.data(), out, blocks * 8, hipMemcpyDeviceToHost);
            std::sort(o.begin(), o.begin() + blocks);
            const double med = (double)o[blocks / 2] / ((double)iters * per[t]);
            printf(" %5.2f[%4.2f]", med, kk / med);
        }
        printf("\n");
    }
    return 0;
}
