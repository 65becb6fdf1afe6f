// tools/ubench.hip -- single-wave instruction timing on gfx950 (what bounds k_maniac_decode).
// hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }

__global__ void k(unsigned long long *out, int *buf, int iters) {
    __shared__ int lds[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) lds[i] = (i * 7 + 1) & 4095;
    __syncthreads();
    unsigned long long t0, t1;
    int s = __builtin_amdgcn_readfirstlane(buf[0]);
    int v = buf[lane];
    int r = 0;
    // 0: dependent SALU adds
    t0 = now();
    for (int i = 0; i < iters; i++) { asm volatile(REP64("s_add_i32 %0, %0, 1\n") : "+s"(s) : : "scc"); }
    t1 = now(); if (lane == 0) out[0] = (t1 - t0); 
    // 1: dependent VALU adds
    t0 = now();
    for (int i = 0; i < iters; i++) { asm volatile(REP64("v_add_u32 %0, %0, 1\n") : "+v"(v)); }
    t1 = now(); if (lane == 0) out[1] = (t1 - t0);
    // 2: readlane -> salu -> v_mov chain (valu<->salu ping-pong), 64 x (v_readlane, s_add, v_mov)
    t0 = now();
    for (int i = 0; i < iters; i++) { asm volatile(REP64("v_readlane_b32 %1, %0, 3\n s_add_i32 %1, %1, 1\n v_mov_b32 %0, %1\n") : "+v"(v), "+s"(s) : : "scc"); }
    t1 = now(); if (lane == 0) out[2] = (t1 - t0);
    // 3: ds_read dependent chain (pointer chasing, uniform address)
    int p = 0;
    t0 = now();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k2 = 0; k2 < 64; k2++) p = __builtin_amdgcn_readfirstlane(lds[p]);
    }
    t1 = now(); if (lane == 0) out[3] = (t1 - t0);
    r += p;
    // 4: ds_bpermute dependent chain
    int q = lane;
    t0 = now();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k2 = 0; k2 < 64; k2++) q = __builtin_amdgcn_ds_bpermute(((q + 1) & 63) << 2, q);
    }
    t1 = now(); if (lane == 0) out[4] = (t1 - t0);
    r += q;
    // 5: taken scalar branches: 64 x (s_cmp, s_cbranch taken over one instruction)
    t0 = now();
    for (int i = 0; i < iters; i++) { asm volatile(REP64("s_cmp_eq_u32 0, 0\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n") ::: "scc"); }
    t1 = now(); if (lane == 0) out[5] = (t1 - t0);
    // 6: not-taken scalar branches
    t0 = now();
    for (int i = 0; i < iters; i++) { asm volatile(REP64("s_cmp_eq_u32 0, 1\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n") ::: "scc"); }
    t1 = now(); if (lane == 0) out[6] = (t1 - t0);
    // 7: global load dependent chain (L2-resident 16 KB table, uniform address)
    int g = 0;
    t0 = now();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k2 = 0; k2 < 16; k2++) g = __builtin_amdgcn_readfirstlane(buf[(g + k2 * 64) & 4095]);
    }
    t1 = now(); if (lane == 0) out[7] = (t1 - t0);
    r += g;
    // 8: v_cmp -> ballot -> s_ff1 -> readlane chain (the walk's resolution step)
    int e = 0;
    t0 = now();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k2 = 0; k2 < 64; k2++) { unsigned long long m = __ballot(v + e > lane); e = __builtin_ctzll(m | (1ull << 63)) & 3; e = __builtin_amdgcn_readlane(v, e) & 3; }
    }
    t1 = now(); if (lane == 0) out[8] = (t1 - t0);
    r += e;
    // 9: independent SALU (4 chains)
    int s1 = s, s2 = s + 1, s3 = s + 2, s4 = s + 3;
    t0 = now();
    for (int i = 0; i < iters; i++) { asm volatile(REP16("s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1\n s_add_i32 %2, %2, 1\n s_add_i32 %3, %3, 1\n") : "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4) : : "scc"); }
    t1 = now(); if (lane == 0) out[9] = (t1 - t0);
    // 10: s_mul_i32 dependent
    t0 = now();
    for (int i = 0; i < iters; i++) { asm volatile(REP64("s_mul_i32 %0, %0, 3\n") : "+s"(s) : : "scc"); }
    t1 = now(); if (lane == 0) out[10] = (t1 - t0);
    // 11: v_mad_u64_u32 dependent
    unsigned long long w64 = v;
    t0 = now();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k2 = 0; k2 < 64; k2++) w64 = (unsigned long long)(unsigned)w64 * 3u + 0x800ull;
    }
    t1 = now(); if (lane == 0) out[11] = (t1 - t0);
    r += (int)w64;
    buf[4096 + lane] = r + s + v + s1 + s2 + s3 + s4;
}

int main() {
    int *buf; unsigned long long *out;
    hipMalloc(&buf, 8192 * 4); hipMalloc(&out, 16 * 8);
    std::vector<int> h(8192);
    for (int i = 0; i < 8192; i++) h[i] = (i * 13 + 5) & 4095;
    hipMemcpy(buf, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    const int iters = 200;
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, buf, iters); hipDeviceSynchronize(); }
    unsigned long long o[16];
    hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
    const char *names[] = {"dependent s_add", "dependent v_add", "readlane+s_add+v_mov (per triple)", "ds_read chain (per hop)", "ds_bpermute chain (per hop)",
                           "taken branch (cmp+branch)", "not-taken branch (cmp+branch+nop)", "global load chain (per hop, /16)", "cmp+ballot+ctz+readlane (per step)",
                           "4 independent s_add (per 4)", "dependent s_mul_i32", "dependent 64-bit mad"};
    const int per[] = {64, 64, 64, 64, 64, 64, 64, 16, 64, 16, 64, 64};
    for (int i = 0; i < 12; i++) printf("%-40s %8.1f cycles\n", names[i], (double)o[i] / (iters * per[i]));
    return 0;
}
