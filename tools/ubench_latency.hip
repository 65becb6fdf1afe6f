// tools/ubench_latency.hip -- ANALYSIS TOOLING: latency of the entropy kernel's two dependent fetches per symbol (a 512-byte
// supernode read by 64 lanes x 8 bytes, then a 64-byte leaf read by 32 lanes x 2 bytes) as a function of how much memory all
// resident wavefronts touch together.  Each wavefront chases pseudo-random slots inside its own region; the next address
// depends on the loaded data, as in the tree walk.  hipcc --offload-arch=gfx950 -O3 -o tools/ubench_latency_bin tools/ubench_latency.hip
//   ubench_latency_bin  -> table: wavefronts x region size per wavefront -> shader cycles per (supernode + leaf) pair
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// prefetch = 1: the leaf line is touched (by 8 lanes, result unused) together with the supernode fetch, the leaf itself is read after the
// supernode has arrived plus ~35 dependent VALU instructions (the second walk round): what "leaf slots next to their supernode,
// prefetched with it" would make of the second fetch
template <int kPrefetch>
__global__ __launch_bounds__(64) void chase(const uint2 *base, size_t region_slots, int iters, unsigned long long *out) {
    const int lane = threadIdx.x;
    const uint2 *region = base + (size_t)blockIdx.x * region_slots * 64;
    const unsigned short *leaves = reinterpret_cast<const unsigned short *>(region);
    unsigned state = blockIdx.x * 2654435761u + 12345u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        const size_t slot = state % region_slots;
        unsigned nstate = state * 1664525u + 1013904223u;
        size_t lslot = (nstate >> 7) % (region_slots * 8);
        unsigned short pf = 0;
        if (kPrefetch && lane < 8) pf = leaves[lslot * 32 + lane * 4];
        const uint2 v = region[slot * 64 + lane];
        unsigned x = (unsigned)__builtin_amdgcn_readfirstlane((int)v.x);
        if (kPrefetch) {
            unsigned a = x + lane;
#pragma unroll
            for (int k = 0; k < 35; k++) a = a * 3u + (a >> 7);   // dependent VALU chain standing in for the second walk round
            x += (unsigned)__builtin_amdgcn_readfirstlane((int)(a & 0u));
        } else {
            nstate += x;
            lslot = (nstate >> 7) % (region_slots * 8);
        }
        state = nstate + x;
        unsigned short l = 0;
        if (lane < 32) l = leaves[lslot * 32 + lane];
        state += (unsigned)__builtin_amdgcn_readfirstlane((int)(l + pf)) + 1u;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { out[blockIdx.x] = t1 - t0; out[8192 + blockIdx.x] = state; }   // (state is stored so that the loop is not dead code)
}

int main() {
    const size_t max_bytes = (size_t)3 << 30;
    uint2 *buf = nullptr;
    unsigned long long *out = nullptr;
    if (hipMalloc((void **)&buf, max_bytes) != hipSuccess || hipMalloc((void **)&out, 8 * 2 * 8192) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, max_bytes);
    const int waves_list[] = {384, 1024, 3072, 6144};
    const size_t kb_list[] = {16, 40, 100, 300, 480};
    printf("%8s %10s %10s %14s %14s\n", "waves", "KB/wave", "total MB", "cycles/pair", "with prefetch");
    for (int waves : waves_list)
        for (size_t kb : kb_list) {
            const size_t slots = kb * 1024 / 512;
            if ((size_t)waves * slots * 512 > max_bytes) continue;
            const int iters = 20000;
            double res[2];
            for (int pf = 0; pf < 2; pf++) {
                if (pf) { hipLaunchKernelGGL(chase<1>, dim3(waves), dim3(64), 0, 0, buf, slots, 2000, out); hipLaunchKernelGGL(chase<1>, dim3(waves), dim3(64), 0, 0, buf, slots, iters, out); }
                else { hipLaunchKernelGGL(chase<0>, dim3(waves), dim3(64), 0, 0, buf, slots, 2000, out); hipLaunchKernelGGL(chase<0>, dim3(waves), dim3(64), 0, 0, buf, slots, iters, out); }
                if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
                std::vector<unsigned long long> h(waves);
                hipMemcpy(h.data(), out, 8 * waves, hipMemcpyDeviceToHost);
                double s = 0;
                for (auto v : h) s += (double)v;
                res[pf] = s / waves / iters;
            }
            printf("%8d %10zu %10.0f %14.0f %14.0f\n", waves, kb, waves * kb / 1024.0, res[0], res[1]);
        }
    return 0;
}
