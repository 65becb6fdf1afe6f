// tools/ubench_context.hip -- ANALYSIS TOOLING (round 6): what a long tile's per-symbol memory chain costs as a function of the CONTEXT LAYOUT.
// Every wavefront plays one long tile: per "symbol" a supernode fetch (64 lanes x NODE bytes), with probability ~0.3 a second one that depends on
// it, then a leaf fetch (LEAF bytes, 32 lanes x 2 or 16 lanes x 2) whose address depends on the last supernode, ~100 dependent VALU instructions
// standing in for the symbol decoder, and the write-back of the leaf.  Addresses are pseudo-random inside the wavefront's own context region
// (supernodes | leaves), the next address depends on loaded data (as in the tree walk).  Regions are spread over the allocation the way the context
// arenas are.   hipcc --offload-arch=gfx950 -O3 -o build/ubench_context_bin tools/ubench_context.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int kNodeBytes, int kLeafBytes>
__global__ __launch_bounds__(64) void chase(unsigned char *base, size_t region_bytes, size_t queue_bytes, int per_queue, size_t leaf_gap, unsigned n_super, unsigned n_leaves, int iters, int alu, unsigned long long *out) {
    const int lane = threadIdx.x;
    // layout: `per_queue` consecutive regions inside a queue's arena of `queue_bytes` (the context arenas of capi.hip: 64 MiB per CU queue)
    unsigned char *region = base + (size_t)(blockIdx.x / per_queue) * queue_bytes + (size_t)(blockIdx.x % per_queue) * region_bytes;
    unsigned char *leaves = region + (leaf_gap ? leaf_gap : (size_t)n_super * 64 * kNodeBytes);   // leaf_gap: the leaves' fixed offset inside a wavefront's scratch area (rounds 1-5: 2 MB behind the supernodes)
    unsigned state = blockIdx.x * 2654435761u + 12345u;
    unsigned cur_leaf = 0;
    unsigned leafv = lane;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        unsigned sn = state % n_super;
        unsigned x;
        if (kNodeBytes == 8) { const uint2 v = reinterpret_cast<const uint2 *>(region)[(size_t)sn * 64 + lane]; x = (unsigned)__builtin_amdgcn_readfirstlane((int)(v.x + v.y)); }
        else { const unsigned v = reinterpret_cast<const unsigned *>(region)[(size_t)sn * 64 + lane]; x = (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
        state = state * 1664525u + 1013904223u + x;
        if ((state >> 11) % 10u < 3u) {   // a second round behind it
            sn = (state >> 5) % n_super;
            if (kNodeBytes == 8) { const uint2 v = reinterpret_cast<const uint2 *>(region)[(size_t)sn * 64 + lane]; x = (unsigned)__builtin_amdgcn_readfirstlane((int)(v.x + v.y)); }
            else { const unsigned v = reinterpret_cast<const unsigned *>(region)[(size_t)sn * 64 + lane]; x = (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
            state = state * 1664525u + 1013904223u + x;
        }
        const unsigned leaf = (state >> 7) % n_leaves;
        // leaf switch: fetch the new one, write the old one back (lanes beyond the record mirror the first ones, as in the kernel)
        const unsigned l2 = (unsigned)(lane & (kLeafBytes / 2 - 1)) * 2u;
        const unsigned fresh = *reinterpret_cast<const unsigned short *>(leaves + (size_t)leaf * kLeafBytes + l2);
        *reinterpret_cast<unsigned short *>(leaves + (size_t)cur_leaf * kLeafBytes + l2) = (unsigned short)leafv;
        cur_leaf = leaf;
        leafv = fresh;
        unsigned a = (unsigned)__builtin_amdgcn_readfirstlane((int)leafv) + lane;
        for (int k = 0; k < alu; k++) a = a * 3u + (a >> 7);   // the symbol decoder's dependent chain
        state += (unsigned)__builtin_amdgcn_readfirstlane((int)(a & 1u));
        leafv = (leafv + (a & 1u)) & 0xFFFu;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { out[blockIdx.x] = t1 - t0; out[8192 + blockIdx.x] = state; }
}

int main(int argc, char **argv) {
    const size_t max_bytes = (size_t)17 << 30;
    unsigned char *buf = nullptr;
    unsigned long long *out = nullptr;
    if (hipMalloc((void **)&buf, max_bytes) != hipSuccess || hipMalloc((void **)&out, 8 * 2 * 8192) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, max_bytes);
    const int alu = argc > 1 ? atoi(argv[1]) : 100;
    printf("# cycles per symbol; 1.3 supernode fetches + leaf fetch + leaf write-back + %d dependent VALU per symbol\n", alu);
    printf("%6s %6s %6s %8s %8s %9s %9s %8s %10s\n", "waves", "node_B", "leaf_B", "supern.", "leaves", "KB/wave", "total_MB", "layout", "cyc/symbol");
    struct Cfg { int node, leaf; unsigned ns, nl; };
    // a long 4K tile: ~120 supernodes, ~900 leaves
    const Cfg cfgs[] = {{8, 64, 120, 900}, {4, 64, 120, 900}, {8, 32, 120, 900}, {4, 32, 120, 900}, {4, 32, 60, 900}, {4, 32, 120, 450}, {8, 64, 30, 225}, {8, 64, 8, 64}};
    const int waves_list[] = {512, 1024, 2048, 3072, 6144};
    for (int waves : waves_list)
        for (const Cfg &c : cfgs) {
            const size_t region = ((size_t)c.ns * 64 * c.node + (size_t)c.nl * c.leaf + 255) / 256 * 256;
            for (int layout = 0; layout < 3; layout++) {
            // layout 0: the regions back to back; 1: as the context arenas place them -- 256 queues of 64 MiB, a queue's tiles 4 regions apart
            // layout 2: the per-wavefront scratch areas of tiles that are not suspendable (streams without group index): 4.7 MB apart, leaves 2 MB behind the supernodes
            const int per_queue = layout == 1 ? (waves + 255) / 256 : waves;
            const size_t stride = layout == 1 ? region * 4 : layout == 2 ? (size_t)4700 << 10 : region;
            const size_t queue_bytes = layout == 1 ? (size_t)64 << 20 : 0;
            const size_t leaf_gap = layout == 2 ? (size_t)2 << 20 : 0;
            if (layout == 1 && ((size_t)per_queue * stride > queue_bytes || (size_t)256 * queue_bytes > max_bytes)) continue;
            if (layout == 2 && (size_t)waves * stride > max_bytes) continue;
            const int iters = 20000;
            for (int pass = 0; pass < 2; pass++) {
                const int it = pass ? iters : 2000;
#define LAUNCH(N, L) hipLaunchKernelGGL((chase<N, L>), dim3(waves), dim3(64), 0, 0, buf, stride, queue_bytes, per_queue, leaf_gap, c.ns, c.nl, it, alu, out)
                if (c.node == 8 && c.leaf == 64) LAUNCH(8, 64); else if (c.node == 4 && c.leaf == 64) LAUNCH(4, 64); else if (c.node == 8) LAUNCH(8, 32); else LAUNCH(4, 32);
            }
            if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
            std::vector<unsigned long long> h(waves);
            hipMemcpy(h.data(), out, 8 * waves, hipMemcpyDeviceToHost);
            double s = 0;
            for (auto v : h) s += (double)v;
            printf("%6d %6d %6d %8u %8u %9.1f %9.0f %8s %10.0f\n", waves, c.node, c.leaf, c.ns, c.nl, region / 1024.0, waves * (double)region / 1048576.0, layout == 1 ? "arenas" : layout == 2 ? "scratch" : "tight", s / waves / iters);
            fflush(stdout);
            }
        }
    return 0;
}
