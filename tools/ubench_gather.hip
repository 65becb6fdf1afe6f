// tools/ubench_gather.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on the access pattern of k_maniac_decode's
// leaf traffic: per wavefront instruction, 32 lanes read (or write) 2 bytes each of ONE 64-byte record at a pseudo-random
// 64-byte-aligned position of a buffer far larger than L2 + Infinity Cache, so every record is one 64-byte line from HBM.
// MI355X_MICROARCH.md calibrates the counters for wide coalesced streams only ("calibrate in your own access pattern").
// Round 3 adds the kernel's OTHER pattern, `snode`: a 512-byte supernode read by one global_load_dwordx2 of 64 lanes (a wide coalesced
// access, the kind the guide says FETCH_SIZE under-reports by 2) at a pseudo-random 512-byte-aligned position.
// Round 6 adds the patterns of the compact context layout: `snode4`, a 256-byte NARROW supernode read by one global_load_dword of 64 lanes, and
// `read32` / `write32`, a 32-byte compact leaf read / written as 2 bytes x 16 lanes (lanes 16..63 mirror them, as in the kernel).
//   ./ubench_gather read|write|snode|snode4|read32|write32 <records per wavefront>      prints the byte count the counters should show
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__global__ __launch_bounds__(64) void k_gather_read(const unsigned short *buf, unsigned long long n_records, int per_wave, unsigned *sink) {
    const int lane = threadIdx.x;
    unsigned long long x = 0x9E3779B97F4A7C15ull * (blockIdx.x + 1);
    unsigned acc = 0;
    for (int i = 0; i < per_wave; i++) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        const unsigned long long rec = (x >> 20) % n_records;
        if (lane < 32) acc += buf[rec * 32 + lane];      // 64-byte record, one global_load_ushort per lane
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}
__global__ __launch_bounds__(64) void k_gather_snode(const uint2 *buf, unsigned long long n_records, int per_wave, unsigned *sink) {
    const int lane = threadIdx.x;
    unsigned long long x = 0x9E3779B97F4A7C15ull * (blockIdx.x + 1);
    unsigned acc = 0;
    for (int i = 0; i < per_wave; i++) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        const unsigned long long rec = (x >> 20) % n_records;
        const uint2 v = buf[rec * 64 + lane];            // 512-byte record, one global_load_dwordx2 per lane
        acc += v.x + v.y;
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}
__global__ __launch_bounds__(64) void k_gather_snode4(const unsigned *buf, unsigned long long n_records, int per_wave, unsigned *sink) {
    const int lane = threadIdx.x;
    unsigned long long x = 0x9E3779B97F4A7C15ull * (blockIdx.x + 1);
    unsigned acc = 0;
    for (int i = 0; i < per_wave; i++) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        const unsigned long long rec = (x >> 20) % n_records;
        acc += buf[rec * 64 + lane];                     // 256-byte record, one global_load_dword per lane
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}
__global__ __launch_bounds__(64) void k_gather_read32(const unsigned short *buf, unsigned long long n_records, int per_wave, unsigned *sink) {
    const int lane = threadIdx.x;
    unsigned long long x = 0x9E3779B97F4A7C15ull * (blockIdx.x + 1);
    unsigned acc = 0;
    for (int i = 0; i < per_wave; i++) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        const unsigned long long rec = (x >> 20) % n_records;
        acc += buf[rec * 16 + (lane & 15)];              // 32-byte record, every lane reads its mirror of 16 chances
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}
__global__ __launch_bounds__(64) void k_gather_write32(unsigned short *buf, unsigned long long n_records, int per_wave) {
    const int lane = threadIdx.x;
    unsigned long long x = 0x9E3779B97F4A7C15ull * (blockIdx.x + 1);
    for (int i = 0; i < per_wave; i++) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        const unsigned long long rec = (x >> 20) % n_records;
        buf[rec * 16 + (lane & 15)] = (unsigned short)(i + (lane & 15));
    }
}
__global__ __launch_bounds__(64) void k_gather_write(unsigned short *buf, unsigned long long n_records, int per_wave) {
    const int lane = threadIdx.x;
    unsigned long long x = 0x9E3779B97F4A7C15ull * (blockIdx.x + 1);
    for (int i = 0; i < per_wave; i++) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        const unsigned long long rec = (x >> 20) % n_records;
        if (lane < 32) buf[rec * 32 + lane] = (unsigned short)(i + lane);
    }
}

int main(int argc, char **argv) {
    const bool wr = argc > 1 && !strcmp(argv[1], "write");
    const bool sn = argc > 1 && !strcmp(argv[1], "snode");
    const bool sn4 = argc > 1 && !strcmp(argv[1], "snode4"), r32 = argc > 1 && !strcmp(argv[1], "read32"), w32 = argc > 1 && !strcmp(argv[1], "write32");
    const int per_wave = argc > 2 ? atoi(argv[2]) : 20000;
    const unsigned long long bytes = 8ull << 30;          // 8 GiB >> 32 MiB of L2 + 256 MiB of Infinity Cache
    const double rec_bytes = sn ? 512.0 : sn4 ? 256.0 : (r32 || w32) ? 32.0 : 64.0;
    const unsigned long long n_records = bytes / (unsigned long long)rec_bytes;
    unsigned short *buf; unsigned *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    const int waves = 4096;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    if (sn4) hipLaunchKernelGGL(k_gather_snode4, dim3(waves), dim3(64), 0, 0, reinterpret_cast<const unsigned *>(buf), n_records, per_wave, sink);
    else if (r32) hipLaunchKernelGGL(k_gather_read32, dim3(waves), dim3(64), 0, 0, buf, n_records, per_wave, sink);
    else if (w32) hipLaunchKernelGGL(k_gather_write32, dim3(waves), dim3(64), 0, 0, buf, n_records, per_wave);
    else if (sn) hipLaunchKernelGGL(k_gather_snode, dim3(waves), dim3(64), 0, 0, reinterpret_cast<const uint2 *>(buf), n_records, per_wave, sink);
    else if (wr) hipLaunchKernelGGL(k_gather_write, dim3(waves), dim3(64), 0, 0, buf, n_records, per_wave);
    else hipLaunchKernelGGL(k_gather_read, dim3(waves), dim3(64), 0, 0, buf, n_records, per_wave, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double known = (double)waves * per_wave * rec_bytes;
    printf("{\"pattern\": \"%s\", \"records\": %llu, \"known_bytes\": %.0f, \"ms\": %.3f, \"GBps\": %.1f}\n",
           sn4 ? "read of 256-byte records, 4 bytes x 64 lanes" : r32 ? "read of 32-byte records, 2 bytes x 16 lanes (mirrored)" : w32 ? "write of 32-byte records, 2 bytes x 16 lanes (mirrored)" :
           sn ? "read of 512-byte records, 8 bytes x 64 lanes" : wr ? "write of 64-byte records, 2 bytes x 32 lanes" : "read of 64-byte records, 2 bytes x 32 lanes",
           (unsigned long long)waves * per_wave, known, ms, known / ms / 1e6);
    return 0;
}
