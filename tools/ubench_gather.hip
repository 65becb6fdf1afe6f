// tools/ubench_gather.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on the access pattern of k_maniac_decode's
// leaf traffic: per wavefront instruction, 32 lanes read (or write) 2 bytes each of ONE 64-byte record at a pseudo-random
// 64-byte-aligned position of a buffer far larger than L2 + Infinity Cache, so every record is one 64-byte line from HBM.
// MI355X_MICROARCH.md calibrates the counters for wide coalesced streams only ("calibrate in your own access pattern").
//   ./ubench_gather read|write <records per wavefront>      prints the byte count the counters should show
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
    const int per_wave = argc > 2 ? atoi(argv[2]) : 20000;
    const unsigned long long bytes = 8ull << 30;          // 8 GiB >> 32 MiB of L2 + 256 MiB of Infinity Cache
    const unsigned long long n_records = bytes / 64;
    unsigned short *buf; unsigned *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    const int waves = 4096;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    if (wr) hipLaunchKernelGGL(k_gather_write, dim3(waves), dim3(64), 0, 0, buf, n_records, per_wave);
    else hipLaunchKernelGGL(k_gather_read, dim3(waves), dim3(64), 0, 0, buf, n_records, per_wave, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double known = (double)waves * per_wave * 64.0;
    printf("{\"pattern\": \"%s of 64-byte records, 2 bytes x 32 lanes\", \"records\": %llu, \"known_bytes\": %.0f, \"ms\": %.3f, \"GBps\": %.1f}\n",
           wr ? "write" : "read", (unsigned long long)waves * per_wave, known, ms, known / ms / 1e6);
    return 0;
}
