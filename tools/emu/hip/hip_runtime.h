// tools/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE: a single-threaded stand-in for the HIP runtime and the
// gfx950 wavefront intrinsics the kernels of fuif_amd/csrc use, so that the SAME kernel sources can be compiled
// with g++ (-DFUIF_EMU -I tools/emu) into tests/_emu/libfuifgpu_emu.so and their LOGIC checked against the
// golden vectors on machines without a GPU (tests/test_emulated_kernels.py).
//
// What it is: every workgroup runs as blockDim.x cooperative fibers on one OS thread; a cross-lane operation
// (readlane, readfirstlane, ballot, ds_bpermute, __syncthreads) is a rendezvous of all fibers of the workgroup.
// By default workgroups run one after the other and the entropy kernel gets ONE persistent wavefront (the work list
// is in dependency order: a tile's producers have finished before it starts).  EMU_WAVES=n EMU_THREADS=n runs n
// persistent wavefronts on n OS threads at once, so tiles really wait for each other (logic of the hand-off; the
// fences are a GPU matter).
// What it is NOT: a performance model, a memory-model checker (the tile-to-tile hand-off protocol is only
// exercised on the GPU), or a product path -- libfuifgpu.so never contains any of this and still fails without a GPU.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local   // one workgroup at a time per OS thread (EMU_THREADS)
#define __constant__ static
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { unsigned x, y; };
struct alignas(16) int4 { int x, y, z, w; };
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

namespace emu {
unsigned coord(int which);                       // 0-2 threadIdx, 3-5 blockIdx, 6-8 blockDim, 9-11 gridDim
struct Coord { int which; operator unsigned() const { return coord(which); } };
struct Triple { Coord x, y, z; };
void launch(dim3 grid, dim3 block, const std::function<void()> &body);
void yield();
void sync();
int readlane(int v, int lane);
int readfirstlane(int v);
int bpermute(int byte_addr, int v);
unsigned long long ballot(bool p);
bool any(bool p);
}  // namespace emu
static const emu::Triple threadIdx = {{0}, {1}, {2}}, blockIdx = {{3}, {4}, {5}}, blockDim = {{6}, {7}, {8}}, gridDim = {{9}, {10}, {11}};

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu::launch((grid), (block), [&]() { (kernel)(__VA_ARGS__); })

// ---- wavefront intrinsics ---------------------------------------------------------------------
#define __builtin_amdgcn_readfirstlane(v) emu::readfirstlane(v)
#define __builtin_amdgcn_readlane(v, l) emu::readlane((v), (l))
#define __builtin_amdgcn_ds_bpermute(a, v) emu::bpermute((a), (v))
#define __builtin_amdgcn_s_sleep(n) emu::yield()
#define __builtin_readcyclecounter() 0ull
#define __ballot(p) emu::ballot(p)
#define __any(p) emu::any(p)
#define __syncthreads() emu::sync()
#define __popcll(x) __builtin_popcountll(x)
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_SEQ_CST)
template <class T> static inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicCAS(T *p, T expected, T desired) { __atomic_compare_exchange_n(p, &expected, desired, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); return expected; }
template <class T> static inline T atomicMax(T *p, T v) {
    T o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (v > o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return o;
}
// round-to-nearest without contraction: the emulator is built with -ffp-contract=off
static inline double __dadd_rn(double a, double b) { return a + b; }
static inline double __dsub_rn(double a, double b) { return a - b; }
static inline double __dmul_rn(double a, double b) { return a * b; }
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
using std::max;
using std::min;

// ---- host API -----------------------------------------------------------------------------------
typedef int hipError_t;
typedef void *hipStream_t;
typedef void *hipEvent_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
struct hipDeviceProp_t { int multiProcessorCount; };
static inline hipError_t hipMalloc(void **p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : 2; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "emulated HIP runtime"; }
// EMU_DEVICES=n: the emulated node has n "devices" (host memory all of them); the current device is per-thread state, as in HIP
static inline int emu_device_count() { const char *e = getenv("EMU_DEVICES"); const int n = e ? atoi(e) : 1; return n < 1 ? 1 : n; }
static inline int &emu_current_device() { static thread_local int d = 0; return d; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = emu_device_count(); return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = emu_current_device(); return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= emu_device_count()) return 101; emu_current_device() = d; return hipSuccess; }
static const hipError_t hipErrorPeerAccessAlreadyEnabled = 704;
static inline hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 1; return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
static inline hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t) { memmove(d, s, n); return hipSuccess; }
// EMU_WAVES: persistent wavefronts of the entropy kernel (each becomes a workgroup; run them on EMU_THREADS >= EMU_WAVES threads)
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { const char *e = getenv("EMU_WAVES"); p->multiProcessorCount = e ? std::max(1, atoi(e)) : 1; return hipSuccess; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }   // v_mul_hi_u32
static inline int __mul24(int a, int b) { return (int)((unsigned)(((a << 8) >> 8)) * (unsigned)(((b << 8) >> 8))); }   // v_mul_i32_i24: the low 24 bits of both factors, sign extended
static inline hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) { *free_b = (size_t)1 << 30; *total_b = (size_t)2 << 30; return hipSuccess; }
template <class K> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, K, int, size_t) { *n = 1; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
