// tools/emu/emu_runtime.cpp -- fiber scheduler behind tools/emu/hip/hip_runtime.h (TEST INFRASTRUCTURE).
// x86-64 SysV only: a 12-instruction context switch keeps a rendezvous of 64 fibers at ~1 us.
#include <atomic>
#include <csignal>
#include <cstdio>
#include <execinfo.h>
#include <sched.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "hip/hip_runtime.h"

extern "C" void emu_switch(void **save_sp, void *new_sp);
asm(".text\n"
    ".globl emu_switch\n"
    ".type emu_switch,@function\n"
    "emu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n");

namespace emu {
namespace {
constexpr size_t kStack = 1 << 20;
struct Fiber {
    Fiber() = default;
    Fiber(Fiber &&o) noexcept { *this = std::move(o); }
    Fiber &operator=(Fiber &&o) noexcept {
        sp = o.sp; stack = o.stack; done = o.done; site = o.site;
        tid[0] = o.tid[0]; tid[1] = o.tid[1]; tid[2] = o.tid[2];
        o.stack = nullptr;
        return *this;
    }
    ~Fiber() { free(stack); }   // a worker thread's fibers go away with the thread
    void *sp = nullptr;
    char *stack = nullptr;
    bool done = true;
    unsigned tid[3] = {0, 0, 0};
    const void *site = nullptr;   // debugging aid: who called the cross-lane operation this fiber is waiting in
};
// Everything below is per OS thread: with EMU_THREADS > 1 the workgroups of a launch run on several threads at once
// (one workgroup at a time per thread, its lanes as fibers of that thread), which is how the tile-to-tile hand-off
// of the entropy kernel gets real concurrency on a CPU.
thread_local std::vector<Fiber> fibers;
thread_local void *main_sp = nullptr;
thread_local int cur = -1, n_alive = 0, n_fibers = 0;   // n_fibers: fibers of the running block (the vector only ever grows)
thread_local dim3 g_grid, g_block, g_bidx;
thread_local const std::function<void()> *g_body = nullptr;
// generation barrier + double-buffered exchange slots
thread_local int bar_count = 0;
thread_local unsigned long long bar_gen = 0;
thread_local long long slot[2][1024];
thread_local unsigned long long stamp[2][1024];   // generation in which a lane deposited: who took part is fixed at the rendezvous, not when a lane reads
thread_local bool threaded = false;

void switch_to(int next) {
    int prev = cur;
    cur = next;
    void **save = prev < 0 ? &main_sp : &fibers[prev].sp;
    void *to = next < 0 ? main_sp : fibers[next].sp;
    emu_switch(save, to);
}
int next_alive(int from) {
    const int n = n_fibers;
    for (int k = 1; k <= n; k++) {
        int i = (from + k) % n;
        if (!fibers[i].done) return i;
    }
    return -1;
}
void release_barrier_if_complete() {
    if (bar_count > 0 && bar_count >= n_alive) { bar_count = 0; bar_gen++; }
}
extern "C" void emu_fiber_main() {
    (*g_body)();
    Fiber &f = fibers[cur];
    f.done = true;
    n_alive--;
    release_barrier_if_complete();   // fibers that returned no longer take part in barriers
    int nx = next_alive(cur);
    switch_to(nx);                   // never returns here
    abort();
}
void rendezvous() {
#ifdef EMU_TRACE_SITES   // needs -fno-omit-frame-pointer
    fibers[cur].site = __builtin_return_address(1);
#endif
    const unsigned long long g = bar_gen;
    if (++bar_count >= n_alive) { bar_count = 0; bar_gen++; return; }
    while (bar_gen == g) {
        int nx = next_alive(cur);
        if (nx < 0 || nx == cur) { fprintf(stderr, "emu: deadlock in a cross-lane operation (divergent control flow?)\n"); abort(); }
        switch_to(nx);
    }
}
}  // namespace

unsigned coord(int which) {
    const Fiber &f = fibers[cur];
    switch (which) {
        case 0: return f.tid[0]; case 1: return f.tid[1]; case 2: return f.tid[2];
        case 3: return g_bidx.x; case 4: return g_bidx.y; case 5: return g_bidx.z;
        case 6: return g_block.x; case 7: return g_block.y; case 8: return g_block.z;
        case 9: return g_grid.x; case 10: return g_grid.y; default: return g_grid.z;
    }
}

void yield() {
    // s_sleep in a spin loop: every lane of the wavefront comes through here; let other THREADS (= other workgroups,
    // the producers this one may be waiting for) run once per pass of the wavefront
    if (threaded && fibers[cur].tid[0] == 0) sched_yield();
    int nx = next_alive(cur);
    if (nx >= 0 && nx != cur) switch_to(nx);
}
void sync() { rendezvous(); }

// every lane deposits its operand, all meet, every lane reads what it needs from the deposit of this generation
struct Exchanged { const long long *v; const unsigned long long *stamp; unsigned long long gen; bool took_part(int lane) const { return stamp[lane] == gen; } };
static Exchanged exchange(long long v) {
    const unsigned long long g = bar_gen;
    const unsigned p = (unsigned)(g & 1u);
    const unsigned lane = fibers[cur].tid[0];
    slot[p][lane] = v;
    stamp[p][lane] = g;
    rendezvous();
    return Exchanged{slot[p], stamp[p], g};
}
int readlane(int v, int lane) { return (int)exchange(v).v[lane & 63]; }
int readfirstlane(int v) {
    Exchanged e = exchange(v);
    for (int i = 0; i < n_fibers; i++) if (e.took_part(i)) return (int)e.v[i];   // lowest active lane
    return v;
}
int bpermute(int byte_addr, int v) { return (int)exchange(v).v[(byte_addr >> 2) & 63]; }
unsigned long long ballot(bool pr) {
    Exchanged e = exchange(pr ? 1 : 0);
    unsigned long long m = 0;
    for (int i = 0; i < n_fibers && i < 64; i++) if (e.took_part(i) && e.v[i]) m |= 1ull << i;
    return m;
}
bool any(bool pr) { return ballot(pr) != 0; }

// EMU_ALARM=<seconds>: print the native stack of whatever is running then and abort (a hung emulation has no other handle)
static void on_alarm(int) {
    void *frames[64];
    int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    fprintf(stderr, "emu: fiber %d running; barrier %d of %d arrived, generation %llu\n", cur, bar_count, n_alive, bar_gen);
    for (int i = 0; i < n_fibers; i++) {
        void *a[1] = {const_cast<void *>(fibers[i].site)};
        fprintf(stderr, "  fiber %2d %s last cross-lane call from ", i, fibers[i].done ? "done" : "live");
        backtrace_symbols_fd(a, 1, 2);
    }
    _exit(97);
}
static void run_block(dim3 grid, dim3 block, dim3 bidx, const std::function<void()> &body) {
    const unsigned nt = block.x * block.y * block.z;
    if (fibers.size() < nt) fibers.resize(nt);
    n_fibers = (int)nt;
    g_grid = grid; g_block = block; g_body = &body; g_bidx = bidx;
    n_alive = (int)nt; bar_count = 0;
    memset(stamp, 0xff, sizeof(stamp));
    for (unsigned t = 0; t < nt; t++) {
        Fiber &f = fibers[t];
        if (!f.stack) f.stack = (char *)malloc(kStack);
        f.done = false;
        f.tid[0] = t % block.x; f.tid[1] = (t / block.x) % block.y; f.tid[2] = t / (block.x * block.y);
        // initial frame: six callee-saved registers (zero) + return address = emu_fiber_main;
        // after the `ret`, rsp is 16n+8 as the ABI expects at a function's first instruction
        uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
        void **sp = (void **)(top - 8);
        *--sp = (void *)emu_fiber_main;
        for (int k = 0; k < 6; k++) *--sp = nullptr;
        f.sp = sp;
    }
    cur = -1;
    switch_to(0);        // returns when the last fiber of the block has finished
    g_body = nullptr;
}

void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
    static bool armed = false;
    if (!armed) {
        armed = true;
        if (const char *e = getenv("EMU_ALARM")) { signal(SIGALRM, on_alarm); alarm((unsigned)atoi(e)); }
    }
    const unsigned nt = block.x * block.y * block.z;
    if (nt > 1024 || nt == 0) abort();
    const unsigned long long blocks = (unsigned long long)grid.x * grid.y * grid.z;
    int want = 1;
    if (const char *e = getenv("EMU_THREADS")) want = atoi(e);
    if (want > 1 && blocks > 1 && blocks <= (unsigned long long)want) {
        // Few workgroups that may wait for each other (the persistent wavefronts of the entropy kernel): one OS thread
        // each, all running at once.  Big grids (the inverse transforms: independent blocks) stay on this thread.
        std::atomic<unsigned long long> next{0};
        const int nthreads = (int)std::min<unsigned long long>((unsigned long long)want, blocks);
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; t++)
            pool.emplace_back([&]() {
                threaded = true;
                for (;;) {
                    const unsigned long long k = next.fetch_add(1);
                    if (k >= blocks) break;
                    run_block(grid, block, dim3((unsigned)(k % grid.x), (unsigned)((k / grid.x) % grid.y), (unsigned)(k / ((unsigned long long)grid.x * grid.y))), body);
                }
            });
        for (auto &th : pool) th.join();
        return;
    }
    threaded = false;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) run_block(grid, block, dim3(bx, by, bz), body);
}
}  // namespace emu
