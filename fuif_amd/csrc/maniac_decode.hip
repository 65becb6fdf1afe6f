// fuif_amd/csrc/maniac_decode.hip -- MANIAC entropy decode of a batch of FUIF streams on gfx950.
//
// One 64-lane wavefront (= one workgroup) per TILE = a run of channel groups whose first byte is
// known.  The format has no intra-stream entry points (a channel group's first byte is only known
// once the previous group is fully decoded, maniac/rac.h:70-104), so a plain stream is one tile: an
// inherently serial chain, and the batch provides the parallelism (1024 streams = one wave per SIMD
// on 256 CUs).  A stream that carries a group index (index.cpp) is one tile per channel group; tiles
// of one image then run concurrently and hand decoded rows to each other (the properties of a pixel
// read up to 6 previously decoded channels, context_predict.h:233-289): see "tile-to-tile hand-off"
// below and DESIGN.md 4.1.  Persistent wavefronts take tiles from a work list in dependency order.
// Inside a tile the kernel's job is to make the per-symbol dependency chain as short as the hardware
// allows.  The wave is used as one scalar
// processor (range coder state, tree position: SGPRs via readlane/readfirstlane) plus a 64-lane
// vector unit:
//   * compressed bytes: one coalesced 256-byte window per refill, lane i holds dword i,
//     bytes are picked with v_readlane (no per-byte memory access);
//   * context tree: after parsing, the tree is cut into complete 6-level "supernodes" (63 node
//     slots in heap order + 64 exits); one walk step evaluates a whole supernode: every lane
//     fetches the property its node tests (ds_bpermute), compares (63 decisions -> one 64-bit
//     mask), and lane e tests whether the mask agrees with the six decisions on the path to exit
//     e.  The root supernode lives in registers, the next 57 in LDS, deeper ones in HBM scratch;
//   * the properties of 64 consecutive pixels that do not depend on the pixel being decoded
//     (reference channels, top row, position) are computed by 64 lanes at once and parked in LDS;
//     per pixel one ds_read puts property p into lane p and the 7 left-dependent ones, all of
//     the form F(c*left + bias), are finished by a handful of vector instructions;
//   * the 31 adaptive chances of the current leaf sit in lanes 0..30; a binary decision is
//     evaluated by all lanes at once for their own chance (64-bit mad, compare -> mask) and the
//     lane of the context in use is picked; (index,bit) pairs are recorded in two scalar masks
//     and all touched chances are advanced together by one vector table lookup after the symbol;
//   * decoded pixels are collected in a VGPR and stored 64 at a time.
// A single wave issues one instruction every ~4.6 cycles, a taken branch costs ~25 and a
// VALU<->SALU hand-over ~14 (tools/ubench.hip), so the kernel is bound by the instruction count of
// the per-symbol chain (and by two dependent HBM round trips per symbol: a supernode, the leaf); every
// item above trades scalar instructions for vector ones.  Measured on the 1024 x 4K launch (round 4,
// profiles/r4_instruction_probes.txt): one more scalar instruction per symbol costs 0.37 % of the launch,
// a vector one 0.19 %, a taken branch 0.44 % -- ~200 instructions per symbol on the ISA of the pixel loop, 122 of them the symbol decoder.
// Round 6: the context of a group comes in two supernode forms (8-byte lane words; NARROW 4-byte ones when every property lies inside
// 13 bits: kLeafFlagN below) and two leaf forms (31 chances in 64 bytes; COMPACT 16 chances in 32 bytes when the symbols have at most 8 magnitude bits:
// LeafRegs::mb), chosen per group and carried in the tile record; context areas have the exact size of their context and come from one two-ended arena;
// the reference loads of a pixel chunk are issued together.  With every wavefront slot busy the launch is bound by what a SIMD ISSUES (LDS-resident
// supernodes in the dense configuration made a long group 7 % faster and the launch 4.7 % slower: removed), so nothing here may add an instruction
// to the per-symbol path lightly.
// The round's last session took instructions OUT (profiles/r6_sq_counters_final.txt: 171 instructions per decoded sample, 81 of them scalar, the scalar issue
// slots the busier ones): fast_symbol_hw with the exponent decisions on chances 2..9 unrolled and an exit per exponent -- 159 instructions per sample, 71 scalar,
// the launch 6.36 -> 6.02 s (profiles/r6_unrolled_decoder.txt).
// Three LDS configurations are built: "wide" for a launch alone = 38.9 KB per wave (58 supernodes = 29 KB, 8.4 KB of chunk properties,
// small state; one wave per SIMD), "wide" for hosts with two batches in flight = 19.7 KB (20 supernodes; exactly two waves per SIMD:
// fuifgpu_batch_set_in_flight, round 5) and "dense" = 5.7 KB (no supernode slots, 32-pixel chunks;
// 80 VGPRs: 24 waves per CU, six per SIMD, which fill each other's stalls when tiles outnumber SIMDs).
// The 16 KB chance transition table is read through L1/L2 instead: its lookups are off the dependency
// chain thanks to the batched update.
//
// What it replaces in the reference:
//   fuif_decode channel loop            encoding/encoding.cpp:708-717
//   fuif_decode_channel                 encoding/encoding.cpp:259-429
//   init_properties / predictors        encoding/context_predict.h:67-120, 124-168, 233-289
//   MetaPropertySymbolCoder::read_tree  maniac/compound.h:277-320   (explicit stack, no recursion)
//   FinalPropertySymbolCoder            maniac/compound.h:135-232
//   reader<15>, SymbolChance            maniac/symbol.h:72-185
//   UniformSymbolCoder                  maniac/symbol.h:44-57
//   RacInput24                          maniac/rac.h:55-117
//   SimpleBitChance::put                maniac/chance.h:77-79
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fuifgpu_internal.h"
#include "maniac_decode.h"

namespace fuifgpu {

namespace {

#define DEV __device__ __forceinline__

// -DFUIF_PROF: per-phase shader-cycle counters (s_memtime) written to DecodeParams::prof[img*8+k]
// -DFUIF_PROF_BY_CHANNEL: the counters are summed per first channel of the tile over all images (row = channel) instead of per image
#ifdef FUIF_PROF_BY_CHANNEL
#define PROF_ROW first_c
#else
#define PROF_ROW img
#endif
#ifdef FUIF_PROF
#define PROF_DECL unsigned long long prof_t0 = 0, prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PROF_START() prof_t0 = __builtin_readcyclecounter()
#define PROF_LAP(k) do { unsigned long long t__ = __builtin_readcyclecounter(); prof_acc[k] += t__ - prof_t0; prof_t0 = t__; } while (0)
#else
#define PROF_DECL
#define PROF_START()
#define PROF_LAP(k)
#endif

constexpr int CH_ZERO = 0, CH_SIGN = 1, CH_EXP = 2, CH_MANT = 16, CH_N = 31;
// Supernodes (512 B each) of the context tree kept in LDS: the kernel is built twice.
//   kLdsWide  : 20 supernodes = 10 KB of tree per wavefront (20 KB of LDS with the property rows): exactly TWO wavefronts per SIMD -- for launches with
//               few tiles (streams without a group index: one tile per image).  Rounds 1-4 kept 58 supernodes (one wavefront per SIMD): a launch
//               alone was 6 % faster (20.8 against 22.1 s for 1024 x 4K), but a lone wavefront per SIMD is latency-bound and nothing could run beside
//               it -- with 20, two 1024-picture launches side by side take 24.5 s instead of 41.6 (693 against 408 Mpixels/s; 12 supernodes = three
//               per SIMD: 662; profiles/r5_overlap_timeline_and_wide_variants.txt)
//   kLdsDense : no tree in LDS (the root supernode lives in registers), 6 wavefronts per SIMD -- best when
//               tiles abound (group index): co-resident wavefronts fill each other's stalls and every round
//               behind the root is one memory fetch (rounds 1-3 kept two slots that served 0.9 % of the rounds)
#ifndef FUIF_LDS_WIDE
#define FUIF_LDS_WIDE 20
#endif
#ifndef FUIF_LDS_DENSE
#define FUIF_LDS_DENSE 0   // round 4: the two slots served 0.9 % of the walk rounds (profiles/r2_walk_locality.txt) and cost every round an LDS read, a compare and two branches
#endif
// A host that decodes ONE batch at a time (fuifgpu_batch_set_in_flight: 1, the default) gets the wide configuration of rounds 1-4 for its launches with few tiles: 58
// supernodes in LDS, one wavefront per SIMD -- a launch alone is 6 % faster that way (20.8 against 22.1 s for 1024 x 4K without index); a host that keeps two batches
// in flight gets the 20-supernode instantiation, whose wavefronts leave room for the other launch's.
#ifndef FUIF_LDS_WIDE_ALONE
#define FUIF_LDS_WIDE_ALONE 58
#endif
constexpr int kLdsWide = FUIF_LDS_WIDE, kLdsDense = FUIF_LDS_DENSE, kLdsWideAlone = FUIF_LDS_WIDE_ALONE;
// (Round 6 tried 11 narrow supernodes resident in LDS in the dense configuration as well -- since the children of the top supernodes are numbered by subtree
// size, the 11 largest second-level supernodes of a long 4K group serve 63-70 % of the rounds behind the root (tools/supernode_packing.py), with the room taken
// from int16 property rows.  A long group alone got 7 % faster, the 1024-picture launch 4.7 % SLOWER: with every wavefront slot busy the launch is bound by what a
// SIMD issues, and the six instructions per round cost more than the hidden round trips gave back.  Removed: profiles/r6_variants_lds_records_vs_none.txt.)
#ifndef FUIF_SIZE_ORDERED
#define FUIF_SIZE_ORDERED 64   // supernodes (breadth first) whose children are numbered by subtree size; 0 = exit order everywhere
#endif
constexpr int kSizeOrdered = FUIF_SIZE_ORDERED;
constexpr uint32_t kLeafFlag = 0x800000u;
constexpr uint32_t kSlowFlag = 0x400000u;   // exit leads to a plain tree node (index in the low 16 bits), not to a supernode
// NARROW supernodes (round 6): 4 bytes per lane instead of 8 -- a supernode is 256 bytes, two cache lines instead of four.  The per-symbol chain of a
// long tile is two dependent memory round trips whose latency grows with everything the ~6000 resident wavefronts keep in flight; halving the supernode
// record is worth 8-15 % of that chain (tools/ubench_context.hip, profiles/r6_ubench_context_layouts.txt).  Lane word:
//     [1:0] exit, bits 13..12 | [6:2] property | [19:7] split value (signed, 13 bits) | [31:20] exit, bits 11..0      (a rotation by 20 makes the exit contiguous)
// exit = kLeafFlagN | leaf id, or the number of the child supernode (14 bits).  A channel group gets narrow supernodes when every property of it stays
// inside +-4095 (8-bit pictures up to 4096 rows / columns do), it has at most 32 properties (default options: 25) and its tree at most kNarrowMaxNodes
// nodes -- then no supernode index, leaf id or split value can overflow its field and the node-by-node walk (kSlowFlag) cannot occur; any other
// group keeps the 8-byte form.  The word IS the ds_bpermute address (bits 7..2: bit 7 is the split's lowest bit, so lanes 32..63 of the property
// vector mirror lanes 0..31 for such groups).
constexpr uint32_t kLeafFlagN = 0x2000u;
constexpr int kNarrowMaxNodes = 13999;   // (7 * 6999 + 5) / 12 = 4083 supernodes at most: inside every area (capi.hip: FUIF_MAX_SUPER = 4096), leaf ids < 7000
constexpr int kNarrowSplitMax = 4095, kNarrowSplitMin = -4096;
DEV uint32_t pack_narrow(int split, uint32_t prop, uint32_t exit14) {
    const int sv = split > kNarrowSplitMax ? kNarrowSplitMax : (split < kNarrowSplitMin ? kNarrowSplitMin : split);   // values lie in [-4095, 4095]: the clamped comparison is the same
    return ((exit14 >> 12) & 3u) | ((prop & 31u) << 2) | (((uint32_t)sv & 0x1FFFu) << 7) | ((exit14 & 0xFFFu) << 20);
}
constexpr int kPropPitch = 33;    // odd pitch: conflict-free column writes / row reads; groups with more than 31 properties use 2 * 33 - 1 = 65 words and half the chunk
constexpr int kPropPitchWide = 65;
// Pixels whose properties are prepared at once (lane = pixel).  The property rows are the largest LDS
// item; a shorter chunk buys resident wavefronts (the real lever of this kernel: it is issue bound).
#ifndef FUIF_CHUNK
#define FUIF_CHUNK 32   // dense configuration; the wide one always prepares 64 pixels at a time
#endif
constexpr int kChunkDense = FUIF_CHUNK;
static_assert(kChunkDense == 64 || kChunkDense == 32 || kChunkDense == 16, "chunk must divide the wavefront");
// The dense configuration is built for 6 wavefronts per SIMD (80 VGPRs, 32-pixel chunks: 5.8 KB of LDS).  The kernel waits
// for memory two thirds of the time even with 4 wavefronts per SIMD (a leaf, a supernode and a table line per symbol,
// profiles/r2_sq_counters_*): measured on 1024 x 4K with the group index, 4 / 5 / 6 / 8 per SIMD = 10.9 / 9.6 / 9.5 /
// 12.0 s (at 8 the 64-register budget spills in the pixel loop).  The wide configuration (LDS bound: one per SIMD)
// ignores the bound.
#ifndef FUIF_WAVES
#define FUIF_WAVES 6
#endif
#define FUIF_OCCUPANCY __attribute__((amdgpu_waves_per_eu(FUIF_WAVES, FUIF_WAVES)))

DEV int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
DEV uint32_t rflu(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
DEV int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
// clang for ROCm 7.2 has no writelane builtin, and v_writelane_b32 cannot take both its value and its lane from SGPRs (one
// constant-bus operand on gfx9): a uniform compare + v_cndmask does the job
DEV int wrlane(int val, int l, int old) { return ((int)threadIdx.x == l) ? val : old; }

// ---- tile-to-tile hand-off (placement independent) ----------------------------------------------
// Tiles of one image run on different wavefronts, possibly on different XCDs whose L2s are not
// coherent.  Everything one tile writes and another reads (coefficient planes, ChannelMeta, the
// progress words) is therefore stored write-through (sc1, agent-scope relaxed atomics) and read
// with agent-scope loads that bypass the CU's L1; a progress word is stored only after
// `s_waitcnt vmcnt(0)` has drained the payload stores of the (single) writing wave.
#ifdef FUIF_EMU   // tools/emu: the same source compiled for the CPU wavefront emulator (test infrastructure)
typedef int32_t gi32;
typedef uint32_t gu32;
typedef int16_t gi16;
#else
typedef __attribute__((address_space(1))) int32_t gi32;
typedef __attribute__((address_space(1))) uint32_t gu32;
typedef __attribute__((address_space(1))) int16_t gi16;
#endif
DEV void st_agent(int32_t *p, int v) { __hip_atomic_store((gi32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV void st_agent(int16_t *p, int v) { __hip_atomic_store((gi16 *)p, (int16_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // coefficient samples
DEV int ld_agent(const int16_t *p) { return (int)__hip_atomic_load((const gi16 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV void st_agent(uint32_t *p, uint32_t v) { __hip_atomic_store((gu32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV int ld_agent(const int32_t *p) { return __hip_atomic_load((const gi32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV uint32_t ld_agent(const uint32_t *p) { return __hip_atomic_load((const gu32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#ifdef FUIF_EMU
DEV void drain_stores() {}
#else
DEV void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
#ifdef FUIF_EMU
DEV unsigned long long realtime() { return 0; }
#else
DEV unsigned long long realtime() { return __builtin_amdgcn_s_memrealtime(); }   // 100 MHz, same clock on every CU
#endif
// -DFUIF_STATS: scheduler statistics (DecodeParams::sched_stats) and the tile log (fuifgpu_batch_tile_log).  The release
// kernel carries neither: nine 64-bit counters live across the whole persistent loop cost it ~20 SGPRs (spills).
#ifdef FUIF_STATS
#define STATS(...) __VA_ARGS__
#else
#define STATS(...)
#endif
// the tile log alone (first start / end / running time of every tile) is also part of the -DFUIF_PROF and -DFUIF_TILELOG builds: it keeps
// one value live across a tile, so that -- unlike the full statistics -- it compiles at 6 wavefronts per SIMD (hipcc 7.2 trips over
// an odd-aligned 64-bit spill reload, 'Subtarget requires even aligned vector registers', when the pressure goes up)
#if defined(FUIF_STATS) || defined(FUIF_PROF) || defined(FUIF_TILELOG)
#define TLOG(...) __VA_ARGS__
#else
#define TLOG(...)
#endif
// A wait gives up (ST_STALLED, reported with ST_CORRUPT) when NOTHING in the launch has made progress for a long while:
// every kStaleCheck polls the waiting wavefront looks at DecodeParams::heartbeat (bumped by every running tile every few
// rows); kStaleLimit looks in a row without a change (~5 s) mean the producer is lost.  Time alone says nothing: the long
// final tiles of a large picture run for seconds while everybody else waits for them.
constexpr uint32_t kStaleCheck = 1u << 16;
constexpr uint32_t kStaleLimit = 64;
constexpr uint32_t kIdleStaleLooks = 1u << 16;   // idle wavefront: looks (>= 0.1 ms apart) without any progress in the launch
constexpr uint32_t kActiveImages = 6;  // images of a queue that are being worked on at a time (context scheduler)

struct Node {  // maniac/compound.h:41-51; property -1 = leaf, child = leaf id
    int32_t splitval;
    uint16_t child;
    int16_t property;
};

// One 64-bit load per tree level.  hipcc otherwise splits the node into two dependent 32-bit
// loads (it sinks the splitval load behind the leaf test), doubling the per-level latency.
#ifdef FUIF_EMU
// emulator-only statistics (tools/emu_walk_stats.py): where the walk rounds behind the root supernode are served from
extern unsigned long long g_emu_stats[6];   // symbols with a walk, rounds from LDS, rounds from scratch memory, suspended tiles (slots 4-5 unused)
#define EMU_COUNT(k) do { if (lane == 0) __atomic_fetch_add(&g_emu_stats[k], 1ull, __ATOMIC_RELAXED); } while (0)
static thread_local const char *emu_lds_base;   // LDS byte addresses are offsets from the supernode array in the emulator
DEV uint2 lds_load_node(uint32_t lds_byte_addr) { return *reinterpret_cast<const uint2 *>(emu_lds_base + lds_byte_addr); }
DEV uint2 global_load_node(const void *p) { return *reinterpret_cast<const uint2 *>(p); }
#else
#define EMU_COUNT(k)
DEV uint2 lds_load_node(uint32_t lds_byte_addr) {
    uint2 v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_byte_addr) : "memory");
    return v;
}
DEV uint2 global_load_node(const void *p) {
    uint2 v;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
#endif
// supernode `sn` of a tree at `base`, lane's record: a scalar base + one 32-bit vector offset (a tree's supernodes span at most 2 MB)
#ifdef FUIF_EMU
DEV uint2 global_load_supernode(const uint2 *base, uint32_t sn, uint32_t lane8) { return *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(base) + sn * 512u + lane8); }
#else
DEV uint2 global_load_supernode(const uint2 *base, uint32_t sn, uint32_t lane8) {
    uint2 v;
    uint32_t off;
    asm volatile("v_lshl_add_u32 %1, %3, 9, %4\n\tglobal_load_dwordx2 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(v), "=&v"(off) : "s"(base), "s"(sn), "v"(lane8) : "memory");
    return v;
}
#endif

// narrow supernode `sn` (256 B), lane's word
#ifdef FUIF_EMU
DEV uint32_t lds_load_node_n(uint32_t lds_byte_addr) { return *reinterpret_cast<const uint32_t *>(emu_lds_base + lds_byte_addr); }
DEV uint32_t global_load_supernode_n(const uint32_t *base, uint32_t sn, uint32_t lane4) { return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(base) + sn * 256u + lane4); }
#else
DEV uint32_t lds_load_node_n(uint32_t lds_byte_addr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_byte_addr) : "memory");
    return v;
}
DEV uint32_t global_load_supernode_n(const uint32_t *base, uint32_t sn, uint32_t lane4) {
    uint32_t v, off;
    asm volatile("v_lshl_add_u32 %1, %3, 8, %4\n\tglobal_load_dword %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(v), "=&v"(off) : "s"(base), "s"(sn), "v"(lane4) : "memory");
    return v;
}
#endif

struct Frame {  // one pending inner node of the pre-order tree parse
    int32_t p, oldmin, oldmax, splitval, child, stage;
};

// Byte source with FileIO / BlobReader end-of-stream semantics (fileio.h:33-143).  `win` is the
// only per-lane member: lane i holds bytes [win_base+4i, win_base+4i+4) of the stream.
struct Stream {
    const uint8_t *p;
    uint32_t size, pos, limit, win_base;
    uint32_t win;
    int eof_flag;   // FileIO::isEOF() = feof(): only after a failed read (fileio.h:55-63)
    int blob_mode;  // BlobReader::isEOF(): pos >= size (fileio.h:100-102)
};

#define LIKELY(x) __builtin_expect(!!(x), 1)
#define UNLIKELY(x) __builtin_expect(!!(x), 0)

DEV int s_getc(Stream &s, int lane) {
    if (UNLIKELY(s.pos >= s.size)) { s.eof_flag = 1; return -1; }
    uint32_t idx = s.pos - s.win_base;
    if (UNLIKELY(idx >= 256u)) {
        s.win_base = s.pos & ~255u;
        s.win = reinterpret_cast<const uint32_t *>(s.p + s.win_base)[lane];
        idx = s.pos - s.win_base;
    }
    const uint32_t word = (uint32_t)rdlane((int)s.win, (int)(idx >> 2));
    s.pos++;
    return (int)((word >> ((idx & 3u) * 8u)) & 0xFFu);
}
DEV bool s_eof(const Stream &s) { return s.blob_mode ? (s.pos >= s.size) : (s.eof_flag != 0); }
DEV bool s_limit_hit(const Stream &s) { return s_eof(s) || (s.limit && s.pos >= s.limit); }

// encoding/encoding.cpp:45-59
DEV int s_varint(Stream &s, int lane) {
    uint32_t result = 0;
    for (int k = 0; k < 10; k++) {
        int b = s_getc(s, lane);
        if (b < 0) return -1;
        if (b < 128) return (int)(result + (uint32_t)b);
        result = (result + (uint32_t)(b - 128)) << 7;
    }
    return -1;
}

DEV int ilog2u(uint32_t l) { return l == 0 ? 0 : 31 - __clz((int)l); }
DEV int slog(int x) {  // encoding/context_predict.h:53-61, branch free: sign(x) * bit_length(|x|)
    const int m = x >> 31;
    const int a = (x ^ m) - m;
    const int r = 32 - __clz(a);  // __clz(0) == 32
    return (r ^ m) - m;
}
DEV int iabs(int x) { return x < 0 ? -x : x; }
// slog(d) > s  <=>  d > slog_threshold(s): slog is monotone, so a tree node that tests one of the four slog properties
// that depend on the pixel to the left (local properties 3, 8, 9, 12) can test the raw difference instead, and the
// per-pixel patch does not have to take a logarithm (s < -32 cannot occur: split values lie inside the property range)
DEV int slog_threshold(int s) {
    if (s >= 31) return 0x7FFFFFFF;
    if (s >= 0) return (1 << s) - 1;
    if (s <= -32) return (int)0x80000000;
    return -(1 << (-s - 1));
}
DEV bool is_raw_slog_prop(int kloc) { return kloc == 3 || kloc == 8 || kloc == 9 || kloc == 12; }
DEV int median3(int a, int b, int c) {  // util.h:9-23
    if (a < b) { if (b < c) return b; return a < c ? c : a; }
    if (a < c) return a;
    return b < c ? c : b;
}

// 24-bit range decoder, maniac/rac.h:55-114.  `low` fits 32 bits before EOF; after EOF the
// reference ORs -1 into a 64-bit low so every later decision is 1 -- a 32-bit all-ones low gives
// the same decisions (low stays >= 2^32-2^24 > range between renormalisations).
struct Rac {
    uint32_t range, low;
};
DEV void rac_input(Rac &r, Stream &s, int lane) {
    // rac.h:70-81: at most two byte shifts; the second is only possible after the first.  Most
    // decisions need none, so the whole renormalisation is kept off the fall-through path.
    if (UNLIKELY(r.range <= 0x10000u)) {
        r.low <<= 8; r.range <<= 8; r.low |= (uint32_t)s_getc(s, lane);
        if (UNLIKELY(r.range <= 0x10000u)) { r.low <<= 8; r.range <<= 8; r.low |= (uint32_t)s_getc(s, lane); }
    }
}
DEV int rac_get(Rac &r, Stream &s, int lane, uint32_t chance) {
    const uint32_t thr = r.range - chance;
    int bit;
    if (r.low >= thr) { r.low -= thr; r.range = chance; bit = 1; }
    else { r.range = thr; bit = 0; }
    rac_input(r, s, lane);
    return bit;
}
DEV void rac_init(Rac &r, Stream &s, int lane) {
    r.range = 1u << 24; r.low = 0;
    for (int k = 0; k < 3; k++) { r.low <<= 8; r.low |= (uint32_t)s_getc(s, lane); }
}
// rac.h:43-52: (range*b12+0x800)>>12 without a 64-bit product
DEV uint32_t chance12(uint32_t range, uint32_t b12) { return (((range & 0xFFFu) * b12 + 0x800u) >> 12) + ((range >> 12) * b12); }
DEV int rac_bit(Rac &r, Stream &s, int lane) { return rac_get(r, s, lane, r.range >> 1); }

// maniac/symbol.h:44-57
DEV int uniform_read(Rac &r, Stream &s, int lane, int min, int len) {
    while (len != 0) {
        int med = len / 2;
        if (rac_bit(r, s, lane)) { min = min + med + 1; len = len - (med + 1); }
        else len = med;
    }
    return min;
}

// ---- memory-resident contexts (tree coder): compound.h:90-95 + chance.h:77-79 -------------------
DEV int ctx_bit(Rac &r, Stream &s, int lane, uint16_t *ch, int idx, const uint16_t *table) {
    const uint32_t c = rflu(ch[idx]);
    const int bit = rac_get(r, s, lane, chance12(r.range, c));
    const uint32_t nc = rflu(table[c * 2 + bit]);
    if (lane == 0) ch[idx] = (uint16_t)nc;
    return bit;
}
// maniac/symbol.h:154-185 on an LDS context
DEV int ctx_symbol(Rac &r, Stream &s, int lane, uint16_t *ch, const uint16_t *table, int min, int max) {
    if (min == max) return min;
    if (ctx_bit(r, s, lane, ch, CH_ZERO, table)) return 0;
    int sign;
    if (min < 0) { if (max > 0) sign = ctx_bit(r, s, lane, ch, CH_SIGN, table); else sign = 0; }
    else sign = 1;
    const int amax = sign ? max : -min;
    const int emax = ilog2u((uint32_t)amax);
    int e = 0;
    for (; e < emax; e++) if (ctx_bit(r, s, lane, ch, CH_EXP + e, table)) break;
    int have = 1 << e;
    for (int pos = e; pos > 0;) {
        pos--;
        int minabs1 = have | (1 << pos);
        if (minabs1 > amax) continue;
        if (ctx_bit(r, s, lane, ch, CH_MANT + pos, table)) have = minabs1;
    }
    return sign ? have : -have;
}
DEV int ctx_symbol2(Rac &r, Stream &s, int lane, uint16_t *ch, const uint16_t *table, int min, int max) {  // symbol.h:235-239
    if (min > 0) return ctx_symbol(r, s, lane, ch, table, 0, max - min) + min;
    if (max < 0) return ctx_symbol(r, s, lane, ch, table, min - max, 0) + max;
    return ctx_symbol(r, s, lane, ch, table, min, max);
}

// ---- register-resident leaf (pixel coder) --------------------------------------------------------
// lane i (< 31) of `leafv` holds chance i of the current leaf.  A decision reads it with
// v_readlane and records (index, bit) in the scalar masks; leaf_commit() advances every touched
// chance at once with one vector lookup in the transition table.
struct LeafRegs {
    int leafv;          // per lane
    uint32_t touched;   // scalar
    uint32_t bits;      // scalar
    // COMPACT leaves (round 6): a group whose symbols have at most 8 magnitude bits (|diff| <= 255: exponent 0..7) touches only the zero, the sign, 7
    // exponent and 7 mantissa chances of a leaf (symbol.h:167-183), so its leaves are 16 chances = 32 bytes instead of 31 padded to 64: slots 0 zero, 1 sign,
    // 2..8 exponent, 9..15 mantissa.  Only the mantissa's first slot moves (16 -> 9); lanes 16..63 mirror lanes 0..15.  Half the leaf bytes of such
    // a group's context (the larger half of it once the supernodes are narrow): profiles/r6_ubench_context_layouts.txt.
    int mb;             // scalar: slot of mantissa chance 0 (CH_MANT = 16, compact leaves: 9)
    uint32_t mirror;    // scalar: 1, compact leaves: 0x10001 -- a 16-bit slot mask times this covers the lanes 16..31 that mirror 0..15
};
constexpr int kMantCompact = 9;
DEV int leaf_bit(Rac &r, Stream &s, int lane, LeafRegs &L, int idx) {
    // Every lane evaluates the decision for ITS chance with the current range (rac.h:43-52,82-95):
    // chance = (range*b12+0x800)>>12 as one 64-bit mad, threshold = range-chance, low >= threshold.
    // Four vector instructions replace ~12 dependent scalar ones; lane `idx` holds the answer.
    const uint32_t ch_v = (uint32_t)(((unsigned long long)r.range * (uint32_t)L.leafv + 0x800ull) >> 12);
    const uint32_t thr_v = r.range - ch_v;
    const unsigned long long ge = __ballot(r.low >= thr_v);
    const uint32_t thr = (uint32_t)rdlane((int)thr_v, idx);
    const int bit = (int)((ge >> idx) & 1ull);
    r.low = bit ? r.low - thr : r.low;
    r.range = bit ? r.range - thr : thr;
    rac_input(r, s, lane);
    L.touched |= 1u << idx;
    L.bits |= (uint32_t)bit << idx;
    return bit;
}
// maniac/symbol.h:154-185
DEV int leaf_symbol(Rac &r, Stream &s, int lane, LeafRegs &L, int min, int max) {
    // single-exit formulation (no early returns / breaks): hipcc's control-flow structuriser turns
    // every early exit of an inlined function into extra mask bookkeeping and branches
    int result = min;
    if (min != max) {
        result = 0;
        if (!leaf_bit(r, s, lane, L, CH_ZERO)) {
            int sign = 1;
            if (min < 0) { sign = 0; if (max > 0) sign = leaf_bit(r, s, lane, L, CH_SIGN); }
            const int amax = sign ? max : -min;
            const int emax = ilog2u((uint32_t)amax);
            int e = 0;
            bool more = emax > 0;
            while (more) {  // unary exponent: stop at the first 1 bit or at emax
                const int b = leaf_bit(r, s, lane, L, CH_EXP + e);
                e += 1 - b;
                more = (b == 0) & (e < emax);
            }
            int have = 1 << e;
            for (int pos = e - 1; pos >= 0; pos--) {
                const int minabs1 = have | (1 << pos);
                if (minabs1 <= amax) {  // else the 1-bit is impossible (symbol.h:180)
                    const int b = leaf_bit(r, s, lane, L, L.mb + pos);
                    have = b ? minabs1 : have;
                }
            }
            result = sign ? have : -have;
        }
    }
    return result;
}
// ---- the common case of leaf_symbol, written for the instruction count ---------------------------------------
// Preconditions (checked by the caller): min < 0 < max (zero and sign are both coded), and the next 64 stream bytes lie
// inside the 256-byte window (a symbol reads at most 2 bytes per decision, 31 decisions), so a renormalisation is a
// register shuffle with no end-of-stream or refill case.  Differences to leaf_symbol: the decision is a SCALAR compare of
// `low` with the threshold of the chance in use (v_readlane of the per-lane thresholds) instead of a vector compare whose
// mask is then picked apart; the (index, bit) bookkeeping of the exponent and mantissa is reconstructed after the loops
// from e and the magnitude instead of being updated per decision; a mantissa bit can only be impossible when e == emax.
struct FastSym {
    int amax_pos, amax_neg, emax_pos, emax_neg;
    int ilast_pos, ilast_neg;   // emax + 1: the index of the last exponent chance of each sign (fast_symbol_hw)
};
DEV uint32_t lane_thresholds(uint32_t range, int leafv) {   // rac.h:43-52 for every lane's own chance: range - chance
    return range - (uint32_t)(((unsigned long long)range * (uint32_t)leafv + 0x800ull) >> 12);
}
DEV void fast_renorm(Rac &r, Stream &s) {   // rac.h:70-81 with the bytes taken from the window registers
    if (UNLIKELY(r.range <= 0x10000u)) {
        uint32_t idx = s.pos - s.win_base;
        uint32_t word = (uint32_t)rdlane((int)s.win, (int)(idx >> 2));
        r.low = (r.low << 8) | ((word >> ((idx & 3u) * 8u)) & 0xFFu); r.range <<= 8; s.pos++;
        if (UNLIKELY(r.range <= 0x10000u)) {
            idx = s.pos - s.win_base;
            word = (uint32_t)rdlane((int)s.win, (int)(idx >> 2));
            r.low = (r.low << 8) | ((word >> ((idx & 3u) * 8u)) & 0xFFu); r.range <<= 8; s.pos++;
        }
    }
}
DEV int fast_symbol(Rac &r, Stream &s, LeafRegs &L, const FastSym &F) {
    uint32_t thr = (uint32_t)rdlane((int)lane_thresholds(r.range, L.leafv), CH_ZERO);
    if (r.low >= thr) {   // the symbol is zero
        r.low -= thr; r.range -= thr;
        fast_renorm(r, s);
        L.touched = 1u; L.bits = 1u;
        return 0;
    }
    r.range = thr;
    fast_renorm(r, s);
    thr = (uint32_t)rdlane((int)lane_thresholds(r.range, L.leafv), CH_SIGN);
    const bool sign = r.low >= thr;
    r.range = sign ? r.range - thr : thr;
    r.low -= sign ? thr : 0u;
    fast_renorm(r, s);
    const int amax = sign ? F.amax_pos : F.amax_neg, emax = sign ? F.emax_pos : F.emax_neg;
    int e = 0;
    uint32_t one = 0;
    while (e < emax) {   // unary exponent: symbol.h:167-170
        thr = (uint32_t)rdlane((int)lane_thresholds(r.range, L.leafv), CH_EXP + e);
        if (r.low >= thr) { r.low -= thr; r.range -= thr; fast_renorm(r, s); one = 1u; break; }
        r.range = thr;
        fast_renorm(r, s);
        e++;
    }
    int have = 1 << e;
    uint32_t skipped = 0;
    if (LIKELY(e < emax)) {
        // have | 1 << pos < 2^(e+1) <= 2^emax <= amax: every mantissa bit is coded (symbol.h:173-183)
        for (int pos = e - 1; pos >= 0; pos--) {
            thr = (uint32_t)rdlane((int)lane_thresholds(r.range, L.leafv), L.mb + pos);
            const bool b = r.low >= thr;
            r.range = b ? r.range - thr : thr;
            r.low -= b ? thr : 0u;
            fast_renorm(r, s);
            have |= b ? (1 << pos) : 0;
        }
    } else {
        for (int pos = e - 1; pos >= 0; pos--) {
            const int minabs1 = have | (1 << pos);
            if (minabs1 > amax) { skipped |= 1u << pos; continue; }   // the 1-bit is impossible (symbol.h:180)
            thr = (uint32_t)rdlane((int)lane_thresholds(r.range, L.leafv), L.mb + pos);
            const bool b = r.low >= thr;
            r.range = b ? r.range - thr : thr;
            r.low -= b ? thr : 0u;
            fast_renorm(r, s);
            have = b ? minabs1 : have;
        }
    }
    const uint32_t emask = (1u << e) - 1u;
    L.touched = 3u | (((1u << (e + (int)one)) - 1u) << CH_EXP) | ((emask & ~skipped) << L.mb);
    L.bits = ((uint32_t)sign << CH_SIGN) | (one << (CH_EXP + e)) | (((uint32_t)have & emask) << L.mb);
    return sign ? have : -have;
}
#ifndef FUIF_EMU
// The same symbol decoder as fast_symbol (the C++ above is the specification and what the CPU emulator runs), hand
// written: hipcc turns the loops with their several exits into ~30 instructions and 4-5 branches per decision (flag
// registers and copies for every exit); this is 13 per exponent bit and 21 per mantissa bit.  three VGPRs at the top of the register budget are scratch
// (clobbers).  Per decision: per-lane thresholds range - ((range * chance + 0x800) >> 12) as one 64-bit mad,
// v_readlane of the lane of the chance in use, scalar compare with `low`; renormalisation out of line.
// scratch VGPRs = the last ones of the register budget of the build (FUIF_WAVES wavefronts per SIMD): the 64-bit
// product, the 64-bit rounding constant 0x800 (loaded at the top of the block; 64-bit tuples must be even aligned on
// gfx950, which an inline-asm "v" operand does not guarantee) and the thresholds
#if FUIF_WAVES >= 8
#define FS_VK "v[58:59]"
#define FS_VK0 "v58"
#define FS_VK1 "v59"
#define FS_VBC "v[60:61]"
#define FS_VB "v60"
#define FS_VC "v61"
#define FS_VA "v63"
#elif FUIF_WAVES == 7
#define FS_VK "v[66:67]"
#define FS_VK0 "v66"
#define FS_VK1 "v67"
#define FS_VBC "v[68:69]"
#define FS_VB "v68"
#define FS_VC "v69"
#define FS_VA "v71"
#elif FUIF_WAVES >= 6
#define FS_VK "v[74:75]"
#define FS_VK0 "v74"
#define FS_VK1 "v75"
#define FS_VBC "v[76:77]"
#define FS_VB "v76"
#define FS_VC "v77"
#define FS_VA "v79"
#elif FUIF_WAVES == 5
#define FS_VK "v[90:91]"
#define FS_VK0 "v90"
#define FS_VK1 "v91"
#define FS_VBC "v[92:93]"
#define FS_VB "v92"
#define FS_VC "v93"
#define FS_VA "v95"
#else
#define FS_VK "v[122:123]"
#define FS_VK0 "v122"
#define FS_VK1 "v123"
#define FS_VBC "v[124:125]"
#define FS_VB "v124"
#define FS_VC "v125"
#define FS_VA "v127"
#endif
// (the _R forms take the register that holds `range` at that point: the unrolled exponent decisions below leave the threshold they read where it is when it
// becomes the new range, instead of copying it)
#define FS_THR_PREP_R(Rr) "v_mad_u64_u32 " FS_VBC ", vcc, " Rr ", %[leafv], " FS_VK "\n\tv_alignbit_b32 " FS_VA ", " FS_VC ", " FS_VB ", 12\n\tv_sub_u32 " FS_VA ", " Rr ", " FS_VA "\n\t"
#define FS_THR_PREP FS_THR_PREP_R("%[R]")
// one or two bytes from the window registers into `low` (rac.h:70-81), then back to label `back`
// (a second byte is needed only after a decision whose chance left less than 1/256 of the range: the stub then runs once more instead of carrying a second copy --
// every decision site has its own stub, and the decoder's code size is what the deeper unrolling lost to: profiles/r6_unrolled_decoder.txt)
#define FS_RENORM_R(lbl, back, Rr) \
    lbl ":\n\t" \
    "s_lshr_b32 %[t0], %[widx], 2\n\tv_readlane_b32 %[t0], %[win], %[t0]\n\ts_lshl_b32 %[t1], %[widx], 3\n\ts_lshr_b32 %[t0], %[t0], %[t1]\n\t" \
    "s_and_b32 %[t0], %[t0], 0xff\n\ts_lshl_b32 %[L], %[L], 8\n\ts_or_b32 %[L], %[L], %[t0]\n\ts_add_u32 %[widx], %[widx], 1\n\ts_lshl_b32 " Rr ", " Rr ", 8\n\t" \
    "s_cmp_gt_u32 " Rr ", 0x10000\n\ts_cbranch_scc1 " back "\n\t" \
    "s_branch " lbl "b\n\t"
#define FS_RENORM(lbl, back) FS_RENORM_R(lbl, back, "%[R]")
#define FS_RN_CHECK_R(lbl, back, Rr) "s_cmp_le_u32 " Rr ", 0x10000\n\ts_cbranch_scc1 " lbl "\n" back ":\n\t"
#define FS_RN_CHECK(lbl, back) FS_RN_CHECK_R(lbl, back, "%[R]")
// Round 6, last session: the first four exponent decisions (chances 2..5) UNROLLED, each with its own exit that knows e: no chance-index add, loop-end test or
// taken back-branch per decision, the threshold read stays where it is when it becomes the new range (Rr / Tr swap roles from one decision to the next), and the
// exit of e = k - 2 runs exactly e mantissa decisions without a counter test and builds the (index, bit) masks from constants.  Needs ilast >= 4 (emax >= 3;
// chances 5..9 are each tested for existence first); smaller ranges and exponents beyond chance 9 take the loops below, as in rounds 4-5.  The exits of
// chances 6..9 (exponent 4..7) share one mantissa ladder and the loops' mask code at label 60.  (Six unrolled decisions WITHOUT the existence tests were slower
// than the loops: the channels whose range ends below chance 7 fell back to them -- profiles/r6_unrolled_decoder.txt.)
#define FS_UEXP(k, Rr, Tr, exitl, rnl, backl) \
    FS_THR_PREP_R(Rr) "s_nop 0\n\tv_readlane_b32 " Tr ", " FS_VA ", " k "\n\t" \
    "s_cmp_ge_u32 %[L], " Tr "\n\ts_cbranch_scc1 " exitl "\n\t" \
    FS_RN_CHECK_R(rnl, backl, Tr)
#define FS_EXIT_HEAD(lbl, Rr, Tr, rnl, backl) \
    lbl ":\n\t" \
    "s_sub_u32 %[L], %[L], " Tr "\n\ts_sub_u32 %[R], " Rr ", " Tr "\n\t" \
    FS_RN_CHECK(rnl, backl)
#define FS_MINIT(e) "s_add_u32 %[midx], %[mb], " e "\n\ts_mov_b32 %[hv], 0\n\t"
#define FS_MBIT(rnl, backl) \
    FS_THR_PREP "s_sub_u32 %[midx], %[midx], 1\n\tv_readlane_b32 %[thr], " FS_VA ", %[midx]\n\t" \
    "s_sub_u32 %[t0], %[R], %[thr]\n\ts_sub_u32 %[t1], %[L], %[thr]\n\t" \
    "s_cselect_b32 %[L], %[L], %[t1]\n\ts_cselect_b32 %[R], %[thr], %[t0]\n\ts_addc_u32 %[hv], %[hv], %[hv]\n\t" \
    FS_RN_CHECK(rnl, backl)
// e >= 1, closing 1 at chance k = e + 2, tl = (1 << (k + 1)) - 1: the zero, sign and exponent chances touched, ml = (1 << e) - 1
#define FS_MASKS(e, k, tl, ml) \
    "s_andn2_b32 %[t0], " ml ", %[hv]\n\ts_lshl_b32 %[t1], %[t0], %[mb]\n\t"                                /* the mantissa as decided (hv holds it inverted); at mb .. */ \
    "s_andn2_b32 %[bits], 2, %[sm]\n\ts_or_b32 %[bits], %[bits], %[t1]\n\ts_bitset1_b32 %[bits], " k "\n\t"   /* + the sign decision + the exponent's closing 1 */ \
    "s_bfm_b32 %[t1], " e ", %[mb]\n\ts_or_b32 %[touched], %[t1], " tl "\n\t" \
    "s_bitset1_b32 %[t0], " e "\n\ts_xor_b32 %[t0], %[t0], %[sm]\n\ts_sub_u32 %[res], %[t0], %[sm]\n\t"       /* magnitude = 1 << e | mantissa, signed */
// (Round 4 measured what one more scalar / vector instruction / taken branch per symbol costs the launch with probe builds of this block: +0.37 % / +0.19 % /
// +0.44 %, profiles/r4_instruction_probes.txt; the probe macros left the source in round 5 -- `git log -S FUIF_PROBE_S` has them.)
// (widx = the stream position inside the window registers `win`: the pixel loop of a chunk whose bytes are all in the stream carries it from symbol to symbol
// instead of converting to and from Stream::pos around every symbol)
DEV int fast_symbol_hw_w(Rac &r, uint32_t &widx_io, const uint32_t win, LeafRegs &L, const FastSym &F) {
    // Round 4 (profiles/r4_instruction_probes.txt: a scalar instruction or a taken branch costs the launch twice a vector one):
    //   * a decision is `s_sub t, low, thr`: SCC = borrow = (low < thr) = NOT the bit, and three s_cselect / s_addc take it from there
    //     (no compare, no separate subtraction of the selected amount);
    //   * the exponent loop counts the chance index itself (no index add per decision); a one-bit ends it with e < emax, which is
    //     the case in which every mantissa bit is coded (symbol.h:173-183): the lean mantissa loop of 8 scalar instructions and 2
    //     branches per bit (was 15 and 3), the bits collected INVERTED by s_addc and turned round once at the end;
    //   * the exhausted exponent (e == emax, under 1 % of the symbols) keeps the careful loop with the amax test per bit;
    //   * (index, bit) masks for leaf_commit from s_bfm_b32 fields.
    uint32_t R = r.range, Lo = r.low, widx = widx_io;
    uint32_t res, touched, bits, t0, t1, thr, idx, ilast, sm, hv, e, midx, amax, have, skipped;
    asm volatile(
        "v_mov_b32 " FS_VK0 ", 0x800\n\tv_mov_b32 " FS_VK1 ", 0\n\t"
        // ---- zero?  (chance 0)
        FS_THR_PREP "s_nop 0\n\tv_readlane_b32 %[thr], " FS_VA ", 0\n\t"
        "s_cmp_ge_u32 %[L], %[thr]\n\ts_cbranch_scc1 70f\n\t"
        FS_RN_CHECK_R("91f", "81", "%[thr]")   // (not zero: the threshold IS the new range and stays in %[thr]; the sign's threshold is read into %[R])
        // ---- sign  (chance 1): bit 1 = positive.  sm = 0 for positive, -1 for negative
        FS_THR_PREP_R("%[thr]") "s_nop 0\n\tv_readlane_b32 %[R], " FS_VA ", 1\n\t"
        "s_sub_u32 %[t0], %[thr], %[R]\n\ts_sub_u32 %[t1], %[L], %[R]\n\t"
        "s_cselect_b32 %[L], %[L], %[t1]\n\ts_cselect_b32 %[R], %[R], %[t0]\n\ts_cselect_b32 %[sm], -1, 0\n\ts_cselect_b32 %[ilast], %[ilastn], %[ilastp]\n\t"
        FS_RN_CHECK("92f", "82")
        // ---- unary exponent  (chances 2 .. emax + 1 = ilast); idx = the chance decided last
        // ---- chances 2..9 unrolled (ilast >= 4; chances 5..9 only where they exist); a 1 leaves through 102..109 with e = 0..7
        "s_cmp_lt_u32 %[ilast], 4\n\ts_cbranch_scc1 19f\n\t"
        FS_UEXP("2", "%[R]", "%[thr]", "102f", "112f", "122")
        FS_UEXP("3", "%[thr]", "%[R]", "103f", "113f", "123")
        FS_UEXP("4", "%[R]", "%[thr]", "104f", "114f", "124")
        "s_cmp_lt_u32 %[ilast], 5\n\ts_cbranch_scc1 18f\n\t"          // emax = 3: chance 4 was the last one, the exponent is exhausted
        FS_UEXP("5", "%[thr]", "%[R]", "105f", "115f", "125")
        // chances 6..9, each behind its own existence test (a channel's range may end anywhere here); their exits share one mantissa ladder
        "s_cmp_lt_u32 %[ilast], 6\n\ts_cbranch_scc1 17f\n\t"
        FS_UEXP("6", "%[R]", "%[thr]", "106f", "116f", "126")
        "s_cmp_lt_u32 %[ilast], 7\n\ts_cbranch_scc1 16f\n\t"
        FS_UEXP("7", "%[thr]", "%[R]", "107f", "117f", "127")
        "s_cmp_lt_u32 %[ilast], 8\n\ts_cbranch_scc1 15f\n\t"
        FS_UEXP("8", "%[R]", "%[thr]", "108f", "118f", "128")
        "s_cmp_lt_u32 %[ilast], 9\n\ts_cbranch_scc1 14f\n\t"
        FS_UEXP("9", "%[thr]", "%[R]", "109f", "119f", "129")
        "s_mov_b32 %[idx], 9\n\ts_cmp_lt_u32 %[idx], %[ilast]\n\ts_cbranch_scc1 20f\n\ts_branch 40f\n"
        "17:\n\ts_mov_b32 %[idx], 5\n\ts_branch 40f\n"
        "16:\n\ts_mov_b32 %[R], %[thr]\n\ts_mov_b32 %[idx], 6\n\ts_branch 40f\n"
        "15:\n\ts_mov_b32 %[idx], 7\n\ts_branch 40f\n"
        "14:\n\ts_mov_b32 %[R], %[thr]\n\ts_mov_b32 %[idx], 8\n\ts_branch 40f\n"
        "18:\n\t"
        "s_mov_b32 %[R], %[thr]\n\ts_mov_b32 %[idx], 4\n\ts_branch 40f\n"
        "19:\n\t"
        "s_mov_b32 %[idx], 1\n\ts_cmp_lt_u32 %[idx], %[ilast]\n\ts_cbranch_scc0 40f\n"
        "20:\n\t"
        FS_THR_PREP "s_add_u32 %[idx], %[idx], 1\n\tv_readlane_b32 %[thr], " FS_VA ", %[idx]\n\t"
        "s_cmp_ge_u32 %[L], %[thr]\n\ts_cbranch_scc1 25f\n\t"
        "s_mov_b32 %[R], %[thr]\n\t"
        FS_RN_CHECK("93f", "83")
        "s_cmp_lt_u32 %[idx], %[ilast]\n\ts_cbranch_scc1 20b\n\t"
        "s_branch 40f\n"
        "25:\n\t"
        "s_sub_u32 %[L], %[L], %[thr]\n\ts_sub_u32 %[R], %[R], %[thr]\n\t"
        FS_RN_CHECK("94f", "84")
        // ---- mantissa, e = idx - 2 < emax: every bit is coded  (chances 16 + pos, top bit first); hv collects the INVERTED bits
        "s_cmp_eq_u32 %[idx], 2\n\ts_mov_b32 %[hv], 0\n\ts_cbranch_scc1 60f\n\t"
        "s_add_u32 %[midx], %[idx], %[mb2]\n"
        "41:\n\t"
        FS_THR_PREP "s_sub_u32 %[midx], %[midx], 1\n\tv_readlane_b32 %[thr], " FS_VA ", %[midx]\n\t"
        "s_sub_u32 %[t0], %[R], %[thr]\n\ts_sub_u32 %[t1], %[L], %[thr]\n\t"
        "s_cselect_b32 %[L], %[L], %[t1]\n\ts_cselect_b32 %[R], %[thr], %[t0]\n\ts_addc_u32 %[hv], %[hv], %[hv]\n\t"
        FS_RN_CHECK("96f", "86")
        "s_cmp_gt_u32 %[midx], %[mb]\n\ts_cbranch_scc1 41b\n"
        // ---- value and the (index, bit) pairs of the decisions taken, for leaf_commit
        "60:\n\t"
        "s_sub_u32 %[e], %[idx], 2\n\ts_bfm_b32 %[t1], %[e], %[mb]\n\t"            // mantissa chances: ((1 << e) - 1) << mb (16; compact leaves: 9)
        "s_add_u32 %[t0], %[idx], 1\n\ts_bfm_b32 %[touched], %[t0], 0\n\ts_or_b32 %[touched], %[touched], %[t1]\n\t"   // chances 0 .. idx
        "s_lshl_b32 %[t0], %[hv], %[mb]\n\ts_andn2_b32 %[t0], %[t1], %[t0]\n\t"    // the mantissa bits as decided, at mb ..
        "s_lshl_b32 %[bits], 1, %[idx]\n\ts_or_b32 %[bits], %[bits], %[t0]\n\t"   // the exponent's closing 1
        "s_andn2_b32 %[t1], 2, %[sm]\n\ts_or_b32 %[bits], %[bits], %[t1]\n\t"     // sign decision
        "s_lshr_b32 %[t0], %[t0], %[mb]\n\ts_bitset1_b32 %[t0], %[e]\n\t"         // magnitude = 1 << e | mantissa
        "s_xor_b32 %[t0], %[t0], %[sm]\n\ts_sub_u32 %[res], %[t0], %[sm]\n\t"
        "s_branch 99f\n"
        // ---- exponent exhausted: e = emax = idx - 1, no closing 1; a mantissa 1 that would exceed amax is not coded (symbol.h:180)
        "40:\n\t"
        "s_sub_u32 %[e], %[idx], 1\n\ts_cmp_eq_u32 %[sm], 0\n\ts_cselect_b32 %[amax], %[amaxp], %[amaxn]\n\t"
        "s_lshl_b32 %[have], 1, %[e]\n\ts_mov_b32 %[skipped], 0\n\ts_mov_b32 %[t1], %[e]\n"
        "42:\n\t"
        "s_sub_u32 %[t1], %[t1], 1\n\ts_cbranch_scc1 61f\n\t"
        "s_lshl_b32 %[t0], 1, %[t1]\n\ts_or_b32 %[res], %[have], %[t0]\n\ts_cmp_gt_i32 %[res], %[amax]\n\ts_cbranch_scc1 45f\n\t"
        FS_THR_PREP "s_add_u32 %[t0], %[t1], %[mb]\n\tv_readlane_b32 %[thr], " FS_VA ", %[t0]\n\t"
        "s_sub_u32 %[t0], %[R], %[thr]\n\ts_cmp_ge_u32 %[L], %[thr]\n\t"
        "s_cselect_b32 %[R], %[t0], %[thr]\n\ts_cselect_b32 %[t0], %[thr], 0\n\ts_cselect_b32 %[have], %[res], %[have]\n\ts_sub_u32 %[L], %[L], %[t0]\n\t"
        // (the renormalisation stub uses t0 and hv as scratch here: t1 is the loop counter)
        "s_cmp_le_u32 %[R], 0x10000\n\ts_cbranch_scc0 42b\n\t"
        "s_lshr_b32 %[t0], %[widx], 2\n\tv_readlane_b32 %[t0], %[win], %[t0]\n\ts_lshl_b32 %[hv], %[widx], 3\n\ts_lshr_b32 %[t0], %[t0], %[hv]\n\t"
        "s_and_b32 %[t0], %[t0], 0xff\n\ts_lshl_b32 %[L], %[L], 8\n\ts_or_b32 %[L], %[L], %[t0]\n\ts_add_u32 %[widx], %[widx], 1\n\ts_lshl_b32 %[R], %[R], 8\n\t"
        "s_cmp_gt_u32 %[R], 0x10000\n\ts_cbranch_scc1 42b\n\t"
        "s_lshr_b32 %[t0], %[widx], 2\n\tv_readlane_b32 %[t0], %[win], %[t0]\n\ts_lshl_b32 %[hv], %[widx], 3\n\ts_lshr_b32 %[t0], %[t0], %[hv]\n\t"
        "s_and_b32 %[t0], %[t0], 0xff\n\ts_lshl_b32 %[L], %[L], 8\n\ts_or_b32 %[L], %[L], %[t0]\n\ts_add_u32 %[widx], %[widx], 1\n\ts_lshl_b32 %[R], %[R], 8\n\t"
        "s_branch 42b\n"
        "45:\n\t"
        "s_or_b32 %[skipped], %[skipped], %[t0]\n\ts_branch 42b\n"
        "61:\n\t"
        "s_bfm_b32 %[t0], %[e], 0\n\ts_bfm_b32 %[t1], %[e], 2\n\t"              // (1 << e) - 1; exponent decisions: chances 2 .. 2 + e - 1
        "s_andn2_b32 %[touched], %[t0], %[skipped]\n\ts_lshl_b32 %[touched], %[touched], %[mb]\n\ts_or_b32 %[touched], %[touched], %[t1]\n\ts_or_b32 %[touched], %[touched], 3\n\t"
        "s_and_b32 %[bits], %[have], %[t0]\n\ts_lshl_b32 %[bits], %[bits], %[mb]\n\ts_andn2_b32 %[t1], 2, %[sm]\n\ts_or_b32 %[bits], %[bits], %[t1]\n\t"
        "s_xor_b32 %[t0], %[have], %[sm]\n\ts_sub_u32 %[res], %[t0], %[sm]\n\t"
        "s_branch 99f\n"
        // ---- the symbol is zero
        "70:\n\t"
        "s_sub_u32 %[L], %[L], %[thr]\n\ts_sub_u32 %[R], %[R], %[thr]\n\ts_mov_b32 %[res], 0\n\ts_mov_b32 %[touched], 1\n\ts_mov_b32 %[bits], 1\n\t"
        FS_RN_CHECK("95f", "85")
        "s_branch 99f\n"
        FS_RENORM_R("91", "81b", "%[thr]") FS_RENORM("92", "82b") FS_RENORM("93", "83b") FS_RENORM("94", "84b") FS_RENORM("95", "85b") FS_RENORM("96", "86b")
        // ---- the exits of the unrolled exponent: e = 0 (value +-1), 2, 3; e = 1 comes last and falls through to the end
        FS_EXIT_HEAD("102", "%[R]", "%[thr]", "132f", "142")
        "s_mov_b32 %[touched], 7\n\ts_andn2_b32 %[t1], 2, %[sm]\n\ts_or_b32 %[bits], %[t1], 4\n\ts_or_b32 %[res], %[sm], 1\n\ts_branch 99f\n"
        FS_EXIT_HEAD("104", "%[R]", "%[thr]", "134f", "144") FS_MINIT("2") FS_MBIT("152f", "162") FS_MBIT("153f", "163") FS_MASKS("2", "4", "31", "3") "s_branch 99f\n"
        FS_EXIT_HEAD("105", "%[thr]", "%[R]", "135f", "145") FS_MINIT("3") FS_MBIT("154f", "164") FS_MBIT("155f", "165") FS_MBIT("156f", "166") FS_MASKS("3", "5", "63", "7") "s_branch 99f\n"
        FS_EXIT_HEAD("106", "%[R]", "%[thr]", "136f", "146") "s_mov_b32 %[idx], 6\n\t" FS_MINIT("4") "s_branch 204f\n"
        FS_EXIT_HEAD("107", "%[thr]", "%[R]", "137f", "147") "s_mov_b32 %[idx], 7\n\t" FS_MINIT("5") "s_branch 205f\n"
        FS_EXIT_HEAD("108", "%[R]", "%[thr]", "138f", "148") "s_mov_b32 %[idx], 8\n\t" FS_MINIT("6") "s_branch 206f\n"
        FS_EXIT_HEAD("109", "%[thr]", "%[R]", "139f", "149") "s_mov_b32 %[idx], 9\n\t" FS_MINIT("7")
        "207:\n\t" FS_MBIT("177f", "187") "206:\n\t" FS_MBIT("176f", "186") "205:\n\t" FS_MBIT("175f", "185") "204:\n\t" FS_MBIT("174f", "184")
        FS_MBIT("173f", "183") FS_MBIT("172f", "182") FS_MBIT("171f", "181") "s_branch 60b\n"
        FS_RENORM_R("116", "126b", "%[thr]") FS_RENORM("117", "127b") FS_RENORM_R("118", "128b", "%[thr]") FS_RENORM("119", "129b")
        FS_RENORM("136", "146b") FS_RENORM("137", "147b") FS_RENORM("138", "148b") FS_RENORM("139", "149b")
        FS_RENORM("171", "181b") FS_RENORM("172", "182b") FS_RENORM("173", "183b") FS_RENORM("174", "184b") FS_RENORM("175", "185b") FS_RENORM("176", "186b") FS_RENORM("177", "187b")
        FS_RENORM("115", "125b") FS_RENORM("135", "145b") FS_RENORM("154", "164b") FS_RENORM("155", "165b") FS_RENORM("156", "166b")
        FS_RENORM_R("112", "122b", "%[thr]") FS_RENORM("113", "123b") FS_RENORM_R("114", "124b", "%[thr]")
        FS_RENORM("132", "142b") FS_RENORM("134", "144b")
        FS_RENORM("152", "162b") FS_RENORM("153", "163b")
        FS_RENORM("133", "143f") FS_RENORM("151", "161f")
        FS_EXIT_HEAD("103", "%[thr]", "%[R]", "133b", "143") FS_MINIT("1") FS_MBIT("151b", "161") FS_MASKS("1", "3", "15", "1")
        "99:\n\t"
        : [R] "+s"(R), [L] "+s"(Lo), [widx] "+s"(widx), [res] "=&s"(res), [touched] "=&s"(touched), [bits] "=&s"(bits), [t0] "=&s"(t0),
          [t1] "=&s"(t1), [thr] "=&s"(thr), [idx] "=&s"(idx), [ilast] "=&s"(ilast), [sm] "=&s"(sm), [hv] "=&s"(hv), [e] "=&s"(e),
          [midx] "=&s"(midx), [amax] "=&s"(amax), [have] "=&s"(have), [skipped] "=&s"(skipped)
        : [leafv] "v"(L.leafv), [win] "v"(win), [amaxp] "s"(F.amax_pos), [amaxn] "s"(F.amax_neg), [ilastp] "s"(F.ilast_pos),
          [ilastn] "s"(F.ilast_neg), [mb] "s"(L.mb), [mb2] "s"(L.mb - 2)
        : "scc", "vcc", FS_VA, FS_VB, FS_VC, FS_VK0, FS_VK1);
    r.range = R; r.low = Lo; widx_io = widx;
    L.touched = touched; L.bits = bits;
    return (int)res;
}
DEV int fast_symbol_hw(Rac &r, Stream &s, LeafRegs &L, const FastSym &F) {
    uint32_t widx = s.pos - s.win_base;
    const int res = fast_symbol_hw_w(r, widx, s.win, L, F);
    s.pos = s.win_base + widx;
    return res;
}
#endif
DEV void leaf_commit(LeafRegs &L, int lane, const uint16_t *table) {
#ifdef FUIF_EMU
    const int slot = lane & (L.mirror == 1u ? 31 : 15);   // (lanes 32..63 mirror 0..31; compact leaves: 16..63 mirror 0..15)
    if ((L.touched >> slot) & 1u) L.leafv = table[L.leafv * 2 + ((L.bits >> slot) & 1u)];
#else
    // the scalar masks ARE lane masks: inverse_ballot hands them to the compiler as per-lane conditions (EXEC and a v_cndmask
    // operand) without a vector test per lane.  (An inline-asm load here would be invisible to the compiler's s_waitcnt
    // placement: tools/test_fast_symbol.hip caught exactly that.)
    // both halves: lanes 32..63 mirror 0..31 (compact leaves: the 16-bit slot masks are first doubled into lanes 16..31, one scalar multiply each)
    const unsigned long long b64 = (unsigned long long)(L.bits * L.mirror) * 0x100000001ull, t64 = (unsigned long long)(L.touched * L.mirror) * 0x100000001ull;
    const uint32_t boff = __builtin_amdgcn_inverse_ballot_w64(b64) ? 2u : 0u;
    if (__builtin_amdgcn_inverse_ballot_w64(t64))
        L.leafv = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(table) + (((uint32_t)L.leafv << 2) | boff));
#endif
    L.touched = 0; L.bits = 0;
}

// maniac/symbol.h:115-138
DEV void symbol_chance_init(uint16_t *ch, int zero_chance) {
    uint32_t rp = 0x1000 - zero_chance;
    ch[CH_ZERO] = (uint16_t)zero_chance;
    ch[CH_SIGN] = 0x800;
    for (int i = 0; i < kMaxBitDepth - 1; i++) {
        if (rp < 0x100) rp = 0x100;
        if (rp > 0xf00) rp = 0xf00;
        ch[CH_EXP + i] = (uint16_t)(0x1000 - rp);
        rp = (rp * rp + 0x800) >> 12;
    }
    for (int i = 0; i < kMaxBitDepth; i++) ch[CH_MANT + i] = 1024;
}

// encoding/encoding.cpp:61-72
DEV bool check_bit_depth(int minv, int maxv, int predictor) {
    int maxav = iabs(maxv);
    if (-minv > maxav) maxav = -minv;
    if (predictor > 0 && maxv - minv > maxav) maxav = maxv - minv;
    if (predictor > 0 && iabs(minv - maxv) > maxav) maxav = iabs(minv - maxv);
    return ilog2u((uint32_t)maxav) + 1 <= kMaxBitDepth;
}

struct RefChan {  // one reference channel of the current group (context_predict.h:233-289)
    int64_t off;  // element offset of the plane inside the image's coefficient slab
    int32_t w, h, hshift, vshift;
    int32_t chan, pad;
};

template <int kLdsSuper>
struct Shared {
    static constexpr bool kDense = kLdsSuper == kLdsDense;
    uint2 snodes[kLdsSuper > 0 ? kLdsSuper * 64 : 1];   // breadth-first top of the supernode tree: lane i = {split_i, prop_i << 2 | exit_i << 8}
    static constexpr int kChunk = kDense ? kChunkDense : 64;
    static constexpr int kPropWords = (kChunk > 4 ? kChunk : 4) * kPropPitch;
    static constexpr int kLdsN = 2 * kLdsSuper;   // narrow supernodes (256 bytes) resident in LDS: twice as many as 512-byte ones fit the wide configurations' `snodes`
    int32_t cprops[kPropWords]; // [pixel of the chunk][property]; the supernode build borrows 256 words
    uint16_t meta_ctx[3][32];            // three SimpleSymbolCoder contexts of the tree coder
    int32_t lo[kMaxProps], hi[kMaxProps];
    RefChan refs[kMaxRefs];
};

// kHandOff = false: every image is one tile, nothing a tile writes is read by another one before the
// kernel ends -- plain cached stores and loads (write-through stores drop the line from L2, and the
// decoder re-reads its own previous rows: ~4 % on a whole-stream decode)
template <bool kHandOff>
DEV void st_plane(coef_t *p, int v) {   // a sample is stored as the reference stores it: narrowed to pixel_type (image/image.h:35)
    if (kHandOff) st_agent(p, v);
    else *p = (coef_t)v;
}
template <bool kHandOff>
DEV int ld_plane(const coef_t *p) { return kHandOff ? ld_agent(p) : (int)*p; }
template <bool kHandOff>
DEV void fill_plane(coef_t *plane, int64_t first, int64_t count, int v, int lane) {
    for (int64_t i = first + lane; i < first + count; i += 64) st_plane<kHandOff>(plane + i, v);
}

}  // namespace

template <int kLdsSuper, bool kHandOff>
__global__ __launch_bounds__(64) FUIF_OCCUPANCY void k_maniac_decode(DecodeParams P) {
    __shared__ Shared<kLdsSuper> sh;
    constexpr int kChunk = Shared<kLdsSuper>::kChunk;
    constexpr int kLdsN = Shared<kLdsSuper>::kLdsN;          // narrow supernodes resident in LDS (behind the root)
    const int lane = threadIdx.x;
    uint32_t *const lds_narrow = reinterpret_cast<uint32_t *>(sh.snodes);   // (a narrow group keeps 2 * kLdsSuper supernodes of 256 bytes there)

    const uint16_t *tree_table = P.tables;          // cut 2, alpha 0xFFFFFFFF/19 (compound.h:262)
    const uint16_t *pixel_table = P.tables + 8192;  // cut 6, alpha 0x0d000000 (encoding.h:54-55)

    uint8_t *scratch = P.scratch + (size_t)blockIdx.x * P.scratch_stride;  // per wavefront, reused from tile to tile
    Node *nodes = reinterpret_cast<Node *>(scratch);                          // parse-order nodes
    uint2 *const snodes_w = reinterpret_cast<uint2 *>(scratch + P.bfs_off);   // supernodes (64 x 8 B each)
    uint16_t *const leaves_w = reinterpret_cast<uint16_t *>(scratch + P.leaves_off);
    Frame *stack = reinterpret_cast<Frame *>(scratch + P.stack_off);
    int32_t *queue = reinterpret_cast<int32_t *>(scratch + P.queue_off);      // breadth-first work list
    uint16_t *subtree = reinterpret_cast<uint16_t *>(scratch + P.subtree_off); // nodes under every tree node (saturating)
    const ChannelGeom *geom = P.geom;
    const int nch = P.n_channels;
#ifdef FUIF_EMU
    emu_lds_base = reinterpret_cast<const char *>(&sh.snodes[0]);
    const uint32_t lds_nodes_addr = 0;
    const uint32_t lds_narrow_addr = (uint32_t)(reinterpret_cast<const char *>(lds_narrow) - reinterpret_cast<const char *>(&sh.snodes[0]));
#else
    const uint32_t lds_nodes_addr = (uint32_t)(uintptr_t)(&sh.snodes[0]);  // LDS byte offset (low half of the flat address)
    const uint32_t lds_narrow_addr = (uint32_t)(uintptr_t)lds_narrow;
#endif
    // geometry of a 6-level supernode in heap order (children of slot k: 2k+1 = "> split", 2k+2 = "<= split"):
    // lane e owns exit e; exp/msk = the decisions its path needs and the slots they sit in
    int exit_q = 0;
    uint32_t exp_lo = 0, exp_hi = 0, msk_lo = 0, msk_hi = 0;
    for (int d = 0; d < 6; d++) {
        const int gt = (lane >> (5 - d)) & 1;
        if (exit_q < 32) { msk_lo |= 1u << exit_q; if (gt) exp_lo |= 1u << exit_q; }
        else { msk_hi |= 1u << (exit_q - 32); if (gt) exp_hi |= 1u << (exit_q - 32); }
        exit_q = 2 * exit_q + (gt ? 1 : 2);
    }

    // ---- persistent wavefront ---------------------------------------------------------------------
    // sched == 0: take tiles from one list until it is empty.  A tile only waits for tiles EARLIER in the list, and a
    // tile that has been taken is running on a resident wavefront, so the earliest unfinished tile can always make
    // progress: no deadlock, whatever the dispatch order or placement of the wavefronts.
    // sched == 1 (maniac_decode.h): the same argument per image -- tiles of an image are started in stream order, a
    // suspended tile waits for rows of earlier tiles of its image only, and the earliest unfinished tile of an image is
    // either running or runnable, so whichever wavefront looks at that image next picks it up.
    const int n_queues = P.n_queues;
    const bool sched = P.sched != 0;
    int home_q = 0;
    uint32_t simd_key = 0, simd_slot = 0;   // simd_slot: this wavefront's entry of P.simd_long (CU key x 4 + SIMD)
    constexpr uint32_t kLongClass = 2, kRecLong = 1u << 12;   // size class of a long tile (>= 1/8 of its picture's samples); its mark in TileRec::flags
    constexpr uint32_t kRecNarrow = 1u << 13;                 // TileRec::flags: the suspended group's supernodes are narrow (4 bytes per lane)
    constexpr uint32_t kRecCompact = 1u << 14;                //                 ... its leaves are compact (16 chances, 32 bytes)
    if (sched) {
        // home queue = dense index of the CU this wavefront sits on (the first arrival numbers it)
#ifdef FUIF_EMU
        const uint32_t key = (uint32_t)blockIdx.x >> 2;
#else
        const uint32_t hw = (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID: cu / sh / se in [15:8]
        const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
        const uint32_t key = ((hw >> 8) & 0xFFu) | ((xcc & 15u) << 8);
#endif
        simd_key = key;
#ifdef FUIF_EMU
        simd_slot = key * 4u + ((uint32_t)blockIdx.x & 3u);
#else
        simd_slot = key * 4u + ((hw >> 4) & 3u);   // HW_ID[5:4] = SIMD
#endif
        uint32_t idx = 0;
        if (lane == 0) {
            atomicAdd(&P.cu_alive[key], 1u);
            if (atomicAdd(&P.simd_claim[2 * key], 1u) == 0u) {
                idx = atomicAdd(&P.simd_claim[2 * 4096], 1u);
                st_agent(&P.simd_claim[2 * key + 1], idx + 1u);
            } else {
                uint32_t v = 0;
                while ((v = ld_agent(&P.simd_claim[2 * key + 1])) == 0u) __builtin_amdgcn_s_sleep(2);
                idx = v - 1u;
            }
        }
        home_q = (int)(rflu(idx) % (uint32_t)n_queues);
    }
    constexpr uint32_t kNoWork = 0xFFFFFFFFu, kResume = 0x80000000u;
    const uint32_t my_pin = (uint32_t)blockIdx.x + 1u;
    // A tile that found no context area (the arenas are bump-allocated per launch and can run out: big trees, a small
    // FUIFGPU_CTX_MB) keeps its supernodes and leaves in THIS wavefront's scratch area.  It is suspended like any other
    // tile when it has to wait -- a waiting tile never holds a wavefront, that is what keeps the earliest unfinished tile of
    // an image runnable -- but only this wavefront can resume it ("pinned"), and until it is finished this wavefront
    // starts no fresh tile (the tree parse would overwrite the area); it still resumes other tiles, which live in arenas.
    uint32_t pinned_tix = kNoWork;
    uint32_t my_turn = (uint32_t)blockIdx.x;   // which active image of the queue a look starts with (wavefront-local: no shared counter to hammer)
    // sched == 1: work in queue q -- the next unstarted tile of the first image that has one, else (resume_ok) a suspended tile of
    // this CU whose awaited rows have arrived
    auto scan_queue = [&](int q, bool fresh_ok, bool resume_ok, bool patient) -> uint32_t {
        const uint32_t ib = P.q_img_begin[q], ie = P.q_img_begin[q + 1];
        // long tiles spread evenly over the SIMDs: while this SIMD runs its share already, a long tile is left to a wavefront of another SIMD of
        // the CU -- unless this wavefront has found nothing else for a while (`patient` false): a runnable tile never waits for long
        const bool simd_full = patient && P.long_per_simd > 0 && rflu(ld_agent(&P.simd_long[simd_slot])) >= (uint32_t)P.long_per_simd;
        // 1. an unstarted tile of one of the first kActiveImages unfinished images of the queue (stream order inside an
        //    image).  Starting comes first: the long final groups of an image must be under way early, and a tile that
        //    has nothing to do yet suspends itself at its first row.
        uint32_t active = 0, k_end = ib;
        for (uint32_t k = ib; k < ie && active < kActiveImages; k++) {
            const uint32_t im = P.q_images[k];
            const uint32_t tb = P.img_tile_begin[im], tn = P.img_tile_begin[im + 1] - tb;
            k_end = k + 1;
            if (rflu(ld_agent(&P.img_done[im])) >= tn) continue;
            active++;
            uint32_t fresh = kNoWork;
            uint32_t peek = lane == 0 ? ld_agent(&P.img_next[im]) : 0u;
            peek = rflu(peek);
            const bool next_is_long = simd_full && peek < tn && ((rflu(P.tiles[tb + peek].flags) >> kTileSizeClassShift) & 15u) <= kLongClass;
            if (fresh_ok && !next_is_long && lane == 0 && peek < tn) {
                const uint32_t n = atomicAdd(&P.img_next[im], 1u);
                if (n < tn) {
                    fresh = tb + n;
                    // the tile is counted as live on this CU BEFORE it is counted as started: a wavefront that sees
                    // started_total == n_tiles sees every cu_live increment (returning atomics: the second is issued
                    // after the first has come back)
                    uint32_t seen = atomicAdd(&P.cu_live[simd_key], 1u);
                    if (q != home_q) { seen |= atomicAdd(&P.cu_foreign[simd_key], 1u); P.tile_rec[fresh].foreign = 1u; }
                    if (seen != 0xFFFFFFFFu) atomicAdd(P.started_total, 1u);
                }
            }
            fresh = rflu(fresh);
            if (fresh != kNoWork) return fresh;
        }
        if (!resume_ok) return kNoWork;
        // 2. a suspended tile whose awaited rows have arrived: the active images take turns (they should finish together,
        //    the last one alone could not keep a CU busy)
        const uint32_t turn = my_turn++;
        const uint32_t nwin = k_end - ib;
        for (uint32_t j = 0; j < nwin; j++) {
            const uint32_t k = ib + (turn + j) % nwin;
            const uint32_t im = P.q_images[k];
            const uint32_t tb = P.img_tile_begin[im], tn = P.img_tile_begin[im + 1] - tb;
            if (rflu(ld_agent(&P.img_done[im])) >= tn) continue;
            // Inside an image the LAST runnable tile first: a tile can only run when the tiles it reads from are ahead of
            // it, so serving the downstream end of that pipeline first keeps every stage moving and the long final
            // groups finish with their leaders instead of after them (lowest-first starved them: measured).
            for (uint32_t t0 = (tn - 1u) & ~63u;; t0 -= 64) {
                bool run = false;
                if (t0 + lane < tn) {
                    TileRec *r = P.tile_rec + tb + t0 + lane;
                    // A suspended tile is resumed on the CU that suspended it: its tree and leaf chances were written with
                    // plain cached stores, and only that CU's own L1 / XCD's L2 are guaranteed to show them (agent-scope
                    // release / acquire fences at every suspension cost an L2 write-back each: measured, 1.3x slower).
                    if (ld_agent(&r->state) == TS_READY && ld_agent(&r->owner) == simd_key + 1u) {
                        const uint32_t pin = ld_agent(&r->pin);
                        if ((pin == 0u || pin == my_pin) && !(simd_full && (ld_agent(&r->flags) & kRecLong))) {
                            const uint32_t wc = ld_agent(&r->wait_chan), wv = ld_agent(&r->wait_val);
                            run = ld_agent(&P.progress[(size_t)im * nch + wc]) >= wv;
                        }
                    }
                }
                unsigned long long m = __ballot(run);
                while (m) {
                    const int l = 63 - __builtin_clzll(m);
                    uint32_t ok = 0;
                    if (lane == 0) ok = atomicCAS(&P.tile_rec[tb + t0 + l].state, (uint32_t)TS_READY, (uint32_t)TS_RUNNING) == TS_READY ? 1u : 0u;
                    if (rflu(ok)) return (tb + t0 + (uint32_t)l) | kResume;
                    m &= ~(1ull << l);
                }
                if (t0 == 0) break;
            }
        }
        return kNoWork;
    };
    uint32_t idle_rounds = 0, stale_looks = 0, last_beat = 0;
    STATS(unsigned long long st_idle = 0, st_picks = 0, st_yields = 0, st_scan = 0, st_spin = 0, st_noctx = 0, st_busy = 0, st_pick_t = 0, st_t0 = realtime();
          const unsigned long long st_begin = st_t0;)   // scheduler statistics of this wavefront
    for (;;) {
    uint32_t tix = kNoWork;
    bool resumed = false;
#ifndef FUIF_EMU
    __builtin_amdgcn_s_setprio(0);   // looking for work never competes with decoding
#endif
    if (!sched) {
        if (lane == 0) { const uint32_t k = atomicAdd(&P.q_head[0], 1u); if (k < (uint32_t)P.n_tiles) tix = k; }
        tix = rflu(tix);
        if (tix == kNoWork) break;
    } else {
        // The home queue first.  Other queues are only worth a look while tiles are still unstarted somewhere (a queue whose
        // CU got no wavefronts of its own must still be served) or when this CU owns a suspended tile of another queue; a
        // tile started by a wavefront of another CU can only be resumed by that CU (such tiles finished seconds late:
        // measured), so a foreign queue is a last resort, reached after a long idle spell.
        STATS(const unsigned long long sc0 = realtime(); if (st_pick_t) { st_busy += sc0 - st_pick_t; st_pick_t = 0; })
        const bool can_start = pinned_tix == kNoWork;
        const bool patient = idle_rounds < 8u;
        tix = scan_queue(home_q, can_start, true, patient);
        const bool all_started = rflu(ld_agent(P.started_total)) >= (uint32_t)P.n_tiles;
        if (tix == kNoWork && idle_rounds > 4096u) {
            const bool foreign_mine = rflu(ld_agent(&P.cu_foreign[simd_key])) != 0u;
            if ((!all_started && can_start) || foreign_mine)
                for (int d = 1; d < n_queues && tix == kNoWork; d++)
                    tix = scan_queue(home_q + d < n_queues ? home_q + d : home_q + d - n_queues, can_start && !all_started, foreign_mine, false);
        }
        if (tix == kNoWork) {
            STATS(if (idle_rounds == 0) st_t0 = sc0;)
            if (rflu(ld_agent(P.done_total)) >= (uint32_t)P.n_tiles) { STATS(st_idle += realtime() - st_t0;) break; }
            // Retire: once every tile of the launch has been started, a wavefront without work can only ever resume tiles of
            // its own CU, and a CU needs no more wavefronts than it has unfinished tiles.  The surplus ones leave (cu_alive
            // never drops below cu_live, and cu_live can only fall from here on): thousands of idle wavefronts polling
            // agent-scope words slow the ones that decode (a launch of 128 pictures ran the same tiles 2.5x slower per
            // symbol than a launch of 1024, profiles/r2_idle_polling.txt), and the slots they free are there for the
            // next launch's wavefronts.
            if (all_started && can_start) {
                uint32_t gone = 0;
                if (lane == 0) {
                    const uint32_t w = ld_agent(&P.cu_alive[simd_key]), l = ld_agent(&P.cu_live[simd_key]);
                    if (w > l && atomicCAS(&P.cu_alive[simd_key], w, w - 1u) == w) gone = 1u;
                }
                if (rflu(gone)) { STATS(st_idle += realtime() - st_t0;) break; }
            }
            // A wavefront without work looks again after 1, then 4, then 32 naps of 8128 cycles (idle_rounds counts naps).
            const uint32_t naps = idle_rounds < 8u ? 1u : (idle_rounds < 64u ? 4u : 32u);
            idle_rounds += naps;
            if (naps == 32u) {
                // only a lost tile gets past this (never observed): nothing in the whole launch has moved for ~7 s
                const uint32_t beat = rflu(ld_agent(P.heartbeat)) + rflu(ld_agent(P.done_total));
                stale_looks = beat == last_beat ? stale_looks + 1u : 0u;
                last_beat = beat;
                if (stale_looks > kIdleStaleLooks) {
                    for (uint32_t k = P.q_img_begin[home_q]; k < P.q_img_begin[home_q + 1]; k++) {
                        const uint32_t im = P.q_images[k];
                        if (lane == 0 && ld_agent(&P.img_done[im]) < P.img_tile_begin[im + 1] - P.img_tile_begin[im]) atomicOr(&P.status[im], ST_STALLED | ST_CORRUPT);
                    }
                    break;
                }
            }
            for (uint32_t k = 0; k < naps; k++) __builtin_amdgcn_s_sleep(127);
            continue;
        }
        STATS(if (idle_rounds) st_idle += sc0 - st_t0; st_pick_t = realtime(); st_scan += st_pick_t - sc0; st_picks++;)
        idle_rounds = 0; stale_looks = 0;
        resumed = (tix & kResume) != 0u;
        tix &= ~kResume;
    }
    const Tile tile = P.tiles[tix];
    const bool long_tile = sched && ((rflu(tile.flags) >> kTileSizeClassShift) & 15u) <= kLongClass;
    if (long_tile && lane == 0) atomicAdd(&P.simd_long[simd_slot], 1u);   // (taken back at the suspension or the end of the tile)
    TLOG(const unsigned long long tile_t0 = realtime();)
#ifndef FUIF_EMU
    {
        // the few tiles that hold most of an image are its critical path: they get the issue slots first (s_setprio),
        // the many short tiles fill what is left (size class: Tile::flags, host side)
        const int cls = (int)((rflu(tile.flags) >> kTileSizeClassShift) & 15u), over = cls - P.prio_base;
        const int prio = P.prio_base < 0 ? 0 : over <= 0 ? 3 : over >= 3 ? 0 : 3 - over;
        if (prio == 3) __builtin_amdgcn_s_setprio(3);
        if (prio == 2) __builtin_amdgcn_s_setprio(2);
        if (prio == 1) __builtin_amdgcn_s_setprio(1);
        if (prio == 0) __builtin_amdgcn_s_setprio(0);
    }
#endif
    STATS(unsigned long long waited = 0;)
    const int img = rfl((int)tile.image);
    const int first_c = rfl(tile.first_channel), last_c = rfl(tile.last_channel);

    const StreamJob job = P.jobs[img];
    Stream s;
    s.p = P.blobs + job.blob_off;
    s.size = rflu(job.blob_size);
    s.pos = rflu(tile.start);
    s.limit = rflu(job.limit);
    s.win_base = 0xFFFFFF00u;
    s.win = 0;
    s.eof_flag = 0;
    s.blob_mode = rfl((int)(job.flags & 1u));

    coef_t *coef = P.coef + (int64_t)img * P.coef_stride;
    ChannelMeta *meta = P.meta + (int64_t)img * P.n_channels;
    uint32_t *progress = P.progress + (size_t)img * nch;
    int status = 0;
    bool stalled = false;
    // sched == 1: this tile's record; supernodes and leaf chances of a tile that may be suspended live in a context area
    TileRec *const rec = sched ? P.tile_rec + tix : nullptr;
    uint2 *snodes_g = snodes_w;
    uint16_t *leaves = leaves_w;
    int ctx_slot = -1;          // >= 0: the tile owns a context area (256-byte units into ctx_scratch)
    bool can_yield = false;     // the tile gives the wavefront back instead of spinning: it owns a context area, or its context is pinned to this wavefront's scratch area
    uint32_t ctx_leaves_units = 0;
    bool yielded = false;
    uint32_t yield_chan = 0, yield_val = 0, resume_y = 0;
    if (resumed) {
        s.pos = rflu(rec->pos);
        const uint32_t fl = rflu(rec->flags);
        status = (int)(fl & 0xFFu);
        s.eof_flag = (int)((fl >> 8) & 1u);
        ctx_slot = rfl((int)rec->ctx);
        resume_y = rflu(rec->y);
        ctx_leaves_units = rflu(rec->ctx_leaves);
        can_yield = true;
        if (ctx_slot >= 0) {
            uint8_t *cb = P.ctx_scratch + (size_t)(uint32_t)ctx_slot * 256u;
            snodes_g = reinterpret_cast<uint2 *>(cb);
            leaves = reinterpret_cast<uint16_t *>(cb + (size_t)ctx_leaves_units * 256u);
        }   // else: pinned to this wavefront -- the scratch area still holds its parse-order nodes, supernodes and leaves
    }
    // a wait inside a tile that cannot be suspended: bounded by "nothing in the launch moves" (kStaleLimit)
    uint32_t stale = 0, beat_seen = 0;
    auto spin_nap = [&](uint32_t &spins) {
        __builtin_amdgcn_s_sleep(8);
        if ((++spins & (kStaleCheck - 1u)) == 0u) {
            const uint32_t beat = rflu(ld_agent(P.heartbeat)) + rflu(ld_agent(P.done_total));
            stale = beat == beat_seen ? stale + 1u : 0u;
            beat_seen = beat;
            if (stale > kStaleLimit) { stalled = true; status |= ST_STALLED | ST_CORRUPT; }
        }
    };
    PROF_DECL;
#ifdef FUIF_PROF
    const unsigned long long prof_rt0 = realtime();   // slot 6: 100 MHz ticks of the run segment (against slot 7: the shader clock the segment ran at)
    const unsigned long long prof_seg0 = __builtin_readcyclecounter();   // slot 7: cycles of the whole run segment (pick-up to suspension / end)
#endif
    // progress word of channel c: 1 = ChannelMeta valid, 1 + r = rows [0,r) final, 1 + h = plane final
    auto publish = [&](int c, uint32_t v) {
        if (kHandOff) {
            drain_stores();
            if (lane == 0) {
                st_agent(progress + c, v);
                if ((v & 31u) == 0u) atomicAdd(P.heartbeat, 1u);   // sign of life for the stall verdicts (every 32nd row is plenty)
            }
        }
    };
    auto wait_header = [&](int c) {
        uint32_t spins = 0;
        if (!stalled && rflu(ld_agent(progress + c)) == 0u) {
            STATS(const unsigned long long w0 = realtime();)
            while (!stalled && rflu(ld_agent(progress + c)) == 0u) spin_nap(spins);
            STATS(waited += realtime() - w0;)
        }
    };

    // ---- fuif_decode channel loop: encoding.cpp:708-717 -------------------------------------
    for (int ci = resumed ? last_c : first_c; ci <= last_c; ci++) {
        // a resumed tile is one single-channel group (its last channel; earlier ones, if any, are empty planes): the
        // header, tree and leaves exist, the code below only rebuilds what lives in registers and LDS
        int beginc = ci, endc = ci, compress = 1, predictor = 0, firstrealc = ci;
        if (resumed) predictor = (int)((rflu(rec->flags) >> 9) & 7u);
        if (!resumed) {
        if (!((s.limit == 0 || s.pos < s.limit) && !s_eof(s))) break;
        if (!rfl(geom[ci].w) || !rfl(geom[ci].h)) continue;

        // ---- fuif_decode_channel: encoding.cpp:259-429 --------------------------------------
        if (s_limit_hit(s)) continue;
        const uint32_t group_pos = s.pos;
        int firstbyte = s_varint(s, lane);
        if (s_limit_hit(s)) continue;
        endc = beginc + (firstbyte >> 4);
        compress = firstbyte & 1;
        predictor = (firstbyte & 14) >> 1;
        int global_minv = 1 - s_varint(s, lane);
        if (s_limit_hit(s)) continue;
        if (global_minv == 1) global_minv = s_varint(s, lane);
        if (s_limit_hit(s)) continue;
        const int global_maxv = global_minv + s_varint(s, lane);
        if (s_limit_hit(s)) continue;
        if (endc > last_c || endc < beginc) { status |= ST_CORRUPT; break; }  // a group never crosses a tile (or the channel list)
        if (lane == 0) P.group_start[(size_t)img * nch + beginc] = group_pos + 1u;

        firstrealc = beginc;
        bool fatal = false, early = false;
        for (int i = beginc; i <= endc; i++) {
            const int gw = rfl(geom[i].w), gh = rfl(geom[i].h);
            const int64_t goff = geom[i].coef_off;
            if ((int64_t)gw * gh <= 0) { publish(i, 1u); continue; }
            int minv = global_minv, maxv = global_maxv;
            if (endc > beginc && global_minv < global_maxv) {
                minv += s_varint(s, lane);
                maxv = minv + s_varint(s, lane);
            }
            int q = 1;
            if (minv == maxv) {
                fill_plane<kHandOff>(coef + goff, 0, (int64_t)gw * gh, minv, lane);
                firstrealc++;
            }
            bool have_q = !(minv == 0 && maxv == 0);
            if (have_q) q = s_varint(s, lane);
            if (lane == 0) { st_agent(&meta[i].minval, minv); st_agent(&meta[i].maxval, maxv); st_agent(&meta[i].q, q); st_agent(&meta[i].decoded, (minv == maxv) ? 1 : 0); }
            publish(i, (minv == maxv) ? (uint32_t)gh + 1u : 1u);  // header known (constant planes are already final)
            if (!have_q) continue;
            if (s_limit_hit(s)) {  // corrupt_or_truncated: encoding.cpp:209-219 (isEOF or limit => zero-fill, true)
                fill_plane<kHandOff>(coef + goff, 0, (int64_t)gw * gh, 0, lane);
                if (lane == 0) st_agent(&meta[i].decoded, 1);
                publish(i, (uint32_t)gh + 1u);
                status |= ST_TRUNCATED;
                early = true;
                break;
            }
            // an inverted range can only come from a corrupt varint; the reference's uniform coder would recurse
            // without end on it (symbol.h:44-57 asserts len >= 0), so it is reported as corruption here
            if (maxv < minv) { fatal = true; break; }
            if (compress && !check_bit_depth(minv, maxv, predictor)) { fatal = true; break; }
        }
        __syncthreads();  // meta[] written by lane 0 is read below by every lane
        if (fatal) { status |= ST_UNSUPPORTED | ST_CORRUPT; break; }
        if (early) continue;
        if (firstrealc > endc) { ci = endc; continue; }
        }  // !resumed

        // ---- init_properties: context_predict.h:67-120 --------------------------------------
        int nrefs = 0;
        int nprops = 0;
        {
            int offset = 0;
            for (int j = beginc - 1; j >= 0 && offset < P.max_properties; j--) {
                if (j < first_c) wait_header(j);  // another tile's channel: its header may still be on its way
                const int jmin = rfl(ld_agent(&meta[j].minval)), jmax = rfl(ld_agent(&meta[j].maxval));
                if (jmin == jmax) continue;
                if (rfl(geom[j].hshift) < 0) continue;
                int mn = jmin; if (mn > 0) mn = 0;
                int mx = jmax; if (mx < 0) mx = 0;
                if (lane == 0) {
                    sh.lo[nprops] = 0; sh.hi[nprops] = iabs(mx > -mn ? mx : mn);
                    sh.lo[nprops + 1] = slog(mn); sh.hi[nprops + 1] = slog(mx);
                    RefChan rc;
                    rc.off = geom[j].coef_off;
                    rc.w = geom[j].w; rc.h = geom[j].h; rc.hshift = geom[j].hshift; rc.vshift = geom[j].vshift;
                    rc.chan = j; rc.pad = 0;
                    sh.refs[nrefs] = rc;
                }
                nprops += 2; offset += 2;
                nrefs++;
            }
            int mn = 0x7FFFFFFF, mx = (int)0x80000001, maxh = 0, maxw = 0;
            for (int j = beginc; j <= endc; j++) {
                // zero-pixel channels keep their constructor range (0,0 for inserted residual
                // channels) in the reference; meta[] is zero-initialised likewise
                const int jmin = rfl(ld_agent(&meta[j].minval)), jmax = rfl(ld_agent(&meta[j].maxval));
                if (jmin < mn) mn = jmin;
                if (jmax > mx) mx = jmax;
                const int jh = rfl(geom[j].h), jw = rfl(geom[j].w);
                if (jh > maxh) maxh = jh;
                if (jw > maxw) maxw = jw;
            }
            if (mn > 0) mn = 0;
            if (mx < 0) mx = 0;
            const int amax = iabs(mn) > iabs(mx) ? iabs(mn) : iabs(mx);
            if (lane == 0) {
                int n = nprops;
                sh.lo[n] = 0; sh.hi[n] = amax; n++;
                sh.lo[n] = 0; sh.hi[n] = amax; n++;
                sh.lo[n] = slog(mn); sh.hi[n] = slog(mx); n++;
                sh.lo[n] = slog(mn); sh.hi[n] = slog(mx); n++;
                sh.lo[n] = 0; sh.hi[n] = maxh - 1; n++;
                sh.lo[n] = 0; sh.hi[n] = maxw - 1; n++;
                sh.lo[n] = mn + mn - mx; sh.hi[n] = mx + mx - mn; n++;
                sh.lo[n] = mn + mn - mx; sh.hi[n] = mx + mx - mn; n++;
                for (int k = 0; k < 5; k++) { sh.lo[n] = slog(mn - mx); sh.hi[n] = slog(mx - mn); n++; }
            }
            nprops += kNonRefProps;
        }
        const int nrefprops = nprops - kNonRefProps;
        // property rows of a chunk: 33 words and the full chunk up to 31 properties, 65 words and half the chunk beyond (the LDS array
        // is sized for kChunk rows of 33 words); lane p of the pixel loop holds property p either way
        const bool wide_props = nprops > 32;
        const int prop_pitch = wide_props ? kPropPitchWide : kPropPitch;
        const int prop_mask = wide_props ? 63 : 31;
        const int chunk_px = wide_props ? kChunk / 2 : kChunk;

        int predictability = 2048;
        Rac rac;
        int tree_size = 1, n_super = 1, cur_leaf = 0;
        bool narrow = false;   // this group's supernodes are 4-byte lane words (kLeafFlagN): decided after the tree parse, kept in TileRec::flags across a suspension
        bool compact = false;  // this group's leaves are 16 chances of 32 bytes (LeafRegs::mb): every symbol of it has at most 8 magnitude bits
        if (resumed) {
            rac.range = rflu(rec->range); rac.low = rflu(rec->low);
            tree_size = rfl((int)rec->tree_size); n_super = rfl((int)rec->n_super); cur_leaf = rfl((int)rec->cur_leaf);
            narrow = (rflu(rec->flags) & kRecNarrow) != 0u;
            compact = (rflu(rec->flags) & kRecCompact) != 0u;
            if (ctx_slot < 0) leaves = reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(snodes_w) + (size_t)n_super * (narrow ? 256u : 512u));   // pinned: see the leaf placement below
            if (narrow) for (int sn = 1; sn <= kLdsN && sn < n_super; sn++) lds_narrow[(sn - 1) * 64 + lane] = reinterpret_cast<const uint32_t *>(snodes_g)[(size_t)sn * 64 + lane];
            else for (int sn = 1; sn <= kLdsSuper && sn < n_super; sn++) sh.snodes[(sn - 1) * 64 + lane] = snodes_g[(size_t)sn * 64 + lane];
            __syncthreads();
        }
        if (!resumed) {
        if (predictor == 0 && compress) {
            int rounded = s_varint(s, lane);
            if (rounded < 1 || rounded > 127) {
                if (s_limit_hit(s)) {
                    fill_plane<kHandOff>(coef + geom[firstrealc].coef_off, 0, (int64_t)geom[firstrealc].w * geom[firstrealc].h, 0, lane);
                    if (lane == 0) st_agent(&meta[firstrealc].decoded, 1);
                    publish(firstrealc, (uint32_t)rfl(geom[firstrealc].h) + 1u);
                    status |= ST_TRUNCATED;
                    continue;
                }
                status |= ST_CORRUPT;
                break;
            }
            predictability = rounded * 32;
        }

        rac_init(rac, s, lane);

        if (!compress) {
            // uncompressed group: encoding.cpp:334-354
            for (int i = beginc; i <= endc; i++) {
                const int gw = rfl(geom[i].w), gh = rfl(geom[i].h);
                const int minv = rfl(ld_agent(&meta[i].minval)), maxv = rfl(ld_agent(&meta[i].maxval));
                if (minv == maxv) continue;
                coef_t *plane = coef + geom[i].coef_off;
                const int zero = minv > 0 ? minv : (maxv < 0 ? maxv : 0);
                int y = 0;
                for (; y < gh; y++) {
                    if (s_limit_hit(s)) break;
                    for (int x0 = 0; x0 < gw; x0 += 64) {
                        const int nx = min(64, gw - x0);
                        int rowv = 0;
                        for (int j = 0; j < nx; j++) rowv = wrlane(uniform_read(rac, s, lane, minv, maxv - minv), j, rowv);
                        if (lane < nx) st_plane<kHandOff>(plane + (int64_t)y * gw + x0 + lane, rowv);
                    }
                    publish(i, (uint32_t)y + 2u);
                }
                // rows the stream never reached: Channel::resize() (encoding.cpp:338) only fills planes it creates
                if (y < gh) { fill_plane<kHandOff>(plane, (int64_t)y * gw, (int64_t)(gh - y) * gw, rfl(geom[i].ctor_data) ? 0 : zero, lane); status |= ST_TRUNCATED; }
                if (lane == 0) st_agent(&meta[i].decoded, 1);
                publish(i, (uint32_t)gh + 1u);
                if (s_limit_hit(s)) break;
            }
            __syncthreads();
            ci = endc;
            continue;
        }

        // ---- MANIAC tree: compound.h:277-320 with an explicit stack -------------------------
        __syncthreads();  // sh.lo/hi, sh.refs
        // narrow supernodes need every property (value and split) inside 13 signed bits and one property per lane of HALF a wavefront
        const bool narrow_ok = nprops <= 32 && __ballot(lane < nprops && (sh.lo[lane] < -kNarrowSplitMax || sh.hi[lane] > kNarrowSplitMax)) == 0ull;
        for (int k = lane; k < 3 * 32; k += 64) sh.meta_ctx[k / 32][k % 32] = 0;
        __syncthreads();
        if (lane == 0) for (int k = 0; k < 3; k++) symbol_chance_init(sh.meta_ctx[k], 1024);
        __syncthreads();
        tree_size = 1;
        int leaf_count = 0;
        bool tree_ok = true;
        {
            int pos = 0, depth = 0;
            while (true) {
                int p = ctx_symbol2(rac, s, lane, sh.meta_ctx[0], tree_table, 0, nprops) - 1;
                if (p != -1) {
                    const int oldmin = rfl(sh.lo[p]), oldmax = rfl(sh.hi[p]);
                    if (oldmin >= oldmax) { tree_ok = false; break; }
                    const int splitval = ctx_symbol2(rac, s, lane, sh.meta_ctx[2], tree_table, oldmin, oldmax - 1);
                    const int child = tree_size;
                    if (tree_size + 2 > P.max_nodes || depth >= kTreeStackDepth) { tree_ok = false; status |= ST_UNSUPPORTED; break; }
                    if (lane == 0) {
                        Node n; n.property = (int16_t)p; n.child = (uint16_t)child; n.splitval = splitval;
                        nodes[pos] = n;
                        Frame f; f.p = p; f.oldmin = oldmin; f.oldmax = oldmax; f.splitval = splitval; f.child = child; f.stage = 0;
                        stack[depth] = f;
                        sh.lo[p] = splitval + 1;
                    }
                    tree_size += 2;
                    depth++;
                    pos = child;
                    // sign of life for the stall verdicts while a big tree is parsed (up to 65535 nodes: tens of milliseconds without a
                    // finished row) -- the row loops bump the same word through publish()
                    if (kHandOff && lane == 0 && (tree_size & 1023) == 1) atomicAdd(P.heartbeat, 1u);
                    __syncthreads();
                    continue;
                }
                // leaf ids only have to be a bijection (each leaf owns its chances, compound.h:213-225): parse order
                if (lane == 0) { Node n; n.property = -1; n.child = (uint16_t)leaf_count; n.splitval = 0; nodes[pos] = n; }
                leaf_count++;
                // return to the nearest ancestor that still has its "<= splitval" branch to read
                bool done = false;
                while (true) {
                    if (depth == 0) { done = true; break; }
                    const int fstage = rfl(stack[depth - 1].stage), fp = rfl(stack[depth - 1].p);
                    if (fstage == 0) {
                        if (lane == 0) { sh.lo[fp] = stack[depth - 1].oldmin; sh.hi[fp] = stack[depth - 1].splitval; stack[depth - 1].stage = 1; }
                        pos = rfl(stack[depth - 1].child) + 1;
                        break;
                    }
                    if (lane == 0) sh.hi[fp] = stack[depth - 1].oldmax;
                    depth--;
                }
                __syncthreads();
                if (done) break;
            }
        }
        __syncthreads();
        if (!tree_ok) {
            // corrupt_or_truncated(io, image.channel[beginc], ...): encoding.cpp:358
            if (s_limit_hit(s)) {
                fill_plane<kHandOff>(coef + geom[beginc].coef_off, 0, (int64_t)geom[beginc].w * geom[beginc].h, 0, lane);
                if (lane == 0) st_agent(&meta[beginc].decoded, 1);
                publish(beginc, (uint32_t)rfl(geom[beginc].h) + 1u);
                status |= ST_TRUNCATED;
                continue;
            }
            status |= ST_CORRUPT;
            break;
        }

        // compact leaves: every channel of the group codes differences of at most 8 magnitude bits (symbol.h:160-183: the exponent stops at
        // ilog2(amax), amax <= maxv - zero or zero - minv with predictor 0, <= maxv - minv with any other)
        {
            bool small = true;
            for (int i = beginc; i <= endc; i++) {
                const int minv = rfl(ld_agent(&meta[i].minval)), maxv = rfl(ld_agent(&meta[i].maxval));
                if (minv == maxv) continue;
                const int zero = minv > 0 ? minv : (maxv < 0 ? maxv : 0);
                const int bound = predictor == 0 ? (maxv - zero > zero - minv ? maxv - zero : zero - minv) : maxv - minv;
                small = small && bound <= 255;
            }
            compact = small;
        }
        const uint32_t leaf_bytes = compact ? 32u : 64u;
        // ---- supernode layout ------------------------------------------------------------------
        // The tree is cut into complete 6-level subtrees ("supernodes", 63 node slots in heap order
        // + 64 exits).  Lane i of a supernode holds {splitval_i, property_i | exit_i << 8}; a walk
        // evaluates all 63 nodes of a supernode at once (see find_leaf below).  Slots below an early
        // leaf are "absent" (splitval INT_MAX = always the <= branch) and every exit under that
        // leaf points to it.  Supernodes are numbered breadth first, so the ones nearest the root
        // are the ones that stay in LDS.
        const int nleaves = (tree_size + 1) / 2;
        n_super = 1;
        // A group that may be suspended: a tile of an image ALL of whose tiles are single one-channel groups (Tile::flags),
        // with references to wait for.  Its supernodes and leaves are built in a context area of its image's queue.
        // (Without a free area the context stays in the wavefront's scratch area and the tile is PINNED to this wavefront: it
        // is suspended and resumed like the others, but by this wavefront only -- see pinned_tix.)
        int max_super_here = P.max_super;
        // ONE arena, two bump pointers in one 64-bit word: the tiles that hold >= 1/16 of their picture -- the long per-symbol chains that bound
        // the launch, whose contexts are what the memory system has to keep close -- are packed from the bottom, everything else from the top.
        // (Rounds 2-5 gave every CU queue its own 64 MiB arena: the hot contexts of a launch lay 512 pages apart instead of ~100; tight
        // placement is worth ~6 % of a long tile's per-symbol chain, profiles/r6_ubench_context_layouts.txt.)  Both counters only grow, and an
        // area is granted from ONE atomic snapshot of both, so areas never overlap; a failed request leaves its units unused (the arena
        // was full: the tile is pinned to this wavefront's scratch area).
        auto grant_area = [&](uint32_t units) -> uint32_t {
            uint32_t off = 0xFFFFFFFFu;
            const bool low_end = ((rflu(tile.flags) >> kTileSizeClassShift) & 15u) <= kLongClass + 1u;
            if (lane == 0) {
                const uint32_t total = P.ctx_units_per_queue * (uint32_t)n_queues;
                const unsigned long long seen = __hip_atomic_load(P.ctx_used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned long long)(uint32_t)seen + (seen >> 32) + units <= total) {
                    const unsigned long long o = atomicAdd(P.ctx_used, low_end ? (unsigned long long)units : (unsigned long long)units << 32);
                    const unsigned long long lo = (uint32_t)o + (low_end ? units : 0u), hi = (o >> 32) + (low_end ? 0u : units);
                    if (lo + hi <= total) off = low_end ? (uint32_t)o : total - (uint32_t)hi;
                }
            }
            return rflu(off);
        };
        // (7n+5)/12 supernodes are enough for n inner nodes (capi.hip), so a build with room for that many never falls to the node-by-node walk
        const uint32_t sn_cap = (7u * ((uint32_t)(tree_size - 1) / 2u) + 5u) / 12u + 1u;
        const bool suspendable_here = sched && kHandOff && (rflu(tile.flags) & kTileSuspendable) && beginc == endc && endc == last_c && nrefs > 0;
        // Round 6: a suspendable tile whose worst case fits the wavefront's scratch area builds its supernodes THERE and moves them, once their real number
        // is known, into a context area of exactly that size (rounds 2-5 reserved the worst case: 526 supernode slots for the ~150 a long 4K group has, with
        // the leaves behind the unused ones -- 2.5x the footprint, and C4's 743 KB per tile ran the arenas dry).  Only trees too large for the scratch
        // area (more than ~7000 inner nodes) still reserve their worst case up front.
        bool exact_area = false;
        narrow = narrow_ok && tree_size <= kNarrowMaxNodes && (int)sn_cap <= P.max_super;
        if (suspendable_here) {
            can_yield = true;
            if ((int)sn_cap <= P.max_super) exact_area = true;
            else {
                const uint32_t sn_units = sn_cap * 2u;   // (such a tree is never narrow)
                const uint32_t off = grant_area(sn_units + ((uint32_t)nleaves * leaf_bytes + 255u) / 256u);
                if (off != 0xFFFFFFFFu) {
                    ctx_slot = (int)off;
                    uint8_t *cb = P.ctx_scratch + (size_t)off * 256u;
                    snodes_g = reinterpret_cast<uint2 *>(cb);
                    leaves = reinterpret_cast<uint16_t *>(cb + (size_t)sn_units * 256u);
                    ctx_leaves_units = sn_units;
                    max_super_here = (int)sn_cap;
                } else {
                    // no area left: the context stays in this wavefront's scratch area and the tile is pinned to the wavefront
                    STATS(st_noctx++;)
                    pinned_tix = tix;
                }
            }
        }
        // Subtree sizes (nodes, saturating): children always have larger indices than their parent in the parse-order
        // array, so one backward sweep does it.  The learner splits contexts that see many samples, so a child
        // supernode with a big subtree is (statistically) a frequently walked one: numbering them big-first puts the
        // busiest part of the tree into the LDS-resident prefix instead of whatever hangs under the first exits.
        if (lane == 0) {
            for (int i = tree_size - 1; i >= 0; i--) {
                const Node n = nodes[i];
                uint32_t sz = 1;
                if (n.property >= 0) sz += (uint32_t)subtree[n.child] + (uint32_t)subtree[n.child + 1];
                subtree[i] = (uint16_t)(sz > 65535u ? 65535u : sz);
            }
        }
        __syncthreads();
        {
            int32_t *slot_node = sh.cprops;        // [127] tree node behind every heap slot (cprops is idle here)
            int32_t *st_split = sh.cprops + 128;   // [64]
            int32_t *st_prop = sh.cprops + 192;    // [64]
            if (lane == 0) queue[0] = 0;
            __syncthreads();
            for (int sn = 0; sn < n_super; sn++) {
                if (kHandOff && lane == 0 && (sn & 255) == 255) atomicAdd(P.heartbeat, 1u);   // (see the tree parse)
                if (lane == 0) slot_node[0] = queue[sn];
                st_split[lane] = 0x7FFFFFFF;
                st_prop[lane] = 0;
                __syncthreads();
                for (int d = 0; d < 6; d++) {
                    if (lane < (1 << d)) {
                        const int k = (1 << d) - 1 + lane;
                        const int t = slot_node[k];
                        const Node n = nodes[t];
                        if (n.property >= 0) {
                            st_split[k] = n.splitval; st_prop[k] = n.property;
                            slot_node[2 * k + 1] = n.child; slot_node[2 * k + 2] = n.child + 1;
                        } else {
                            slot_node[2 * k + 1] = t; slot_node[2 * k + 2] = t;
                        }
                    }
                    __syncthreads();
                }
                const int t = slot_node[exit_q];
                const Node n = nodes[t];
                const bool inner = n.property >= 0;
                const unsigned long long im = __ballot(inner);
                int rank = __popcll(im & ((1ull << lane) - 1ull));
                if (sn < kSizeOrdered) {
                    // children of the top supernodes are numbered by subtree size (descending; ties by exit)
                    const int mine = inner ? (int)subtree[t] : -1;
                    rank = 0;
                    for (int j = 0; j < 64; j++) {
                        const int other = rdlane(mine, j);
                        rank += (other > mine) | ((other == mine) & (j < lane));
                    }
                }
                uint32_t tgt;
                // The scratch area holds P.max_super supernodes.  Subtrees beyond that (only trees with tens of
                // thousands of nodes get there) are walked node by node from the parse-order array instead.
                const bool admit = inner && (n_super + rank < max_super_here);
                if (admit) { tgt = (uint32_t)(n_super + rank); queue[n_super + rank] = t; }
                else if (inner) tgt = kSlowFlag | (uint32_t)t;
                else tgt = kLeafFlag | (uint32_t)n.child;
                n_super += __popcll(__ballot(admit));
                uint2 out;
                out.x = (uint32_t)(is_raw_slog_prop(st_prop[lane] - nrefprops) && st_split[lane] != 0x7FFFFFFF ? slog_threshold(st_split[lane]) : st_split[lane]);
                out.y = (((uint32_t)st_prop[lane] << 2) & 0xFFu) | (tgt << 8);   // the property as a ds_bpermute byte address (lane * 4): no shift or mask per walk round
                if (narrow) {
                    // (every inner exit is admitted: the area holds (7n+5)/12 + 1 supernodes, which is what made the group narrow)
                    const uint32_t w = pack_narrow((int)out.x, (uint32_t)st_prop[lane], inner ? (tgt & 0x1FFFu) : (kLeafFlagN | (uint32_t)n.child));
                    reinterpret_cast<uint32_t *>(snodes_g)[(size_t)sn * 64 + lane] = w;
                    if (sn >= 1 && sn <= kLdsN) lds_narrow[(sn - 1) * 64 + lane] = w;
                } else {
                    snodes_g[(size_t)sn * 64 + lane] = out;
                    if (sn >= 1 && sn <= kLdsSuper) sh.snodes[(sn - 1) * 64 + lane] = out;   // the root (0) lives in registers: LDS holds 1..kLdsSuper
                }
                __syncthreads();
            }
        }
        if (exact_area) {
            const uint32_t sn_units = narrow ? (uint32_t)n_super : (uint32_t)n_super * 2u;   // a narrow supernode is one 256-byte unit
            const uint32_t off = grant_area(sn_units + ((uint32_t)nleaves * leaf_bytes + 255u) / 256u);
            if (off != 0xFFFFFFFFu) {
                ctx_slot = (int)off;
                uint8_t *cb = P.ctx_scratch + (size_t)off * 256u;
                uint32_t *dst = reinterpret_cast<uint32_t *>(cb);
                const uint32_t *src = reinterpret_cast<const uint32_t *>(snodes_w);
                for (uint32_t k = (uint32_t)lane; k < sn_units * 64u; k += 64u) dst[k] = src[k];   // (the supernodes just built, 256 bytes per unit)
                snodes_g = reinterpret_cast<uint2 *>(cb);
                leaves = reinterpret_cast<uint16_t *>(cb + (size_t)sn_units * 256u);
                ctx_leaves_units = sn_units;
                __syncthreads();
            } else {
                // no area left: the context stays in this wavefront's scratch area and the tile is pinned to the wavefront
                STATS(st_noctx++;)
                pinned_tix = tix;
            }
        }
        // A context that lives in this wavefront's scratch area (streams without group index, pinned tiles): the leaf chances start right behind the
        // supernodes the tree really has, not at the area's fixed leaf offset 2 MB further on -- the two halves of the per-symbol chain then share a page
        // (thousands of wavefronts x two pages each was more than the address translation caches hold; profiles/r6_ubench_context_layouts.txt, "scratch")
        if (ctx_slot < 0) leaves = reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(snodes_w) + (size_t)n_super * (narrow ? 256u : 512u));
        // FinalPropertySymbolCoder ctor: every leaf starts from SymbolChance(zero_chance) (compound.h:213-219)
        {
            if (lane == 0) {
                if (compact) {   // zero, sign, exponent 0..6, mantissa 0..6 (symbol_chance_init's values for these slots)
                    uint16_t full[32];
                    symbol_chance_init(full, predictability);
                    for (int k = 0; k < kMantCompact; k++) leaves[k] = full[k];
                    for (int k = 0; k < 16 - kMantCompact; k++) leaves[kMantCompact + k] = full[CH_MANT + k];
                } else { symbol_chance_init(leaves, predictability); leaves[31] = 0; }
            }
            __syncthreads();
            const uint32_t *l0 = reinterpret_cast<const uint32_t *>(leaves);
            uint32_t *lw = reinterpret_cast<uint32_t *>(leaves);
            const int words = compact ? 8 : 16;   // per leaf
            const uint32_t mine = l0[lane & (words - 1)];
            for (int64_t i = words + lane; i < (int64_t)nleaves * words; i += 64) lw[i] = mine;  // (i & (words - 1)) == (lane & (words - 1))
            __syncthreads();
        }
        }  // !resumed
        const uint2 root_nd = narrow ? uint2{0u, 0u} : snodes_g[lane];  // the root supernode lives in registers
        const uint32_t root_w = narrow ? reinterpret_cast<const uint32_t *>(snodes_g)[lane] : 0u;
        LeafRegs L;
        const uint32_t leaf_shift = compact ? 5u : 6u, leaf_l2 = (uint32_t)(lane & (compact ? 15 : 31)) * 2u;   // a leaf's bytes; this lane's chance inside it
        L.leafv = (int)*reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(leaves) + (((uint32_t)cur_leaf << leaf_shift) + leaf_l2));   // lanes 32..63 mirror lanes 0..31 (switch_leaf)
        // the shift as a per-lane value the compiler cannot see through: (id << shift) + offset is then ONE vector instruction (v_lshl_add_u32 with the shift
        // in a VGPR) instead of a scalar shift + a vector add per address -- the scalar pipe is the loaded one (two addresses per symbol)
        uint32_t leaf_shift_v = leaf_shift;
#ifndef FUIF_EMU
        asm volatile("v_mov_b32 %0, %1" : "=v"(leaf_shift_v) : "s"(leaf_shift));
#endif
        L.touched = 0; L.bits = 0;
        L.mb = rfl(compact ? kMantCompact : CH_MANT); L.mirror = rflu(compact ? 0x10001u : 1u);
        auto switch_leaf = [&](int id) {
            // Lanes 32..63 mirror lanes 0..31 (same addresses, same values): the write-back and the fetch need no lane mask, and the
            // compiler tracks both (32-bit offsets from the leaves' base: a tree has at most 32768 leaves of 64 bytes)
            if (LIKELY(id != cur_leaf)) {
                char *lb = reinterpret_cast<char *>(leaves);
                const uint32_t l2 = leaf_l2;
                // the fetch is issued FIRST: it does not depend on the chances in leafv, the write-back does (the commit's table lookup may
                // still be landing in them) -- where the walk is short (wide configuration: every round from LDS) that wait would
                // otherwise sit in front of the leaf's memory round trip
                const int fresh = (int)*reinterpret_cast<const uint16_t *>(lb + (((uint32_t)id << leaf_shift_v) + l2));
                *reinterpret_cast<uint16_t *>(lb + (((uint32_t)cur_leaf << leaf_shift_v) + l2)) = (uint16_t)L.leafv;
                L.leafv = fresh;
                cur_leaf = id;
            }
        };

        // ---- pixel loops: encoding.cpp:365-425 ----------------------------------------------
        for (int i = beginc; i <= endc; i++) {
            const int w = rfl(geom[i].w), h = rfl(geom[i].h);
            const int ghs = rfl(geom[i].hshift), gvs = rfl(geom[i].vshift);
            const int minv = rfl(ld_agent(&meta[i].minval)), maxv = rfl(ld_agent(&meta[i].maxval));
            if (minv == maxv) continue;
            coef_t *plane = coef + geom[i].coef_off;
            const int zero = minv > 0 ? minv : (maxv < 0 ? maxv : 0);
            int y = resumed ? (int)resume_y : 0;
            uint32_t ref_seen = 0;  // lane k: last progress word seen for reference channel k
            if (tree_size == 1 && predictor == 0 && zero == 0) {
                // fast track: encoding.cpp:371-383 (single leaf, no properties)
                for (; y < h; y++) {
                    if (s_limit_hit(s)) break;
                    for (int x0 = 0; x0 < w; x0 += 64) {
                        const int nx = min(64, w - x0);
                        int rowv = 0;
                        for (int j = 0; j < nx; j++) {
                            rowv = wrlane(leaf_symbol(rac, s, lane, L, minv, maxv), j, rowv);
                            leaf_commit(L, lane, pixel_table);
                        }
                        if (lane < nx) st_plane<kHandOff>(plane + (int64_t)y * w + x0 + lane, rowv);
                    }
                    publish(i, (uint32_t)y + 2u);
                }
            } else {
                // PRED0 = predictor 0 (all Squeeze residual / DCT coefficient channels): guess is a constant
                FastSym fsym;
                fsym.amax_pos = maxv - zero; fsym.amax_neg = zero - minv;
                fsym.emax_pos = ilog2u((uint32_t)fsym.amax_pos); fsym.emax_neg = ilog2u((uint32_t)fsym.amax_neg);
                fsym.ilast_pos = rfl(fsym.emax_pos + 1); fsym.ilast_neg = rfl(fsym.emax_neg + 1);
                const bool sym_fast = minv < zero && zero < maxv;   // both signs possible: symbol.h:160-165 codes zero and sign
                auto rows = [&](auto pred0_tag, auto narrow_tag) {
                    constexpr bool PRED0 = decltype(pred0_tag)::value;
                    constexpr bool NARROW = decltype(narrow_tag)::value;   // 4-byte supernode lane words (kLeafFlagN)
                    for (; y < h; y++) {
                        if (s_limit_hit(s)) break;
                        __syncthreads();  // the previous row's stores are complete before it is re-read as `top`
                        if (kHandOff && nrefs) {
                            // the rows of the reference channels this row looks at must be final; they may be the
                            // work of other tiles that are still running (lane k watches reference k)
                            uint32_t need = 0;
                            const uint32_t *fp = progress;
                            if (lane < nrefs) {
                                const RefChan rc = sh.refs[lane];
                                int ry = (y << gvs) >> rc.vshift; if (ry >= rc.h) ry = rc.h - 1;
                                need = (uint32_t)ry + 2u;
                                fp = progress + rc.chan;
                            }
                            uint32_t spins = 0;
                            if (__any(ref_seen < need)) {
                                if (ref_seen < need) ref_seen = ld_agent(fp);
                                if (__any(ref_seen < need) && can_yield) {
                                    // suspend: the scheduler resumes this tile when the first missing reference is yield_slack rows ahead
                                    const int bl = __builtin_ctzll(__ballot(ref_seen < need));
                                    const RefChan rcb = sh.refs[bl];
                                    const uint32_t nb = (uint32_t)rdlane((int)need, bl) + P.yield_slack, fin = (uint32_t)rfl(rcb.h) + 1u;
                                    yield_chan = (uint32_t)rfl(rcb.chan); yield_val = nb < fin ? nb : fin;
                                    resume_y = (uint32_t)y;
                                    EMU_COUNT(3);
                                    yielded = true;
                                    break;
                                }
                                if (__any(ref_seen < need)) {
                                    STATS(const unsigned long long w0 = realtime();)
                                    while (__any(ref_seen < need) && !stalled) {
                                        spin_nap(spins);
                                        if (ref_seen < need) ref_seen = ld_agent(fp);
                                    }
                                    STATS(waited += realtime() - w0;)
                                }
                            }
                        }
                        const coef_t *row1 = plane + (int64_t)(y - 1) * w;
                        const coef_t *row2 = plane + (int64_t)(y - 2) * w;
                        // left / leftleft start as `zero`, which is exactly what the edge rules
                        // give at x == 0 (context_predict.h:126,131)
                        int left = zero, leftleft = zero;
                        // The 7 left-dependent local properties (context_predict.h:136-154) are all of the form
                        // F(c*left + bias): the vector phase stores the bias, the scalar phase finishes them with
                        // ~16 vector instructions for all lanes at once.  k = local property index of this lane.
                        //   k:  1 |left|   3 slog(left)   6 left+top-topleft   7 topleft+topright-top
                        //       8 slog(left-topleft)   9 slog(topleft-top)   12 slog(left-leftleft)
                        // In row 0 topleft IS left (context_predict.h:128), which moves the left term from 6,8 to 7,9.
                        const int kloc = (lane & prop_mask) - nrefprops;   // (up to 32 properties: lanes 32..63 mirror lanes 0..31, which narrow supernodes rely on)
                        // (3, 8, 9, 12 stay raw differences: the supernodes hold slog_threshold(split) for them)
                        const bool f_abs = (kloc == 1);
                        // per-lane masks: d = bias + (left & m_left) - (leftleft & m_ll) -- two ANDs and one add3 instead of two 24-bit multiplies
                        const int m_left = ((kloc == 1) | (kloc == 3) | (kloc == 12) | (y ? ((kloc == 6) | (kloc == 8)) : ((kloc == 7) | (kloc == 9)))) ? -1 : 0;
                        const int m_ll = (kloc == 12) ? -1 : 0;
                        for (int x0 = 0; x0 < w; x0 += chunk_px) {
                            const int nx = min(chunk_px, w - x0);
                            // ---- vector phase: lane j prepares pixel x0+j ------------------------
                            PROF_START();
                            const int x = min(x0 + lane, w - 1);
                            const int vtop = y ? row1[x] : zero;
                            const int vtl = (y && x) ? row1[x - 1] : zero;                 // x == 0: topleft = left = zero
                            const int vtr = (x + 1 < w && y) ? row1[x + 1] : vtop;         // context_predict.h:129
                            const int vtt = y > 1 ? row2[x] : vtop;                        // :133
                            if (lane < chunk_px) {
                                int32_t *cp = sh.cprops + lane * prop_pitch;
                                // (more than kFastRefs references, -E > 18: the rest one by one -- rare, and its chunks are half as long)
                                for (int k = kFastRefs; k < nrefs; k++) {
                                    const RefChan rc = sh.refs[k];
                                    int ry = (y << gvs) >> rc.vshift; if (ry >= rc.h) ry = rc.h - 1;
                                    int rx = ghs < 0 ? rc.w - 1 : (x << ghs) >> rc.hshift; if (rx >= rc.w) rx = rc.w - 1;
                                    const int v = ld_plane<kHandOff>(coef + rc.off + (int64_t)ry * rc.w + rx);
                                    cp[2 * k] = iabs(v); cp[2 * k + 1] = slog(v);
                                }
                                // The reference samples of this pixel: every load is a memory round trip (another tile's plane, read past the L1), so the
                                // loads of up to 6 references -- what the default options give -- are ISSUED TOGETHER and consumed afterwards.  (Rounds 1-5
                                // had one `if (k < nrefs)` block per reference: a uniform branch between two loads, which hipcc does not hoist a load
                                // across -- six round trips one after the other, 7200 cycles per 32-pixel chunk on the long 4K groups,
                                // profiles/r6_phases_by_channel_narrow.txt.)  A slot beyond nrefs reads reference 0 again and is dropped.
                                auto ref_sample = [&](int k) -> int {
                                    // rx = min((x<<hshift)>>ref.hshift, ref.w-1) covers the three cases of context_predict.h:241-284
                                    const RefChan rc = sh.refs[k < nrefs ? k : 0];
                                    int ry = (y << gvs) >> rc.vshift; if (ry >= rc.h) ry = rc.h - 1;
                                    // a meta-channel (hshift -1) coded AFTER ordinary channels (Approximate on a palette, approximate.h:76)
                                    // takes the `ch.hshift < rc.hshift` branch with stepsize (1<<rc.hshift) >> -1, which the reference's
                                    // x86 build evaluates as 0: every x then reads the LAST sample of the reference row (:253-262)
                                    int rx = ghs < 0 ? rc.w - 1 : (x << ghs) >> rc.hshift; if (rx >= rc.w) rx = rc.w - 1;
                                    return ld_plane<kHandOff>(coef + rc.off + (int64_t)ry * rc.w + rx);
                                };
                                static_assert(kFastRefs == 9, "the reference loads are issued in batches of 6 + 3");
                                if (nrefs > 0) {
                                    const int r0 = ref_sample(0), r1 = ref_sample(1), r2 = ref_sample(2), r3 = ref_sample(3), r4 = ref_sample(4), r5 = ref_sample(5);
                                    cp[0] = iabs(r0); cp[1] = slog(r0);
                                    if (nrefs > 1) { cp[2] = iabs(r1); cp[3] = slog(r1); }
                                    if (nrefs > 2) { cp[4] = iabs(r2); cp[5] = slog(r2); }
                                    if (nrefs > 3) { cp[6] = iabs(r3); cp[7] = slog(r3); }
                                    if (nrefs > 4) { cp[8] = iabs(r4); cp[9] = slog(r4); }
                                    if (nrefs > 5) { cp[10] = iabs(r5); cp[11] = slog(r5); }
                                    if (nrefs > 6) {
                                        const int r6 = ref_sample(6), r7 = ref_sample(7), r8 = ref_sample(8);
                                        cp[12] = iabs(r6); cp[13] = slog(r6);
                                        if (nrefs > 7) { cp[14] = iabs(r7); cp[15] = slog(r7); }
                                        if (nrefs > 8) { cp[16] = iabs(r8); cp[17] = slog(r8); }
                                    }
                                }
                                int32_t *q = cp + nrefprops;
                                q[0] = iabs(vtop); q[2] = slog(vtop); q[4] = y; q[5] = (x0 + lane);
                                q[10] = slog(vtop - vtr); q[11] = slog(vtop - vtt);
                                q[1] = 0; q[3] = 0; q[12] = 0;
                                if (y) { q[6] = (vtop - vtl); q[7] = (vtl + vtr - vtop); q[8] = -vtl; q[9] = (vtl - vtop); }
                                else { q[6] = zero; q[7] = 0; q[8] = 0; q[9] = -zero; }
                            }
                            __syncthreads();
                            PROF_LAP(0);
                            int rowv = 0;
                            // ---- scalar phase: one pixel at a time ---------------------------------
                            // kChunkFast: every symbol of this chunk has its (at most 62) bytes inside the stream -- known once per
                            // chunk, so the hand-written decoder needs no end-of-stream test per pixel (the loop is compiled twice)
                            auto pixels = [&](auto chunk_fast_tag) {
                            constexpr bool kChunkFast = decltype(chunk_fast_tag)::value;
                            // leftleft = value at x-2, except at x == 1 where the rule is leftleft = left (context_predict.h:131): the first
                            // pixel of a row is a loop part of its own, so that the rule costs nothing per pixel
                            const int j_split = x0 == 0 ? 1 : 0;
                            const int32_t *prow = &sh.cprops[lane & prop_mask];
#ifndef FUIF_EMU
                            uint32_t widx = s.pos - s.win_base;   // (used by the kChunkFast instance only)
#endif
                            for (int part = 0; part < 2; part++) {
                            const int j_end = part ? nx : j_split;
                            for (int j = part ? j_split : 0; j < j_end; j++) {
                                PROF_START();
                                int pv = *prow; prow += prop_pitch;   // cprops[j * prop_pitch + (lane & prop_mask)]
                                const int l = left;
                                {
                                    const int d = pv + (l & m_left) + (-leftleft & m_ll);
                                    pv = f_abs ? iabs(d) : d;
                                }
                                int guess = zero;
                                if (!PRED0) {
                                    const int top = rdlane(vtop, j);
                                    const int topleft = y ? rdlane(vtl, j) : l;   // y == 0: topleft = left (context_predict.h:128)
                                    const int topright = rdlane(vtr, j);
                                    switch (predictor) {  // context_predict.h:157-166
                                        case 0: guess = zero; break;
                                        case 1: guess = (l + top) / 2; break;
                                        case 2: guess = median3(l + top - topleft, l, top); break;
                                        case 3: guess = l; break;
                                        case 4: guess = top; break;
                                        case 5: guess = (l + topleft + top + topright) / 4; break;
                                        case 6: { int t = l + top - topleft; guess = t < minv ? minv : (t > maxv ? maxv : t); break; }
                                        default: guess = median3(l + top - topleft, l, top); break;
                                    }
                                }
                                const int mn = minv - guess, mx = maxv - guess;
                                int diff = mn;  // compound.h:228: min == max needs no symbol
                                PROF_LAP(1);
                                if (LIKELY(mn != mx)) {
                                    // find_leaf: compound.h:142-153, six tree levels per step.  Lane i fetches the
                                    // property its node tests (ds_bpermute from lane `prop` of pv) and compares;
                                    // the 63 outcomes form a mask; lane e checks whether the mask agrees with the
                                    // six decisions on the path to exit e -- exactly one exit matches.
                                    // Control flow is kept to one `while` with a single condition and if-without-else
                                    // bodies: hipcc's structuriser turns every if/else into 2-3 branches (~25 cycles each).
                                    auto walk_round = [&](const uint2 nd) -> uint32_t {
                                        const int val = __builtin_amdgcn_ds_bpermute((int)nd.y, pv);   // the source lane is address bits 7..2: the exit bits above are ignored
                                        const unsigned long long m = __ballot(val > (int)nd.x);
                                        const uint32_t mlo = (uint32_t)m, mhi = (uint32_t)(m >> 32);
                                        const bool hit = ((((mlo ^ exp_lo) & msk_lo) | ((mhi ^ exp_hi) & msk_hi)) == 0u);
                                        const int e = __builtin_ctzll(__ballot(hit));
                                        return (uint32_t)rdlane((int)nd.y, e) >> 8;
                                    };
                                    uint32_t tgt;
                                    if (NARROW) {
                                        // the same round on 4-byte lane words: the word is the ds_bpermute address, the split a 13-bit field, the exit 14 bits
                                        // that a rotation makes contiguous
                                        auto walk_round_n = [&](const uint32_t w) -> uint32_t {
                                            const int val = __builtin_amdgcn_ds_bpermute((int)w, pv);   // source lane = bits 7..2 = property | split bit 0 << 5: lanes 32..63 of pv mirror 0..31
                                            const int split = ((int)(w << 12)) >> 19;                  // bits 19..7, sign extended (v_bfe_i32)
                                            const unsigned long long m = __ballot(val > split);
                                            const uint32_t mlo = (uint32_t)m, mhi = (uint32_t)(m >> 32);
                                            const bool hit = ((((mlo ^ exp_lo) & msk_lo) | ((mhi ^ exp_hi) & msk_hi)) == 0u);
                                            const int e = __builtin_ctzll(__ballot(hit));
                                            const uint32_t ex = ((w >> 20) | (w << 12)) & 0x3FFFu;    // exit: v_alignbit_b32 + v_and (masked on the vector side: the scalar pipe is the loaded one)
                                            return (uint32_t)rdlane((int)ex, e);
                                        };
                                        tgt = walk_round_n(root_w);
                                        EMU_COUNT(0);
                                        while (!(tgt & kLeafFlagN)) {
                                            EMU_COUNT(tgt <= (uint32_t)kLdsN ? 1 : 2);
                                            uint32_t w;
                                            if (kLdsN > 0) {
                                                const uint32_t li = tgt <= (uint32_t)kLdsN ? tgt : (uint32_t)kLdsN;
                                                w = lds_load_node_n(lds_narrow_addr + (li - 1u) * 256u + (uint32_t)lane * 4u);
                                                if (tgt > (uint32_t)kLdsN) w = global_load_supernode_n(reinterpret_cast<const uint32_t *>(snodes_g), tgt, (uint32_t)lane * 4u);
                                            } else w = global_load_supernode_n(reinterpret_cast<const uint32_t *>(snodes_g), tgt, (uint32_t)lane * 4u);
                                            tgt = walk_round_n(w);
                                        }
                                        tgt &= kLeafFlagN - 1u;
                                    } else {
                                    tgt = walk_round(root_nd);
                                    EMU_COUNT(0);
                                    while (!(tgt & (kLeafFlag | kSlowFlag))) {
                                        EMU_COUNT(tgt <= (uint32_t)kLdsSuper ? 1 : 2);
                                        // LDS-resident supernodes are the common case; the load is issued unconditionally
                                        // (index clamped) and replaced in the rare deep case
                                        // (supernode i sits in LDS slot i-1: the -512 folds into the base address)
                                        uint2 nd;
                                        if (kLdsSuper > 0) {
                                            const uint32_t li = tgt <= (uint32_t)kLdsSuper ? tgt : (uint32_t)kLdsSuper;
                                            nd = lds_load_node(lds_nodes_addr + (li - 1u) * 512u + (uint32_t)lane * 8u);
                                            if (UNLIKELY(tgt > (uint32_t)kLdsSuper)) nd = global_load_node(&snodes_g[(size_t)tgt * 64 + lane]);
                                        } else nd = global_load_supernode(snodes_g, tgt, (uint32_t)lane * 8u);   // dense configuration: every supernode behind the root comes from memory
                                        tgt = walk_round(nd);
                                    }
                                    if (UNLIKELY(tgt & kSlowFlag)) {
                                        // beyond the supernode cap: compound.h:142-153 as written, one node per step
                                        // (lane p of pv holds property p).  Kept OUT of the loop above: a second exit
                                        // would make the structuriser add ~4 branches to every round.
                                        int t = (int)(tgt & 0xFFFFu);
                                        Node n = nodes[t];
                                        while (n.property >= 0) {
                                            const int np = rfl((int)n.property);
                                            const int sv = is_raw_slog_prop(np - nrefprops) ? slog_threshold(rfl(n.splitval)) : rfl(n.splitval);
                                            t = rdlane(pv, np) > sv ? (int)n.child : (int)n.child + 1;
                                            t = rfl(t);
                                            n = nodes[t];
                                        }
                                        tgt = kLeafFlag | (uint32_t)rfl((int)n.child);
                                    }
                                    }
                                    const int leaf = (int)(tgt & (kLeafFlag - 1u));
                                    PROF_LAP(2);
                                    switch_leaf(leaf);
#ifdef FUIF_PROF
                                    asm volatile("; the leaf's chances are here" :: "v"(L.leafv));  // force the leaf load to complete inside this lap (an input operand: hipcc waits for it)
#endif
                                    PROF_LAP(3);
                                    if (kChunkFast || (PRED0 && sym_fast && LIKELY(s.pos + 64u <= s.size))) {
                                        // the symbol's bytes (at most 62) are in the stream; keep them in the window registers
                                        // (the reload starts at a 4-byte boundary; the 256 bytes it reads lie inside the allocation)
#ifdef FUIF_EMU
                                        if (UNLIKELY(s.pos - s.win_base > 188u)) {
                                            s.win_base = s.pos & ~3u;
                                            s.win = reinterpret_cast<const uint32_t *>(s.p + s.win_base)[lane];
                                        }
                                        diff = fast_symbol(rac, s, L, fsym);
#else
                                        if constexpr (kChunkFast) {   // the position lives in widx for the whole chunk
                                            if (UNLIKELY(widx > 188u)) {
                                                s.pos = s.win_base + widx;
                                                s.win_base = s.pos & ~3u;
                                                widx = s.pos - s.win_base;
                                                s.win = reinterpret_cast<const uint32_t *>(s.p + s.win_base)[lane];
                                            }
                                            diff = fast_symbol_hw_w(rac, widx, s.win, L, fsym);
                                        } else {
                                            if (UNLIKELY(s.pos - s.win_base > 188u)) {
                                                s.win_base = s.pos & ~3u;
                                                s.win = reinterpret_cast<const uint32_t *>(s.p + s.win_base)[lane];
                                            }
                                            diff = fast_symbol_hw(rac, s, L, fsym);
                                        }
#endif
                                    } else diff = leaf_symbol(rac, s, lane, L, mn, mx);
                                    PROF_LAP(4);
                                    // advance the touched chances now: the table lookup overlaps the next pixel's tree walk
                                    leaf_commit(L, lane, pixel_table);
                                }
                                const int v = diff + guess;
                                rowv = wrlane(v, j, rowv);
                                leftleft = l;
                                left = v;
                                PROF_LAP(5);
                            }
                            if (part == 0 && j_split) leftleft = left;   // after x == 0
                            }
#ifndef FUIF_EMU
                            if constexpr (kChunkFast) s.pos = s.win_base + widx;
#endif
                            };
                            if (PRED0 && sym_fast && s.pos + 64u * (uint32_t)(nx + 1) <= s.size) pixels(std::true_type{}); else pixels(std::false_type{});
                            PROF_START();
                            if (lane < nx) st_plane<kHandOff>(plane + (int64_t)y * w + x0 + lane, rowv);
                            __syncthreads();  // cprops is rewritten by the next chunk
                            PROF_LAP(5);   // (the row store is counted with the rest of the pixel loop: slot 6 holds the segment's wall-clock ticks)
                        }
                        publish(i, (uint32_t)y + 2u);
                    }
                };
                if (predictor == 0) { if (narrow) rows(std::true_type{}, std::true_type{}); else rows(std::true_type{}, std::false_type{}); }
                else { if (narrow) rows(std::false_type{}, std::true_type{}); else rows(std::false_type{}, std::false_type{}); }
            }
            if (yielded) break;
            // rows the stream never reached keep what Channel::resize() (encoding.cpp:368) left there: `zero` in a plane
            // it created, the constructor's 0 in a plane that already had its samples (image.h:64-65,73-75)
            if (y < h) { __syncthreads(); fill_plane<kHandOff>(plane, (int64_t)y * w, (int64_t)(h - y) * w, rfl(geom[i].ctor_data) ? 0 : zero, lane); status |= ST_TRUNCATED; }
            if (lane == 0) st_agent(&meta[i].decoded, 1);
            publish(i, (uint32_t)h + 1u);
            if (s_limit_hit(s)) break;
        }
        if (yielded) {
            // suspend: current leaf back to its slot, coder state into the tile's record, then release the record
            if (lane < (compact ? 16 : 32)) leaves[((int64_t)cur_leaf << (leaf_shift - 1u)) + lane] = (uint16_t)L.leafv;
            if (lane == 0) {
                rec->wait_chan = yield_chan; rec->wait_val = yield_val; rec->y = resume_y;
                rec->range = rac.range; rec->low = rac.low; rec->pos = s.pos;
                rec->flags = ((uint32_t)status & 0xFFu) | ((uint32_t)(s.eof_flag & 1) << 8) | ((uint32_t)predictor << 9) | (long_tile ? kRecLong : 0u) | (narrow ? kRecNarrow : 0u) | (compact ? kRecCompact : 0u);
                if (long_tile) atomicAdd(&P.simd_long[simd_slot], 0xFFFFFFFFu);
                rec->ctx = (uint32_t)ctx_slot; rec->ctx_leaves = ctx_leaves_units; rec->tree_size = (uint32_t)tree_size; rec->n_super = (uint32_t)n_super; rec->cur_leaf = (uint32_t)cur_leaf;
                rec->pin = ctx_slot >= 0 ? 0u : my_pin;
                rec->owner = simd_key + 1u;
            }
            drain_stores();   // the record and the leaf are in this XCD's L2 before the state says so
            if (lane == 0) st_agent(&rec->state, (uint32_t)TS_READY);
            break;
        }
        __syncthreads();
        ci = endc;
    }
#ifdef FUIF_PROF
    if (yielded) { prof_acc[7] += __builtin_readcyclecounter() - prof_seg0; prof_acc[6] += realtime() - prof_rt0; }
    if (yielded && lane == 0 && P.prof) for (int k = 0; k < 8; k++) atomicAdd(&P.prof[(size_t)PROF_ROW * 8 + k], prof_acc[k]);   // (a suspended tile's laps count too)
#endif
    TLOG(if (yielded && lane == 0 && P.tile_log) {
        unsigned long long *tl = P.tile_log + (size_t)tix * 4;
        if (!resumed) tl[1] = tile_t0;
        tl[3] = (tl[3] + (realtime() - tile_t0)) & 0xFFFFFFFFFFFFull;
    })
    if (yielded) { STATS(st_yields++;) __syncthreads(); continue; }   // the tile goes on later, on whichever wavefront picks it up
    if (s_limit_hit(s)) status |= ST_TRUNCATED;
    // The group index is untrusted input (a stale or crafted trailer): a tile that was decoded in full must have stopped
    // exactly where the next tile starts, as it does when the stream is decoded front to back; otherwise the picture is
    // flagged instead of being silently different from what the reference (which ignores the trailer) decodes.
    if (rflu(tile.end) != 0u && !(status & (ST_TRUNCATED | ST_CORRUPT)) && s.pos != rflu(tile.end)) status |= ST_CORRUPT;
    // planes the tile never reached read as zeros in the reference (empty Channel::data,
    // image/image.h:82-85; zero-filled residuals, transform/squeeze.h:379-383).  Every channel of
    // the tile ends up published as final, whatever path led here: nobody waits for ever.
    __syncthreads();
    for (int c = first_c; c <= last_c; c++) {
        const int gw = rfl(geom[c].w), gh = rfl(geom[c].h);
        const uint32_t done = (uint32_t)gh + 1u;
        if (!kHandOff || rflu(ld_agent(progress + c)) != done) {
            if ((int64_t)gw * gh > 0 && rfl(ld_agent(&meta[c].decoded)) == 0) fill_plane<kHandOff>(coef + geom[c].coef_off, 0, (int64_t)gw * gh, 0, lane);
            publish(c, done);
        }
    }
    if (lane == 0) { atomicOr(&P.status[img], status); atomicMax(&P.consumed[img], s.pos); }
    STATS(st_spin += waited;)
    if (sched) {
        if (pinned_tix == tix) pinned_tix = kNoWork;   // the scratch area is free again
        if (lane == 0) {
            st_agent(&rec->state, (uint32_t)TS_DONE);
            if (ld_agent(&rec->foreign)) atomicAdd(&P.cu_foreign[simd_key], 0xFFFFFFFFu);
            if (long_tile) atomicAdd(&P.simd_long[simd_slot], 0xFFFFFFFFu);
            atomicAdd(&P.cu_live[simd_key], 0xFFFFFFFFu);
            atomicAdd(&P.img_done[img], 1u);
            atomicAdd(P.done_total, 1u);
        }
    }
#if defined(FUIF_STATS) || defined(FUIF_PROF) || defined(FUIF_TILELOG)
    if (lane == 0 && P.tile_log) {
        // {image << 32 | first channel, first start, end, ticks some wavefront was running the tile | CU key << 48}: the log keeps the
        // first start and the running time itself (nothing extra stays live in registers across the tile)
        unsigned long long *tl = P.tile_log + (size_t)tix * 4;
        const unsigned long long t_end = realtime();
        tl[0] = ((unsigned long long)(uint32_t)img << 32) | (uint32_t)first_c;
        if (!resumed) tl[1] = tile_t0;
        tl[2] = t_end;
        // (CU key in bits 48..59, the SIMD the LAST run segment sat on in bits 60..61: HW_REG_HW_ID[5:4])
#ifdef FUIF_EMU
        const unsigned long long simd_id = (unsigned long long)blockIdx.x & 3ull;
#else
        const unsigned long long simd_id = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 4) & 3ull;
#endif
        tl[3] = ((tl[3] + (t_end - tile_t0)) & 0xFFFFFFFFFFFFull) | ((unsigned long long)simd_key << 48) | (simd_id << 60);
    }
#endif
#ifdef FUIF_PROF
    prof_acc[7] += __builtin_readcyclecounter() - prof_seg0; prof_acc[6] += realtime() - prof_rt0;
    if (lane == 0 && P.prof) for (int k = 0; k < 8; k++) atomicAdd(&P.prof[(size_t)PROF_ROW * 8 + k], prof_acc[k]);
#endif
    __syncthreads();
    }  // tile loop
#ifdef FUIF_STATS
    if (sched && lane == 0 && P.sched_stats) {
        atomicAdd(&P.sched_stats[0], st_idle); atomicAdd(&P.sched_stats[1], st_picks); atomicAdd(&P.sched_stats[2], st_yields); atomicAdd(&P.sched_stats[3], st_scan); atomicAdd(&P.sched_stats[4], st_spin); atomicAdd(&P.sched_stats[5], st_noctx); atomicAdd(&P.sched_stats[6], st_busy); atomicAdd(&P.sched_stats[7], realtime() - st_begin);
    }
#endif
}

// config: 0 = wide for hosts with two batches in flight (kLdsWide supernodes in LDS), 1 = dense, 2 = wide for a launch alone (kLdsWideAlone)
int maniac_max_waves(int config, int *per_simd) {
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    hipError_t e = config == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_maniac_decode<kLdsDense, true>, 64, 0)
                 : config == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_maniac_decode<kLdsWideAlone, true>, 64, 0)
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_maniac_decode<kLdsWide, true>, 64, 0);
    if (e != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_simd) *per_simd = per_cu / 4 > 0 ? per_cu / 4 : 1;   // a CU has 4 SIMDs
    return per_cu * prop.multiProcessorCount;
}

// hand_off = 0 promises that no tile reads what another tile of the launch writes (one tile per image)
void launch_maniac_decode(const DecodeParams &P, int n_waves, int config, int hand_off, hipStream_t stream) {
    if (config == 1) hipLaunchKernelGGL((k_maniac_decode<kLdsDense, true>), dim3(n_waves), dim3(64), 0, stream, P);
    else if (config == 2 && hand_off) hipLaunchKernelGGL((k_maniac_decode<kLdsWideAlone, true>), dim3(n_waves), dim3(64), 0, stream, P);
    else if (config == 2) hipLaunchKernelGGL((k_maniac_decode<kLdsWideAlone, false>), dim3(n_waves), dim3(64), 0, stream, P);
    else if (hand_off) hipLaunchKernelGGL((k_maniac_decode<kLdsWide, true>), dim3(n_waves), dim3(64), 0, stream, P);
    else hipLaunchKernelGGL((k_maniac_decode<kLdsWide, false>), dim3(n_waves), dim3(64), 0, stream, P);
}

}  // namespace fuifgpu


#ifdef FUIF_EMU
namespace fuifgpu { namespace { unsigned long long g_emu_stats[6]; } }
// emulator builds only (tools/emu_walk_stats.py): {symbols that walked the tree, rounds served from LDS, rounds from scratch}
extern "C" void fuifgpu_emu_walk_stats(unsigned long long *out6, int reset) {
    for (int k = 0; k < 6; k++) { out6[k] = fuifgpu::g_emu_stats[k]; if (reset) fuifgpu::g_emu_stats[k] = 0; }
}
#endif
