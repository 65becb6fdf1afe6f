// fuif_amd/csrc/maniac_decode.h -- launch interface of the entropy kernel
#pragma once
#include <hip/hip_runtime.h>

#include "fuifgpu_internal.h"

namespace fuifgpu {

// Saved state of a suspended tile (sched == 1).  Written by the wavefront that suspends it, then `state` is released;
// read by the wavefront that wins the READY -> RUNNING exchange.
enum : uint32_t { TS_NEW = 0, TS_READY = 1, TS_RUNNING = 2, TS_DONE = 3 };
struct TileRec {
    uint32_t state;
    uint32_t wait_chan, wait_val;   // runnable once progress[wait_chan] >= wait_val
    uint32_t y;                     // next row
    uint32_t range, low, pos;       // range coder + stream position
    uint32_t flags;                 // status bits 0..7 | eof_flag << 8 | predictor << 9
    uint32_t ctx, ctx_leaves;       // context area: offset into ctx_scratch, offset of the leaf chances inside it (256-byte units)
    uint32_t tree_size, n_super, cur_leaf;
    uint32_t t_first_lo, t_first_hi, run_ticks;
    uint32_t pin;                   // 0: the context lives in a context area; else 1 + the wavefront (workgroup id) in whose scratch area it lives
                                    // (no area was free): only that wavefront can resume the tile
    uint32_t foreign;               // the tile was started by a CU whose home queue is not its image's queue (cu_foreign counts them)
    uint32_t pad;
    uint32_t owner;                 // 1 + CU key of the wavefronts that may resume it (the CU that suspended it)
};

struct DecodeParams {
    const uint8_t *blobs;        // all streams of the batch, each 16-byte aligned and padded
    const StreamJob *jobs;
    int32_t n_images;
    int32_t n_channels;
    const ChannelGeom *geom;     // coded channel table (shared by the batch)
    coef_t *coef;                // [n_images][coef_stride] int16 samples (fuifgpu_internal.h)
    int64_t coef_stride;
    ChannelMeta *meta;           // [n_images][n_channels]
    int32_t *status;             // [n_images]
    uint32_t *consumed;          // [n_images]
    const uint16_t *tables;      // [0,8192): tree coder table, [8192,16384): pixel coder table
    // work list: tiles in dependency order (a tile only reads channels of tiles before it), handed
    // out to persistent wavefronts through *queue_head
    const Tile *tiles;
    int32_t n_tiles;
    // Two ways of handing tiles to the persistent wavefronts:
    //  sched == 0: one list, one head counter (q_head[0]); a tile runs to completion on the wavefront that took it and
    //              spins when it needs rows another tile has not finished yet;
    //  sched == 1: "contexts" (dense launches with more tiles than wavefronts).  The tiles of an image are contiguous
    //              and in stream order; images are dealt to n_queues queues, one per CU (simd_claim: physical CU ->
    //              dense index, numbered by first arrival; an affinity, never a requirement).  A wavefront looks for
    //              work in its home queue, then in the next few: first a SUSPENDED tile whose awaited rows have
    //              arrived, else the next unstarted tile of an image.  A tile that meets unfinished reference rows
    //              saves its coder state in its TileRec (its tree and leaf chances live in a context area, not in the
    //              wavefront's scratch) and gives the wavefront back instead of spinning; a wavefront of the same CU
    //              resumes it later.
    int32_t sched;
    int32_t prio_base;           // tiles of size class <= prio_base run at wavefront priority 3, +1 at 2, +2 at 1 (negative: off)
    uint32_t yield_slack;        // a suspended tile is runnable again once the rows it waits for are this many rows ahead (or final)
    int32_t n_queues;
    uint32_t *q_head;            // sched 0: [1] head of the single list; zeroed before the launch
    const uint32_t *q_img_begin; // sched 1: [n_queues + 1] into q_images
    const uint32_t *q_images;    // sched 1: image ids, queue by queue
    const uint32_t *img_tile_begin; // sched 1: [n_images + 1]: tiles of image i are tiles[img_tile_begin[i] .. img_tile_begin[i+1])
    uint32_t *img_next;          // sched 1: [n_images] tiles started; zeroed before the launch
    uint32_t *img_done;          // sched 1: [n_images] tiles finished; zeroed before the launch
    uint32_t *q_turn;            // sched 1: [n_queues] whose turn it is to start a tile; zeroed before the launch
    uint32_t *done_total;        // sched 1: tiles finished; zeroed before the launch
    uint32_t *started_total;     // sched 1: tiles started; once it reaches n_tiles a wavefront without work can only ever resume tiles of its own CU
    uint32_t *heartbeat;         // bumped every few rows by every running tile: "no progress anywhere" is what a stall verdict needs, not "I was idle long"
    uint32_t *cu_alive;          // sched 1: [4096] per CU key: wavefronts that have arrived and not retired
    uint32_t *cu_live;           // sched 1: [4096] per CU key: tiles started on that CU and not finished (running or suspended)
    uint32_t *cu_foreign;        // sched 1: [4096] per CU key: those of them that belong to another queue than the CU's home queue
    uint32_t *simd_long;         // sched 1: [4096 * 4] per CU key and SIMD: long tiles (>= 1/8 of their picture) running on that SIMD right now
    int32_t long_per_simd;       // a wavefront does not start / resume a long tile while its SIMD runs this many already, unless it has been idle for a
                                 // while (0: no such rule).  Long tiles are the launch's critical path; by chance a SIMD got 1 to 7 of them and each extra
                                 // one costs the others ~1 % (profiles/r3_stragglers.txt)
    TileRec *tile_rec;           // sched 1: [n_tiles]; zeroed before the launch
    uint8_t *ctx_scratch;        // sched 1: the arena of the context areas (supernodes | leaf chances) of suspendable tiles: ctx_units_per_queue * n_queues units
    uint32_t ctx_units_per_queue; //         arena size in 256-byte units per queue (the arena is sized per queue and used as ONE since round 6)
    unsigned long long *ctx_used; // sched 1: units handed out from the bottom (low word: long tiles) and from the top (high word: the others) of the arena -- bump
                                 //          allocation, a launch never frees; zeroed before the launch
    unsigned long long *sched_stats; // -DFUIF_STATS builds, sched 1: {ticks wavefronts spent without work before the last tile finished, tiles picked up, suspensions, ...}; zeroed before the launch
    uint32_t *simd_claim;        // [2 * 4096 + 1] {arrivals, 1 + dense index} per physical CU key, then the CU counter; zeroed before the launch
    uint32_t *progress;          // [n_images][n_channels] 0 = nothing yet, 1 + rows finished once the header is known; zeroed before the launch
    uint32_t *group_start;       // [n_images][n_channels] 1 + byte offset of the group that starts at this channel (0 = none); zeroed before the launch
    uint8_t *scratch;            // per wavefront: parse-order nodes | breadth-first nodes | leaves | parse stack | BFS queue
    size_t scratch_stride, bfs_off, leaves_off, stack_off, queue_off, subtree_off;
    int32_t max_properties;
    int32_t max_nodes;
    int32_t max_super;           // supernodes the scratch area holds
    unsigned long long *prof;    // -DFUIF_PROF builds: 8 cycle counters per stream (else unused)
    unsigned long long *tile_log; // -DFUIF_STATS builds: [n_tiles][4] {image << 32 | first channel, start, end, waited} in s_memrealtime ticks (100 MHz); waited's top 16 bits = SIMD key
};

int maniac_max_supernodes(int max_nodes);
size_t maniac_scratch_bytes(int max_nodes, size_t *bfs_off, size_t *leaves_off, size_t *stack_off, size_t *queue_off, size_t *subtree_off);
// The kernel exists in two LDS configurations: wide (1 wavefront per SIMD, most of the context tree in
// LDS) and dense (4 per SIMD).  maniac_max_waves = wavefronts the device holds at once in that configuration.
int maniac_max_waves(int config, int *per_simd = nullptr);   // config: 0 wide (two batches in flight), 1 dense, 2 wide (a launch alone)
void launch_maniac_decode(const DecodeParams &P, int n_waves, int config, int hand_off, hipStream_t stream);

}  // namespace fuifgpu
