// fuif_amd/csrc/maniac_decode.h -- launch interface of the entropy kernel
#pragma once
#include <hip/hip_runtime.h>

#include "fuifgpu_internal.h"

namespace fuifgpu {

struct DecodeParams {
    const uint8_t *blobs;        // all streams of the batch, each 16-byte aligned and padded
    const StreamJob *jobs;
    int32_t n_images;
    int32_t n_channels;
    const ChannelGeom *geom;     // coded channel table (shared by the batch)
    int32_t *coef;               // [n_images][coef_stride]
    int64_t coef_stride;
    ChannelMeta *meta;           // [n_images][n_channels]
    int32_t *status;             // [n_images]
    uint32_t *consumed;          // [n_images]
    const uint16_t *tables;      // [0,8192): tree coder table, [8192,16384): pixel coder table
    // work list: tiles in dependency order (a tile only reads channels of tiles before it), handed
    // out to persistent wavefronts through *queue_head
    const Tile *tiles;
    int32_t n_tiles;
    // The list is cut into n_queues queues (tiles [q_begin[q], q_begin[q+1]), each with its own head counter).  A
    // wavefront's home queue follows from the SIMD it runs on (simd_claim: physical id -> dense index, filled in by
    // the first wavefront of every SIMD); with one image per queue a SIMD's resident wavefronts work through one
    // image, so every SIMD gets the same amount of work whatever the order wavefronts finish in.  A wavefront whose
    // home queue is empty goes through the other queues (the mapping is an affinity, never a requirement).
    const uint32_t *q_begin;     // [n_queues + 1]
    uint32_t *q_head;            // [n_queues], zeroed before the launch
    int32_t n_queues;
    uint32_t *simd_claim;        // [2 * 16384 + 1] {arrivals, 1 + dense index} per physical SIMD key, then the SIMD counter; zeroed before the launch
    uint32_t *progress;          // [n_images][n_channels] 0 = nothing yet, 1 + rows finished once the header is known; zeroed before the launch
    uint32_t *group_start;       // [n_images][n_channels] 1 + byte offset of the group that starts at this channel (0 = none); zeroed before the launch
    uint8_t *scratch;            // per wavefront: parse-order nodes | breadth-first nodes | leaves | parse stack | BFS queue
    size_t scratch_stride, bfs_off, leaves_off, stack_off, queue_off, subtree_off;
    int32_t max_properties;
    int32_t max_nodes;
    int32_t max_super;           // supernodes the scratch area holds
    unsigned long long *prof;    // -DFUIF_PROF builds: 8 cycle counters per stream (else unused)
    unsigned long long *tile_log; // [n_tiles][4] {image << 32 | first channel, start, end, waited} in s_memrealtime ticks (100 MHz); waited's top 16 bits = SIMD key
};

int maniac_max_supernodes(int max_nodes);
size_t maniac_scratch_bytes(int max_nodes, size_t *bfs_off, size_t *leaves_off, size_t *stack_off, size_t *queue_off, size_t *subtree_off);
// The kernel exists in two LDS configurations: wide (1 wavefront per SIMD, most of the context tree in
// LDS) and dense (4 per SIMD).  maniac_max_waves = wavefronts the device holds at once in that configuration.
int maniac_max_waves(int dense, int *per_simd = nullptr);
void launch_maniac_decode(const DecodeParams &P, int n_waves, int dense, int hand_off, hipStream_t stream);

}  // namespace fuifgpu
