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
    uint32_t *queue_head;        // zeroed before the launch
    uint32_t *progress;          // [n_images][n_channels] 0 = nothing yet, 1 + rows finished once the header is known; zeroed before the launch
    uint32_t *group_start;       // [n_images][n_channels] 1 + byte offset of the group that starts at this channel (0 = none); zeroed before the launch
    uint8_t *scratch;            // per wavefront: parse-order nodes | breadth-first nodes | leaves | parse stack | BFS queue
    size_t scratch_stride, bfs_off, leaves_off, stack_off, queue_off, subtree_off;
    int32_t max_properties;
    int32_t max_nodes;
    int32_t max_super;           // supernodes the scratch area holds
    unsigned long long *prof;    // -DFUIF_PROF builds: 8 cycle counters per stream (else unused)
};

int maniac_max_supernodes(int max_nodes);
size_t maniac_scratch_bytes(int max_nodes, size_t *bfs_off, size_t *leaves_off, size_t *stack_off, size_t *queue_off, size_t *subtree_off);
// The kernel exists in two LDS configurations: wide (1 wavefront per SIMD, most of the context tree in
// LDS) and dense (4 per SIMD).  maniac_max_waves = wavefronts the device holds at once in that configuration.
int maniac_max_waves(int dense);
void launch_maniac_decode(const DecodeParams &P, int n_waves, int dense, int hand_off, hipStream_t stream);

}  // namespace fuifgpu
