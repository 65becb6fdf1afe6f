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
    uint8_t *scratch;            // per stream: parse-order nodes | breadth-first nodes | leaves | parse stack | BFS queue
    size_t scratch_stride, bfs_off, leaves_off, stack_off, queue_off;
    int32_t max_properties;
    int32_t max_nodes;
    unsigned long long *prof;    // -DFUIF_PROF builds: 8 cycle counters per stream (else unused)
};

size_t maniac_scratch_bytes(int max_nodes, size_t *bfs_off, size_t *leaves_off, size_t *stack_off, size_t *queue_off);
void launch_maniac_decode(const DecodeParams &P, hipStream_t stream);

}  // namespace fuifgpu
