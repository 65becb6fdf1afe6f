// fuif_amd/csrc/transforms.h -- launch interface of the inverse-transform kernels
#pragma once
#include <hip/hip_runtime.h>

#include "fuifgpu_internal.h"

namespace fuifgpu {

// per-launch slab bases: address = base[buf] + z*stride[buf] + plane offset (z = image inside the chunk)
struct Bases {
    int32_t *base[3];
    int64_t stride[3];
};

void launch_op(const Op &op, const Bases &b, const PlaneRef *dev_list, ChannelMeta *meta, int n_channels, int img_first,
               int n_images, hipStream_t stream, int32_t *status = nullptr);   // status: per-image FUIFGPU_ST_* words (optional)

}  // namespace fuifgpu
