// fuif_amd/csrc/transforms.h -- launch interface of the inverse-transform kernels
#pragma once
#include <hip/hip_runtime.h>

#include "fuifgpu_internal.h"

namespace fuifgpu {

// per-launch slab bases: address = base[buf] + z*stride[buf] + plane offset (z = image inside the chunk)
struct Bases {
    int32_t *base[3];
    int64_t stride[3];
    const coef_t *c16;      // the chunk's coefficient slab as the entropy kernel wrote it (int16 samples; same plane offsets and image stride
                            // as base[BUF_COEF]): what the squeeze kernels read their residuals from when Op::r16 is set; may be NULL otherwise
};

void launch_op(const Op &op, const Bases &b, const PlaneRef *dev_list, ChannelMeta *meta, int n_channels, int img_first,
               int n_images, hipStream_t stream, int32_t *status = nullptr);   // status: per-image FUIFGPU_ST_* words (optional)

// coefficient samples as the entropy kernel stores them (int16, fuifgpu_internal.h) -> the int32 planes the inverse kernels work on:
// the planes of Plan::widen ({element offset, elements} pairs, device array) of n_images images: src / dst advance by `stride` elements per image
void launch_widen_planes(const coef_t *src, int32_t *dst, int64_t stride, const int64_t *dev_pairs, int n_planes, int64_t max_elems, int n_images, hipStream_t stream);

// interleaved 8/16-bit samples of up to 5 final planes (export/write_pam.h:136-150)
struct PackedPlanes {
    int32_t n;
    PlaneRef p[5];
};
void launch_pack(const Bases &b, const PackedPlanes &pp, int w, int h, int lo, int hi, int bytes_per_sample, uint8_t *dst, int64_t dst_stride,
                 int n_images, hipStream_t stream);

// per-image position-weighted 64-bit sums of n_images runs of `elems` int32 samples, `stride` samples apart (sums[] is zeroed on the stream first)
void launch_plane_checksums(const int32_t *planes, int64_t elems, int64_t stride, int n_images, unsigned long long *sums, hipStream_t stream);

// forward YCoCg (in place, three contiguous planes of n samples) and forward Squeeze of one plane (transform/ycocg.h:65-95,
// transform/squeeze.h:135-170,227-263): raw device pointers, the writer's optional GPU path
void launch_scale(int32_t *plane, int64_t n, int q, hipStream_t stream);   // transform/quantize.h:32-49 on one plane
void launch_fwd_ycocg(int32_t *c0, int32_t *c1, int32_t *c2, int64_t n, hipStream_t stream);
void launch_fwd_squeeze(bool horizontal, const int32_t *in, int w, int h, int32_t *avg, int32_t *res, hipStream_t stream);

}  // namespace fuifgpu
