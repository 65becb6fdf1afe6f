/* include/fuifgpu.h -- C-ABI of libfuifgpu.so: the MI355X (gfx950) FUIF decode path.
 *
 * This is the drop-in boundary for the ONE hot path of cloudinary/fuif: channel-group entropy
 * decode followed by the inverse transform chain.  Each entry point names the reference
 * interface it replaces (paths relative to the reference tree).  Plain pointers and sizes only;
 * no C++ or torch types cross this boundary.  See INTEGRATION.md for the reference-side binding.
 *
 * Threading: a fuifgpu_batch is owned by one host thread at a time.  All device work is enqueued
 * on the hipStream_t passed as `void *stream` (NULL = the default stream).
 */
#ifndef FUIFGPU_H
#define FUIFGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FUIFGPU_ABI_VERSION 3   /* 2 (round 4): fuifgpu_encode_options starts with struct_size; sibling batches freeze the launch resources.
                                 * 3 (rounds 5-6): every fuifgpu_batch_* call runs on the batch's own device (the caller's is restored on return);
                                 *   the device / peer-copy / checksum / in-flight entry points exist; fuifgpu_batch_create reads FUIFGPU_IN_FLIGHT.
                                 *   A host checks fuifgpu_abi_version() >= the version whose entry points it binds before looking them up. */

/* error codes (0 = success).  The reference reports these conditions as `return false` +
 * e_printf (encoding/encoding.cpp:601-605, 276-279, 697-700). */
#define FUIFGPU_OK 0
#define FUIFGPU_E_NOT_FUIF 1      /* bad magic / short header */
#define FUIFGPU_E_CORRUPT 2       /* header or transform list is inconsistent */
#define FUIFGPU_E_UNSUPPORTED 3   /* feature outside the hot-path scope (`-E k` > 50: more than 25 reference channels per context tree, the CLI default is 12; a data-driven Permute over channels of unequal geometry) */
#define FUIFGPU_E_ARG 4
#define FUIFGPU_E_HIP 5           /* a HIP runtime call failed; see fuifgpu_last_error() */
#define FUIFGPU_E_MISMATCH 6      /* image does not share the batch's plan signature */
#define FUIFGPU_E_NOMEM 7

/* per-image status bits written by the entropy kernel (fuifgpu_batch_status) */
#define FUIFGPU_ST_TRUNCATED 1    /* EOF or preview limit reached: remaining samples are zero-filled
                                     (not an error, encoding/encoding.cpp:209-219) */
#define FUIFGPU_ST_CORRUPT 2      /* the reference would return false */
#define FUIFGPU_ST_UNSUPPORTED 4
#define FUIFGPU_ST_STALLED 8 /* internal error: a group gave up waiting for rows of another group (always with CORRUPT) */

typedef struct fuifgpu_plan fuifgpu_plan;    /* host: parsed header + channel table + inverse schedule */
typedef struct fuifgpu_batch fuifgpu_batch;  /* device: buffers and state for N same-geometry images */

/* replaces the header half of fuif_decode<IO>() (encoding/encoding.cpp:599-702): what
 * `Image(w,h,maxval,nb_channels)` + every Transform::meta_apply() (transform/transform.cpp:66-81)
 * compute, without touching pixel data. */
typedef struct {
    int32_t w, h, bit_depth, maxval, nb_channels, colormodel, max_properties, nb_frames;
    int32_t nb_transforms, nb_coded_channels, nb_output_channels, nb_ops;
    int32_t responsive_offsets[5];
    int32_t data_start;
    int64_t coef_elems, out_elems, tmp_elems; /* elements per image: int16 samples in the coefficient slab (a coded sample is the
                                                 reference's pixel_type, image/image.h:35), int32 in the output and scratch slabs */
    uint64_t signature;
} fuifgpu_image_info;

/* mirrors the geometry fields of `class Channel` (image/image.h:54-91) */
typedef struct {
    int32_t w, h, hshift, vshift, hcshift, vcshift, component, reserved;
    int64_t offset; /* element offset inside the per-image coefficient (coded) or output slab */
} fuifgpu_channel_desc;

const char *fuifgpu_strerror(int code);
const char *fuifgpu_last_error(void);
int fuifgpu_abi_version(void);

/* ---- host side: header parse + planning (no GPU needed) ------------------------------------ */
int fuifgpu_plan_create(const uint8_t *blob, size_t size, fuifgpu_plan **out);
void fuifgpu_plan_destroy(fuifgpu_plan *plan);
int fuifgpu_plan_info(const fuifgpu_plan *plan, fuifgpu_image_info *info);
int fuifgpu_plan_coded_channel(const fuifgpu_plan *plan, int index, fuifgpu_channel_desc *desc);
int fuifgpu_plan_output_channel(const fuifgpu_plan *plan, int index, fuifgpu_channel_desc *desc);
/* transform list as stored in Image::transform after meta_apply (encoding.cpp:673-693);
 * params_out receives at most cap ints, *nparams the real count */
int fuifgpu_plan_transform(const fuifgpu_plan *plan, int index, int32_t *id, int32_t *params_out, int cap, int32_t *nparams);
/* maniac/chance.cpp:31-65 build_table(), table[2*chance+bit]; host helper used by tests */
void fuifgpu_build_chance_table(uint16_t *table8192, uint32_t alpha, int cut);

/* ---- device side --------------------------------------------------------------------------- */
/* Creates device state for `n_images` streams that all share `plan`'s signature.
 * coef_ext / out_ext: optional caller-owned device slabs of n_images*coef_elems /
 * n_images*out_elems int32 (e.g. a torch tensor's data_ptr); NULL = allocated by the library.
 * tmp_images: number of images whose inverse-transform scratch is resident at once (0 = default). */
int fuifgpu_batch_create(const fuifgpu_plan *plan, int n_images, size_t blob_capacity_bytes,
                         int16_t *coef_ext, int32_t *out_ext, int tmp_images, fuifgpu_batch **out);
void fuifgpu_batch_destroy(fuifgpu_batch *batch);

/* A batch WITHOUT an output slab, for batches whose outputs do not fit next to their coefficients (BASELINE config C4: 256 x
 * 8192x8192x4 -- 137 GB of int16 coefficients, 275 GB of int32 outputs): every image is entropy-decoded in ONE launch, then the
 * inverse transforms run range by range into caller memory (fuifgpu_batch_undo_transforms_to) and the caller consumes each
 * slice before the next.  fuifgpu_batch_undo_transforms, _out_ptr, _download_out and _pack_out are refused on such a batch.
 * The reference has no counterpart: it decodes one image at a time (fuif.cpp:213-233). */
int fuifgpu_batch_create_streaming(const fuifgpu_plan *plan, int n_images, size_t blob_capacity_bytes, int tmp_images, fuifgpu_batch **out);
/* Image::undo_transforms(0) for images [first_image, first_image + n_images) of the current decode; their output planes go to
 * out_device (n_images * out_elems int32, device memory; plan output-channel offsets apply inside each image's slice).  Any
 * batch takes it; every image once per decode (a range that repeats an image, or a mix with fuifgpu_batch_undo_transforms, is
 * FUIFGPU_E_ARG).  A range counts as done once the call has queued it; after a failed call (a HIP error) decode again before
 * retrying.  fuifgpu_batch_last_timing's transform_ms is the time of the LAST call's range, not of all ranges of a decode. */
int fuifgpu_batch_undo_transforms_to(fuifgpu_batch *batch, int first_image, int n_images, int32_t *out_device, void *stream);

/* A second set of stream buffers for the SAME slabs, so that the upload of the next batch (host parse + H2D copies, on a copy
 * stream, from another host thread) runs while the previous batch decodes: the sibling owns what fuifgpu_batch_upload writes
 * (stream bytes, tile lists, per-image status / consumed / metadata) and launches with the primary's coefficient and output
 * slabs, decoder scratch, context arenas and transform arena -- ~12 MB per 4K stream instead of a second 45 GB of launch state.
 * Creating the first sibling sizes the primary's decoder scratch and context arenas for the worst case (the device's wavefront
 * capacity, a full batch) and they never move again while a sibling exists, so an upload into one batch on another host thread
 * cannot pull them from under a decode of the other; either batch may be loaded first, with any number of images <= n_images.
 * Rules: decode / undo_transforms of the
 * two on ONE stream (they share the slabs); destroy the sibling first (a sibling that outlives its primary refuses every call).  The pattern, per step k: thread U uploads into batch
 * (k+1)%2 on the copy stream while the caller runs decode + undo_transforms of batch k%2; join; consume; repeat
 * (bench.py's `value_incl_h2d`, tests/test_gpu_synthetic.py).  The reference has no counterpart: it reads one file at a time
 * (encoding/encoding.cpp:745-753). */
int fuifgpu_batch_create_sibling(fuifgpu_batch *primary, size_t blob_capacity_bytes, fuifgpu_batch **out);

/* Stage compressed streams (host pointers) into HBM.  Every blob must parse to the batch's
 * signature.  preview: -1 = full decode, 0..4 = responsive truncation point
 * (fuif_options::preview, encoding/encoding.h:34).  Replaces the IO object handed to
 * fuif_decode<IO>() (fileio.h:33-143). */
int fuifgpu_batch_upload(fuifgpu_batch *batch, const uint8_t *const *blobs, const size_t *sizes,
                         int n_images, int preview, void *stream);

/* replaces the channel loop of fuif_decode (encoding/encoding.cpp:708-717) and
 * fuif_decode_channel (encoding/encoding.cpp:259-429) for every stream of the batch:
 * fills the coefficient slab and the per-channel {minval,maxval,q}. */
int fuifgpu_batch_decode(fuifgpu_batch *batch, void *stream);

/* replaces Image::undo_transforms(0) (image/image.cpp:94-115) for every image of the batch:
 * inverse Squeeze / Quantize / DCT / ChromaSubsample / YCoCg / YCbCr + final clamp into the
 * output slab. */
int fuifgpu_batch_undo_transforms(fuifgpu_batch *batch, void *stream);
/* (once per decode: the inverse of Approximate rewrites the per-channel metadata, so a second call is refused with FUIFGPU_E_ARG
 * until the batch is decoded again.  The coefficient slab itself is not touched -- the kernels work on a widened copy of a chunk
 * of images -- and fuifgpu_batch_download_coef stays legal after it.) */

int fuifgpu_batch_sync(fuifgpu_batch *batch, void *stream);
/* status[n_images] (FUIFGPU_ST_* bits), bytes_consumed[n_images] (io.ftell() at the end) */
int fuifgpu_batch_status(fuifgpu_batch *batch, int32_t *status, uint32_t *bytes_consumed);
/* {minval,maxval,q,decoded} of every coded channel of one image (Channel::minval/maxval/q) */
int fuifgpu_batch_channel_meta(fuifgpu_batch *batch, int image, int32_t *meta4_per_channel);
int16_t *fuifgpu_batch_coef_ptr(fuifgpu_batch *batch, int image);   /* device pointer: int16 samples (ABI 2; fuifgpu_batch_download_coef hands them out as int32) */
int32_t *fuifgpu_batch_out_ptr(fuifgpu_batch *batch, int image);    /* device pointer */
int fuifgpu_batch_download_coef(fuifgpu_batch *batch, int image, int32_t *host, void *stream);
int fuifgpu_batch_download_out(fuifgpu_batch *batch, int image, int32_t *host, void *stream);
/* ---- packed output: what export/write_pam.h:136-150 puts in a PNM/PAM file ---------------------------------
 * For every pixel of the w x h image the first `components` output channels (0 = all, at most 4), clamped to
 * [0,maxval], 1 byte per sample when maxval < 256 else 2 bytes big-endian, interleaved.  The packing runs on
 * the GPU after fuifgpu_batch_undo_transforms; a host that only wants the picture pulls 1/4 (16-bit) or 1/8
 * (8-bit) of the bytes of the int32 planes over PCIe. */
size_t fuifgpu_plan_packed_bytes(const fuifgpu_plan *plan, int components);                 /* bytes per image */
/* n_images images starting at first_image into a DEVICE buffer of n_images * packed_bytes */
int fuifgpu_batch_pack_out(fuifgpu_batch *batch, int first_image, int n_images, int components, uint8_t *dst_device, void *stream);
/* one image into HOST memory (packs into a temporary device buffer, copies, synchronises the stream) */
int fuifgpu_batch_download_packed(fuifgpu_batch *batch, int image, int components, uint8_t *host, void *stream);

/* Verification aid -- no counterpart in the reference (its tests compare files with `cmp`): sums_device[k] = sum over the
 * elems_per_image int32 samples of image k (image_stride samples apart, a multiple of 4; planes_device 16-byte aligned) of
 * sample * (index mod 65521 + 1), as a wrapping 64-bit integer; asynchronous on `stream`.  A host that decodes step after step into
 * the same planes (bench.py's overlapped steps, fuifgpu_batch_undo_transforms_to slice by slice) keeps 8 bytes per image and step
 * and compares them when the steps are done, instead of holding or downloading the planes of every step. */
int fuifgpu_plane_checksums(const int32_t *planes_device, int64_t elems_per_image, int64_t image_stride, int n_images, uint64_t *sums_device, void *stream);

/* kernel time of the last decode / undo_transforms launch set, measured with hipEvents on the
 * caller's stream (ms); used by bench.py for the roofline */
int fuifgpu_batch_last_timing(fuifgpu_batch *batch, float *decode_ms, float *transform_ms);
/* diagnostic builds (-DFUIF_PROF) only: 8 counters per stream of the last decode, shader cycles
 * {vector phase, property patch, tree walk, leaf switch, symbol decode, per-pixel rest (incl. the row store),
 * 100 MHz ticks of the run segments, shader cycles of the run segments}; all zero in release builds
 * (-DFUIF_PROF_BY_CHANNEL: one row per first channel of a tile, summed over the images, instead of one per stream) */
int fuifgpu_batch_profile(fuifgpu_batch *batch, uint64_t *out8_per_image);
/* diagnostic builds (-DFUIF_STATS or -DFUIF_PROF; FUIFGPU_E_UNSUPPORTED from the release library, whose kernel carries no
 * statistics): the schedule of the last decode launch, 4 words per tile of the work list {image << 32 | first channel,
 * first start, end, ticks some wavefront was running the tile | CU key << 48}, times in 100 MHz s_memrealtime ticks.
 * Logging starts with the first call (which returns *n_tiles = 0); tools/tile_timeline.py turns it into a report. */
int fuifgpu_batch_tile_log(fuifgpu_batch *batch, uint64_t *out4_per_tile, int cap, int *n_tiles);
/* diagnostic: counters of the tile scheduler for the last dense launch {ticks (100 MHz) wavefronts spent without work while
 * tiles were unfinished, tiles picked up (starts + resumptions), suspensions, ticks spent looking for the tile picked up, ticks spent spinning inside tiles (waits a tile could not be
 * suspended for), suspendable tiles that found their context arena full, ticks between picking a tile up and looking for
 * the next one, ticks the wavefronts lived} */
int fuifgpu_batch_sched_stats(fuifgpu_batch *batch, uint64_t *out8);

/* ---- group index (csrc/index.cpp) --------------------------------------
 * A FUIF stream is a chain of channel groups (fuif_decode_channel, encoding/encoding.cpp:259-429),
 * each with its own range coder, whose byte boundaries the encoder knows (encoding.cpp:525-527,542)
 * but does not store.  The index stores them in a trailer BEHIND the stream
 *     <stream> <payload> <u32 LE payload length> "FGIX"
 *     payload = varint 1 ; varint n ; n x { varint channel delta ; varint byte-offset delta }
 * which the reference decoder never reads (it stops after the last group, encoding.cpp:708-717):
 * indexed files still decode with the unmodified reference.  fuifgpu_batch_upload() finds the
 * trailer by itself and then decodes every group of every image on its own wavefront; streams
 * without one are decoded one wavefront per image.  Ways to get an index: the writer below
 * (emit_index), or decode once and keep fuifgpu_batch_group_index()'s result with the asset. */
/* groups of the trailer of `blob` (n_groups = 0: none / not valid for this stream) */
int fuifgpu_index_parse(const uint8_t *blob, size_t size, int32_t *first_channel, uint32_t *start, int cap, int *n_groups);
/* copy of the stream with a trailer for the given groups (replaces an existing one); free with fuifgpu_free_blob */
int fuifgpu_index_append(const uint8_t *blob, size_t size, const int32_t *first_channel, const uint32_t *start, int n_groups,
                         uint8_t **blob_out, size_t *size_out);
/* after a decode: the groups the entropy kernel went through for one image (any stream, indexed or not) */
int fuifgpu_batch_group_index(fuifgpu_batch *batch, int image, int32_t *first_channel, uint32_t *start, int cap, int *n_groups);
/* enable = 0: ignore trailers from the next upload on (A/B measurements; default 1) */
int fuifgpu_batch_set_group_parallel(fuifgpu_batch *batch, int enable);
/* How many batches the host keeps in flight on this device (default 1); applies from the next upload on.  It only matters for launches with few tiles
 * (streams without group index: one wavefront per picture).  1: such a launch runs with 58 supernodes of every context tree in LDS, one wavefront per
 * SIMD -- the fastest a launch ALONE can be (20.8 s for 1024 x 4K).  2 or more: 20 supernodes, two wavefronts per SIMD -- 6 % slower alone, but the
 * launch of a second batch on a second stream runs beside it: 24.5 s for two such launches instead of 41.6 s (profiles/r5_overlap_timeline_and_wide_variants.txt).
 * The reference decodes one file at a time (fuif.cpp:213-233); no counterpart. */
int fuifgpu_batch_set_in_flight(fuifgpu_batch *batch, int n_batches);

/* ---- several GPUs of one node (round 5) ---------------------------------------------------------------------
 * The reference decodes file after file on one core (fuif.cpp:213-233: fuif_decode_file + undo_transforms per file); a batch of
 * independent images shards across the GPUs of a node with no data-path exchange.  Inside ONE process the unit is
 * the calling thread's current device, as in HIP: fuifgpu_set_device() selects it, a batch lives on the device that was current when
 * it was created, and every fuifgpu_batch_* call runs on the batch's own device whatever the calling thread's current device is (it
 * is restored on return) -- so a host runs one thread per device (fuif_decode_files in fuif_amd/boundary does) or one thread over
 * the batches of several devices.  A `stream` handed to a fuifgpu_batch_* call must be a stream OF THE BATCH'S DEVICE (NULL = that
 * device's null stream); the library does not check it, and HIP reports a foreign stream as an invalid handle at the first launch.
 * fuifgpu_dev_* and the single-transform entry points use the calling thread's current device.
 * Several PROCESSES (one per GPU, torch.distributed / RCCL: bench.py, fuif_amd/dist.py) each simply see their own device. */
int fuifgpu_device_count(int *n_devices);
int fuifgpu_set_device(int device);                                   /* FUIFGPU_E_ARG: no such device */
int fuifgpu_get_device(int *device);
int fuifgpu_batch_device(const fuifgpu_batch *batch, int *device);
/* The final gather's building block inside one process: `bytes` from src_device's memory into dst_device's over xGMI (peer access
 * is enabled on first use; without it the runtime stages the copy), asynchronous on `stream` (a stream of the calling thread's
 * current device; NULL = its null stream).  What a host uses to collect the packed pictures (fuifgpu_batch_pack_out) of every
 * GPU's shard on one GPU; the multi-process form of the same gather goes through RCCL (fuif_amd/dist.py gather_packed). */
int fuifgpu_peer_copy(void *dst_device_ptr, int dst_device, const void *src_device_ptr, int src_device, size_t bytes, void *stream);

/* ---- device memory for hosts that are not HIP programs (the boundary layer is plain g++ code) ------------- */
void *fuifgpu_dev_alloc(size_t bytes);                       /* NULL on failure (fuifgpu_last_error) */
void fuifgpu_dev_free(void *device_ptr);
int fuifgpu_dev_mem_info(size_t *free_bytes, size_t *total_bytes);   /* hipMemGetInfo: what a host sizes its batches with */
int fuifgpu_dev_upload(void *dst_device, const void *src_host, size_t bytes);
int fuifgpu_dev_download(void *dst_host, const void *src_device, size_t bytes);   /* waits for the null stream */

/* ---- single-transform entry points on raw device planes (row-major int32) ------------------
 * These are what Transform::apply(image, true) (transform/transform.cpp:48-63) dispatches to in the C++ boundary
 * layer (fuif_amd/boundary/fuif_gpu_boundary.cpp binds Transform::apply for the Squeeze, YCoCg, YCbCr, DCT, Quantize and
 * ChromaSubsample inverses to them: the path of Image::undo_transforms(keep != 0)); tests/test_gpu_transform_exports.py
 * checks each against the oracle. */
/* transform/squeeze.h:81-132 inv_hsqueeze: avg w1 x h + residual w2 x h -> out (w1+w2) x h */
int fuifgpu_inv_hsqueeze(const int32_t *avg, int w1, const int32_t *res, int w2, int h, int32_t *out, int n_planes,
                         int64_t avg_stride, int64_t res_stride, int64_t out_stride, void *stream);
/* transform/squeeze.h:173-224 inv_vsqueeze: avg w x h1 + residual w x h2 -> out w x (h1+h2) */
int fuifgpu_inv_vsqueeze(const int32_t *avg, int h1, const int32_t *res, int h2, int w, int32_t *out, int n_planes,
                         int64_t avg_stride, int64_t res_stride, int64_t out_stride, void *stream);
/* transform/ycocg.h:33-63 inv_YCoCg, in place on three planes with row pitches p0,p1,p2 */
int fuifgpu_inv_ycocg(int32_t *c0, int32_t *c1, int32_t *c2, int w, int h, int p0, int p1, int p2, int maxval, void *stream);
/* transform/ycbcr.h:33-63 inv_YCbCr */
int fuifgpu_inv_ycbcr(int32_t *c0, int32_t *c1, int32_t *c2, int w, int h, int p0, int p1, int p2, int minval, int maxval, void *stream);
/* transform/quantize.h:32-49 inv_quantize of one plane: every sample times the channel's quantisation constant, in place */
int fuifgpu_inv_quantize(int32_t *plane, int64_t n_samples, int q, void *stream);
/* transform/dct.h:88-107 + 282-291: 64 coefficient planes (bw x bh each, src[i] in the
 * reference's own zig-zag position order i=0..63) -> (8bw) x (8bh) samples; DC offset (maxval+1)*4 */
int fuifgpu_idct8x8(const int32_t *const *src64_dev, int bw, int bh, int32_t *out, int maxval, void *stream);
/* transform/subsample.h:90-126 chroma upsampling: the "fancy" filter for srh, srv in {1,2}, plain replication when either is larger (4:1:1) */
int fuifgpu_upsample(const int32_t *in, int w, int h, int srh, int srv, int32_t *out, void *stream);
/* transform/palette.h:57-64 inv_palette, one component per call: out[i] = palette_row[CLAMP(index[i], 0, colours-1)] over w x h samples
 * (palette_row = row `component` of the palette meta-channel; colours == 0 reads Channel::zero, image.h:82). out may not be index. */
int fuifgpu_inv_palette(const int32_t *index, int w, int h, const int32_t *palette_row, int colours, int32_t *out, void *stream);
/* transform/approximate.h:44-57 inv_approximate of one channel, in place: plane[i] = plane[i] * q + remainder[i] (q = parameter + 1);
 * remainder == NULL = "the remainder channel is not available" (:49,54): nothing is added */
int fuifgpu_inv_approximate(int32_t *plane, const int32_t *remainder, int64_t n_samples, int q, void *stream);
/* transform/2dmatch.h:112-177 inv_match, in place on n_planes (<= 64) w x h planes; match = the match meta-channel, match_q / match_maxval its
 * Channel::q and ::maxval (the mode is data: q == 1 free offsets :136-146, q == 2*fh*fh+(fh&1) previous frames :147-171, anything else
 * FUIFGPU_E_CORRUPT like :172-175). Synchronous. FUIFGPU_E_UNSUPPORTED (planes untouched) for a forward reference (an image narrower than the
 * offset spiral: the one case where the reference's scan order matters). */
int fuifgpu_inv_match(const int32_t *match, int w, int h, int32_t *const *planes, int n_planes, int softmatch, int match_q, int match_maxval,
                      int nb_frames, void *stream);
/* ---- forward transforms of the writer, raw device planes, contiguous rows ----
 * transform/ycocg.h:65-95 fwd_YCoCg, in place on three w x h planes */
int fuifgpu_fwd_ycocg(int32_t *c0, int32_t *c1, int32_t *c2, int w, int h, void *stream);
/* transform/squeeze.h:135-170 fwd_hsqueeze: in w x h -> avg ((w+1)/2) x h + residual (w/2) x h */
int fuifgpu_fwd_hsqueeze(const int32_t *in, int w, int h, int32_t *avg, int32_t *res, void *stream);
/* transform/squeeze.h:227-263 fwd_vsqueeze: in w x h -> avg w x ((h+1)/2) + residual w x (h/2) */
int fuifgpu_fwd_vsqueeze(const int32_t *in, int w, int h, int32_t *avg, int32_t *res, void *stream);

/* ---- stream writer (host C++; the input generator) --------------------
 * Writes a lossless FUIF stream the reference decoder accepts, with the format decisions of the
 * reference CLI for photographic PNM input (fuif.cpp:380-455,580-588; encoding/encoding.cpp:455-573).
 * tree_mode 0 = single-leaf MANIAC trees (byte-identical to `fuif -I 0`), 1 = trees learned by the
 * writer's own greedy learner.  *blob_out is malloc'd; release with fuifgpu_free_blob. */
typedef struct {
    uint32_t struct_size;   /* = sizeof(fuifgpu_encode_options) of the CALLER's header.  The library reads that many bytes and takes
                               every field behind them as 0, so the struct can grow at its end without breaking callers built
                               against an older header (ABI 2, round 4: round 3 appended gpu_entropy to the unversioned struct and a
                               caller built before that had 4 bytes read past its object).  0 or a size that is not a multiple of
                               4 is FUIFGPU_E_ARG. */
    int32_t ycocg;          /* 1: YCoCg when nch >= 3 (CLI default) */
    int32_t squeeze;        /* 1: default Squeeze (CLI default "responsive") */
    int32_t max_properties; /* CLI default 12 (-E) */
    int32_t tree_mode;      /* 0 none, 1 learned */
    int32_t max_tree_nodes; /* cap for learned trees (<= 65535) */
    int32_t emit_index;     /* 1: append the group index trailer (see fuifgpu_index_*) */
    int32_t split_bits;     /* learned trees: > 0 = a split must save this many bits (flat); 0 = the default rule, the description
                               length of the extra leaf ((k/2) log2 pixels), which follows the reference encoder's tree sizes */
    int32_t gpu_forward;    /* 1: forward YCoCg and Squeeze run on the GPU (fuifgpu_fwd_*).
                               Same bytes as with 0.  No GPU: FUIFGPU_E_HIP (no silent host route). */
    int32_t gpu_entropy;    /* 1: the MANIAC pixel loop of every compressed channel group runs on the GPU (csrc/maniac_encode.hip: context
                               model of all pixels in parallel, then one wavefront per group for symbol binarisation, chance updates and
                               the range coder, maniac/rac_enc.h:28-100); group headers, tree learning and the tree itself stay host code.
                               Same bytes as with 0.  No GPU: FUIFGPU_E_HIP. */
} fuifgpu_encode_options;
int fuifgpu_encode_image(const int32_t *planes, int w, int h, int nch, int bit_depth, const fuifgpu_encode_options *opt,
                         uint8_t **blob_out, size_t *size_out);
/* A batch of pictures of one size (planes[m]: nch planes of w*h samples): the host prepares every channel group of every picture
 * (header, learned tree, the coder's state behind it), then the MANIAC pixel loops of ALL groups run in one launch pair -- the
 * context model of every pixel in parallel, one wavefront per group for the range coder (csrc/maniac_encode.hip; the way
 * k_maniac_decode runs one wavefront per group of a batch) -- and the host assembles the streams.  blobs_out[m] / sizes_out[m]
 * are what fuifgpu_encode_image writes for picture m, byte for byte (gpu_entropy is implied; opt NULL = CLI defaults).
 * A picture in flight holds its channels on the host and on the device plus 12 bytes of coder scratch per sample (~0.5 GB per 4K
 * RGB picture): the caller sizes its batches to the device (fuifgpu_dev_mem_info) and chunks larger sets.
 * Replaces N runs of the reference's `fuif_encode_file` (encoding/encoding.cpp:727-735). */
int fuifgpu_encode_images(const int32_t *const *planes, int n_images, int w, int h, int nch, int bit_depth, const fuifgpu_encode_options *opt,
                          uint8_t **blobs_out, size_t *sizes_out);
/* channels already in a transform domain (e.g. the quantised DCT coefficient planes import/read_jpeg.h:56-184
 * builds): geometry + q + samples per channel, `transforms` = flat words {id, nparams, params...} of the
 * transforms that produced them; opt->squeeze adds the default Squeeze of the first nb_channels channels */
typedef struct {
    int32_t w, h, hshift, vshift, hcshift, vcshift, component, q;
    const int32_t *data;
} fuifgpu_raw_channel;
int fuifgpu_encode_channels(const fuifgpu_raw_channel *channels, int n_channels, int w, int h, int nb_channels, int bit_depth,
                            const int32_t *transforms, int n_transform_words, const fuifgpu_encode_options *opt,
                            uint8_t **blob_out, size_t *size_out);
void fuifgpu_free_blob(uint8_t *blob);

#ifdef __cplusplus
}
#endif
#endif /* FUIFGPU_H */
