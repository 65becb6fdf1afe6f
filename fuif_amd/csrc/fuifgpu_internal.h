// fuif_amd/csrc/fuifgpu_internal.h -- structures shared by the host planner, the C-ABI layer and
// the gfx950 kernels.  Everything here is plain-old-data so it can be passed to kernels by value.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace fuifgpu {

// transform ids (reference: transform/transform.h:29-70)
enum : int {
    TR_YCBCR = 0, TR_YCOCG = 1, TR_SUBSAMPLE = 3, TR_DCT = 4, TR_QUANTIZE = 5,
    TR_PALETTE = 6, TR_SQUEEZE = 7, TR_2DMATCH = 8, TR_PERMUTE = 9, TR_APPROXIMATE = 10
};

constexpr int kMaxBitDepth = 15;        // config.h:5 (MAX_BIT_DEPTH)
constexpr int kNonRefProps = 13;        // encoding/context_predict.h:210
constexpr int kMaxProps = 64;           // 2*6 reference properties + 13 local ones = 25 at default options; one property per lane of the wavefront
constexpr int kMaxRefs = 25;            // 2*25 + 13 = 63 properties: what the 64 lanes hold (-E 50).  Up to 9 references (31 properties, -E 18) a chunk's
                                        // property rows have 33 words and the chunk its full length; beyond, 65 words and half the pixels (round 4)
constexpr int kFastRefs = 9;            // references whose loads the vector phase issues together (unrolled); further ones go through a loop
constexpr int kMaxNodes = 65535;        // childID is uint16_t (maniac/compound.h:46)
constexpr int kLeafStride = 32;         // 31 chances (maniac/symbol.h:72-77) padded to 64 bytes
constexpr int kTreeStackDepth = 2048;   // explicit stack replacing the recursion of compound.h:277-308
constexpr int kPlaneAlign = 128;        // planes start on 256-byte boundaries inside a slab (coefficient slab: 128 int16 elements)
// A coded sample is a pixel_type = int16_t in the reference (image/image.h:35; check_bit_depth caps compressed samples at 15 bits
// of magnitude, encoding.cpp:61-72): the coefficient slab the entropy kernel writes holds int16 samples (round 4: half the slab,
// twice the images per launch for C4).  The inverse transforms compute in int32: they run on a widened copy of a chunk of images.
using coef_t = int16_t;

// status word per image (bit flags)
enum : int {
    ST_OK = 0,
    ST_TRUNCATED = 1,       // ran into EOF / the preview byte limit (not an error: encoding.cpp:209-219)
    ST_CORRUPT = 2,         // the reference would return false
    ST_UNSUPPORTED = 4,     // needs a feature outside SURVEY.md §8 (bit depth, tree size, ...)
    ST_STALLED = 8,         // internal: a tile gave up waiting for another tile's rows (never expected; reported with ST_CORRUPT)
};

// Geometry of one coded channel after all meta transforms (image/image.h:54-91 minus the data)
struct ChannelGeom {
    int32_t w, h;
    int32_t hshift, vshift, hcshift, vcshift;
    int32_t component;
    int32_t ctor_data; // 1: the reference's Channel owns w*h zero samples before decoding (Image constructor planes, palette /
                       // match meta-channels; image.h:64-65), so rows a truncated stream never reaches stay 0; 0: the plane is
                       // created by Channel::resize() at decode time, which fills with Channel::zero (image.h:73-75)
    int64_t coef_off;  // element offset of the plane inside one image's coefficient slab
};

// Per-image, per-coded-channel values that only the bitstream knows (written by the entropy kernel)
struct ChannelMeta {
    int32_t minval, maxval, q, decoded;  // decoded: 0 = untouched (reads as zeros), 1 = has data
};

// One compressed stream (= one image) of a batch
struct StreamJob {
    uint64_t blob_off;    // byte offset inside the batch's blob buffer (16-byte aligned)
    uint32_t blob_size;
    uint32_t data_start;  // first byte after the header = first channel group
    uint32_t limit;       // bytes_to_load for responsive decodes (0 = none), encoding.cpp:704-705
    uint32_t flags;       // bit 0: BlobReader EOF semantics (fileio.h:100-102) instead of FileIO/feof
};

// One unit of entropy-decoding work: a run of consecutive channel groups of one image whose first
// byte is known.  Without a group index (index.cpp) that is the whole stream; with one, every
// group is its own tile and gets its own wavefront.
struct Tile {
    uint32_t image;
    uint32_t start;                       // byte offset of the first group header of the tile
    int32_t first_channel, last_channel;  // coded channels [first,last] are decoded (or zero-filled) by this tile
    uint32_t end;                         // where the index says the NEXT tile starts (0 = last tile / unknown): a tile that is
                                          // decoded in full must stop exactly there, else the index does not belong to the stream
    uint32_t flags;                       // bit 0: every tile of this image is one single-channel group, so every tile that can wait for
                                          // another tile's rows can be SUSPENDED (context scheduler).  Otherwise none of the image's
                                          // tiles is: a suspended tile needs a free wavefront to go on, and tiles that wait by
                                          // spinning could hold all of them (round 1's invariant -- a taken tile runs -- per image)
};
constexpr uint32_t kTileSuspendable = 1u;
constexpr uint32_t kTileSizeClassShift = 4;   // bits 4..7: floor(log2(image samples / tile samples)), 15 = empty tile
constexpr int kDefaultPrioBase = 2;    // tiles holding >= 1/8 of an image: priority 3, >= 1/16: 2, >= 1/32: 1 (measured: -3 % launch time)

// --- inverse-transform schedule ---------------------------------------------------------------
enum : int { BUF_COEF = 0, BUF_OUT = 1, BUF_TMP = 2,
              // only in the source list of an OP_IDCT: a coded plane as the entropy kernel stored it (int16 samples in the coefficient slab, same offset as
              // BUF_COEF) whose dequantisation is folded into the load -- sample * ChannelMeta::q of `qsrc` (planner peephole fuse_dequant_into_idct)
              BUF_COEF16Q = 3 };
struct PlaneRef {
    int32_t buf;
    int32_t w, h;
    int32_t qsrc;     // coded channel whose ChannelMeta::q this plane carries (Channel::q)
    int64_t off;
};
enum : int {
    OP_HSQUEEZE = 1, OP_VSQUEEZE = 2, OP_YCOCG = 3, OP_YCBCR = 4, OP_QUANT = 5, OP_IDCT = 6,
    OP_UPSAMPLE = 7, OP_COPY_CLAMP = 8, OP_CLAMP = 9,
    OP_PALETTE = 10,   // dst = palette[p0][clamp(index)]: src[0] index plane, src[1] palette plane (p1 colours wide)
    OP_APPROX = 11,    // src[0] = src[0]*p0 + src[1] in place when the remainder src[1] was decoded (p1: it has constructor data)
    OP_MATCH = 12,     // 2D match against previous frames, in place on the listed planes: src[0] match plane, p0 softmatch, p1 frame height
    // 2D match with free offsets (2dmatch.h:136-146), exact matches only: source map by pointer jumping
    OP_MATCH_INIT = 13,   // dst[0] = linear index every sample copies from (itself / -1 = before the first sample); src[0] match plane, p0 softmatch
    OP_MATCH_JUMP = 14,   // dst[0][p] = src[0][src[0][p]]: one doubling step; src[1] match plane (mode check)
    OP_MATCH_APPLY = 15,  // listed planes[p] = planes[src[0][p]] in place; src[1] match plane (mode check)
    // soft matches (p0 = 1 on all three): the list is [planes (n), accumulators (n), accumulators (n)], pad = 3n; INIT fills the first copy of the
    // accumulators, JUMP reads copy p1 and writes the other, APPLY adds copy p1 to the root's sample
    OP_PERMUTE = 16,      // dst[0] = listed plane number perm[p0], perm = the samples of the 1-row meta plane src[0] (p1 = its length): transform/permute.h:31-54
    // the last three ops of a default YCoCg + Squeeze chain in one pass (planner peephole, plan.cpp finalize()): the horizontal
    // unsqueeze of Co (src[0] avg, src[1] residual) and of Cg (src[2] avg, ext[0] residual) followed by the inverse YCoCg with
    // the finished Y plane dst[0]; R, G, B go to dst[0], dst[1], dst[2] (squeeze.h:81-132 twice + ycocg.h:49-61)
    OP_HSQ2_YCOCG = 17,
    // the last three ops of a JPEG-transcoded 4:2:0 chain in one pass (planner peephole fuse_upsample_ycbcr): the 2x2 "fancy" upsampling of Cb (src[1]) and
    // Cr (src[2]) (subsample.h:90-115), the inverse YCbCr with the finished Y plane dst[0] over its p0 x p1 samples (ycbcr.h:49-60) and the final clamp of
    // the chroma planes' samples outside that region (image.cpp:107-113); R, G, B go to dst[0], dst[1], dst[2]
    OP_UPS2_YCBCR = 18
};
struct Op {
    int32_t kind;
    int32_t clamp_out;         // fused final clamp (image/image.cpp:107-113) on the stores
    int32_t lo, hi;            // clamp bounds / image minval,maxval
    int32_t p0, p1;            // op specific (upsample: srh,srv)
    PlaneRef src[3];
    PlaneRef dst[3];
    int32_t idct_first;        // OP_IDCT: index into Plan::idct_src of the 64 source planes
    int32_t pad;
    PlaneRef ext[1];           // OP_HSQ2_YCOCG: a fourth source
    int32_t r16;               // squeeze family: the residual plane(s) are coded planes nobody has rewritten -- the kernel reads them as int16 samples
                               // straight from the coefficient slab (no widened copy of them is made)
    int32_t pad2;              // OP_IDCT: 1 = the AC source planes (entries 1..63 of the list) are all BUF_COEF16Q -- the kernel instantiation with int16 AC loads
};

struct TransformDesc {
    int id;
    std::vector<int> params;
};

struct OutputChannel {
    PlaneRef plane;   // always BUF_OUT
    int32_t hshift, vshift, hcshift, vcshift, component;
};

struct Plan {
    // header (encoding/encoding.cpp:599-657)
    int w = 0, h = 0, bit_depth = 0, maxval = 0, minval = 0;
    int nb_channels = 0, colormodel = 0, nb_frames = 1, max_properties = 0;
    int responsive_offsets[5] = {0, 0, 0, 0, 0};
    size_t data_start = 0;
    std::vector<TransformDesc> transforms;      // with default parameters expanded like meta_apply does
    std::vector<ChannelGeom> coded;             // channel table the entropy stage fills
    int64_t coef_elems = 0;                     // per-image coefficient slab (int32 elements)
    // inverse schedule (image/image.cpp:94-115)
    std::vector<Op> ops;
    std::vector<PlaneRef> idct_src;
    std::vector<int64_t> widen;                 // {element offset, elements} pairs: the coded planes some inverse kernel reads (or rewrites) as int32 --
                                                // they are widened into the int32 copy before the schedule runs; squeeze residuals are not (Op::r16)
    std::vector<OutputChannel> outputs;
    int64_t out_elems = 0, tmp_elems = 0;
    uint64_t signature = 0;                     // equal signature <=> same geometry & schedule
    int error = 0;                              // FUIFGPU_E_* (0 = ok)
    std::string message;
};

// One channel group of a stream: where its header starts and the first channel it codes (index.cpp)
struct GroupEntry {
    uint32_t start;
    int32_t first_channel;
};
void build_index_trailer(const std::vector<GroupEntry> &groups, std::vector<uint8_t> &out);
bool parse_index_trailer(const uint8_t *blob, size_t n, size_t data_start, int nch, std::vector<GroupEntry> &groups, size_t *stream_end);

// host planner (plan.cpp)
int parse_and_plan(const uint8_t *blob, size_t n, Plan &plan);
void build_chance_table(uint16_t *table8192, uint32_t alpha, int cut);

}  // namespace fuifgpu
