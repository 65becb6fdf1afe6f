// fuif_amd/boundary/fuif_gpu_boundary.cpp -- the reference-side binding of libfuifgpu.so.
//
// This is the ONE file a maintainer of cloudinary/fuif adds to route the decode hot path to the
// MI355X.  It is compiled against the reference's OWN headers (Image, Channel,
// Transform, fuif_options, FileIO, BlobReader -- included from the reference tree, nothing copied)
// and defines the three entry points the CLI uses for decoding:
//
//     bool fuif_decode_file(const char*, Image&, fuif_options)      encoding/encoding.cpp:745-753
//     template <IO> bool fuif_decode(IO&, Image&, fuif_options)     encoding/encoding.cpp:599-720
//     void Image::undo_transforms(int keep)                         image/image.cpp:94-115
//
// The reference's own definitions of these three stay in the link under the names
// fuif_decode_file_cpu / fuif_decode_cpu / Image::undo_transforms_cpu (the Makefile compiles
// encoding.cpp and image.cpp with -Dname=name_cpu: encoder and decoder are one translation unit), but THIS FILE DOES NOT CALL THE
// REFERENCE'S DECODER (round 6): -i/--identify is answered from the header bytes here (identify_header below), and a stream the library
// reports as FUIFGPU_E_UNSUPPORTED / FUIFGPU_ST_UNSUPPORTED (a data-driven Permute over channels of different geometry; a 2D
// match with a forward reference; more than 50 reference properties) is a loud error -- a planner regression or a failed launch in
// a deployment can never silently produce the output of the code this path replaces.  Only the ten-line LOOP of
// undo_transforms(keep != 0) (image.cpp:94-115) is the reference's: its Transform::apply calls come back here.
// A maintainer who wants the reference's decoder as a route for out-of-scope streams builds this one file with
// -DFUIFGPU_WITH_CPU_FALLBACK (`make WITH_CPU_FALLBACK=1`: binaries under _build_with_cpu_fallback/, never the shipped _build/);
// that build honours FUIFGPU_ALLOW_CPU_FALLBACK=1 and FUIFGPU_CPU_TRANSFORMS=1 at run time.  Stills and
// animations (FUAF), Squeeze / YCoCg / YCbCr / DCT / Quantize / Subsample / Palette / Approximate / 2D-match / Permute
// chains all decode on the GPU; FUIFGPU_VERBOSE=1 reports on stderr which path decoded.  Everything else (fuif.cpp, import/export code, the encoder) is
// compiled and linked UNCHANGED; fuif_encode_file is bound as well, only to append the group index to the file the
// reference's encoder wrote when FUIFGPU_WRITE_INDEX=1 asks for it (off by default).
//
// Ownership: fuif_decode() fills `image` exactly like the reference (channel vector = coded
// channels with geometry, ranges, q and samples narrowed to pixel_type; transform list with
// expanded parameters) and keeps the device batch alive in a registry keyed by &image so that the
// following image.undo_transforms() can run the inverse-transform schedule on the coefficients
// that are still resident in HBM.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <memory>
#include <utility>
#include <vector>

#include "encoding/encoding.h"
#include "fileio.h"
#include "image/image.h"
#include "io.h"
#include "transform/transform.h"

#include "fuifgpu.h"

// The reference's DCT helpers (scan script, zig-zag table, default parameters) are definitions inside a header that
// transform.cpp already instantiates; a second copy in a namespace of its own gives this file the reference's OWN
// tables without a duplicate symbol (every header dct.h includes has been included above, so only its body lands here).
namespace refdct {
#include "transform/dct.h"
#include "transform/subsample.h"
}

#ifdef FUIFGPU_WITH_CPU_FALLBACK
// the reference's CPU implementations, renamed at compile time (see Makefile): declared -- and callable -- in the opt-in build only
bool fuif_decode_file_cpu(const char *filename, Image &image, fuif_options options);
template <typename IO> bool fuif_decode_cpu(IO &io, Image &image, fuif_options options);
#define FUIFGPU_OUTSIDE_HINT "set FUIFGPU_ALLOW_CPU_FALLBACK=1 to decode it with the reference's CPU code"
#else
#define FUIFGPU_OUTSIDE_HINT "this binding was built without the reference's CPU decoder (-DFUIFGPU_WITH_CPU_FALLBACK)"
#endif

namespace {

struct Resident {
    fuifgpu_plan *plan = nullptr;
    fuifgpu_batch *batch = nullptr;
    std::vector<uint8_t> bytes;          // the stream, for the CPU route if the inverse chain turns out to be unsupported
    int preview = -1;                    // fuif_options::preview of the decode (fuif_options itself is not assignable: it holds an Image)
    size_t n_channels = 0;               // identity of the Image this batch belongs to (the key is only an address)
    const pixel_type *first_plane = nullptr;
    ~Resident() {
        if (batch) fuifgpu_batch_destroy(batch);
        if (plan) fuifgpu_plan_destroy(plan);
    }
};
// Never destroyed: static destructors run after the HIP runtime may be gone (hipFree / hipEventDestroy at exit);
// the process is ending anyway.
std::map<const Image *, std::unique_ptr<Resident>> &registry() {
    static auto *r = new std::map<const Image *, std::unique_ptr<Resident>>();
    return *r;
}
bool env_flag(const char *name) { const char *e = getenv(name); return e && *e && strcmp(e, "0") != 0; }
// A stream or transform the GPU path does not take is a loud error.  Only a binding compiled with -DFUIFGPU_WITH_CPU_FALLBACK holds call
// sites of the reference's decoder at all, and even there they are opt-in at run time (FUIFGPU_ALLOW_CPU_FALLBACK=1).
#ifdef FUIFGPU_WITH_CPU_FALLBACK
bool cpu_fallback_allowed() { return env_flag("FUIFGPU_ALLOW_CPU_FALLBACK"); }
#else
constexpr bool cpu_fallback_allowed() { return false; }
#endif
// den / num / loops of an animation header (encoding.cpp:611-622): the four varints after the magic, then these
struct Cursor {
    const std::vector<uint8_t> &b; size_t pos;
    int varint() { int r = 0; for (int k = 0; k < 10 && pos < b.size(); k++) { int c = b[pos++]; if (c < 128) return r + c; r = (r + c - 128) << 7; } return -1; }
};

template <typename IO> std::vector<uint8_t> slurp(IO &io) {
    std::vector<uint8_t> b;
    for (int c = io.get_c(); c != io.EOS; c = io.get_c()) b.push_back((uint8_t)c);
    return b;
}

bool gpu_decode_bytes(const std::vector<uint8_t> &bytes, Image &image, const fuif_options &options, bool *unsupported) {
    *unsupported = false;
    auto res = std::make_unique<Resident>();
    int rc = fuifgpu_plan_create(bytes.data(), bytes.size(), &res->plan);
    if (rc == FUIFGPU_E_UNSUPPORTED) { *unsupported = true; return false; }
    if (rc == FUIFGPU_E_NOT_FUIF) { e_printf("%s is not a FUIF file\n", "input"); return false; }
    if (rc != FUIFGPU_OK) { e_printf("Corrupt file. Aborting. (%s)\n", fuifgpu_last_error()); return false; }
    fuifgpu_image_info info;
    fuifgpu_plan_info(res->plan, &info);
    rc = fuifgpu_batch_create(res->plan, 1, bytes.size(), nullptr, nullptr, 1, &res->batch);
    if (rc != FUIFGPU_OK) { e_printf("fuifgpu: %s (%s)\n", fuifgpu_strerror(rc), fuifgpu_last_error()); return false; }
    const uint8_t *blobs[1] = {bytes.data()};
    const size_t sizes[1] = {bytes.size()};
    if ((rc = fuifgpu_batch_upload(res->batch, blobs, sizes, 1, options.preview, nullptr)) != FUIFGPU_OK ||
        (rc = fuifgpu_batch_decode(res->batch, nullptr)) != FUIFGPU_OK || (rc = fuifgpu_batch_sync(res->batch, nullptr)) != FUIFGPU_OK) {
        e_printf("fuifgpu: %s (%s)\n", fuifgpu_strerror(rc), fuifgpu_last_error());
        return false;
    }
    int32_t status = 0;
    uint32_t used = 0;
    fuifgpu_batch_status(res->batch, &status, &used);
    if (status & FUIFGPU_ST_UNSUPPORTED) { *unsupported = true; return false; }
    if (status & FUIFGPU_ST_CORRUPT) { e_printf("Corruption detected.\n"); return false; }

    // Image(w,h,maxval,nb_channels,colormodel) + what meta_apply and the channel loop leave behind
    image = Image(info.w, info.h, info.maxval, info.nb_channels, info.colormodel);
    if (info.nb_frames > 1 && bytes.size() > 4) {   // FUAF: filmstrip geometry + timing (encoding.cpp:611-622,641-645)
        Cursor c{bytes, 4};
        for (int k = 0; k < 4; k++) c.varint();
        image.nb_frames = c.varint() + 2;
        image.den = c.varint() + 1;
        image.num.clear();
        const int numerator = c.varint();
        if (numerator > 0) { image.num.push_back(numerator); for (int i = 1; i < image.nb_frames; i++) image.num.push_back(c.varint()); }
        image.loops = c.varint();
    }
    std::vector<int32_t> slab((size_t)(info.coef_elems > 0 ? info.coef_elems : 1));
    fuifgpu_batch_download_coef(res->batch, 0, slab.data(), nullptr);
    std::vector<int32_t> meta((size_t)info.nb_coded_channels * 4);
    fuifgpu_batch_channel_meta(res->batch, 0, meta.data());
    image.channel.clear();
    for (int c = 0; c < info.nb_coded_channels; c++) {
        fuifgpu_channel_desc d;
        fuifgpu_plan_coded_channel(res->plan, c, &d);
        Channel ch;
        ch.w = d.w; ch.h = d.h; ch.hshift = d.hshift; ch.vshift = d.vshift; ch.hcshift = d.hcshift; ch.vcshift = d.vcshift;
        ch.component = d.component;
        const int32_t *m = &meta[(size_t)c * 4];
        const bool parsed = (m[0] != 0 || m[1] != 0 || m[2] != 0);
        ch.minval = (pixel_type)m[0]; ch.maxval = (pixel_type)m[1]; ch.q = parsed ? m[2] : 1;
        if (!parsed && c < info.nb_channels) { ch.minval = 0; ch.maxval = (pixel_type)info.maxval; }  // Image ctor range of a never-reached base channel
        ch.setzero();
        if (m[3]) {  // plane was decoded (or constant / zero-filled by the truncation rules)
            const size_t n = (size_t)d.w * d.h;
            ch.data.resize(n);
            for (size_t i = 0; i < n; i++) ch.data[i] = (pixel_type)slab[(size_t)d.offset + i];
        }
        image.channel.push_back(ch);
    }
    image.transform.clear();
    for (int t = 0; t < info.nb_transforms; t++) {
        int32_t id = 0, np = 0;
        std::vector<int32_t> params(4096);
        fuifgpu_plan_transform(res->plan, t, &id, params.data(), (int)params.size(), &np);
        Transform tr(id);
        for (int k = 0; k < np && k < (int)params.size(); k++) tr.parameters.push_back(params[k]);
        image.transform.push_back(tr);
    }
    // Image::nb_meta_channels / nb_channels as the meta steps leave them (transform/palette.h:87-88, 2dmatch.h:191, permute.h:60)
    for (const Transform &tr : image.transform) {
        if (tr.ID == TRANSFORM_PALETTE && tr.parameters.size() == 3) {
            image.nb_meta_channels++;
            image.nb_channels -= tr.parameters[1] - tr.parameters[0];
        }
        if (tr.ID == TRANSFORM_2DMATCH) image.nb_meta_channels++;
        if (tr.ID == TRANSFORM_PERMUTE && tr.parameters.empty()) image.nb_meta_channels++;   // the permutation is a meta-channel (permute.h:58-63)
    }
    image.error = false;
    res->bytes = bytes;
    res->preview = options.preview;
    res->n_channels = image.channel.size();
    res->first_plane = image.channel.empty() ? nullptr : image.channel[0].data.data();
    if (env_flag("FUIFGPU_VERBOSE")) fprintf(stderr, "fuifgpu: %d coded channels entropy-decoded on the GPU\n", info.nb_coded_channels);
    registry()[&image] = std::move(res);
    return true;
}


// -i / --identify: the header lines the reference prints (encoding.cpp:601-702 with options.identify), from the header bytes alone --
// magic, the basic-info varints, the responsive offsets, the transform list; no Image is touched, no channel data is read and nothing
// of the reference's decoder runs.  Same text at every verbosity level (v_printf is the reference's, io.cpp:58-65).
template <typename IO> bool identify_header(IO &io) {
    char buff[5];
    if (!io.gets(buff, 5)) { e_printf("Could not read header from file: %s\n", io.getName()); return false; }
    const bool multi_frame = !strcmp(buff, "FUAF");
    if (!multi_frame && strcmp(buff, "FUIF")) { e_printf("%s is not a FUIF file\n", io.getName()); return false; }
    auto varint = [&io]() { int r = 0; for (int k = 0; k < 10; k++) { const int c = io.get_c(); if (c < 0) break; if (c < 128) return r + c; r = (r + c - 128) << 7; } return -1; };   // encoding.cpp:45-59
    const int nb_channels = varint() - '0', bit_depth = varint() - '&';
    const int w = varint() + 1, h = varint() + 1;
    int nb_frames = 1;
    if (multi_frame) {
        nb_frames = varint() + 2;
        varint();                                                     // den
        if (varint()) for (int i = 1; i < nb_frames; i++) varint();   // num
        varint();                                                     // loops
    }
    const int colormodel = varint();
    v_printf(1, "%s: %i-channel, %i-bit, ", io.getName(), nb_channels, bit_depth);
    if (multi_frame) v_printf(1, "%ix%i %s%s animation (%i frames)\n", w, h / nb_frames, colormodel_name(colormodel, nb_channels), colorprofile_name(colormodel), nb_frames);
    else v_printf(1, "%ix%i %s%s image\n", w, h, colormodel_name(colormodel, nb_channels), colorprofile_name(colormodel));
    const int max_properties = varint();
    v_printf(4, "Global option: up to %i back-referencing MANIAC properties.\n", max_properties);
    v_printf(7, "First part of header decoded (basic info). Read %i bytes so far.\n", io.ftell());
    if (nb_channels < 1) return true;
    static const int sizes[5] = {0, 16, 8, 4, 2};   // responsive_sizes, encoding/encoding.h (LQIP, 1/16 .. 1/2)
    int offsets[5], relative = 0;
    for (int s = 0; s < 5; s++) { offsets[s] = varint() * TRUNCATION_OFFSET_RESOLUTION + relative; relative = offsets[s]; }
    relative = io.ftell();
    for (int s = 0; s < 5; s++) {
        offsets[s] += relative;
        if (s) v_printf(3, "Responsive truncation point for size 1/%i at position %i\n", sizes[s], offsets[s]);
        else v_printf(3, "Responsive truncation point for LQIP at position %i\n", offsets[s]);
    }
    v_printf(7, "Second part of header decoded (responsive offsets and global parameters; before transforms). Read %i bytes so far.\n", io.ftell());
    const int nb_transforms = varint();
    v_printf(2, "Image data underwent %i transformations: ", nb_transforms);
    for (int i = 0; i < nb_transforms; i++) {
        const int id_and_nb_params = varint();
        if (id_and_nb_params < 0) break;
        Transform t(id_and_nb_params & 0xf);
        if (t.has_parameters()) for (int j = 0, n = id_and_nb_params >> 4; j < n; j++) t.parameters.push_back(varint());
        if (i) v_printf(2, ", ");
        v_printf(2, "%s", t.name());
        if (t.ID == TRANSFORM_PALETTE && t.parameters.size() >= 3) {
            if (t.parameters[0] == t.parameters[1]) v_printf(3, "[Compact channel %i to ", t.parameters[0]);
            else v_printf(3, "[channels %i-%i with ", t.parameters[0], t.parameters[1]);
            v_printf(3, "%i colors]", t.parameters[2]);
        }
    }
    v_printf(2, "\n");
    v_printf(6, "Header decoded. Read %i bytes so far.\n", io.ftell());
    return true;
}

}  // namespace

template <typename IO> bool fuif_decode(IO &io, Image &image, fuif_options options) {
    registry().erase(&image);   // whatever was decoded into an Image at this address before is gone now
    if (options.identify) return identify_header(io);
    std::vector<uint8_t> bytes = slurp(io);
    bool unsupported = false;
    if (gpu_decode_bytes(bytes, image, options, &unsupported)) return true;
    if (!unsupported) return false;
    if (!cpu_fallback_allowed()) {
        e_printf("fuifgpu: this stream needs a feature outside the GPU path (%s); " FUIFGPU_OUTSIDE_HINT "\n", fuifgpu_last_error());
        return false;
    }
#ifdef FUIFGPU_WITH_CPU_FALLBACK
    if (env_flag("FUIFGPU_VERBOSE")) fprintf(stderr, "fuifgpu: outside the GPU path, decoding with the reference's CPU code\n");
    BlobReader again(bytes.data(), bytes.size());  // outside the GPU scope: the reference's own decoder
    return fuif_decode_cpu(again, image, options);
#else
    return false;
#endif
}
template bool fuif_decode(FileIO &io, Image &image, fuif_options options);
template bool fuif_decode(BlobReader &io, Image &image, fuif_options options);

bool fuif_decode_file(const char *filename, Image &image, fuif_options options) {
    FILE *file = !strcmp(filename, "-") ? stdin : fopen(filename, "rb");
    if (!file) return false;
    FileIO fio(file, (file == stdin ? "from standard input" : filename));
    return fuif_decode(fio, image, options);
}

// fuif_encode_file (encoding/encoding.cpp:727-735, called by fuif.cpp:615).  The reference's encoder does all the work under the name
// fuif_encode_file_cpu (the Makefile's rename); with FUIFGPU_WRITE_INDEX=1 the file it wrote is decoded once on the GPU (entropy decode
// only) and gets the group index trailer: byte for byte the reference's stream, then the trailer its decoder never reads --
// so that `fuif in.ppm out.fuif` produces files whose channel groups decode in parallel from then on.  Off by default: the CLI's output stays the
// reference's file exactly; a file the GPU path cannot index (out of scope, no device) is left as it is, with a note on stderr.
bool fuif_encode_file_cpu(const char *filename, const Image &image, fuif_options &options);
bool fuif_encode_file(const char *filename, const Image &image, fuif_options &options) {
    if (!fuif_encode_file_cpu(filename, image, options)) return false;
    if (!env_flag("FUIFGPU_WRITE_INDEX") || !strcmp(filename, "-")) return true;
    std::vector<uint8_t> bytes;
    {
        FILE *f = fopen(filename, "rb");
        if (!f) return true;
        FileIO io(f, filename);
        bytes = slurp(io);
    }
    fuifgpu_plan *plan = nullptr;
    fuifgpu_batch *batch = nullptr;
    uint8_t *indexed = nullptr;
    size_t indexed_size = 0;
    int ng = 0;
    int rc = fuifgpu_plan_create(bytes.data(), bytes.size(), &plan);
    if (rc == FUIFGPU_OK) rc = fuifgpu_batch_create_streaming(plan, 1, bytes.size() + 4096, 1, &batch);
    const uint8_t *ptr = bytes.data();
    const size_t size = bytes.size();
    if (rc == FUIFGPU_OK) rc = fuifgpu_batch_upload(batch, &ptr, &size, 1, -1, nullptr);
    if (rc == FUIFGPU_OK) rc = fuifgpu_batch_decode(batch, nullptr);
    if (rc == FUIFGPU_OK) rc = fuifgpu_batch_sync(batch, nullptr);
    int32_t status = 0;
    if (rc == FUIFGPU_OK) { fuifgpu_batch_status(batch, &status, nullptr); if (status) rc = FUIFGPU_E_CORRUPT; }
    if (rc == FUIFGPU_OK) {
        fuifgpu_image_info info;
        fuifgpu_plan_info(plan, &info);
        std::vector<int32_t> first((size_t)info.nb_coded_channels + 1);
        std::vector<uint32_t> start((size_t)info.nb_coded_channels + 1);
        rc = fuifgpu_batch_group_index(batch, 0, first.data(), start.data(), (int)first.size(), &ng);
        if (rc == FUIFGPU_OK) rc = fuifgpu_index_append(bytes.data(), bytes.size(), first.data(), start.data(), ng, &indexed, &indexed_size);
    }
    if (rc == FUIFGPU_OK) {
        // the indexed stream = the file's own bytes + the trailer: only the trailer is APPENDED, so a short write (disk full) can never damage
        // the valid file the reference encoder has just written -- at worst it ends in an incomplete trailer, which the reader ignores
        bool appended = indexed_size >= bytes.size() && !memcmp(indexed, bytes.data(), bytes.size());
        if (appended) {
            FILE *f = fopen(filename, "ab");
            appended = f != nullptr;
            if (f) {
                const size_t extra = indexed_size - bytes.size();
                appended = fwrite(indexed + bytes.size(), 1, extra, f) == extra;
                appended = (fclose(f) == 0) && appended;
            }
        }
        if (!appended) fprintf(stderr, "fuifgpu: %s: the group index could not be appended (write failed); the file is as the encoder wrote it\n", filename);
        else if (env_flag("FUIFGPU_VERBOSE")) fprintf(stderr, "fuifgpu: %s: group index of %d groups appended (%zu + %zu bytes)\n", filename, ng, bytes.size(), indexed_size - bytes.size());
    } else {
        fprintf(stderr, "fuifgpu: %s written without group index (%s)\n", filename, status ? "the decode was flagged" : fuifgpu_last_error());
    }
    fuifgpu_free_blob(indexed);
    if (batch) fuifgpu_batch_destroy(batch);
    if (plan) fuifgpu_plan_destroy(plan);
    return true;
}

void fuifgpu_boundary_undo_transforms(Image *self, int keep) __asm__("_ZN5Image15undo_transformsEi");

// ---- the batch entry (fuifgpu_boundary.h) ----------------------------------------------------------------------------
#include "fuifgpu_boundary.h"
namespace {
// what Image::undo_transforms() leaves behind, from the output planes of image `index` of a batch (the single-image path below
// builds the same thing)
void image_from_outputs(Image &img, fuifgpu_plan *plan, fuifgpu_batch *batch, int index, const fuifgpu_image_info &info) {
    img = Image(info.w, info.h, info.maxval, info.nb_channels, info.colormodel);
    std::vector<int32_t> slab((size_t)(info.out_elems > 0 ? info.out_elems : 1));
    fuifgpu_batch_download_out(batch, index, slab.data(), nullptr);
    std::vector<Channel> outch;
    for (int c = 0; c < info.nb_output_channels; c++) {
        fuifgpu_channel_desc d;
        fuifgpu_plan_output_channel(plan, c, &d);
        Channel ch(d.w, d.h, (pixel_type)img.minval, (pixel_type)img.maxval, 1, d.hshift, d.vshift, d.hcshift, d.vcshift);
        ch.component = d.component;
        const size_t n = (size_t)d.w * d.h;
        for (size_t i = 0; i < n; i++) ch.data[i] = (pixel_type)slab[(size_t)d.offset + i];
        outch.push_back(ch);
    }
    img.channel = outch;
    img.nb_meta_channels = 0;
    img.nb_channels = (int)outch.size();
    img.transform.clear();
    img.error = false;
}
#ifdef FUIFGPU_WITH_CPU_FALLBACK
bool cpu_decode_whole(const std::vector<uint8_t> &bytes, Image &img, const fuif_options &options) {
    BlobReader io(bytes.data(), bytes.size());
    if (!fuif_decode_cpu(io, img, options)) return false;
    img.undo_transforms(0);   // (macro-renamed: the reference's CPU implementation)
    return !img.error;
}
#endif
}  // namespace

namespace {
// the files `idx` (one geometry + transform chain: one plan) on the calling thread's current device: one batch object, chunk by chunk
// `sharers` = host threads of this call that work on the same device (a device list may name one twice): each sizes its chunks for its share of the memory
void decode_group_here(fuifgpu_plan *plan, const std::vector<int> &idx, const std::vector<std::vector<uint8_t>> &bytes, const char *const *filenames,
                       Image *images, const fuif_options &options, std::vector<char> &ok, std::vector<char> &cpu_route, bool verbose, int sharers) {
    fuifgpu_image_info info;
    fuifgpu_plan_info(plan, &info);
    int device = 0;
    fuifgpu_get_device(&device);
    // How many of the group's files fit on the device at once: per picture its coefficient and output slabs (int32), its
    // stream and a context arena; a quarter of the free memory (at most 40 GiB) stays for the decoder scratch and the
    // transform arena.  A larger group goes through ONE batch object chunk by chunk (FUIFGPU_BOUNDARY_CHUNK: tests).
    const int n = (int)idx.size();
    size_t max_stream = 0;
    for (int i : idx) max_stream = std::max(max_stream, bytes[i].size());
    int chunk = n;
    {
        size_t free_b = 0, total_b = 0;
        if (fuifgpu_dev_mem_info(&free_b, &total_b) == FUIFGPU_OK && free_b) {
            const size_t reserve = std::min<size_t>(free_b / 4, (size_t)40 << 30);
            const size_t per_image = 2 * (size_t)info.coef_elems + 4 * (size_t)info.out_elems + max_stream + ((size_t)16 << 20);   // int16 coefficients, int32 outputs
            chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, (free_b - reserve) / (size_t)std::max(1, sharers) / std::max<size_t>(per_image, 1)));
        }
        if (const char *e = getenv("FUIFGPU_BOUNDARY_CHUNK")) chunk = std::max(1, std::min(n, atoi(e)));
    }
    size_t cap = 0;
    for (int c0 = 0; c0 < n; c0 += chunk) {
        size_t t = 0;
        for (int k = c0; k < std::min(n, c0 + chunk); k++) t += bytes[idx[k]].size();
        cap = std::max(cap, t);
    }
    fuifgpu_batch *batch = nullptr;
    int rc = fuifgpu_batch_create(plan, chunk, cap, nullptr, nullptr, 0, &batch);
    int n_chunks = 0;
    for (int c0 = 0; c0 < n && rc == FUIFGPU_OK; c0 += chunk, n_chunks++) {
        const int cnt = std::min(chunk, n - c0);
        std::vector<const uint8_t *> ptr;
        std::vector<size_t> len;
        for (int k = c0; k < c0 + cnt; k++) { ptr.push_back(bytes[idx[k]].data()); len.push_back(bytes[idx[k]].size()); }
        rc = fuifgpu_batch_upload(batch, ptr.data(), len.data(), cnt, options.preview, nullptr);
        if (rc == FUIFGPU_OK) rc = fuifgpu_batch_decode(batch, nullptr);
        if (rc == FUIFGPU_OK) rc = fuifgpu_batch_undo_transforms(batch, nullptr);
        if (rc == FUIFGPU_OK) rc = fuifgpu_batch_sync(batch, nullptr);
        if (rc != FUIFGPU_OK) break;
        std::vector<int32_t> status((size_t)cnt, 0);
        fuifgpu_batch_status(batch, status.data(), nullptr);
        for (int k = 0; k < cnt; k++) {
            const int i = idx[c0 + k];
            if (status[k] & FUIFGPU_ST_UNSUPPORTED) { cpu_route[i] = 1; continue; }
            if (status[k] & FUIFGPU_ST_CORRUPT) { e_printf("%s: corruption detected.\n", filenames[i]); continue; }
            image_from_outputs(images[i], plan, batch, k, info);
            ok[i] = 1;
        }
    }
    if (rc != FUIFGPU_OK) {
        e_printf("fuifgpu: %s (%s)\n", fuifgpu_strerror(rc), fuifgpu_last_error());
        if (batch) fuifgpu_batch_destroy(batch);
        return;
    }
    if (verbose) {
        if (n_chunks <= 1) fprintf(stderr, "fuifgpu: %d file(s) of %dx%d decoded in one batch on the GPU (device %d)\n", n, info.w, info.h, device);
        else fprintf(stderr, "fuifgpu: %d file(s) of %dx%d decoded in %d batches of up to %d on the GPU (device %d)\n", n, info.w, info.h, n_chunks, chunk, device);
    }
    fuifgpu_batch_destroy(batch);
}

// FUIFGPU_DEVICES = "all" | "0,2,3": the GPUs fuif_decode_files() spreads a list of files over (unset: the calling thread's current device).
// A list that does not parse, or a device count that cannot be read, is an ERROR (false): a typo must not quietly leave a GPU out.
bool devices_from_env(std::vector<int> &out) {
    out.clear();
    const char *e = getenv("FUIFGPU_DEVICES");
    if (!e || !*e) return true;
    int n = 0;
    if (fuifgpu_device_count(&n) != FUIFGPU_OK) { e_printf("fuifgpu: FUIFGPU_DEVICES is set but the device count cannot be read (%s)\n", fuifgpu_last_error()); return false; }
    if (!strcmp(e, "all")) { for (int d = 0; d < n; d++) out.push_back(d); return true; }
    for (const char *p = e; *p;) {
        char *end = nullptr;
        const long d = strtol(p, &end, 10);
        if (end == p || (*end != ',' && *end != 0) || (*end == ',' && end[1] == 0)) {
            e_printf("fuifgpu: FUIFGPU_DEVICES=\"%s\" is not \"all\" or a comma-separated list of GPU numbers\n", e);
            out.clear();
            return false;
        }
        out.push_back((int)d);
        p = *end == ',' ? end + 1 : end;
    }
    return true;
}
}  // namespace

int fuif_decode_files(const char *const *filenames, int n_files, Image *images, fuif_options options, bool *ok_out) {
    std::vector<int> dev;
    if (!devices_from_env(dev)) { if (ok_out) for (int i = 0; i < n_files; i++) ok_out[i] = false; return 0; }
    return fuif_decode_files_on(filenames, n_files, images, options, dev.empty() ? nullptr : dev.data(), (int)dev.size(), ok_out);
}

int fuif_decode_files_on(const char *const *filenames, int n_files, Image *images, fuif_options options, const int *devices, int n_devices, bool *ok_out) {
    if (!filenames || !images || n_files < 1 || n_devices < 0 || (n_devices > 0 && !devices)) return 0;
    const bool verbose = env_flag("FUIFGPU_VERBOSE"), no_cpu = !cpu_fallback_allowed();
    {   // a device list is checked before anything is decoded: a typo must not quietly leave a GPU out
        int n_dev = 0;
        if (n_devices > 0 && fuifgpu_device_count(&n_dev) != FUIFGPU_OK) { e_printf("fuifgpu: %s\n", fuifgpu_last_error()); return 0; }
        for (int k = 0; k < n_devices; k++)
            if (devices[k] < 0 || devices[k] >= n_dev) { e_printf("fuifgpu: no GPU %d on this node (%d visible)\n", devices[k], n_dev); return 0; }
    }
    std::vector<std::vector<uint8_t>> bytes((size_t)n_files);
    std::vector<fuifgpu_plan *> plans((size_t)n_files, nullptr);
    std::vector<char> ok((size_t)n_files, 0), cpu_route((size_t)n_files, 0);
    std::map<uint64_t, std::vector<int>> groups;   // plan signature -> files
    for (int i = 0; i < n_files; i++) {
        FILE *f = fopen(filenames[i], "rb");
        if (!f) { e_printf("Could not open %s\n", filenames[i]); continue; }
        FileIO io(f, filenames[i]);
        bytes[i] = slurp(io);
        const int rc = fuifgpu_plan_create(bytes[i].data(), bytes[i].size(), &plans[i]);
        if (rc == FUIFGPU_E_UNSUPPORTED) { cpu_route[i] = 1; continue; }
        if (rc != FUIFGPU_OK) { e_printf("%s: %s (%s)\n", filenames[i], fuifgpu_strerror(rc), fuifgpu_last_error()); continue; }
        fuifgpu_image_info info;
        fuifgpu_plan_info(plans[i], &info);
        if (info.nb_frames > 1) { cpu_route[i] = 2; continue; }   // animations go one by one through fuif_decode (timing header)
        groups[info.signature].push_back(i);
    }
    // Images are independent units: the files of every signature group are dealt round-robin to the devices of the list and
    // one host thread per device runs its share -- its own batch objects, launches and downloads; nothing is exchanged between the devices,
    // the decoded Images land in host memory (the reference's Image owns std::vectors).  One device (or none named): the calling thread.
    struct Share { fuifgpu_plan *plan; std::vector<int> idx; };
    const int n_workers = std::max(1, n_devices);
    std::vector<std::vector<Share>> work((size_t)n_workers);
    for (auto &g : groups) {
        std::vector<std::vector<int>> dealt((size_t)n_workers);
        for (size_t k = 0; k < g.second.size(); k++) dealt[k % (size_t)n_workers].push_back(g.second[k]);
        for (int wkr = 0; wkr < n_workers; wkr++)
            if (!dealt[wkr].empty()) work[wkr].push_back(Share{plans[g.second[0]], dealt[wkr]});
    }
    auto run = [&](int wkr) {
        if (n_devices > 0 && fuifgpu_set_device(devices[wkr]) != FUIFGPU_OK) { e_printf("fuifgpu: GPU %d: %s\n", devices[wkr], fuifgpu_last_error()); return; }
        int sharers = 1;
        if (n_devices > 0) { sharers = 0; for (int k = 0; k < n_devices; k++) sharers += devices[k] == devices[wkr] ? 1 : 0; }
        for (const Share &sh : work[wkr]) decode_group_here(sh.plan, sh.idx, bytes, filenames, images, options, ok, cpu_route, verbose, sharers);
    };
    if (n_workers == 1) run(0);
    else {
        std::vector<std::thread> threads;
        for (int wkr = 0; wkr < n_workers; wkr++) threads.emplace_back(run, wkr);
        for (std::thread &t : threads) t.join();
    }
    for (int i = 0; i < n_files; i++) {
        if (cpu_route[i] == 2) {   // an animation: the single-file path (GPU as well)
            BlobReader io(bytes[i].data(), bytes[i].size());
            if (fuif_decode(io, images[i], options)) { fuifgpu_boundary_undo_transforms(&images[i], 0); ok[i] = !images[i].error; }
        } else if (cpu_route[i] == 1) {
            if (no_cpu) { e_printf("fuifgpu: %s needs a feature outside the GPU path; " FUIFGPU_OUTSIDE_HINT "\n", filenames[i]); continue; }
#ifdef FUIFGPU_WITH_CPU_FALLBACK
            if (verbose) fprintf(stderr, "fuifgpu: %s is outside the GPU path, decoding with the reference's CPU code\n", filenames[i]);
            ok[i] = cpu_decode_whole(bytes[i], images[i], options) ? 1 : 0;
#endif
        }
        if (plans[i]) fuifgpu_plan_destroy(plans[i]);
    }
    int n_ok = 0;
    for (int i = 0; i < n_files; i++) { n_ok += ok[i]; if (ok_out) ok_out[i] = ok[i] != 0; }
    return n_ok;
}

// Image::undo_transforms(int) -- exported under the member's Itanium-ABI name.  This file is compiled
// with -Dundo_transforms=undo_transforms_cpu (like image.cpp), so inside this translation unit the class
// declares the reference's CPU implementation as Image::undo_transforms_cpu, and the GPU version is a
// free function taking `this` explicitly (same calling convention) bound to the original symbol.
void fuifgpu_boundary_undo_transforms(Image *self, int keep) __asm__("_ZN5Image15undo_transformsEi");
void fuifgpu_boundary_undo_transforms(Image *self, int keep) {
    auto it = registry().find(self);
    // the batch must belong to THIS image: same channel table as fuif_decode left it (an Image constructed later at the
    // same address, or one whose channels were replaced, is not ours)
    if (it != registry().end() && (it->second->n_channels != self->channel.size() ||
                                   it->second->first_plane != (self->channel.empty() ? nullptr : self->channel[0].data.data()))) {
        registry().erase(it);
        it = registry().end();
    }
    if (it == registry().end() || keep != 0) {
        if (it != registry().end()) registry().erase(it);
        self->undo_transforms(keep);  // macro-renamed: the reference's own CPU implementation
        return;
    }
    Resident &res = *it->second;
    fuifgpu_image_info info;
    fuifgpu_plan_info(res.plan, &info);
    int rc = fuifgpu_batch_undo_transforms(res.batch, nullptr);
    if (rc == FUIFGPU_OK) rc = fuifgpu_batch_sync(res.batch, nullptr);
    if (rc != FUIFGPU_OK) {
        e_printf("Error while undoing transforms on the GPU: %s\n", fuifgpu_last_error());
        self->error = true;
        registry().erase(it);
        return;
    }
    // the inverse kernels can still flag the image (forward references of a 2D match, a match channel of neither mode: k_match_init / k_inv_match_frames)
    int32_t status = 0;
    fuifgpu_batch_status(res.batch, &status, nullptr);
    if (status & FUIFGPU_ST_UNSUPPORTED) {
        if (!cpu_fallback_allowed()) {
            e_printf("fuifgpu: the transform chain needs a feature outside the GPU path; " FUIFGPU_OUTSIDE_HINT "\n");
            self->error = true;
        } else {
#ifdef FUIFGPU_WITH_CPU_FALLBACK
            // the planes on the host are still the coded ones: decode again with the reference's code and undo there
            BlobReader again(res.bytes.data(), res.bytes.size());
            Image redo;
            fuif_options opt = default_fuif_options;
            opt.preview = res.preview;
            if (fuif_decode_cpu(again, redo, opt)) {
                redo.undo_transforms(0);   // (macro-renamed: the CPU implementation)
                self->channel.swap(redo.channel); self->transform.clear();
                self->nb_channels = redo.nb_channels; self->nb_meta_channels = redo.nb_meta_channels; self->error = redo.error;
            } else self->error = true;
#endif
        }
        registry().erase(self);
        return;
    }
    if (status & FUIFGPU_ST_CORRUPT) {
        e_printf("Corruption detected while undoing transforms.\n");
        self->error = true;
        registry().erase(it);
        return;
    }
    std::vector<int32_t> slab((size_t)(info.out_elems > 0 ? info.out_elems : 1));
    fuifgpu_batch_download_out(res.batch, 0, slab.data(), nullptr);
    std::vector<Channel> outch;
    for (int c = 0; c < info.nb_output_channels; c++) {
        fuifgpu_channel_desc d;
        fuifgpu_plan_output_channel(res.plan, c, &d);
        Channel ch(d.w, d.h, (pixel_type)self->minval, (pixel_type)self->maxval, 1, d.hshift, d.vshift, d.hcshift, d.vcshift);
        ch.component = d.component;
        const size_t n = (size_t)d.w * d.h;
        for (size_t i = 0; i < n; i++) ch.data[i] = (pixel_type)slab[(size_t)d.offset + i];
        outch.push_back(ch);
    }
    self->channel = outch;
    self->nb_meta_channels = 0;               // every Palette has been expanded again (palette.h:65-67)
    self->nb_channels = (int)outch.size();
    self->transform.clear();
    registry().erase(self);
}


// =====================================================================================================
// Transform::apply(image, inverse) -- transform/transform.cpp:48-63.  The reference's own definition is kept in the
// link as Transform::apply_cpu (the Makefile compiles transform.cpp with -Dapply=apply_cpu, and this file too, so the
// class declares it under that name here); the symbol Transform::apply is bound to the function below.  The INVERSE of
// Squeeze, YCoCg, YCbCr, DCT, Quantize, ChromaSubsample, Palette, Approximate and 2D-match runs on the MI355X through the single-transform
// entry points of the C-ABI (the inverse of Permute only reorders the channel list; it stays the reference's statement): this is the path
// of Image::undo_transforms(keep != 0) (fuif.cpp:220,230, image.cpp:94-115 calls t.apply(*this, true) per transform),
// while a plain undo_transforms() replays the whole chain on the planes that are still resident (above).  Everything
// else -- forward transforms, the other inverses, an image a kernel precondition does not hold for -- goes to apply_cpu.
namespace {
struct DevPlane {
    int32_t *d = nullptr;
    size_t n = 0;
    ~DevPlane() { fuifgpu_dev_free(d); }
    bool put(const Channel &ch) {   // pixel_type -> int32, H2D
        n = (size_t)ch.w * ch.h;
        if (ch.data.size() < n) return false;
        std::vector<int32_t> wide(n);
        for (size_t i = 0; i < n; i++) wide[i] = ch.data[i];
        d = (int32_t *)fuifgpu_dev_alloc(n * 4);
        return d && fuifgpu_dev_upload(d, wide.data(), n * 4) == FUIFGPU_OK;
    }
    bool alloc(size_t count) { n = count; d = (int32_t *)fuifgpu_dev_alloc(n * 4); return d != nullptr; }
    bool zeros(size_t count) {   // a channel a partial decode never reached: every Channel::value() of it reads zero (image.h:82)
        std::vector<int32_t> z(count, 0);
        return alloc(count) && fuifgpu_dev_upload(d, z.data(), n * 4) == FUIFGPU_OK;
    }
    bool get(Channel &ch) const {   // D2H, int32 -> pixel_type
        std::vector<int32_t> wide(n);
        if (fuifgpu_dev_download(wide.data(), d, n * 4) != FUIFGPU_OK) return false;
        ch.data.resize(n);
        for (size_t i = 0; i < n; i++) ch.data[i] = (pixel_type)wide[i];
        return true;
    }
};

// transform/ycocg.h:33-63 / transform/ycbcr.h:33-63 (same preconditions, checked by the caller's CPU twin otherwise)
bool gpu_inv_color(Image &img, bool ycbcr) {
    const int m = ycbcr ? 0 : img.nb_meta_channels;
    if ((int)img.channel.size() < m + 3 || (!ycbcr && img.nb_channels < 3)) return false;
    Channel &c0 = img.channel[m], &c1 = img.channel[m + 1], &c2 = img.channel[m + 2];
    const int w = c0.w, h = c0.h;
    if (w < 1 || h < 1 || c1.w < w || c1.h < h || c2.w < w || c2.h < h) return false;
    DevPlane p0, p1, p2;
    if (!p0.put(c0) || !p1.put(c1) || !p2.put(c2)) return false;
    const int rc = ycbcr ? fuifgpu_inv_ycbcr(p0.d, p1.d, p2.d, w, h, c0.w, c1.w, c2.w, img.minval, img.maxval, nullptr)
                         : fuifgpu_inv_ycocg(p0.d, p1.d, p2.d, w, h, c0.w, c1.w, c2.w, img.maxval, nullptr);
    if (rc != FUIFGPU_OK) return false;
    Channel n0 = c0, n1 = c1, n2 = c2;    // all three or none
    if (!p0.get(n0) || !p1.get(n1) || !p2.get(n2)) return false;
    c0 = n0; c1 = n1; c2 = n2;
    return true;
}

// transform/squeeze.h:363-388, inverse branch, with explicit parameters (after a decode they always are: meta_apply
// expands the defaults in place, squeeze.h:323-326)
bool gpu_inv_squeeze(Image &img, const std::vector<int> &par) {
    if (par.empty() || par.size() % 3) return false;
    // dry run of the channel bookkeeping: every step must be one the kernels take (sizes as forward squeeze leaves them)
    {
        std::vector<std::pair<int, int>> dims;
        for (const Channel &c : img.channel) dims.emplace_back(c.w, c.h);
        for (int i = (int)par.size() - 3; i >= 0; i -= 3) {
            const bool horizontal = par[i] & 1, in_place = !(par[i] & 2);
            const int beginc = par[i + 1], endc = par[i + 2];
            const int offset = in_place ? endc + 1 : img.nb_meta_channels + img.nb_channels;
            if (beginc < 0 || endc < beginc || offset + (endc - beginc) >= (int)dims.size()) return false;
            for (int c = beginc; c <= endc; c++) {
                const auto a = dims[c], r = dims[offset + c - beginc];
                if (a.first < 1 || a.second < 1) return false;
                if (horizontal) { if (r.second != a.second || a.first - r.first < 0 || a.first - r.first > 1) return false; dims[c].first += r.first; }
                else { if (r.first != a.first || a.second - r.second < 0 || a.second - r.second > 1) return false; dims[c].second += r.second; }
            }
            dims.erase(dims.begin() + offset, dims.begin() + offset + (endc - beginc + 1));
        }
    }
    // (on a copy of the channel table: a step that fails -- allocation, upload, kernel -- must leave the image as it was, so that the
    // caller's CPU twin, or the error it reports, sees the untouched input and not a half-unsqueezed one)
    std::vector<Channel> work = img.channel;
    for (int i = (int)par.size() - 3; i >= 0; i -= 3) {
        const bool horizontal = par[i] & 1, in_place = !(par[i] & 2);
        const int beginc = par[i + 1], endc = par[i + 2];
        const int offset = in_place ? endc + 1 : img.nb_meta_channels + img.nb_channels;
        for (int c = beginc; c <= endc; c++) {
            Channel &chin = work[c];
            Channel &res = work[offset + c - beginc];
            if (res.data.size() == 0) res.resize();   // zero-filled residuals of a partial decode (squeeze.h:379-383)
            DevPlane a, r, o;
            if (!a.put(chin) || (res.w * res.h > 0 && !r.put(res))) return false;
            int rc;
            Channel out = horizontal ? Channel(chin.w + res.w, chin.h, chin.minval, chin.maxval, chin.q, chin.hshift - 1, chin.vshift, chin.hcshift - 1, chin.vcshift)
                                     : Channel(chin.w, chin.h + res.h, chin.minval, chin.maxval, chin.q, chin.hshift, chin.vshift - 1, chin.hcshift, chin.vcshift - 1);
            out.component = chin.component;
            if (!o.alloc((size_t)out.w * out.h)) return false;
            if (horizontal) rc = fuifgpu_inv_hsqueeze(a.d, chin.w, r.d, res.w, chin.h, o.d, 1, 0, 0, 0, nullptr);
            else rc = fuifgpu_inv_vsqueeze(a.d, chin.h, r.d, res.h, chin.w, o.d, 1, 0, 0, 0, nullptr);
            if (rc != FUIFGPU_OK || !o.get(out)) return false;
            work[c] = out;
        }
        work.erase(work.begin() + offset, work.begin() + offset + (endc - beginc + 1));
    }
    img.channel.swap(work);
    return true;
}

// transform/quantize.h:32-49
bool gpu_inv_quantize(Image &img) {
    std::vector<Channel> work = img.channel;
    for (size_t c = (size_t)img.nb_meta_channels; c < work.size(); c++) {
        Channel &ch = work[c];
        if (ch.data.size() == 0) continue;
        const int q = ch.q;
        if (q == 1) continue;
        DevPlane p;
        if (!p.put(ch) || fuifgpu_inv_quantize(p.d, (int64_t)p.n, q, nullptr) != FUIFGPU_OK || !p.get(ch)) return false;
        ch.minval *= q;     // (pixel_type arithmetic, as in the reference)
        ch.maxval *= q;
        ch.q = 1;
    }
    img.channel.swap(work);
    return true;
}

// transform/dct.h:249-296.  The 64 coefficient planes of a component must share the block grid (they do in every stream an
// encoder writes: meta_DCT gives them one geometry); anything else is left to the reference's own loop.
bool gpu_inv_dct(Image &img, std::vector<int> par) {
    if (par.empty()) refdct::default_DCT_parameters(par, img);
    if (par.size() < 2) return false;
    const int beginc = img.nb_meta_channels + par[0], endc = img.nb_meta_channels + par[1];
    const int nb = endc - beginc + 1;
    const int offset = (int)img.channel.size() - 63 * nb;
    if (nb < 1 || beginc < 0 || offset <= endc) return false;
    std::vector<std::vector<int>> ordering;
    std::vector<int> comp, coeff;
    refdct::default_DCT_scanscript(nb, ordering, comp, coeff);
    std::vector<Channel> work = img.channel;
    for (int c = beginc; c <= endc; c++) {
        int bw = img.channel[c - beginc + offset].w, bh = img.channel[c - beginc + offset].h;
        if (img.channel[c].w < bw) bw = img.channel[c].w;
        if (img.channel[c].h < bh) bh = img.channel[c].h;
        if (bw < 1 || bh < 1) return false;
        std::vector<DevPlane> planes(64);
        const int32_t *src[64];
        for (int i = 0; i < 64; i++) {
            const Channel &sc = img.channel[i == 0 ? c : offset - nb + ordering[c - beginc][refdct::jpeg_zigzag[i]]];
            if (sc.data.size() == 0) { if (!planes[i].zeros((size_t)bw * bh)) return false; }   // responsive decode: coefficients not loaded are zero (dct.h:285-286 via value())
            else if (sc.w != bw || sc.h < bh || sc.data.size() < (size_t)sc.w * sc.h) return false;
            else if (!planes[i].put(sc)) return false;
            src[i] = planes[i].d;
        }
        DevPlane o;
        if (!o.alloc((size_t)bw * 8 * bh * 8)) return false;
        if (fuifgpu_idct8x8(src, bw, bh, o.d, img.maxval, nullptr) != FUIFGPU_OK) return false;
        Channel outch(bw * 8, bh * 8, 0, 0);
        outch.component = img.channel[c].component;
        outch.hshift = img.channel[c].hshift - 3;
        outch.vshift = img.channel[c].vshift - 3;
        outch.hcshift = img.channel[c].hcshift - 3;
        outch.vcshift = img.channel[c].hcshift - 3;   // (sic: dct.h:278 takes hcshift for both)
        if (!o.get(outch)) return false;
        work[c] = outch;
    }
    work.erase(work.begin() + offset, work.begin() + offset + nb * 63);
    img.channel.swap(work);
    return true;
}

// transform/subsample.h:73-127 (the "fancy" filter for factors 1 and 2, replication for larger ones: both in k_upsample)
bool gpu_inv_subsample(Image &img, std::vector<int> par) {
    refdct::check_subsample_parameters(par);
    std::vector<Channel> work = img.channel;
    for (size_t i = 0; i + 3 < par.size(); i += 4) {
        const int c1 = par[i], c2 = par[i + 1], srh = par[i + 2], srv = par[i + 3];
        if (c1 < 0 || c2 >= (int)work.size() || srh < 1 || srv < 1 || srh > 8 || srv > 8) return false;
        for (int c = c1; c <= c2; c++) {
            const int ow = work[c].w, oh = work[c].h;
            if (ow >= work[img.nb_meta_channels].w && oh >= work[img.nb_meta_channels].h) continue;   // subsample.h:87-91
            if (ow < 1 || oh < 1) return false;
            DevPlane in, out;
            Channel up(ow * srh, oh * srv, work[c].minval, work[c].maxval);
            if (!in.put(work[c]) || !out.alloc((size_t)up.w * up.h)) return false;
            if (fuifgpu_upsample(in.d, ow, oh, srh, srv, out.d, nullptr) != FUIFGPU_OK || !out.get(up)) return false;
            work[c] = up;
        }
    }
    img.channel.swap(work);
    return true;
}

// transform/palette.h:32-68.  The palette is meta-channel 0 (colours wide, one row per component); the index channel becomes
// component 0 and nb-1 new channels follow it.  One gather per component (fuifgpu_inv_palette).
bool gpu_inv_palette(Image &img, const std::vector<int> &par) {
    if (img.nb_meta_channels < 1 || par.size() != 3 || img.channel.empty()) return false;
    const int nb = img.channel[0].h, colours = img.channel[0].w;
    const int c0 = img.nb_meta_channels + par[0];
    if (nb < 1 || colours < 0 || c0 < 1 || c0 >= (int)img.channel.size()) return false;
    const int w = img.channel[c0].w, h = img.channel[c0].h;
    if (w < 0 || h < 0 || img.channel[c0].data.size() != (size_t)w * h) return false;   // a partly decoded index channel: the reference's loop has its own reading of that
    std::vector<Channel> work = img.channel;
    for (int i = 1; i < nb; i++) {      // palette.h:52-55, statement by statement (which of the new channels ends up labelled follows from it)
        work.insert(work.begin() + c0 + 1, Channel(w, h, 0, 1));
        work[c0 + i].component = par[0] + i;
    }
    if ((size_t)w * h > 0) {
        const Channel &palette = work[0];
        DevPlane idx;
        if (!idx.put(img.channel[c0])) return false;
        for (int c = 0; c < nb; c++) {
            Channel row(std::max(colours, 1), 1, 0, 0);
            for (int i = 0; i < colours; i++) row.data[i] = palette.value(c, i);   // Channel::value: samples a partial decode never reached read as zero
            DevPlane pr, out;
            if (!pr.put(row) || !out.alloc((size_t)w * h)) return false;
            if (fuifgpu_inv_palette(idx.d, w, h, pr.d, colours, out.d, nullptr) != FUIFGPU_OK || !out.get(work[c0 + c])) return false;
        }
    }
    work.erase(work.begin(), work.begin() + 1);
    img.channel.swap(work);
    img.nb_channels += nb - 1;
    img.nb_meta_channels--;
    return true;
}

// transform/approximate.h:32-62: channel = channel * (parameter + 1) + remainder, the remainder channels sit at the end of the list
bool gpu_inv_approximate(Image &img, const std::vector<int> &par) {
    if (par.size() < 3) return false;
    const int beginc = par[0], endc = par[1];
    if (beginc < 0 || endc < beginc || endc >= (int)img.channel.size()) return false;
    auto factor = [&](int c) { return (c + 2 - beginc < (int)par.size() ? par[c + 2 - beginc] : par.back()); };
    int offset = (int)img.channel.size() - (endc - beginc + 1);
    for (int c = beginc; c <= endc; c++) if (!factor(c)) offset++;
    if (offset <= endc || offset > (int)img.channel.size()) return false;
    std::vector<Channel> work = img.channel;
    int i = 0;
    for (int c = beginc; c <= endc; c++) {
        const int q = factor(c) + 1;
        if (q == 1) continue;
        if (offset + i >= (int)work.size()) return false;
        Channel &ch = work[c];
        const Channel &chr = work[offset + i];
        i++;
        const size_t n = (size_t)ch.w * ch.h;
        const bool have = chr.data.size() != 0;
        // a responsive decode that reached neither channel: every read is Channel::zero and every store goes to it (image.h:82-85); 0*q + 0 changes nothing
        if (ch.data.size() == 0 && !have && ch.zero == 0) continue;
        if (ch.w < 0 || ch.h < 0 || ch.data.size() != n || (have && (chr.w != ch.w || chr.h != ch.h || chr.data.size() != n))) return false;
        if (have) ch.q = chr.q;
        if (n == 0) continue;
        DevPlane p, r;
        if (!p.put(ch) || (have && !r.put(chr))) return false;
        if (fuifgpu_inv_approximate(p.d, have ? r.d : nullptr, (int64_t)n, q, nullptr) != FUIFGPU_OK || !p.get(ch)) return false;
    }
    work.erase(work.begin() + offset, work.end());
    img.channel.swap(work);
    return true;
}

// transform/2dmatch.h:112-177: the match meta-channel says, per sample, which earlier sample (free offsets, q == 1) or which earlier
// frame (q == 2*fh*fh + (fh&1)) it repeats -- or, for a soft match, is a difference to.  Chains are resolved on the GPU by pointer
// jumping (fuifgpu_inv_match); the mode is data (Channel::q of the match channel).
bool gpu_inv_match(Image &img, std::vector<int> par) {
    if (img.nb_meta_channels < 1 || img.channel.empty()) return false;
    if (par.empty()) { par = {0, img.nb_channels - 1, 0, 1000000}; }   // default_match_parameters, 2dmatch.h:104-110
    if (par.size() < 3) return false;
    const Channel &m = img.channel[0];
    const int c0 = img.nb_meta_channels + par[0], cn = img.nb_meta_channels + par[1];
    if (c0 < 1 || cn < c0 || cn >= (int)img.channel.size() || cn - c0 + 1 > 64 || img.nb_frames < 1) return false;
    const int w = img.channel[c0].w, h = img.channel[c0].h;
    const size_t n = (size_t)w * h;
    if (w < 0 || h < 0 || m.w != w || m.h != h || m.data.size() != n) return false;
    std::vector<Channel> work = img.channel;
    if (n > 0) {
        std::vector<DevPlane> planes(cn - c0 + 1);
        std::vector<int32_t *> ptrs;
        for (int c = c0; c <= cn; c++) {
            if (work[c].w != w || work[c].h != h || work[c].data.size() != n || !planes[c - c0].put(work[c])) return false;
            ptrs.push_back(planes[c - c0].d);
        }
        DevPlane dm;
        if (!dm.put(m)) return false;
        if (fuifgpu_inv_match(dm.d, w, h, ptrs.data(), (int)ptrs.size(), par[2] ? 1 : 0, m.q, m.maxval, img.nb_frames, nullptr) != FUIFGPU_OK) return false;
        for (int c = c0; c <= cn; c++) if (!planes[c - c0].get(work[c])) return false;
    }
    work.erase(work.begin(), work.begin() + 1);
    img.channel.swap(work);
    img.nb_meta_channels--;
    return true;
}
}  // namespace

bool fuifgpu_boundary_transform_apply(Transform *self, Image &input, bool inverse) __asm__("_ZN9Transform5applyER5Imageb");
bool fuifgpu_boundary_transform_apply(Transform *self, Image &input, bool inverse) {
#ifdef FUIFGPU_WITH_CPU_FALLBACK
    const bool cpu_transforms = env_flag("FUIFGPU_CPU_TRANSFORMS");   // opt-in build only: every inverse through the reference's code
#else
    constexpr bool cpu_transforms = false;
#endif
    if (inverse && !cpu_transforms) {
        bool done = false, claimed = true;
        switch (self->ID) {
            case TRANSFORM_YCoCg: done = gpu_inv_color(input, false); break;
            case TRANSFORM_YCbCr: done = gpu_inv_color(input, true); break;
            case TRANSFORM_SQUEEZE: done = gpu_inv_squeeze(input, self->parameters); break;
            case TRANSFORM_DCT: done = gpu_inv_dct(input, self->parameters); break;
            case TRANSFORM_QUANTIZE: done = gpu_inv_quantize(input); break;
            case TRANSFORM_ChromaSubsample: done = gpu_inv_subsample(input, self->parameters); break;
            case TRANSFORM_PALETTE: done = gpu_inv_palette(input, self->parameters); break;
            case TRANSFORM_APPROXIMATE: done = gpu_inv_approximate(input, self->parameters); break;
            case TRANSFORM_2DMATCH: done = gpu_inv_match(input, self->parameters); break;
            default: claimed = false; break;   // Permute (permute.h:31-54) reorders the channel list and touches no sample: nothing for a kernel to do
        }
        if (done) {
            if (env_flag("FUIFGPU_VERBOSE")) fprintf(stderr, "fuifgpu: inverse %s on the GPU (Transform::apply)\n", self->name());
            return true;
        }
        // a transform this layer binds, on an image its kernels do not take (or a device error): the image is untouched (every
        // gpu_inv_* works on a copy), so the reference's own loop can run -- unless the caller asked for the GPU path or nothing
        if (claimed && !cpu_fallback_allowed()) {
            e_printf("fuifgpu: inverse %s could not run on the GPU (%s); " FUIFGPU_OUTSIDE_HINT "\n", self->name(), fuifgpu_last_error());
            return false;
        }
        if (claimed && env_flag("FUIFGPU_VERBOSE")) fprintf(stderr, "fuifgpu: inverse %s with the reference's CPU code (Transform::apply)\n", self->name());
    }
    return self->apply(input, inverse);   // macro-renamed: Transform::apply_cpu, the reference's own dispatcher
}
