// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A thin extern "C" driver around the *real* cloudinary/fuif sources, compiled from where they
// lie under /root/reference by oracle/Makefile into oracle/_ref/libfuifref.so.  No reference
// source is copied: this file only #includes the reference's public headers and calls
//   fuif_decode<FileIO|BlobReader>   (encoding/encoding.h:67-68, encoding.cpp:599-720)
//   Image::undo_transforms           (image/image.h:126, image.cpp:94-115)
//   Image::do_transform / fuif_prepare_encode / fuif_encode<BlobIO>  (to make .fuif inputs)
// so that tests can compare the repo's oracle restatement and the HIP path plane-by-plane
// against the shipped reference (pixel_type = int16_t, image/image.h:35; widened to int32 here).
//
// The encode helper restates the *policy* of the CLI's encode branch for PNM input
// (fuif.cpp:380-393 colour transform, :441-455 squeeze + max_group=1, :580-588 default
// predictors) because main() is not callable as a library; the unmodified CLI itself is also
// built (oracle/_ref/fuif) and used to make the committed golden fixtures.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>

#include "encoding/encoding.h"
#include "image/image.h"
#include "transform/transform.h"
#include "fileio.h"
#include "io.h"

extern "C" {

// ---- decode ---------------------------------------------------------------------------------
// io_kind: 0 = FileIO over fmemopen (what the CLI uses: feof() semantics), 1 = BlobReader.
void *fuifref_decode(const uint8_t *blob, size_t n, int preview, int io_kind, int *ok) {
    Image *img = new Image();
    fuif_options options = default_fuif_options;
    options.preview = preview;
    bool r;
    if (io_kind == 0) {
        FILE *f = fmemopen((void *)blob, n, "rb");
        if (!f) { *ok = 0; return img; }
        FileIO fio(f, "mem");   // closes f in its destructor (fileio.h:49-51)
        r = fuif_decode(fio, *img, options);
    } else {
        BlobReader br(blob, n);
        r = fuif_decode(br, *img, options);
    }
    *ok = r ? 1 : 0;
    return img;
}

int fuifref_undo_transforms(void *h, int keep) {
    Image *img = (Image *)h;
    img->undo_transforms(keep);
    return img->error ? 0 : 1;
}

void fuifref_free(void *h) { delete (Image *)h; }

// out[0..9] = w,h,minval,maxval,nb_channels,real_nb_channels,nb_meta_channels,#channels,#transforms,error
void fuifref_image_info(void *h, int32_t *out) {
    Image *img = (Image *)h;
    out[0] = img->w; out[1] = img->h; out[2] = img->minval; out[3] = img->maxval;
    out[4] = img->nb_channels; out[5] = img->real_nb_channels; out[6] = img->nb_meta_channels;
    out[7] = (int32_t)img->channel.size(); out[8] = (int32_t)img->transform.size();
    out[9] = img->error ? 1 : 0;
}

// out[0..11] = w,h,minval,maxval,q,hshift,vshift,hcshift,vcshift,component,zero,data.size()
void fuifref_channel_info(void *h, int c, int32_t *out) {
    const Channel &ch = ((Image *)h)->channel[c];
    out[0] = ch.w; out[1] = ch.h; out[2] = ch.minval; out[3] = ch.maxval; out[4] = ch.q;
    out[5] = ch.hshift; out[6] = ch.vshift; out[7] = ch.hcshift; out[8] = ch.vcshift;
    out[9] = ch.component; out[10] = ch.zero; out[11] = (int32_t)ch.data.size();
}

void fuifref_channel_data(void *h, int c, int32_t *out) {
    const Channel &ch = ((Image *)h)->channel[c];
    for (size_t i = 0; i < ch.data.size(); i++) out[i] = ch.data[i];
}

// transform t: out[0]=ID, out[1]=#params, out[2..] = params (at most cap-2)
void fuifref_transform_info(void *h, int t, int32_t *out, int cap) {
    const Transform &tr = ((Image *)h)->transform[t];
    out[0] = tr.ID; out[1] = (int32_t)tr.parameters.size();
    for (int i = 0; i < (int)tr.parameters.size() && i + 2 < cap; i++) out[i + 2] = tr.parameters[i];
}

// ---- encode (input generator for tests) ------------------------------------------------------
// planes: nch planes of w*h int32, values in [0,maxval].
// opts[0] colorspace: -1 default (YCoCg when >=3 channels), 0 none
// opts[1] squeeze:    1 = default squeeze (CLI default "responsive"), 0 = off
// opts[2] max_group:  -1 = CLI default
// opts[3] nb_repeats*1000 (CLI default 500; 0 = no tree learning => single-leaf trees)
// opts[4] max_properties (CLI default 12)
// opts[5] compress (1 default; 0 = -U uncompressed groups)
// opts[6] predictor override for all channels (-1 = CLI defaults)
// opts[7], opts[8..11]: Permute (see below)
// opts[12], opts[13]: 2D match with explicit parameters {0, nch-1, opts[12] = softmatch, opts[13] = max distance} when opts[13] != 0, applied where the
//            CLI applies its (never soft) match: after the colour transform, before Squeeze (fuif.cpp:438-448).  A negative distance = the
//            previous-frame mode, which needs opts[14] = number of frames (>= 2; the planes are the vertical film strip, h = frames * frame height)
// opts[15]: one quantization constant for every non-meta channel (0 / 1 = none): Transform(TRANSFORM_QUANTIZE) with explicit parameters, after
//            Squeeze like the CLI's (fuif.cpp:458-505) -- a lossy stream whose soft matches carry non-zero differences
// callers pass at least 16 ints
// returns malloc'd blob in *out (caller frees with fuifref_free_blob), size as return value; 0 on failure
size_t fuifref_encode(int w, int h, int nch, int maxval, const int32_t *planes, const int32_t *opts, uint8_t **out) {
    *out = nullptr;
    Image img(w, h, maxval, nch);
    for (int c = 0; c < nch; c++)
        for (int i = 0; i < w * h; i++) img.channel[c].data[i] = (pixel_type)planes[(size_t)c * w * h + i];
    fuif_options options = default_fuif_options;
    options.max_group = opts[2];
    options.nb_repeats = opts[3] / 1000.0f;
    options.max_properties = opts[4];
    options.compress = opts[5] != 0;

    img.recompute_minmax();
    // Permute (transform/permute.h): the CLI never applies it (fuif.cpp:362-368 is commented out), so fixtures come from here.
    // opts[7]: 0 none; 1 explicit form (the permutation is a transform parameter: leading -1, permute.h:86-89);
    //          2 channel form (the permutation is the content of a 1-row meta-channel, permute.h:58-63; the forward step
    //            leaves the permutation in Transform::parameters as well, which a decoder would take for the explicit form --
    //            they are cleared here so that the stream says "read it from the meta-channel");  opts[8..8+nch) = permutation
    if (opts[7] == 1 || opts[7] == 2) {
        Transform perm(TRANSFORM_PERMUTE);
        if (opts[7] == 1) perm.parameters.push_back(-1);
        for (int c = 0; c < nch; c++) perm.parameters.push_back(opts[8 + c]);
        if (!img.do_transform(perm)) return 0;
        if (opts[7] == 2) img.transform.back().parameters.clear();
    }
    // fuif.cpp:380-393 (no palette here: photographic inputs)
    if (opts[0] < 0) img.do_transform(Transform(TRANSFORM_YCoCg));
    if (opts[14] >= 2) {
        if (h % opts[14]) return 0;
        img.nb_frames = opts[14];
    }
    if (opts[13] != 0) {
        Transform match(TRANSFORM_2DMATCH);
        match.parameters.push_back(0);
        match.parameters.push_back(img.nb_channels - 1);
        match.parameters.push_back(opts[12] ? 1 : 0);
        match.parameters.push_back(opts[13]);
        if (!img.do_transform(match)) return 0;
    }
    // fuif.cpp:449-455
    if (opts[1] && img.channel[0].w * img.channel[0].h > 20) {
        img.do_transform(Transform(TRANSFORM_SQUEEZE));
        if (options.max_group < 0) options.max_group = 1;
    }
    if (opts[15] > 1) {
        Transform quantize(TRANSFORM_QUANTIZE);
        for (int i = 0; i < img.nb_meta_channels; i++) quantize.parameters.push_back(1);
        for (size_t i = img.nb_meta_channels; i < img.channel.size(); i++) quantize.parameters.push_back(opts[15]);
        if (!img.do_transform(quantize)) return 0;
    }
    // fuif.cpp:580-588
    if (opts[6] >= 0) {
        options.predictor.push_back(opts[6]);
    } else {
        for (int i = 0; i < img.nb_meta_channels; i++) options.predictor.push_back(3);
        for (int i = 0; i < img.nb_channels; i++) options.predictor.push_back(2);
        options.predictor.push_back(0);
    }
    fuif_prepare_encode(img, options);
    BlobIO bio;
    if (!fuif_encode(bio, img, options)) return 0;
    size_t n = 0;
    uint8_t *p = bio.release(&n);
    *out = p;
    return n;
}

void fuifref_free_blob(uint8_t *p) { delete[] p; }

void fuifref_set_verbosity(int v) { increase_verbosity(v - get_verbosity()); }

}  // extern "C"
