// fuif_amd/boundary/fuifgpu_boundary.h -- the one entry point the binding ADDS to the reference's decode interface.
//
// The reference decodes one file per call: fuif_decode_file(filename, image, options) followed by image.undo_transforms()
// (encoding/encoding.cpp:745-753, fuif.cpp:213-233).  One file is one launch of a few dozen wavefronts on a device that holds
// 6144 -- the MI355X path is built for batches.  fuif_decode_files() is the batch form of those two calls: N files in, N
// finished Images out (entropy decode AND the whole inverse-transform chain), files of equal geometry and transform chain
// sharing one fuifgpu_batch, i.e. one k_maniac_decode launch.  Everything else about an Image is as fuif_decode_file +
// undo_transforms leave it, so the reference's writers (export/write_pam.h ...) take it unchanged.
#pragma once
#include "encoding/encoding.h"
#include "image/image.h"

// images[i] receives file i; ok[i] (optional) says whether it decoded.  Returns the number of files decoded.
// A file outside the GPU scope is an error (ok[i] = false); only a binding built with -DFUIFGPU_WITH_CPU_FALLBACK can route it to the reference's own decoder (FUIFGPU_ALLOW_CPU_FALLBACK=1).
// The files are decoded on the calling thread's current GPU -- or, with FUIFGPU_DEVICES="all" / "0,1,..." in the environment, spread over those
// GPUs of the node as fuif_decode_files_on does.
int fuif_decode_files(const char *const *filenames, int n_files, Image *images, fuif_options options, bool *ok = nullptr);

// Several GPUs of one node (round 5): images are independent units, so the files of every geometry are dealt round-robin to the
// `n_devices` GPUs of `devices` and one host thread per GPU decodes its share (fuifgpu_set_device + its own fuifgpu_batch objects:
// one k_maniac_decode launch per geometry and GPU); nothing moves between the GPUs, every finished Image lands in host memory like
// fuif_decode_file + undo_transforms leave it.  A device that does not exist fails the call before anything is decoded (returns 0).
// n_devices = 0: the calling thread's current GPU.  The same device may be named twice (two host threads, two batches on it).
int fuif_decode_files_on(const char *const *filenames, int n_files, Image *images, fuif_options options, const int *devices, int n_devices,
                         bool *ok = nullptr);
