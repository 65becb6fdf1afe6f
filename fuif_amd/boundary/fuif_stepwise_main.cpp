// fuif_amd/boundary/fuif_stepwise_main.cpp -- a front end that undoes the transform chain ONE TRANSFORM AT A TIME:
//
//     fuif_gpu_stepwise [-R n] in.fuif out.pam
//
// = fuif_decode_file, then Image::undo_transforms(k) for k = n-1 ... 0 (image/image.cpp:94-115: each call undoes the last
// transform through Transform::apply(image, true) and keeps the first k), then write_PAM_file.  An application that looks at
// intermediate stages (the CLI's own .yuv output is undo_transforms(2), fuif.cpp:228-231) works this way, and it is the path
// on which the binding's per-transform entry (fuif_gpu_boundary.cpp, Transform::apply) does ALL the work -- every inverse the
// C-ABI has a single-transform entry point for, Palette, Approximate and 2D-match included.  The output must equal
// `fuif -d in.fuif out.pam` of the unmodified CLI byte for byte (tests/test_boundary_cli.py, tests/test_emulated_kernels.py).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "encoding/encoding.h"
#include "export/write_pam.h"
#include "image/image.h"
#include "io.h"

int main(int argc, char **argv) {
    int a = 1;
    fuif_options options = default_fuif_options;
    if (a + 1 < argc && !strcmp(argv[a], "-R")) { options.preview = atoi(argv[a + 1]); a += 2; }
    if (argc - a != 2) { fprintf(stderr, "usage: %s [-R 0..4] in.fuif out.pam\n", argv[0]); return 2; }
    Image image;
    if (!fuif_decode_file(argv[a], image, options)) { fprintf(stderr, "%s: not decoded\n", argv[a]); return 1; }
    for (int keep = (int)image.transform.size() - 1; keep >= 0 && !image.error; keep--) image.undo_transforms(keep);
    if (!image.error) image.undo_transforms(0);   // a chain that was empty all along still gets the final clamp (image.cpp:107-113)
    if (image.error) { fprintf(stderr, "%s: a transform could not be undone\n", argv[a]); return 1; }
    write_PAM_file(argv[a + 1], image);   // (its return value is no verdict: the reference's function ends in `return 0`, export/write_pam.h)
    return 0;
}
