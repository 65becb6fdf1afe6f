// fuif_amd/boundary/fuif_batch_main.cpp -- a many-files front end for the batch entry of the binding:
//
//     fuif_gpu_batch [-R n] [--devices 0,1,...] OUTDIR a.fuif b.fuif ...      ->  OUTDIR/a.pam, OUTDIR/b.pam, ...
//
// i.e. `fuif -d x.fuif x.pam` for every file, with all files of one geometry decoded in ONE launch (fuif_decode_files,
// fuifgpu_boundary.h).  It is built from the reference's own Image / export code like the CLI (write_PAM_file is the
// reference's, export/write_pam.h) and exists so that the batch entry can be tested against the unmodified CLI's output
// files byte for byte (tests/test_boundary_cli.py); a real consumer calls fuif_decode_files() directly.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "encoding/encoding.h"
#include "export/write_pam.h"
#include "image/image.h"
#include "io.h"

#include "fuifgpu_boundary.h"

int main(int argc, char **argv) {
    int a = 1;
    fuif_options options = default_fuif_options;
    std::vector<int> devices;      // --devices 0,1,... : the GPUs of the node the files are spread over (one host thread each)
    for (bool more = true; more && a + 1 < argc;) {
        more = false;
        if (!strcmp(argv[a], "-R")) { options.preview = atoi(argv[a + 1]); a += 2; more = true; }
        else if (!strcmp(argv[a], "--devices")) {
            for (const char *p = argv[a + 1]; *p;) {
                char *end = nullptr;
                const long d = strtol(p, &end, 10);
                if (end == p) { fprintf(stderr, "--devices wants a comma-separated list of GPU numbers\n"); return 2; }
                devices.push_back((int)d);
                p = *end == ',' ? end + 1 : end;
            }
            a += 2; more = true;
        }
    }
    if (argc - a < 2) { fprintf(stderr, "usage: %s [-R 0..4] [--devices 0,1,...] OUTDIR file.fuif ...\n", argv[0]); return 2; }
    const std::string outdir = argv[a++];
    const int n = argc - a;
    std::vector<Image> images((size_t)n);
    std::vector<char> ok((size_t)n, 0);
    bool *okp = new bool[n];
    const int done = devices.empty() ? fuif_decode_files(argv + a, n, images.data(), options, okp)
                                     : fuif_decode_files_on(argv + a, n, images.data(), options, devices.data(), (int)devices.size(), okp);
    for (int i = 0; i < n; i++) {
        if (!okp[i]) { fprintf(stderr, "%s: not decoded\n", argv[a + i]); continue; }
        std::string base = argv[a + i];
        const size_t slash = base.find_last_of('/');
        if (slash != std::string::npos) base = base.substr(slash + 1);
        const size_t dot = base.find_last_of('.');
        if (dot != std::string::npos) base = base.substr(0, dot);
        const std::string out = outdir + "/" + base + ".pam";
        write_PAM_file(out.c_str(), images[i]);
    }
    delete[] okp;
    return done == n ? 0 : 1;
}
