// fuif_amd/boundary/fuif_index_main.cpp -- gives EXISTING FUIF files the group index:
//
//     fuif_gpu_index OUTDIR a.fuif b.fuif ...      ->  OUTDIR/a.fuif, OUTDIR/b.fuif, ... = the input bytes + the FGIX trailer
//
// A stream as the reference encoder writes it has no entry points (a channel group ends where its range coder stopped reading,
// maniac/rac.h:70-104), so it decodes on ONE wavefront.  Decoding it once tells where every group starts
// (fuifgpu_batch_group_index); the trailer keeps those offsets behind the stream, where the reference decoder never looks
// (encoding/encoding.cpp:708-717: it stops after the last group), and from then on every group of the file decodes on a wavefront
// of its own (3x the throughput on 1024 x 4K).  All files of one geometry go through ONE launch; only the entropy decode runs (a
// streaming batch: no output slab, no inverse transforms).  Files that already carry a valid trailer, and files the GPU path does
// not take, are copied unchanged (and said so).  Plain C++ over include/fuifgpu.h: no reference code is linked.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "fuifgpu.h"

static bool read_file(const char *path, std::vector<uint8_t> &out) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
    fclose(f);
    return true;
}
static bool write_file(const std::string &path, const uint8_t *p, size_t n) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(p, 1, n, f) == n;
    return fclose(f) == 0 && ok;
}
static std::string out_name(const std::string &dir, const char *in) {
    std::string base = in;
    const size_t slash = base.find_last_of('/');
    if (slash != std::string::npos) base = base.substr(slash + 1);
    return dir + "/" + base;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s OUTDIR file.fuif ...\n", argv[0]); return 2; }
    const std::string outdir = argv[1];
    const int n = argc - 2;
    char **names = argv + 2;
    // two inputs of one basename (from different directories) would land on one output file and the second would silently replace the
    // first: refused before anything is written
    for (int i = 0; i < n; i++)
        for (int k = 0; k < i; k++)
            if (out_name(outdir, names[i]) == out_name(outdir, names[k])) {
                fprintf(stderr, "%s and %s would both be written to %s: give files of one name separate runs / output directories\n", names[k], names[i], out_name(outdir, names[i]).c_str());
                return 2;
            }
    std::vector<std::vector<uint8_t>> bytes((size_t)n);
    std::vector<fuifgpu_plan *> plans((size_t)n, nullptr);
    std::map<uint64_t, std::vector<int>> groups;   // plan signature -> files
    int failed = 0, copied = 0, indexed = 0;
    for (int i = 0; i < n; i++) {
        if (!read_file(names[i], bytes[i])) { fprintf(stderr, "%s: cannot read\n", names[i]); failed++; continue; }
        int have = 0;
        if (fuifgpu_index_parse(bytes[i].data(), bytes[i].size(), nullptr, nullptr, 0, &have) == FUIFGPU_OK && have > 0) {
            fprintf(stderr, "%s: already indexed (%d groups), copied\n", names[i], have);
            if (write_file(out_name(outdir, names[i]), bytes[i].data(), bytes[i].size())) copied++; else failed++;
            bytes[i].clear();
            continue;
        }
        const int rc = fuifgpu_plan_create(bytes[i].data(), bytes[i].size(), &plans[i]);
        if (rc != FUIFGPU_OK) {
            fprintf(stderr, "%s: %s (%s), copied without index\n", names[i], fuifgpu_strerror(rc), fuifgpu_last_error());
            if (rc == FUIFGPU_E_UNSUPPORTED && write_file(out_name(outdir, names[i]), bytes[i].data(), bytes[i].size())) copied++; else failed++;
            bytes[i].clear();
            continue;
        }
        fuifgpu_image_info info;
        fuifgpu_plan_info(plans[i], &info);
        groups[info.signature].push_back(i);
    }
    for (auto &g : groups) {
        const std::vector<int> &idx = g.second;
        fuifgpu_image_info info;
        fuifgpu_plan_info(plans[idx[0]], &info);
        size_t max_stream = 0;
        for (int i : idx) max_stream = std::max(max_stream, bytes[i].size());
        // how many of the group's files go into one launch: int16 coefficients + stream + context arena per picture, a quarter of the
        // free memory (at most 40 GiB) kept for the decoder scratch
        int chunk = (int)idx.size();
        size_t free_b = 0, total_b = 0;
        if (fuifgpu_dev_mem_info(&free_b, &total_b) == FUIFGPU_OK && free_b) {
            const size_t reserve = std::min<size_t>(free_b / 4, (size_t)40 << 30);
            const size_t per_image = 2 * (size_t)info.coef_elems + max_stream + ((size_t)16 << 20);
            chunk = (int)std::max<size_t>(1, std::min<size_t>(idx.size(), (free_b - reserve) / std::max<size_t>(per_image, 1)));
        }
        if (const char *e = getenv("FUIFGPU_BOUNDARY_CHUNK")) chunk = std::max(1, std::min((int)idx.size(), atoi(e)));
        fuifgpu_batch *batch = nullptr;
        int rc = fuifgpu_batch_create_streaming(plans[idx[0]], chunk, (max_stream + 4096) * (size_t)chunk, 1, &batch);
        if (rc != FUIFGPU_OK) { fprintf(stderr, "batch of %d: %s (%s)\n", chunk, fuifgpu_strerror(rc), fuifgpu_last_error()); failed += (int)idx.size(); continue; }
        for (size_t at = 0; at < idx.size(); at += (size_t)chunk) {
            const int cnt = (int)std::min<size_t>((size_t)chunk, idx.size() - at);
            std::vector<const uint8_t *> ptrs((size_t)cnt);
            std::vector<size_t> sizes((size_t)cnt);
            for (int k = 0; k < cnt; k++) { ptrs[k] = bytes[idx[at + k]].data(); sizes[k] = bytes[idx[at + k]].size(); }
            rc = fuifgpu_batch_upload(batch, ptrs.data(), sizes.data(), cnt, -1, nullptr);
            if (rc == FUIFGPU_OK) rc = fuifgpu_batch_decode(batch, nullptr);
            if (rc == FUIFGPU_OK) rc = fuifgpu_batch_sync(batch, nullptr);
            if (rc != FUIFGPU_OK) { fprintf(stderr, "decode of %d file(s): %s (%s)\n", cnt, fuifgpu_strerror(rc), fuifgpu_last_error()); failed += cnt; continue; }
            std::vector<int32_t> status((size_t)cnt);
            std::vector<uint32_t> used((size_t)cnt);
            fuifgpu_batch_status(batch, status.data(), used.data());
            for (int k = 0; k < cnt; k++) {
                const int i = idx[at + k];
                // a truncated, damaged or out-of-scope stream gets no index: its group starts are not those of the whole file
                if (status[k] != 0) { fprintf(stderr, "%s: status %d, copied without index\n", names[i], status[k]); if (write_file(out_name(outdir, names[i]), bytes[i].data(), bytes[i].size())) copied++; else failed++; continue; }
                std::vector<int32_t> first((size_t)info.nb_coded_channels + 1);
                std::vector<uint32_t> start((size_t)info.nb_coded_channels + 1);
                int ng = 0;
                uint8_t *out = nullptr;
                size_t out_size = 0;
                rc = fuifgpu_batch_group_index(batch, k, first.data(), start.data(), (int)first.size(), &ng);
                if (rc == FUIFGPU_OK) rc = fuifgpu_index_append(bytes[i].data(), bytes[i].size(), first.data(), start.data(), ng, &out, &out_size);
                if (rc != FUIFGPU_OK || !write_file(out_name(outdir, names[i]), out, out_size)) { fprintf(stderr, "%s: %s\n", names[i], rc != FUIFGPU_OK ? fuifgpu_last_error() : "cannot write"); failed++; }
                else { indexed++; if (getenv("FUIFGPU_VERBOSE")) fprintf(stderr, "%s: %d groups, %zu + %zu bytes\n", names[i], ng, bytes[i].size(), out_size - bytes[i].size()); }
                fuifgpu_free_blob(out);
            }
        }
        fuifgpu_batch_destroy(batch);
    }
    for (fuifgpu_plan *p : plans) if (p) fuifgpu_plan_destroy(p);
    fprintf(stderr, "fuifgpu: %d file(s) indexed, %d copied unchanged, %d failed\n", indexed, copied, failed);
    return failed ? 1 : 0;
}
