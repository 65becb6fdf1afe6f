// fuif_amd/csrc/maniac_encode.h -- the writer's MANIAC pixel loop on the GPU ("fixed-tree MANIAC writer",
// maniac/rac_enc.h:28-100, maniac/symbol_enc.h, encoding/encoding.cpp:74-207).  Host interface of maniac_encode.hip.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "fuifgpu_internal.h"

namespace fuifgpu {

struct EncRef {            // a reference channel of the group (context_predict.h:233-289), samples on the device
    const int32_t *data;
    int32_t w, h, hshift, vshift;
};
struct EncNode {           // decoder-layout tree node (maniac/compound.h:41-56): prop < 0 = leaf number `leaf`
    int32_t prop, split, child, leaf;
};
struct EncGroup {          // one single-channel group whose samples are all known
    const int32_t *plane;  // device, w x h contiguous
    int32_t w, h, hshift, vshift;
    int32_t minval, maxval, zero, predictor;
    int32_t nrefs;
    EncRef refs[kMaxRefs];
};
struct RacEncState {       // RacOutput24 between two symbols: range, low, the delayed byte (-1 = none yet) and the run of 0xFF behind it
    uint32_t range, low;
    int32_t delayed, pending;
};

// ---- many groups in one launch pair (a batch of pictures) ----------------------------------------------------------------
struct EncJobDev {         // what the kernels see of one job; all pointers are device pointers
    EncGroup g;
    const EncNode *tree;
    int32_t n_nodes, pad;
    int32_t *guess, *leaf; // per pixel, written by k_enc_model_jobs, read by k_enc_rac_jobs
    uint16_t *leaves;      // n_leaves x 31 chances
    uint32_t *state;       // 8 words: RacEncState in / out, bytes emitted, overflow flag
    uint8_t *out;
    uint32_t out_cap, pad2;
    int64_t n;             // pixels
};
struct EncJob {            // host side of one job
    EncGroup g;                        // device pointers of the planes
    std::vector<EncNode> tree;
    int n_leaves = 1;
    uint16_t leaf_init[31];
    RacEncState state;                 // in: behind the tree; out: behind the last symbol
    std::vector<uint8_t> body;         // out: the bytes the coder emitted for the pixels
};
// Runs every job's context model (one launch, blockIdx.y = job) and then every job's coder (one wavefront per job).
int maniac_encode_jobs_gpu(std::vector<EncJob> &jobs, const uint16_t *pixel_table);

// grow-only device buffers of one encode call
struct EncScratch {
    int32_t *d_guess = nullptr, *d_leaf = nullptr;
    size_t pixel_cap = 0;
    uint8_t *d_bytes = nullptr;
    size_t bytes_cap = 0;
    EncNode *d_tree = nullptr;
    size_t tree_cap = 0;
    uint16_t *d_leaves = nullptr;
    size_t leaves_cap = 0;
    uint16_t *d_table = nullptr;      // the pixel coder's chance transition table (8192 entries)
    uint32_t *d_state = nullptr;      // RacEncState + byte count + overflow flag
    void release();
};

// Encodes every sample of the group in row-major order with the given tree (n_nodes >= 1) into the range coder whose state is
// *state (the tree has already gone into it on the host); appends the bytes the coder emits to `out` and leaves the state
// behind the last symbol (the caller flushes).  leaf_init = the 31 chances every leaf starts with (symbol.h:115-138),
// pixel_table = the 2 x 4096 transition table of the pixel coder.  Returns FUIFGPU_OK or FUIFGPU_E_HIP / FUIFGPU_E_NOMEM.
int maniac_encode_group_gpu(const EncGroup &g, const EncNode *tree, int n_nodes, int n_leaves, const uint16_t *leaf_init, const uint16_t *pixel_table,
                            RacEncState *state, std::vector<uint8_t> &out, EncScratch &scratch);

}  // namespace fuifgpu
