// fuif_amd/csrc/capi.hip -- the extern "C" layer of libfuifgpu.so (see include/fuifgpu.h).
//
// Host-side batch management for the MI355X FUIF decode path: staging of compressed streams in
// HBM, one entropy-kernel launch per batch, and the inverse-transform schedule replayed over
// chunks of images so that the TMP slab stays small next to the 288 GB of HBM the coefficient and
// output slabs are sized for.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fuifgpu.h"
#include "fuifgpu_internal.h"
#include "maniac_decode.h"
#include "transforms.h"

using namespace fuifgpu;

struct fuifgpu_plan {
    Plan plan;
};

struct fuifgpu_batch {
    Plan plan;
    int device = 0;                   // the HIP device the batch was created on: every call on the batch runs there (DeviceGuard)
    int n = 0;
    int n_loaded = 0;
    // device state
    uint8_t *d_blobs = nullptr;
    size_t blob_cap = 0;
    StreamJob *d_jobs = nullptr;
    ChannelGeom *d_geom = nullptr;
    ChannelMeta *d_meta = nullptr;
    int32_t *d_status = nullptr;
    uint32_t *d_consumed = nullptr;
    uint16_t *d_tables = nullptr;
    uint8_t *d_scratch = nullptr;
    size_t scratch_stride = 0, bfs_off = 0, leaves_off = 0, stack_off = 0, queue_off = 0, subtree_off = 0;
    int scratch_waves = 0;            // wavefronts d_scratch is sized for
    int max_waves[3] = {0, 0, 0};     // resident wavefronts the device holds in the kernel configurations: wide for two batches in flight / dense / wide for a launch alone
    int in_flight = 1;                // batches the host keeps in flight on this device (fuifgpu_batch_set_in_flight): picks the wide configuration
    int n_waves = 0, dense = 0, cfg = 1;   // persistent wavefronts and configuration (index into max_waves) of the next decode launch
    bool group_parallel = true;       // use group indices (index.cpp) when streams carry them
    Tile *d_tiles = nullptr;
    int tiles_cap = 0, n_tiles = 0;
    uint32_t *d_progress = nullptr, *d_group_start = nullptr;
    // scheduler state of the entropy kernel (maniac_decode.h), zeroed per launch, and the queue / image layout tables
    uint32_t *d_sched = nullptr, *d_layout = nullptr;
    size_t sched_words = 0, layout_cap = 0;
    int sched = 0, n_queues = 1, waves_per_simd = 4;
    uint8_t *d_ctx = nullptr;         // context areas of suspendable tiles (sched == 1)
    size_t ctx_bytes = 0;
    uint32_t ctx_units_per_queue = 0;
    std::vector<Tile> tiles;
    int max_nodes = kMaxNodes;
    coef_t *d_coef = nullptr;         // int16 samples: what the entropy kernel writes (fuifgpu_internal.h)
    int32_t *d_out = nullptr, *d_tmp = nullptr;
    int32_t *d_coef32 = nullptr;      // the coefficients of tmp_images images widened to int32: what the inverse transforms read and rewrite (a launch resource, like d_tmp)
    bool own_coef = false, own_out = false;
    int tmp_images = 0;
    PlaneRef *d_list = nullptr;
    int64_t *d_widen = nullptr;       // Plan::widen on the device: the coded planes the inverse kernels want as int32
    unsigned long long *d_prof = nullptr, *d_tile_log = nullptr;   // d_tile_log: only allocated once fuifgpu_batch_tile_log has been asked for
    int tile_log_cap = 0; bool want_tile_log = false;
    // host staging (pinned)
    uint8_t *h_blobs = nullptr;
    std::vector<StreamJob> jobs;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool decode_timed = false, transform_timed = false;
    bool no_out = false;              // fuifgpu_batch_create_streaming: no output slab; fuifgpu_batch_undo_transforms_to writes into caller memory
    std::vector<char> undone;         // ... which images of the current decode have been through it (each once: the channel metadata is rewritten)
    bool coef_consumed = false;   // undo_transforms has run on the current decode: several inverse steps work in place on the coefficients
    // A sibling (fuifgpu_batch_create_sibling) owns only what an UPLOAD writes -- stream bytes, tile lists, per-image results -- and
    // decodes with the primary's slabs, decoder scratch, context arenas and transform arena (everything a LAUNCH uses)
    fuifgpu_batch *share = nullptr;
    std::vector<fuifgpu_batch *> siblings;   // of a primary: told when it goes away
    bool orphan = false;                      // a sibling whose primary was destroyed first: every call is refused
};
static inline const fuifgpu_batch *launch_res(const fuifgpu_batch *b) { return b->share ? b->share : b; }

static thread_local std::string g_last_error;

static int hip_fail(hipError_t e, const char *what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return FUIFGPU_E_HIP;
}
static int fail_msg(int code, const char *what) {
    g_last_error = what;
    return code;
}
#define HIPCHK(call)                                      \
    do {                                                  \
        hipError_t e__ = (call);                          \
        if (e__ != hipSuccess) return hip_fail(e__, #call); \
    } while (0)

namespace fuifgpu {
// Supernodes a wavefront's scratch area holds.  A tree of n inner nodes needs at most (7n+5)/12 of them
// (M supernodes with children hold >= 6 inner nodes each, T without hold >= 1: inner >= 2T-1 and
// >= 6M+T); beyond kMaxSuper the kernel walks the remaining subtrees node by node, which keeps the
// area at ~5 MB per wavefront instead of 13 MB for a case no encoder in sight produces.
#ifndef FUIF_MAX_SUPER
#define FUIF_MAX_SUPER 4096
#endif
int maniac_max_supernodes(int max_nodes) { return (int)std::min<int64_t>(FUIF_MAX_SUPER, ((int64_t)max_nodes + 1) * 5 / 16 + 66); }
size_t maniac_scratch_bytes(int max_nodes, size_t *bfs_off, size_t *leaves_off, size_t *stack_off, size_t *queue_off, size_t *subtree_off) {
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    size_t nodes = up((size_t)(max_nodes + 1) * 8);
    size_t snodes = up((size_t)maniac_max_supernodes(max_nodes) * 512);
    size_t leaves = up((size_t)((max_nodes + 1) / 2 + 1) * kLeafStride * 2);
    size_t stack = up((size_t)kTreeStackDepth * 24);
    size_t queue = up((size_t)(maniac_max_supernodes(max_nodes) + 64) * 4);
    *bfs_off = nodes;
    *leaves_off = nodes + snodes;
    *stack_off = nodes + snodes + leaves;
    *queue_off = nodes + snodes + leaves + stack;
    *subtree_off = nodes + snodes + leaves + stack + queue;
    return nodes + snodes + leaves + stack + queue + up((size_t)(max_nodes + 1) * 2);
}
}  // namespace fuifgpu

extern "C" {

// A HIP device is per-thread state.  A batch lives on the device that was current when it was created (fuifgpu_set_device); every entry
// point that takes a batch switches the calling thread to that device for the duration of the call, so one host thread per device -- or one
// thread walking over the batches of several devices -- both work, and the caller's current device is what it was when the call returns.
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int want) {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != want && hipSetDevice(want) == hipSuccess) prev = cur;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define ON_BATCH_DEVICE(b) DeviceGuard device_guard__((b) ? (b)->device : 0)

const char *fuifgpu_strerror(int code) {
    switch (code) {
        case FUIFGPU_OK: return "ok";
        case FUIFGPU_E_NOT_FUIF: return "not a FUIF stream";
        case FUIFGPU_E_CORRUPT: return "corrupt header or transform list";
        case FUIFGPU_E_UNSUPPORTED: return "feature outside the MI355X hot-path scope";
        case FUIFGPU_E_ARG: return "invalid argument";
        case FUIFGPU_E_HIP: return "HIP runtime error";
        case FUIFGPU_E_MISMATCH: return "stream does not match the batch plan";
        case FUIFGPU_E_NOMEM: return "out of memory";
        default: return "unknown error";
    }
}
const char *fuifgpu_last_error(void) { return g_last_error.c_str(); }
int fuifgpu_abi_version(void) { return FUIFGPU_ABI_VERSION; }

void fuifgpu_build_chance_table(uint16_t *table8192, uint32_t alpha, int cut) { build_chance_table(table8192, alpha, cut); }

int fuifgpu_plan_create(const uint8_t *blob, size_t size, fuifgpu_plan **out) {
    if (!blob || !out) return FUIFGPU_E_ARG;
    fuifgpu_plan *p = new fuifgpu_plan();
    int r = parse_and_plan(blob, size, p->plan);
    if (r != FUIFGPU_OK) {
        g_last_error = p->plan.message;
        delete p;
        *out = nullptr;
        return r;
    }
    *out = p;
    return FUIFGPU_OK;
}
void fuifgpu_plan_destroy(fuifgpu_plan *plan) { delete plan; }

int fuifgpu_plan_info(const fuifgpu_plan *plan, fuifgpu_image_info *info) {
    if (!plan || !info) return FUIFGPU_E_ARG;
    const Plan &p = plan->plan;
    memset(info, 0, sizeof(*info));
    info->w = p.w; info->h = p.h; info->bit_depth = p.bit_depth; info->maxval = p.maxval; info->nb_channels = p.nb_channels;
    info->colormodel = p.colormodel; info->max_properties = p.max_properties; info->nb_frames = p.nb_frames;
    info->nb_transforms = (int)p.transforms.size(); info->nb_coded_channels = (int)p.coded.size();
    info->nb_output_channels = (int)p.outputs.size(); info->nb_ops = (int)p.ops.size();
    for (int s = 0; s < 5; s++) info->responsive_offsets[s] = p.responsive_offsets[s];
    info->data_start = (int)p.data_start;
    info->coef_elems = p.coef_elems; info->out_elems = p.out_elems; info->tmp_elems = p.tmp_elems;
    info->signature = p.signature;
    return FUIFGPU_OK;
}
int fuifgpu_plan_coded_channel(const fuifgpu_plan *plan, int index, fuifgpu_channel_desc *d) {
    if (!plan || !d || index < 0 || index >= (int)plan->plan.coded.size()) return FUIFGPU_E_ARG;
    const ChannelGeom &g = plan->plan.coded[index];
    d->w = g.w; d->h = g.h; d->hshift = g.hshift; d->vshift = g.vshift; d->hcshift = g.hcshift; d->vcshift = g.vcshift;
    d->component = g.component; d->reserved = 0; d->offset = g.coef_off;
    return FUIFGPU_OK;
}
int fuifgpu_plan_output_channel(const fuifgpu_plan *plan, int index, fuifgpu_channel_desc *d) {
    if (!plan || !d || index < 0 || index >= (int)plan->plan.outputs.size()) return FUIFGPU_E_ARG;
    const OutputChannel &o = plan->plan.outputs[index];
    d->w = o.plane.w; d->h = o.plane.h; d->hshift = o.hshift; d->vshift = o.vshift; d->hcshift = o.hcshift; d->vcshift = o.vcshift;
    d->component = o.component; d->reserved = 0; d->offset = o.plane.off;
    return FUIFGPU_OK;
}
int fuifgpu_plan_transform(const fuifgpu_plan *plan, int index, int32_t *id, int32_t *params_out, int cap, int32_t *nparams) {
    if (!plan || index < 0 || index >= (int)plan->plan.transforms.size()) return FUIFGPU_E_ARG;
    const TransformDesc &t = plan->plan.transforms[index];
    if (id) *id = t.id;
    if (nparams) *nparams = (int)t.params.size();
    if (params_out) for (int i = 0; i < (int)t.params.size() && i < cap; i++) params_out[i] = t.params[i];
    return FUIFGPU_OK;
}

// -------------------------------------------------------------------------------------------------
void fuifgpu_batch_destroy(fuifgpu_batch *b) {
    if (!b) return;
    ON_BATCH_DEVICE(b);
    // the documented order is "sibling first"; the other order must not leave a sibling launching with freed memory
    for (fuifgpu_batch *s : b->siblings) { s->share = nullptr; s->orphan = true; s->n_loaded = 0; s->d_coef = nullptr; s->d_out = nullptr; }
    if (b->share) {
        auto &v = b->share->siblings;
        v.erase(std::remove(v.begin(), v.end(), b), v.end());
    }
    hipFree(b->d_blobs); hipFree(b->d_jobs); hipFree(b->d_geom); hipFree(b->d_meta); hipFree(b->d_status); hipFree(b->d_consumed);
    hipFree(b->d_tables); hipFree(b->d_scratch); hipFree(b->d_tmp); hipFree(b->d_coef32); hipFree(b->d_list); hipFree(b->d_widen); hipFree(b->d_prof); hipFree(b->d_tile_log);
    hipFree(b->d_tiles); hipFree(b->d_progress); hipFree(b->d_group_start); hipFree(b->d_sched); hipFree(b->d_layout); hipFree(b->d_ctx);
    if (b->own_coef) hipFree(b->d_coef);
    if (b->own_out) hipFree(b->d_out);
    if (b->h_blobs) hipHostFree(b->h_blobs);
    for (int i = 0; i < 4; i++) if (b->ev[i]) hipEventDestroy(b->ev[i]);
    delete b;
}

static int32_t *const kNoOutSlab = reinterpret_cast<int32_t *>(~(uintptr_t)0);   // batch_create_impl: a batch without an output slab (streaming)
static int batch_create_impl(const Plan &plan_in, int n_images, size_t blob_capacity_bytes, coef_t *coef_ext, int32_t *out_ext,
                             int tmp_images, fuifgpu_batch *share, fuifgpu_batch **out) {
    if (!out || n_images < 1 || n_images > 65535) return FUIFGPU_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        g_last_error = "no HIP device visible: libfuifgpu has no CPU fallback";
        return FUIFGPU_E_HIP;
    }
    fuifgpu_batch *b = new fuifgpu_batch();
    if (share) b->device = share->device;                        // (a sibling lives where its primary lives: the caller holds that device, see below)
    else if (hipGetDevice(&b->device) != hipSuccess) b->device = 0;
    b->plan = plan_in;
    b->n = n_images;
    b->share = share;
    const Plan &p = b->plan;
    const int nch = (int)p.coded.size();
    b->blob_cap = blob_capacity_bytes + (size_t)n_images * 32 + 1024;  // + slack for the 256-byte read window
#define CHK(call)                                                        \
    do {                                                                 \
        hipError_t e__ = (call);                                         \
        if (e__ != hipSuccess) { int rc = hip_fail(e__, #call); fuifgpu_batch_destroy(b); return rc; } \
    } while (0)
    CHK(hipMalloc((void **)&b->d_blobs, b->blob_cap));
    CHK(hipMalloc((void **)&b->d_jobs, sizeof(StreamJob) * n_images));
    CHK(hipMalloc((void **)&b->d_geom, sizeof(ChannelGeom) * std::max(nch, 1)));
    CHK(hipMemcpy(b->d_geom, p.coded.data(), sizeof(ChannelGeom) * nch, hipMemcpyHostToDevice));
    CHK(hipMalloc((void **)&b->d_meta, sizeof(ChannelMeta) * (size_t)n_images * std::max(nch, 1)));
    CHK(hipMalloc((void **)&b->d_status, sizeof(int32_t) * n_images));
    CHK(hipMalloc((void **)&b->d_consumed, sizeof(uint32_t) * n_images));
    {
        std::vector<uint16_t> tables(16384);
        build_chance_table(tables.data(), 0xFFFFFFFFu / 19, 2);        // tree coder: maniac/compound.h:262
        build_chance_table(tables.data() + 8192, 0x0d000000u, 6);      // pixel coder: encoding/encoding.h:54-55
        CHK(hipMalloc((void **)&b->d_tables, tables.size() * 2));
        CHK(hipMemcpy(b->d_tables, tables.data(), tables.size() * 2, hipMemcpyHostToDevice));
    }
    b->scratch_stride = maniac_scratch_bytes(b->max_nodes, &b->bfs_off, &b->leaves_off, &b->stack_off, &b->queue_off, &b->subtree_off);
    b->max_waves[0] = maniac_max_waves(0);
    b->max_waves[1] = maniac_max_waves(1, &b->waves_per_simd);
    b->max_waves[2] = maniac_max_waves(2);
    if (const char *e = getenv("FUIFGPU_IN_FLIGHT")) b->in_flight = std::max(1, atoi(e));     // (tests and tools: what fuifgpu_batch_set_in_flight sets)
    if (b->max_waves[0] < 1 || b->max_waves[1] < 1 || b->max_waves[2] < 1) { g_last_error = "cannot query the device occupancy of the entropy kernel"; fuifgpu_batch_destroy(b); return FUIFGPU_E_HIP; }
    CHK(hipMalloc((void **)&b->d_progress, sizeof(uint32_t) * (size_t)n_images * std::max(nch, 1)));
    CHK(hipMalloc((void **)&b->d_group_start, sizeof(uint32_t) * (size_t)n_images * std::max(nch, 1)));
    if (coef_ext) b->d_coef = coef_ext;
    else { CHK(hipMalloc((void **)&b->d_coef, sizeof(coef_t) * (size_t)std::max<int64_t>(p.coef_elems, 1) * n_images)); b->own_coef = true; }
    if (out_ext == kNoOutSlab) { b->d_out = nullptr; b->no_out = true; }
    else if (out_ext) b->d_out = out_ext;
    else { CHK(hipMalloc((void **)&b->d_out, sizeof(int32_t) * (size_t)std::max<int64_t>(p.out_elems, 1) * n_images)); b->own_out = true; }
    if (tmp_images <= 0) {
        // default: keep the TMP slab + the widened coefficients of the images it serves under ~12 GiB
        int64_t per = (std::max<int64_t>(p.tmp_elems, 1) + std::max<int64_t>(p.coef_elems, 1)) * 4;
        tmp_images = (int)std::max<int64_t>(1, std::min<int64_t>(n_images, (12LL << 30) / per));
    }
    b->tmp_images = std::min(tmp_images, n_images);
    if (!share) {
        CHK(hipMalloc((void **)&b->d_tmp, sizeof(int32_t) * (size_t)std::max<int64_t>(p.tmp_elems, 1) * b->tmp_images));
        CHK(hipMalloc((void **)&b->d_coef32, sizeof(int32_t) * (size_t)std::max<int64_t>(p.coef_elems, 1) * b->tmp_images));
    }
    if (!p.idct_src.empty()) {
        CHK(hipMalloc((void **)&b->d_list, sizeof(PlaneRef) * p.idct_src.size()));
        CHK(hipMemcpy(b->d_list, p.idct_src.data(), sizeof(PlaneRef) * p.idct_src.size(), hipMemcpyHostToDevice));
    }
    if (!p.widen.empty()) {
        CHK(hipMalloc((void **)&b->d_widen, sizeof(int64_t) * p.widen.size()));
        CHK(hipMemcpy(b->d_widen, p.widen.data(), sizeof(int64_t) * p.widen.size(), hipMemcpyHostToDevice));
    }
    CHK(hipMalloc((void **)&b->d_prof, sizeof(unsigned long long) * 8 * n_images));
    CHK(hipMemset(b->d_prof, 0, sizeof(unsigned long long) * 8 * n_images));
    for (int i = 0; i < 4; i++) CHK(hipEventCreate(&b->ev[i]));
#undef CHK
    *out = b;
    return FUIFGPU_OK;
}

int fuifgpu_batch_create(const fuifgpu_plan *plan, int n_images, size_t blob_capacity_bytes, int16_t *coef_ext, int32_t *out_ext,
                         int tmp_images, fuifgpu_batch **out) {
    if (!plan) return FUIFGPU_E_ARG;
    return batch_create_impl(plan->plan, n_images, blob_capacity_bytes, coef_ext, out_ext, tmp_images, nullptr, out);
}

// The launch resources a sibling borrows must not move while it may be decoding: the documented pipeline runs an upload into one
// batch on a host thread while the other batch decodes.  When the first sibling is created the primary's decoder scratch and
// context arenas are therefore sized ONCE for the worst case -- as many wavefronts as the device holds (or the batch can ever
// have tiles), arenas for a full batch -- and are never reallocated while a sibling exists (fuifgpu_batch_upload clamps to them).
static size_t ctx_bytes_per_image() {
    size_t per_image = 16u << 20;
    if (const char *e = getenv("FUIFGPU_CTX_MB")) per_image = (size_t)std::max(1, atoi(e)) << 20;
    if (const char *e = getenv("FUIFGPU_CTX_KB")) per_image = (size_t)std::max(0, atoi(e)) << 10;   // tests: arenas that run out (pinned tiles)
    return per_image;
}
static int freeze_launch_resources(fuifgpu_batch *b) {
    const int64_t cap = std::max(std::max(b->max_waves[0], b->max_waves[1]), b->max_waves[2]);
    const int waves = (int)std::max<int64_t>(1, std::min<int64_t>(cap, (int64_t)b->n * std::max<int64_t>((int64_t)b->plan.coded.size(), 1)));
    if (waves > b->scratch_waves) {
        HIPCHK(hipDeviceSynchronize());
        hipFree(b->d_scratch); b->d_scratch = nullptr; b->scratch_waves = 0;
        HIPCHK(hipMalloc((void **)&b->d_scratch, b->scratch_stride * (size_t)waves));
        b->scratch_waves = waves;
    }
    size_t need = ctx_bytes_per_image() * (size_t)b->n;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) need = std::min(need, (free_b + b->ctx_bytes) / 2);
    need = need / 256 * 256;
    if (need > b->ctx_bytes) {
        HIPCHK(hipDeviceSynchronize());
        hipFree(b->d_ctx); b->d_ctx = nullptr; b->ctx_bytes = 0;
        HIPCHK(hipMalloc((void **)&b->d_ctx, std::max<size_t>(need, 256)));
        b->ctx_bytes = need;
    }
    return FUIFGPU_OK;
}

int fuifgpu_batch_create_streaming(const fuifgpu_plan *plan, int n_images, size_t blob_capacity_bytes, int tmp_images, fuifgpu_batch **out) {
    if (!plan) return FUIFGPU_E_ARG;
    return batch_create_impl(plan->plan, n_images, blob_capacity_bytes, nullptr, kNoOutSlab, tmp_images, nullptr, out);
}

int fuifgpu_batch_create_sibling(fuifgpu_batch *primary, size_t blob_capacity_bytes, fuifgpu_batch **out) {
    if (!primary || primary->share || primary->no_out) return FUIFGPU_E_ARG;
    ON_BATCH_DEVICE(primary);
    if (primary->siblings.empty()) { const int frc = freeze_launch_resources(primary); if (frc != FUIFGPU_OK) return frc; }
    const int rc = batch_create_impl(primary->plan, primary->n, blob_capacity_bytes, primary->d_coef, primary->d_out, primary->tmp_images, primary, out);
    if (rc == FUIFGPU_OK) primary->siblings.push_back(*out);
    return rc;
}

int fuifgpu_batch_upload(fuifgpu_batch *b, const uint8_t *const *blobs, const size_t *sizes, int n_images, int preview, void *stream) {
    ON_BATCH_DEVICE(b);
    if (!b || !blobs || !sizes || n_images < 1 || n_images > b->n || preview < -1 || preview > 4) return FUIFGPU_E_ARG;
    if (b->orphan) { g_last_error = "sibling batch: its primary has been destroyed"; return FUIFGPU_E_ARG; }
    hipStream_t st = (hipStream_t)stream;
    b->jobs.assign(n_images, StreamJob{});
    size_t off = 0;
    Plan tmp;
    // Distinct host blobs are copied H2D once; replicas (same host pointer) are duplicated D2D so a
    // 1024-image batch built from K distinct streams moves K streams over PCIe, not 1024.
    std::vector<std::pair<const uint8_t *, int>> seen;
    const int nch = (int)b->plan.coded.size();
    std::vector<std::vector<GroupEntry>> groups;   // per distinct stream: its tiles' first bytes / channels
    std::vector<int> group_of(n_images, -1);
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < n_images; i++) {
        if (sizes[i] > 0xFFFFFFF0ull) return FUIFGPU_E_ARG;
        int src = -1;
        for (auto &sp : seen) if (sp.first == blobs[i] && b->jobs[sp.second].blob_size == sizes[i]) { src = sp.second; break; }
        size_t padded = (sizes[i] + 15) / 16 * 16 + 16;
        if (off + padded + 256 > b->blob_cap) { g_last_error = "blob capacity exceeded"; return FUIFGPU_E_NOMEM; }  // the last 256-byte read window stays inside the allocation
        StreamJob &j = b->jobs[i];
        if (src < 0) {
            int r = parse_and_plan(blobs[i], sizes[i], tmp);
            if (r != FUIFGPU_OK) { g_last_error = tmp.message; return r; }
            if (tmp.signature != b->plan.signature) { g_last_error = "image " + std::to_string(i) + " has a different geometry/transform chain"; return FUIFGPU_E_MISMATCH; }
            std::vector<GroupEntry> idx;
            if (b->group_parallel) parse_index_trailer(blobs[i], sizes[i], tmp.data_start, nch, idx, nullptr);
            if (idx.empty()) idx.push_back(GroupEntry{(uint32_t)tmp.data_start, 0});
            groups.push_back(std::move(idx));
            group_of[i] = (int)groups.size() - 1;
            HIPCHK(hipMemsetAsync(b->d_blobs + off + (padded - 32), 0, 32, st));
            HIPCHK(hipMemcpyAsync(b->d_blobs + off, blobs[i], sizes[i], hipMemcpyHostToDevice, st));
            j.data_start = (uint32_t)tmp.data_start;
            j.limit = preview >= 0 ? (uint32_t)tmp.responsive_offsets[preview] : 0u;
            seen.emplace_back(blobs[i], i);
        } else {
            HIPCHK(hipMemcpyAsync(b->d_blobs + off, b->d_blobs + b->jobs[src].blob_off, padded, hipMemcpyDeviceToDevice, st));
            j.data_start = b->jobs[src].data_start;
            j.limit = b->jobs[src].limit;
            group_of[i] = group_of[src];
        }
        j.blob_off = off; j.blob_size = (uint32_t)sizes[i];
        j.flags = 0;
        off += padded;
    }
    HIPCHK(hipMemcpyAsync(b->d_jobs, b->jobs.data(), sizeof(StreamJob) * n_images, hipMemcpyHostToDevice, st));
    // Work list.  Which configuration runs is known from the tile count alone: more tiles than the wide configuration
    // has wavefronts -> dense (4 wavefronts per SIMD).
    size_t total_tiles = 0, deepest = 0;
    for (int i = 0; i < n_images; i++) total_tiles += groups[group_of[i]].size();
    for (auto &g : groups) deepest = std::max(deepest, g.size());
    const int wide_cfg = b->in_flight > 1 ? 0 : 2;      // a host with two batches in flight leaves room for the other launch's wavefronts (20 LDS supernodes: two per SIMD)
    b->dense = (int64_t)total_tiles > b->max_waves[wide_cfg] ? 1 : 0;
    b->cfg = b->dense ? 1 : wide_cfg;
    b->n_waves = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)total_tiles, b->max_waves[b->cfg]));
    // an image whose every tile holds exactly one non-empty channel (one single-channel group per tile) may have its tiles suspended
    std::vector<char> suspendable(groups.size(), 0);
    for (size_t gi = 0; gi < groups.size(); gi++) {
        const std::vector<GroupEntry> &g = groups[gi];
        bool ok = g.size() > 1;
        for (size_t k = 0; k < g.size() && ok; k++) {
            const int first = k == 0 ? 0 : g[k].first_channel, last = k + 1 < g.size() ? g[k + 1].first_channel - 1 : nch - 1;
            int nonempty = 0;
            for (int c = first; c <= last; c++) nonempty += (int64_t)b->plan.coded[c].w * b->plan.coded[c].h > 0 ? 1 : 0;
            ok = nonempty <= 1;
        }
        suspendable[gi] = ok ? 1 : 0;
    }
    int64_t image_samples = 0;
    for (int c = 0; c < nch; c++) image_samples += (int64_t)b->plan.coded[c].w * b->plan.coded[c].h;
    auto push_tile = [&](int i, size_t k) {
        const std::vector<GroupEntry> &g = groups[group_of[i]];
        if (k >= g.size()) return;
        Tile t;
        t.image = (uint32_t)i;
        t.start = g[k].start;
        t.first_channel = k == 0 ? 0 : g[k].first_channel;
        t.last_channel = k + 1 < g.size() ? g[k + 1].first_channel - 1 : nch - 1;
        t.end = k + 1 < g.size() ? g[k + 1].start : 0u;
        t.flags = suspendable[group_of[i]] ? kTileSuspendable : 0u;
        // size class = floor(log2(samples of the image / samples of the tile)): the few tiles that hold most of an image are its
        // critical path (one range coder each), the kernel runs them at a higher wavefront priority
        int64_t mine = 0;
        for (int c = (int)t.first_channel; c <= (int)t.last_channel; c++) mine += (int64_t)b->plan.coded[c].w * b->plan.coded[c].h;
        uint32_t cls = 15;
        if (mine > 0) { cls = 0; while (cls < 15 && (mine << (cls + 1)) <= image_samples) cls++; }
        t.flags |= cls << kTileSizeClassShift;
        b->tiles.push_back(t);
    };
    // Dense launches with more tiles than wavefronts use the context scheduler (maniac_decode.h, sched == 1): tiles image by
    // image in stream order, images dealt to one queue per CU, a tile that would wait for another tile's rows is
    // suspended instead of holding its wavefront (measured on 1024 x 4K: a quarter of all wavefront time was spent in
    // such waits, profiles/r2_tile_timeline_baseline.txt).  Otherwise -- and with FUIFGPU_TILE_ORDER=group, a diagnostic
    // -- one group-major list as in round 1: tile k of every image before tile k+1 of any.
    const char *ord = getenv("FUIFGPU_TILE_ORDER");
    b->sched = b->dense && (int64_t)total_tiles > b->n_waves && !(ord && !strcmp(ord, "group")) ? 1 : 0;
    b->tiles.clear();
    std::vector<uint32_t> layout;   // sched: q_img_begin [Q+1] | q_images [n] | img_tile_begin [n+1]
    if (!b->sched) {
        b->n_queues = 1;
        for (size_t k = 0; k < deepest; k++)
            for (int i = 0; i < n_images; i++) push_tile(i, k);
    } else {
        const int waves_per_cu = 4 * std::max(1, b->waves_per_simd);
        b->n_queues = std::max(1, std::min(n_images, b->n_waves / waves_per_cu));
        const int Q = b->n_queues;
        layout.resize((size_t)Q + 1 + n_images + n_images + 1);
        uint32_t *qib = layout.data(), *qim = qib + Q + 1, *itb = qim + n_images;
        // Images are dealt to the queues longest stream first, back and forth (0..Q-1, Q-1..0, ...): every queue gets the same
        // number of images and a similar number of bytes.  Dealing them in caller order put the copies of one picture on
        // one CU (a batch of K pictures replicated, K dividing Q), and the CUs holding the longest pictures finished 1.3 s
        // after the median one (profiles/r2_priority_and_balance.txt); the stream length is the best predictor of the
        // decoding time the host has.
        std::vector<int> by_size(n_images);
        for (int i = 0; i < n_images; i++) by_size[i] = i;
        std::stable_sort(by_size.begin(), by_size.end(), [&](int x, int y) { return sizes[x] > sizes[y]; });
        std::vector<std::vector<uint32_t>> dealt(Q);
        for (int k = 0; k < n_images; k++) {
            const int round = k / Q, at = k % Q;
            dealt[(round & 1) ? Q - 1 - at : at].push_back((uint32_t)by_size[k]);
        }
        uint32_t pos = 0;
        for (int q = 0; q < Q; q++) {
            qib[q] = pos;
            std::sort(dealt[q].begin(), dealt[q].end());   // caller order inside a queue
            for (uint32_t i : dealt[q]) qim[pos++] = i;
        }
        qib[Q] = pos;
        for (int i = 0; i < n_images; i++) {
            itb[i] = (uint32_t)b->tiles.size();
            for (size_t k = 0; k < groups[group_of[i]].size(); k++) push_tile(i, k);
        }
        itb[n_images] = (uint32_t)b->tiles.size();
    }
    b->n_tiles = (int)b->tiles.size();
    {
        // scheduler state, zeroed before every launch: q_head | done_total | statistics | started_total | heartbeat | cu claim table |
        // cu_alive | cu_live | cu_foreign | img_next | img_done | ctx_used | tile records
        const size_t words = 24 + (2 * 4096 + 1) + 3 * 4096 + 4 * 4096 + 2 * (size_t)n_images + 2 * (size_t)b->n_queues + (b->sched ? (size_t)b->n_tiles * (sizeof(TileRec) / 4) : 0);
        if (words > b->sched_words) {
            hipFree(b->d_sched); b->d_sched = nullptr; b->sched_words = 0;
            HIPCHK(hipMalloc((void **)&b->d_sched, words * 4));
            b->sched_words = words;
        }
        if (layout.size() > b->layout_cap) {
            hipFree(b->d_layout); b->d_layout = nullptr; b->layout_cap = 0;
            HIPCHK(hipMalloc((void **)&b->d_layout, layout.size() * 4));
            b->layout_cap = layout.size();
        }
        if (!layout.empty()) HIPCHK(hipMemcpyAsync(b->d_layout, layout.data(), layout.size() * 4, hipMemcpyHostToDevice, st));
        if (b->sched && !b->share) {
            // Context arenas: a suspendable tile keeps its supernodes and leaf chances in its image's queue arena (bump
            // allocation inside a launch).  16 MiB per image covers trees of ~2000 nodes on every tile of a 61-tile image
            // three times over (FUIFGPU_CTX_MB overrides); a tile that finds the arena full is simply not suspendable.
            const size_t per_image = ctx_bytes_per_image();
            const size_t images_per_queue = ((size_t)n_images + b->n_queues - 1) / b->n_queues;
            size_t per_queue = per_image * images_per_queue;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const size_t budget = (free_b + b->ctx_bytes) / 2;   // never more than half of what is left
                if (per_queue * (size_t)b->n_queues > budget) per_queue = budget / (size_t)b->n_queues;
            }
            // a primary with siblings never moves its arenas (a sibling may be decoding out of them right now): it lives with what
            // fuifgpu_batch_create_sibling froze
            if (!b->siblings.empty()) per_queue = std::min(per_queue, b->ctx_bytes / (size_t)b->n_queues);
            per_queue = std::min<size_t>(per_queue / 256 * 256, (size_t)0xFFFFFF00u / (size_t)b->n_queues * 256);
            b->ctx_units_per_queue = (uint32_t)(per_queue / 256);
            const size_t need = per_queue * (size_t)b->n_queues;
            if (need > b->ctx_bytes) {
                hipFree(b->d_ctx); b->d_ctx = nullptr; b->ctx_bytes = 0;
                HIPCHK(hipMalloc((void **)&b->d_ctx, std::max<size_t>(need, 256)));
                b->ctx_bytes = need;
            }
        }
        HIPCHK(hipStreamSynchronize(st));  // layout is a local
    }
    if (b->n_tiles > b->tiles_cap) {
        hipFree(b->d_tiles); b->d_tiles = nullptr; b->tiles_cap = 0;
        HIPCHK(hipMalloc((void **)&b->d_tiles, sizeof(Tile) * (size_t)b->n_tiles));
        b->tiles_cap = b->n_tiles;
    }
    if (b->n_tiles) HIPCHK(hipMemcpyAsync(b->d_tiles, b->tiles.data(), sizeof(Tile) * (size_t)b->n_tiles, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));  // b->tiles / b->jobs may be rebuilt by the next upload
    // one persistent wavefront per tile up to what the device holds at once; each owns a scratch area
    if (b->share) {
        // a sibling launches with the primary's decoder scratch, sized for the device's wavefront capacity when the sibling was created
        if (b->n_waves > b->share->scratch_waves) { g_last_error = "sibling batch: more wavefronts than the primary's decoder scratch was sized for"; return FUIFGPU_E_ARG; }
    } else if (b->n_waves > b->scratch_waves) {
        if (!b->siblings.empty()) { g_last_error = "primary batch with siblings: decoder scratch cannot grow (internal sizing error)"; return FUIFGPU_E_ARG; }
        hipFree(b->d_scratch); b->d_scratch = nullptr; b->scratch_waves = 0;
        HIPCHK(hipMalloc((void **)&b->d_scratch, b->scratch_stride * (size_t)b->n_waves));
        b->scratch_waves = b->n_waves;
    }
    b->n_loaded = n_images;
    if (getenv("FUIFGPU_VERBOSE"))
        fprintf(stderr, "fuifgpu: %d images, %d tiles, %s configuration, %d persistent wavefronts (%d per SIMD), %d queues, context scheduler %s\n", n_images, b->n_tiles,
                b->cfg == 1 ? "dense" : b->cfg == 2 ? "wide" : "wide (two batches in flight)", b->n_waves, b->waves_per_simd, b->n_queues, b->sched ? "on" : "off");
    return FUIFGPU_OK;
}

int fuifgpu_batch_decode(fuifgpu_batch *b, void *stream) {
    ON_BATCH_DEVICE(b);
    if (!b || b->n_loaded < 1) return FUIFGPU_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int nch = (int)b->plan.coded.size();
    HIPCHK(hipMemsetAsync(b->d_meta, 0, sizeof(ChannelMeta) * (size_t)b->n_loaded * std::max(nch, 1), st));
    // every word the tiles poll or accumulate into is zeroed before every launch
    HIPCHK(hipMemsetAsync(b->d_progress, 0, sizeof(uint32_t) * (size_t)b->n_loaded * std::max(nch, 1), st));
    HIPCHK(hipMemsetAsync(b->d_group_start, 0, sizeof(uint32_t) * (size_t)b->n_loaded * std::max(nch, 1), st));
    HIPCHK(hipMemsetAsync(b->d_sched, 0, b->sched_words * 4, st));
    HIPCHK(hipMemsetAsync(b->d_status, 0, sizeof(int32_t) * b->n_loaded, st));
    HIPCHK(hipMemsetAsync(b->d_consumed, 0, sizeof(uint32_t) * b->n_loaded, st));
    HIPCHK(hipMemsetAsync(b->d_prof, 0, sizeof(unsigned long long) * 8 * b->n_loaded, st));
    DecodeParams P{};
    P.blobs = b->d_blobs; P.jobs = b->d_jobs; P.n_images = b->n_loaded; P.n_channels = nch; P.geom = b->d_geom;
    P.coef = b->d_coef; P.coef_stride = b->plan.coef_elems; P.meta = b->d_meta; P.status = b->d_status; P.consumed = b->d_consumed;
    const fuifgpu_batch *r = launch_res(b);
    P.tables = b->d_tables; P.scratch = r->d_scratch; P.scratch_stride = b->scratch_stride; P.bfs_off = b->bfs_off; P.leaves_off = b->leaves_off;
    P.stack_off = b->stack_off; P.queue_off = b->queue_off; P.subtree_off = b->subtree_off; P.max_properties = b->plan.max_properties; P.max_nodes = b->max_nodes; P.max_super = maniac_max_supernodes(b->max_nodes); P.prof = b->d_prof;
    if (b->want_tile_log && b->tile_log_cap < b->n_tiles) {
        hipFree(b->d_tile_log); b->d_tile_log = nullptr; b->tile_log_cap = 0;
        HIPCHK(hipMalloc((void **)&b->d_tile_log, sizeof(unsigned long long) * 4 * (size_t)b->n_tiles));
        b->tile_log_cap = b->n_tiles;
    }
    P.tile_log = b->want_tile_log ? b->d_tile_log : nullptr;   // (only -DFUIF_STATS / -DFUIF_PROF / -DFUIF_TILELOG kernels write it)
    if (P.tile_log) HIPCHK(hipMemsetAsync(b->d_tile_log, 0, sizeof(unsigned long long) * 4 * (size_t)b->n_tiles, st));   // the running time accumulates over a tile's run segments
    P.tiles = b->d_tiles; P.n_tiles = b->n_tiles; P.sched = b->sched; P.n_queues = b->n_queues;
    {
        uint32_t *w = b->d_sched;
        P.q_head = w; P.done_total = w + 1; P.sched_stats = reinterpret_cast<unsigned long long *>(w + 2); P.started_total = w + 18; P.heartbeat = w + 19;
        P.ctx_used = reinterpret_cast<unsigned long long *>(w + 20);   // (byte offset 80: 8-byte aligned)
        w += 24;
        P.yield_slack = 4;   // (round 4, profiles/r4_scheduler_knobs.txt: 4 -> 7.30 s, 8 -> 7.37 s, 16 -> 7.60 s, 32 -> 7.83 s on the trimmed kernel)
        if (const char *e = getenv("FUIFGPU_YIELD_SLACK")) P.yield_slack = (uint32_t)std::max(0, atoi(e));
        P.prio_base = kDefaultPrioBase;   // size classes <= base run at wavefront priority 3, base+1 at 2, base+2 at 1; negative: all 0
        if (const char *e = getenv("FUIFGPU_PRIO_BASE")) P.prio_base = atoi(e);
        P.simd_claim = w; w += 2 * 4096 + 1;
        P.cu_alive = w; w += 4096; P.cu_live = w; w += 4096; P.cu_foreign = w; w += 4096;
        P.simd_long = w; w += 4 * 4096;
        P.long_per_simd = 3;
        if (const char *e = getenv("FUIFGPU_LONG_PER_SIMD")) P.long_per_simd = std::max(0, atoi(e));
        P.img_next = w; w += b->n_loaded;
        P.img_done = w; w += b->n_loaded;
        w += b->n_queues;   // (rounds 2-5: one bump counter per queue)
        P.q_turn = w; w += b->n_queues;
        P.tile_rec = reinterpret_cast<TileRec *>(w);
        P.q_img_begin = b->d_layout; P.q_images = b->d_layout + b->n_queues + 1; P.img_tile_begin = b->d_layout + b->n_queues + 1 + b->n_loaded;
        P.ctx_scratch = r->d_ctx;
        P.ctx_units_per_queue = b->share ? (uint32_t)std::min<size_t>(r->ctx_bytes / (size_t)std::max(b->n_queues, 1) / 256, (size_t)0xFFFFFF00u / (size_t)std::max(b->n_queues, 1)) : b->ctx_units_per_queue;
    } P.progress = b->d_progress; P.group_start = b->d_group_start;
    HIPCHK(hipEventRecord(b->ev[0], st));
    launch_maniac_decode(P, b->n_waves, b->cfg, b->n_tiles > b->n_loaded ? 1 : 0, st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(b->ev[1], st));
    b->decode_timed = true;
    b->coef_consumed = false;
    b->undone.clear();
    return FUIFGPU_OK;
}

static int undo_range(fuifgpu_batch *b, int first, int count, int32_t *out_base, hipStream_t st) {
    const Plan &p = b->plan;
    const int nch = (int)p.coded.size();
    const fuifgpu_batch *r = launch_res(b);
    for (int i0 = first; i0 < first + count; i0 += r->tmp_images) {
        const int cnt = std::min(r->tmp_images, first + count - i0);
        Bases bases{};
        // The coded planes some kernel reads as int32 (or rewrites: dequantisation, Approximate, the match transforms) are copied,
        // widened, into the chunk's int32 coefficient copy (Plan::widen); Squeeze residuals -- nearly all coded samples of a Squeeze
        // chain -- are read as int16 straight from the slab the entropy kernel wrote (Op::r16).  The slab itself is never written here.
        {
            int64_t widest = 0;
            for (size_t k = 1; k < p.widen.size(); k += 2) widest = std::max(widest, p.widen[k]);
            launch_widen_planes(b->d_coef + (int64_t)i0 * p.coef_elems, r->d_coef32, p.coef_elems, b->d_widen, (int)(p.widen.size() / 2), widest, cnt, st);
        }
        bases.base[BUF_COEF] = r->d_coef32; bases.stride[BUF_COEF] = p.coef_elems;
        bases.c16 = b->d_coef + (int64_t)i0 * p.coef_elems;
        bases.base[BUF_OUT] = out_base + (int64_t)(i0 - first) * p.out_elems; bases.stride[BUF_OUT] = p.out_elems;
        bases.base[BUF_TMP] = r->d_tmp; bases.stride[BUF_TMP] = p.tmp_elems;
        for (const Op &op : p.ops) launch_op(op, bases, b->d_list, b->d_meta, nch, i0, cnt, st, b->d_status);
    }
    HIPCHK(hipGetLastError());
    return FUIFGPU_OK;
}

int fuifgpu_batch_undo_transforms_to(fuifgpu_batch *b, int first_image, int n_images, int32_t *out_device, void *stream) {
    ON_BATCH_DEVICE(b);
    if (!b || b->orphan || !out_device || first_image < 0 || n_images < 1 || first_image + n_images > b->n_loaded) return FUIFGPU_E_ARG;
    if (b->coef_consumed) { g_last_error = "fuifgpu_batch_undo_transforms_to: fuifgpu_batch_undo_transforms already ran on this decode"; return FUIFGPU_E_ARG; }
    if ((int)b->undone.size() != b->n_loaded) b->undone.assign((size_t)b->n_loaded, 0);
    for (int i = first_image; i < first_image + n_images; i++)
        if (b->undone[i]) { g_last_error = "fuifgpu_batch_undo_transforms_to: an image of the range has been through the inverse transforms already (once per decode)"; return FUIFGPU_E_ARG; }
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipEventRecord(b->ev[2], st));
    const int rc = undo_range(b, first_image, n_images, out_device, st);
    // (marked only once the range has been queued: a call that failed -- a HIP error, after which the decode's metadata is not to be trusted anyway -- does not
    // make the range unreachable for the caller's retry after the next decode)
    if (rc != FUIFGPU_OK) return rc;
    for (int i = first_image; i < first_image + n_images; i++) b->undone[i] = 1;
    HIPCHK(hipEventRecord(b->ev[3], st));
    b->transform_timed = true;
    return FUIFGPU_OK;
}

int fuifgpu_batch_undo_transforms(fuifgpu_batch *b, void *stream) {
    ON_BATCH_DEVICE(b);
    if (!b || b->n_loaded < 1) return FUIFGPU_E_ARG;
    if (b->no_out) { g_last_error = "fuifgpu_batch_undo_transforms: a streaming batch has no output slab (fuifgpu_batch_undo_transforms_to)"; return FUIFGPU_E_ARG; }
    for (char u : b->undone) if (u) { g_last_error = "fuifgpu_batch_undo_transforms: part of this decode went through fuifgpu_batch_undo_transforms_to already"; return FUIFGPU_E_ARG; }
    // The inverse kernels work on a widened COPY of the coefficients (round 4), so the slab itself stays as decoded; but
    // Approximate rewrites ChannelMeta::q, which a second pass over the same decode would apply twice
    if (b->coef_consumed) { g_last_error = "fuifgpu_batch_undo_transforms: already run on this decode (it rewrites the channel metadata); decode again first"; return FUIFGPU_E_ARG; }
    b->coef_consumed = true;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipEventRecord(b->ev[2], st));
    const int rc = undo_range(b, 0, b->n_loaded, b->d_out, st);
    if (rc != FUIFGPU_OK) return rc;
    HIPCHK(hipEventRecord(b->ev[3], st));
    b->transform_timed = true;
    return FUIFGPU_OK;
}

int fuifgpu_batch_sync(fuifgpu_batch *b, void *stream) {
    ON_BATCH_DEVICE(b);
    if (!b) return FUIFGPU_E_ARG;
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return FUIFGPU_OK;
}

int fuifgpu_batch_status(fuifgpu_batch *b, int32_t *status, uint32_t *bytes_consumed) {
    ON_BATCH_DEVICE(b);
    if (!b || b->n_loaded < 1) return FUIFGPU_E_ARG;
    HIPCHK(hipDeviceSynchronize());
    if (status) HIPCHK(hipMemcpy(status, b->d_status, sizeof(int32_t) * b->n_loaded, hipMemcpyDeviceToHost));
    if (bytes_consumed) HIPCHK(hipMemcpy(bytes_consumed, b->d_consumed, sizeof(uint32_t) * b->n_loaded, hipMemcpyDeviceToHost));
    return FUIFGPU_OK;
}

int fuifgpu_batch_set_in_flight(fuifgpu_batch *b, int n_batches) {
    if (!b || n_batches < 1) return FUIFGPU_E_ARG;
    b->in_flight = n_batches;
    return FUIFGPU_OK;
}

int fuifgpu_batch_set_group_parallel(fuifgpu_batch *b, int enable) {
    if (!b) return FUIFGPU_E_ARG;
    b->group_parallel = enable != 0;
    return FUIFGPU_OK;
}

int fuifgpu_batch_group_index(fuifgpu_batch *b, int image, int32_t *first_channel, uint32_t *start, int cap, int *n_groups) {
    ON_BATCH_DEVICE(b);
    if (!b || image < 0 || image >= b->n_loaded || !n_groups) return FUIFGPU_E_ARG;
    const int nch = (int)b->plan.coded.size();
    std::vector<uint32_t> gs(std::max(nch, 1));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(gs.data(), b->d_group_start + (size_t)image * nch, sizeof(uint32_t) * nch, hipMemcpyDeviceToHost));
    int n = 0;
    for (int c = 0; c < nch; c++) {
        if (!gs[c]) continue;
        if (n < cap) {
            if (first_channel) first_channel[n] = c;
            if (start) start[n] = gs[c] - 1u;
        }
        n++;
    }
    *n_groups = n;
    return FUIFGPU_OK;
}

int fuifgpu_batch_channel_meta(fuifgpu_batch *b, int image, int32_t *meta4) {
    ON_BATCH_DEVICE(b);
    if (!b || image < 0 || image >= b->n_loaded || !meta4) return FUIFGPU_E_ARG;
    const int nch = (int)b->plan.coded.size();
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(meta4, b->d_meta + (size_t)image * nch, sizeof(ChannelMeta) * nch, hipMemcpyDeviceToHost));
    return FUIFGPU_OK;
}

int16_t *fuifgpu_batch_coef_ptr(fuifgpu_batch *b, int image) { return (b && !b->orphan && image >= 0 && image < b->n) ? b->d_coef + (int64_t)image * b->plan.coef_elems : nullptr; }
int32_t *fuifgpu_batch_out_ptr(fuifgpu_batch *b, int image) { return (b && !b->orphan && !b->no_out && image >= 0 && image < b->n) ? b->d_out + (int64_t)image * b->plan.out_elems : nullptr; }

int fuifgpu_batch_download_coef(fuifgpu_batch *b, int image, int32_t *host, void *stream) {
    ON_BATCH_DEVICE(b);
    if (!b || image < 0 || image >= b->n || !host || b->orphan) return FUIFGPU_E_ARG;
    // the slab holds int16 samples; the caller gets them as int32, like every other plane of the interface
    const size_t n = (size_t)b->plan.coef_elems;
    std::vector<coef_t> narrow(n);
    HIPCHK(hipMemcpyAsync(narrow.data(), fuifgpu_batch_coef_ptr(b, image), sizeof(coef_t) * n, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    for (size_t k = 0; k < n; k++) host[k] = narrow[k];
    return FUIFGPU_OK;
}
int fuifgpu_batch_download_out(fuifgpu_batch *b, int image, int32_t *host, void *stream) {
    ON_BATCH_DEVICE(b);
    if (!b || image < 0 || image >= b->n || !host || b->orphan || b->no_out) return FUIFGPU_E_ARG;
    HIPCHK(hipMemcpyAsync(host, fuifgpu_batch_out_ptr(b, image), sizeof(int32_t) * (size_t)b->plan.out_elems, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return FUIFGPU_OK;
}

// ---- packed output (export/write_pam.h:29-160, the part that turns planes into the bytes of a PNM/PAM) ------
static int packed_layout(const Plan &p, int components, PackedPlanes *pp, int *bps) {
    const int nout = (int)p.outputs.size();
    if (components <= 0) components = std::min(nout, 4);
    if (components > nout || components > 5) return FUIFGPU_E_ARG;
    pp->n = components;
    for (int c = 0; c < components; c++) {
        const OutputChannel &o = p.outputs[c];
        if (o.plane.w < p.w || o.plane.h < p.h) return FUIFGPU_E_ARG;   // write_pam.h:50-54 refuses such channels too
        pp->p[c] = o.plane;
    }
    *bps = p.maxval > 255 ? 2 : 1;
    return FUIFGPU_OK;
}
size_t fuifgpu_plan_packed_bytes(const fuifgpu_plan *plan, int components) {
    if (!plan) return 0;
    PackedPlanes pp; int bps = 1;
    if (packed_layout(plan->plan, components, &pp, &bps) != FUIFGPU_OK) return 0;
    return (size_t)plan->plan.w * plan->plan.h * pp.n * bps;
}
int fuifgpu_batch_pack_out(fuifgpu_batch *b, int first_image, int n_images, int components, uint8_t *dst_device, void *stream) {
    ON_BATCH_DEVICE(b);
    if (!b || !dst_device || first_image < 0 || n_images < 1 || first_image + n_images > b->n_loaded || b->no_out) return FUIFGPU_E_ARG;
    const Plan &p = b->plan;
    PackedPlanes pp; int bps = 1;
    int rc = packed_layout(p, components, &pp, &bps);
    if (rc != FUIFGPU_OK) return rc;
    Bases bases{};
    bases.base[BUF_COEF] = nullptr; bases.stride[BUF_COEF] = 0;   // final planes always live in the output slab (plan.cpp finalize())
    bases.base[BUF_OUT] = b->d_out + (int64_t)first_image * p.out_elems; bases.stride[BUF_OUT] = p.out_elems;
    bases.base[BUF_TMP] = launch_res(b)->d_tmp; bases.stride[BUF_TMP] = p.tmp_elems;
    launch_pack(bases, pp, p.w, p.h, p.minval, p.maxval, bps, dst_device, (int64_t)p.w * p.h * pp.n * bps, n_images, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return FUIFGPU_OK;
}
int fuifgpu_batch_download_packed(fuifgpu_batch *b, int image, int components, uint8_t *host, void *stream) {
    ON_BATCH_DEVICE(b);
    if (!b || !host || image < 0 || image >= b->n_loaded) return FUIFGPU_E_ARG;
    PackedPlanes pp; int bps = 1;
    int rc = packed_layout(b->plan, components, &pp, &bps);
    if (rc != FUIFGPU_OK) return rc;
    const size_t bytes = (size_t)b->plan.w * b->plan.h * pp.n * bps;
    uint8_t *staging = nullptr;
    HIPCHK(hipMalloc((void **)&staging, bytes ? bytes : 1));
    rc = fuifgpu_batch_pack_out(b, image, 1, components, staging, stream);
    hipError_t e = hipSuccess;
    if (rc == FUIFGPU_OK) {
        e = hipMemcpyAsync(host, staging, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream);
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    }
    hipFree(staging);
    if (rc != FUIFGPU_OK) return rc;
    if (e != hipSuccess) return hip_fail(e, "download_packed");
    return FUIFGPU_OK;
}

int fuifgpu_plane_checksums(const int32_t *planes_device, int64_t elems_per_image, int64_t image_stride, int n_images, uint64_t *sums_device, void *stream) {
    if (!planes_device || !sums_device || elems_per_image < 0 || image_stride < elems_per_image || n_images < 1 || n_images > 65535 ||
        ((uintptr_t)planes_device & 15) || (image_stride & 3)) return FUIFGPU_E_ARG;
    launch_plane_checksums(planes_device, elems_per_image, image_stride, n_images, (unsigned long long *)sums_device, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return FUIFGPU_OK;
}

int fuifgpu_batch_last_timing(fuifgpu_batch *b, float *decode_ms, float *transform_ms) {
    ON_BATCH_DEVICE(b);
    if (!b) return FUIFGPU_E_ARG;
    if (decode_ms) {
        *decode_ms = -1.f;
        if (b->decode_timed) { HIPCHK(hipEventSynchronize(b->ev[1])); HIPCHK(hipEventElapsedTime(decode_ms, b->ev[0], b->ev[1])); }
    }
    if (transform_ms) {
        *transform_ms = -1.f;
        if (b->transform_timed) { HIPCHK(hipEventSynchronize(b->ev[3])); HIPCHK(hipEventElapsedTime(transform_ms, b->ev[2], b->ev[3])); }
    }
    return FUIFGPU_OK;
}

// diagnostic: per-stream phase cycle counters of the last decode (only filled by -DFUIF_PROF builds)
int fuifgpu_batch_profile(fuifgpu_batch *b, uint64_t *out8_per_image) {
    ON_BATCH_DEVICE(b);
    if (!b || !out8_per_image || b->n_loaded < 1) return FUIFGPU_E_ARG;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out8_per_image, b->d_prof, sizeof(unsigned long long) * 8 * b->n_loaded, hipMemcpyDeviceToHost));
    return FUIFGPU_OK;
}

// diagnostic: schedule of the last decode launch.  The first call (cap 0 is fine) switches logging on for later launches.
// Only the -DFUIF_STATS build of the library records it (the release kernel carries no statistics): FUIFGPU_E_UNSUPPORTED otherwise.
int fuifgpu_batch_tile_log(fuifgpu_batch *b, uint64_t *out4_per_tile, int cap, int *n_tiles) {
    ON_BATCH_DEVICE(b);
    if (!b || !n_tiles) return FUIFGPU_E_ARG;
#if !defined(FUIF_STATS) && !defined(FUIF_PROF) && !defined(FUIF_TILELOG)
    *n_tiles = 0;
    g_last_error = "fuifgpu_batch_tile_log: this build of libfuifgpu carries no scheduler statistics (build with -DFUIF_STATS)";
    return FUIFGPU_E_UNSUPPORTED;
#endif
    const bool had = b->want_tile_log && b->d_tile_log && b->tile_log_cap >= b->n_tiles;
    b->want_tile_log = true;
    *n_tiles = had ? b->n_tiles : 0;
    if (!had || !out4_per_tile) return FUIFGPU_OK;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out4_per_tile, b->d_tile_log, sizeof(unsigned long long) * 4 * (size_t)std::min(cap, b->n_tiles), hipMemcpyDeviceToHost));
    return FUIFGPU_OK;
}

// diagnostic: scheduler counters of the last dense launch {idle ticks (100 MHz) summed over wavefronts, tiles picked up, suspensions, ticks spent picking, ticks spent spinning inside tiles, suspendable tiles that found the arena full}
int fuifgpu_batch_sched_stats(fuifgpu_batch *b, uint64_t *out8) {
    ON_BATCH_DEVICE(b);
    if (!b || !out8 || !b->d_sched) return FUIFGPU_E_ARG;
#ifndef FUIF_STATS
    g_last_error = "fuifgpu_batch_sched_stats: this build of libfuifgpu carries no scheduler statistics (build with -DFUIF_STATS)";
    return FUIFGPU_E_UNSUPPORTED;
#endif
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out8, b->d_sched + 2, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return FUIFGPU_OK;
}

// ---- device memory for callers that are not HIP programs themselves (the C++ boundary layer is compiled with g++) ---
// ---- device selection (one node, several GPUs: images are independent units) ---------------------------------------
int fuifgpu_device_count(int *n_devices) {
    if (!n_devices) return FUIFGPU_E_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { *n_devices = 0; g_last_error = "no HIP device visible: libfuifgpu has no CPU fallback"; return FUIFGPU_E_HIP; }
    *n_devices = n;
    return FUIFGPU_OK;
}
int fuifgpu_set_device(int device) {
    int n = 0;
    const int rc = fuifgpu_device_count(&n);
    if (rc != FUIFGPU_OK) return rc;
    if (device < 0 || device >= n) { g_last_error = "fuifgpu_set_device: no such device"; return FUIFGPU_E_ARG; }
    HIPCHK(hipSetDevice(device));
    return FUIFGPU_OK;
}
int fuifgpu_get_device(int *device) {
    if (!device) return FUIFGPU_E_ARG;
    HIPCHK(hipGetDevice(device));
    return FUIFGPU_OK;
}
int fuifgpu_batch_device(const fuifgpu_batch *b, int *device) {
    if (!b || !device) return FUIFGPU_E_ARG;
    *device = b->device;
    return FUIFGPU_OK;
}
// the final gather's building block inside ONE process: bytes from one GPU's memory into another's over xGMI (peer access is enabled on
// first use; where the platform has none the runtime stages the copy through the host).  Asynchronous on `stream` of the CURRENT device.
int fuifgpu_peer_copy(void *dst_device_ptr, int dst_device, const void *src_device_ptr, int src_device, size_t bytes, void *stream) {
    if ((!dst_device_ptr || !src_device_ptr) && bytes) return FUIFGPU_E_ARG;
    int n = 0;
    const int rc = fuifgpu_device_count(&n);
    if (rc != FUIFGPU_OK) return rc;
    if (dst_device < 0 || dst_device >= n || src_device < 0 || src_device >= n) { g_last_error = "fuifgpu_peer_copy: no such device"; return FUIFGPU_E_ARG; }
    if (!bytes) return FUIFGPU_OK;
    if (dst_device == src_device) {
        DeviceGuard g(dst_device);
        HIPCHK(hipMemcpyAsync(dst_device_ptr, src_device_ptr, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return FUIFGPU_OK;
    }
    {
        DeviceGuard g(dst_device);
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dst_device, src_device) == hipSuccess && can) {
            const hipError_t e = hipDeviceEnablePeerAccess(src_device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();   // not fatal: the copy below still works, staged
            else (void)hipGetLastError();
        }
    }
    HIPCHK(hipMemcpyPeerAsync(dst_device_ptr, dst_device, src_device_ptr, src_device, bytes, (hipStream_t)stream));
    return FUIFGPU_OK;
}

void *fuifgpu_dev_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { g_last_error = "hipMalloc failed (no HIP device or out of memory): libfuifgpu has no CPU fallback"; return nullptr; }
    return p;
}
void fuifgpu_dev_free(void *p) { if (p) hipFree(p); }
int fuifgpu_dev_mem_info(size_t *free_bytes, size_t *total_bytes) {
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return FUIFGPU_OK;
}
int fuifgpu_dev_upload(void *dst_device, const void *src_host, size_t bytes) {
    if ((!dst_device || !src_host) && bytes) return FUIFGPU_E_ARG;
    if (bytes) HIPCHK(hipMemcpy(dst_device, src_host, bytes, hipMemcpyHostToDevice));
    return FUIFGPU_OK;
}
int fuifgpu_dev_download(void *dst_host, const void *src_device, size_t bytes) {
    if ((!dst_host || !src_device) && bytes) return FUIFGPU_E_ARG;
    if (bytes) HIPCHK(hipMemcpy(dst_host, src_device, bytes, hipMemcpyDeviceToHost));   // synchronises with the null stream's kernels
    return FUIFGPU_OK;
}

// ---- single-transform entry points -----------------------------------------------------------
static Op raw_op(int kind) {
    Op op{};
    op.kind = kind;
    return op;
}
static PlaneRef raw_plane(int buf, int w, int h) {
    PlaneRef p{};
    p.buf = buf; p.w = w; p.h = h; p.qsrc = -1; p.off = 0;
    return p;
}

int fuifgpu_inv_hsqueeze(const int32_t *avg, int w1, const int32_t *res, int w2, int h, int32_t *out, int n_planes, int64_t sa, int64_t sr,
                         int64_t so, void *stream) {
    if (!avg || !out || (!res && w2 > 0) || w1 < 1 || h < 1 || w1 - w2 < 0 || w1 - w2 > 1 || n_planes < 1 || n_planes > 65535) return FUIFGPU_E_ARG;
    Bases b{};
    b.base[0] = const_cast<int32_t *>(avg); b.stride[0] = sa;
    b.base[1] = const_cast<int32_t *>(res ? res : avg); b.stride[1] = sr;
    b.base[2] = out; b.stride[2] = so;
    Op op = raw_op(OP_HSQUEEZE);
    op.src[0] = raw_plane(0, w1, h); op.src[1] = raw_plane(1, w2, h); op.dst[0] = raw_plane(2, w1 + w2, h);
    launch_op(op, b, nullptr, nullptr, 0, 0, n_planes, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return FUIFGPU_OK;
}
int fuifgpu_inv_vsqueeze(const int32_t *avg, int h1, const int32_t *res, int h2, int w, int32_t *out, int n_planes, int64_t sa, int64_t sr,
                         int64_t so, void *stream) {
    if (!avg || !out || (!res && h2 > 0) || h1 < 1 || w < 1 || h1 - h2 < 0 || h1 - h2 > 1 || n_planes < 1 || n_planes > 65535) return FUIFGPU_E_ARG;
    Bases b{};
    b.base[0] = const_cast<int32_t *>(avg); b.stride[0] = sa;
    b.base[1] = const_cast<int32_t *>(res ? res : avg); b.stride[1] = sr;
    b.base[2] = out; b.stride[2] = so;
    Op op = raw_op(OP_VSQUEEZE);
    op.src[0] = raw_plane(0, w, h1); op.src[1] = raw_plane(1, w, h2); op.dst[0] = raw_plane(2, w, h1 + h2);
    launch_op(op, b, nullptr, nullptr, 0, 0, n_planes, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return FUIFGPU_OK;
}
static int color_raw(int kind, int32_t *c0, int32_t *c1, int32_t *c2, int w, int h, int p0, int p1, int p2, int minval, int maxval, void *stream) {
    if (!c0 || !c1 || !c2 || w < 1 || h < 1 || p0 < w || p1 < w || p2 < w) return FUIFGPU_E_ARG;
    Bases b{};
    b.base[0] = c0; b.base[1] = c1; b.base[2] = c2;
    b.stride[0] = b.stride[1] = b.stride[2] = 0;
    Op op = raw_op(kind);
    op.src[0] = op.dst[0] = raw_plane(0, p0, h);
    op.src[1] = op.dst[1] = raw_plane(1, p1, h);
    op.src[2] = op.dst[2] = raw_plane(2, p2, h);
    op.p0 = w; op.p1 = h; op.lo = minval; op.hi = maxval;
    launch_op(op, b, nullptr, nullptr, 0, 0, 1, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return FUIFGPU_OK;
}
int fuifgpu_inv_ycocg(int32_t *c0, int32_t *c1, int32_t *c2, int w, int h, int p0, int p1, int p2, int maxval, void *stream) {
    return color_raw(OP_YCOCG, c0, c1, c2, w, h, p0, p1, p2, 0, maxval, stream);
}
int fuifgpu_inv_ycbcr(int32_t *c0, int32_t *c1, int32_t *c2, int w, int h, int p0, int p1, int p2, int minval, int maxval, void *stream) {
    return color_raw(OP_YCBCR, c0, c1, c2, w, h, p0, p1, p2, minval, maxval, stream);
}
int fuifgpu_inv_quantize(int32_t *plane, int64_t n_samples, int q, void *stream) {
    if (!plane || n_samples < 0) return FUIFGPU_E_ARG;
    launch_scale(plane, n_samples, q, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return FUIFGPU_OK;
}
int fuifgpu_idct8x8(const int32_t *const *src64, int bw, int bh, int32_t *out, int maxval, void *stream) {
    if (!src64 || !out || bw < 1 || bh < 1) return FUIFGPU_E_ARG;
    // planes are addressed as element offsets from a null base: every plane pointer is 4-byte aligned
    std::vector<PlaneRef> list(64);
    for (int i = 0; i < 64; i++) {
        if (!src64[i]) return FUIFGPU_E_ARG;
        list[i] = raw_plane(0, bw, bh);
        list[i].off = (int64_t)(reinterpret_cast<uintptr_t>(src64[i]) / 4);
    }
    PlaneRef *d_list = nullptr;
    HIPCHK(hipMalloc((void **)&d_list, sizeof(PlaneRef) * 64));
    hipError_t e = hipMemcpy(d_list, list.data(), sizeof(PlaneRef) * 64, hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(d_list); return hip_fail(e, "hipMemcpy"); }
    Bases b{};
    b.base[0] = nullptr; b.stride[0] = 0; b.base[1] = out; b.stride[1] = 0; b.base[2] = nullptr; b.stride[2] = 0;
    Op op = raw_op(OP_IDCT);
    op.p0 = bw; op.p1 = bh; op.hi = maxval; op.idct_first = 0; op.pad = 64;
    op.dst[0] = raw_plane(1, bw * 8, bh * 8);
    launch_op(op, b, d_list, nullptr, 0, 0, 1, (hipStream_t)stream);
    e = hipStreamSynchronize((hipStream_t)stream);
    hipFree(d_list);
    if (e != hipSuccess) return hip_fail(e, "idct8x8");
    return FUIFGPU_OK;
}
int fuifgpu_fwd_ycocg(int32_t *c0, int32_t *c1, int32_t *c2, int w, int h, void *stream) {
    if (!c0 || !c1 || !c2 || w < 1 || h < 1) return FUIFGPU_E_ARG;
    launch_fwd_ycocg(c0, c1, c2, (int64_t)w * h, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return FUIFGPU_OK;
}
int fuifgpu_fwd_hsqueeze(const int32_t *in, int w, int h, int32_t *avg, int32_t *res, void *stream) {
    if (!in || !avg || (w > 1 && !res) || w < 1 || h < 1) return FUIFGPU_E_ARG;
    launch_fwd_squeeze(true, in, w, h, avg, res, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return FUIFGPU_OK;
}
int fuifgpu_fwd_vsqueeze(const int32_t *in, int w, int h, int32_t *avg, int32_t *res, void *stream) {
    if (!in || !avg || (h > 1 && !res) || w < 1 || h < 1) return FUIFGPU_E_ARG;
    launch_fwd_squeeze(false, in, w, h, avg, res, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return FUIFGPU_OK;
}
// a plane anywhere in device memory as an element offset from a null base (every plane pointer is 4-byte aligned)
static PlaneRef raw_plane_at(const int32_t *ptr, int w, int h) {
    PlaneRef p = raw_plane(0, w, h);
    p.off = (int64_t)(reinterpret_cast<uintptr_t>(ptr) / 4);
    return p;
}
int fuifgpu_inv_palette(const int32_t *index, int w, int h, const int32_t *palette_row, int colours, int32_t *out, void *stream) {
    if (!index || !out || out == index || w < 0 || h < 0 || colours < 0 || (colours > 0 && !palette_row)) return FUIFGPU_E_ARG;
    if ((int64_t)w * h == 0) return FUIFGPU_OK;
    Bases b{};
    Op op = raw_op(OP_PALETTE);
    op.src[0] = raw_plane_at(index, w, h);
    op.src[1] = raw_plane_at(colours > 0 ? palette_row : index, colours, 1);
    op.dst[0] = raw_plane_at(out, w, h);
    op.p0 = 0; op.p1 = colours;
    launch_op(op, b, nullptr, nullptr, 0, 0, 1, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return FUIFGPU_OK;
}
int fuifgpu_inv_approximate(int32_t *plane, const int32_t *remainder, int64_t n_samples, int q, void *stream) {
    if (!plane || n_samples < 0 || n_samples > 0x7fffffffLL * 32768) return FUIFGPU_E_ARG;
    if (n_samples == 0) return FUIFGPU_OK;
    // the kernel walks w*h samples linearly: any factorisation with both factors in int range will do
    int h = 1;
    int64_t w = n_samples;
    while (w > 0x7fffffffLL) { h *= 2; w = (n_samples + h - 1) / h; }
    if (w * h != n_samples) {   // not a clean factorisation: a head of w*(h-1) samples and a tail
        const int64_t head = w * (h - 1);
        int rc = fuifgpu_inv_approximate(plane, remainder, head, q, stream);
        if (rc != FUIFGPU_OK) return rc;
        return fuifgpu_inv_approximate(plane + head, remainder ? remainder + head : nullptr, n_samples - head, q, stream);
    }
    Bases b{};
    Op op = raw_op(OP_APPROX);
    op.src[0] = op.dst[0] = raw_plane_at(plane, (int)w, h);
    op.src[1] = raw_plane_at(remainder ? remainder : plane, (int)w, h);
    op.p0 = q; op.p1 = remainder ? 1 : 0;   // p1: "the remainder has samples" (no per-image ChannelMeta here: qsrc = -1)
    launch_op(op, b, nullptr, nullptr, 0, 0, 1, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return FUIFGPU_OK;
}
int fuifgpu_inv_match(const int32_t *match, int w, int h, int32_t *const *planes, int n_planes, int softmatch, int match_q, int match_maxval, int nb_frames,
                      void *stream) {
    if (!match || !planes || w < 0 || h < 0 || n_planes < 1 || n_planes > 64 || nb_frames < 1) return FUIFGPU_E_ARG;
    for (int k = 0; k < n_planes; k++) if (!planes[k]) return FUIFGPU_E_ARG;
    const int64_t n = (int64_t)w * h;
    if (n == 0) return FUIFGPU_OK;
    if (n > 0x7fffffffLL) return fail_msg(FUIFGPU_E_UNSUPPORTED, "match over more than 2^31 samples");
    const int fh = h / nb_frames;
    if (match_q != 1 && match_q != 2 * fh * fh + (fh & 1)) return fail_msg(FUIFGPU_E_CORRUPT, "match transform with unexpected quantization factor");   // 2dmatch.h:172-175
    hipStream_t st = (hipStream_t)stream;
    const bool free_mode = match_q == 1, soft = softmatch != 0;
    // device side: the plane list (a soft free-offset match appends two copies of one accumulator per plane), one ChannelMeta for the
    // match channel (its q and maxval are what the kernels decide on), one status word, and for the free-offset mode two maps of source indices
    struct Side { PlaneRef list[192]; ChannelMeta meta; int32_t status; } *d_side = nullptr;
    Side side{};
    int32_t *d_work = nullptr;
    if (free_mode) {
        hipError_t ea = hipMalloc((void **)&d_work, (size_t)n * 4 * (2 + (soft ? 2 * n_planes : 0)));
        if (ea != hipSuccess) return hip_fail(ea, "hipMalloc");
    }
    for (int k = 0; k < n_planes; k++) side.list[k] = raw_plane_at(planes[k], w, h);
    if (free_mode && soft) for (int k = 0; k < 2 * n_planes; k++) side.list[n_planes + k] = raw_plane_at(d_work + (int64_t)(2 + k) * n, w, h);
    const int n_list = free_mode && soft ? 3 * n_planes : n_planes;
    side.meta = ChannelMeta{0, match_maxval, match_q, 1};
    side.status = 0;
    auto cleanup = [&]() { hipFree(d_side); hipFree(d_work); };
    hipError_t e = hipMalloc((void **)&d_side, sizeof(Side));
    if (e != hipSuccess) { cleanup(); return hip_fail(e, "hipMalloc"); }
    e = hipMemcpyAsync(d_side, &side, sizeof(Side), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) { cleanup(); return hip_fail(e, "hipMemcpyAsync"); }
    ChannelMeta *d_meta = &d_side->meta;
    int32_t *d_status = &d_side->status;
    const PlaneRef *d_list = d_side->list;
    Bases b{};
    PlaneRef pm = raw_plane_at(match, w, h);
    pm.qsrc = 0;
    if (!free_mode) {
        Op op = raw_op(OP_MATCH);
        op.src[0] = pm; op.p0 = soft; op.p1 = fh; op.idct_first = 0; op.pad = n_planes;
        launch_op(op, b, d_list, d_meta, 1, 0, 1, st, d_status);
    } else {   // the same op sequence Plan builds (plan.cpp, inv_match)
        PlaneRef cur = raw_plane_at(d_work, w, h), other = raw_plane_at(d_work + n, w, h);
        Op init = raw_op(OP_MATCH_INIT);
        init.src[0] = pm; init.dst[0] = cur; init.p0 = soft; init.idct_first = 0; init.pad = soft ? n_list : 0;
        launch_op(init, b, d_list, d_meta, 1, 0, 1, st, d_status);
        int steps = 1;
        while ((1LL << steps) < n) steps++;
        for (int k = 0; k < steps; k++) {
            Op j = raw_op(OP_MATCH_JUMP);
            j.src[0] = cur; j.src[1] = pm; j.dst[0] = other; j.p0 = soft; j.p1 = k & 1; j.idct_first = 0; j.pad = soft ? n_list : 0;
            launch_op(j, b, d_list, d_meta, 1, 0, 1, st, d_status);
            std::swap(cur, other);
        }
        Op ap = raw_op(OP_MATCH_APPLY);
        ap.src[0] = cur; ap.src[1] = pm; ap.p0 = soft; ap.p1 = steps & 1; ap.idct_first = 0; ap.pad = n_list;
        launch_op(ap, b, d_list, d_meta, 1, 0, 1, st, d_status);
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&side.status, d_status, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    cleanup();
    if (e != hipSuccess) return hip_fail(e, "inv_match");
    if (side.status & ST_UNSUPPORTED) return fail_msg(FUIFGPU_E_UNSUPPORTED, "2D match with a forward reference (an image narrower than the offset spiral)");
    if (side.status & ST_CORRUPT) return fail_msg(FUIFGPU_E_CORRUPT, "2D match channel holds an offset code outside its table");
    return FUIFGPU_OK;
}
int fuifgpu_upsample(const int32_t *in, int w, int h, int srh, int srv, int32_t *out, void *stream) {
    if (!in || !out || w < 1 || h < 1 || srh < 1 || srh > 8 || srv < 1 || srv > 8) return FUIFGPU_E_ARG;
    Bases b{};
    b.base[0] = const_cast<int32_t *>(in); b.stride[0] = 0; b.base[1] = out; b.stride[1] = 0; b.base[2] = nullptr; b.stride[2] = 0;
    Op op = raw_op(OP_UPSAMPLE);
    op.src[0] = raw_plane(0, w, h); op.dst[0] = raw_plane(1, w * srh, h * srv); op.p0 = srh; op.p1 = srv;
    launch_op(op, b, nullptr, nullptr, 0, 0, 1, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return FUIFGPU_OK;
}

}  // extern "C"
