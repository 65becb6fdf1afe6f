// fuif_amd/csrc/maniac_decode.hip -- MANIAC entropy decode of a batch of FUIF streams on gfx950.
//
// One 64-lane wavefront (= one workgroup) per stream.  The format has no intra-stream entry
// points (a channel group's first byte is only known once the previous group is fully decoded,
// maniac/rac.h:70-104), so a stream is an inherently serial chain; the batch provides the
// parallelism (>=1024 streams = one per SIMD on 256 CUs).  Within the wave, control flow is
// wave-uniform: every lane executes the same scalar program; lanes are used as a vector unit for
// the bulk work (table staging, plane fills, leaf initialisation).
//
// What it replaces in the reference:
//   fuif_decode channel loop            encoding/encoding.cpp:708-717
//   fuif_decode_channel                 encoding/encoding.cpp:259-429
//   init_properties / predictors        encoding/context_predict.h:67-120, 124-168, 233-289
//   MetaPropertySymbolCoder::read_tree  maniac/compound.h:277-320   (explicit stack, no recursion)
//   FinalPropertySymbolCoder            maniac/compound.h:135-232
//   reader<15>, SymbolChance            maniac/symbol.h:72-185
//   UniformSymbolCoder                  maniac/symbol.h:44-57
//   RacInput24                          maniac/rac.h:55-117
//   SimpleBitChance::put                maniac/chance.h:77-79
#include <hip/hip_runtime.h>

#include "fuifgpu_internal.h"
#include "maniac_decode.h"

namespace fuifgpu {

namespace {

#define DEV __device__ __forceinline__

constexpr int CH_ZERO = 0, CH_SIGN = 1, CH_EXP = 2, CH_MANT = 16, CH_N = 31;

struct Node {  // maniac/compound.h:41-51
    int16_t property;
    uint16_t child;
    int32_t splitval;
};

struct Frame {  // one pending inner node of the pre-order tree parse
    int32_t p, oldmin, oldmax, splitval, child, stage;
};

struct Stream {
    const uint8_t *p;
    uint32_t size, pos, limit;
    int eof_flag;   // FileIO::isEOF() = feof(): only after a failed read (fileio.h:55-63)
    int blob_mode;  // BlobReader::isEOF(): pos >= size (fileio.h:100-102)
};

DEV int s_getc(Stream &s) {
    if (s.pos >= s.size) { s.eof_flag = 1; return -1; }
    return s.p[s.pos++];
}
DEV bool s_eof(const Stream &s) { return s.blob_mode ? (s.pos >= s.size) : (s.eof_flag != 0); }
DEV bool s_limit_hit(const Stream &s) { return s_eof(s) || (s.limit && s.pos >= s.limit); }

// encoding/encoding.cpp:45-59
DEV int s_varint(Stream &s) {
    uint32_t result = 0;
    for (int k = 0; k < 10; k++) {
        int b = s_getc(s);
        if (b < 0) return -1;
        if (b < 128) return (int)(result + (uint32_t)b);
        result = (result + (uint32_t)(b - 128)) << 7;
    }
    return -1;
}

DEV int ilog2u(uint32_t l) { return l == 0 ? 0 : 31 - __clz((int)l); }
DEV int slog(int x) {  // encoding/context_predict.h:53-61
    if (x == 0) return 0;
    if (x > 0) return 32 - __clz(x);
    return -(32 - __clz(-x));
}
DEV int iabs(int x) { return x < 0 ? -x : x; }
DEV int median3(int a, int b, int c) {  // util.h:9-23
    if (a < b) { if (b < c) return b; return a < c ? c : a; }
    if (a < c) return a;
    return b < c ? c : b;
}

// 24-bit range decoder, maniac/rac.h:55-114.  `low` fits 32 bits before EOF; after EOF the
// reference ORs -1 into a 64-bit low so every later decision is 1 -- a 32-bit all-ones low gives
// the same decisions (low stays >= 2^32-2^24 > range between renormalisations).
struct Rac {
    uint32_t range, low;
};
DEV void rac_input(Rac &r, Stream &s) {
    if (r.range <= 0x10000u) { r.low <<= 8; r.range <<= 8; r.low |= (uint32_t)s_getc(s); }
    if (r.range <= 0x10000u) { r.low <<= 8; r.range <<= 8; r.low |= (uint32_t)s_getc(s); }
}
DEV int rac_get(Rac &r, Stream &s, uint32_t chance) {
    uint32_t thr = r.range - chance;
    int bit;
    if (r.low >= thr) { r.low -= thr; r.range = chance; bit = 1; }
    else { r.range = thr; bit = 0; }
    rac_input(r, s);
    return bit;
}
DEV void rac_init(Rac &r, Stream &s) {
    r.range = 1u << 24; r.low = 0;
    for (int k = 0; k < 3; k++) { r.low <<= 8; r.low |= (uint32_t)s_getc(s); }
}
// rac.h:43-52: (range*b12+0x800)>>12 without a 64-bit product
DEV uint32_t chance12(uint32_t range, uint32_t b12) { return (((range & 0xFFFu) * b12 + 0x800u) >> 12) + ((range >> 12) * b12); }
DEV int rac_bit(Rac &r, Stream &s) { return rac_get(r, s, r.range >> 1); }

// maniac/symbol.h:44-57
DEV int uniform_read(Rac &r, Stream &s, int min, int len) {
    while (len != 0) {
        int med = len / 2;
        if (rac_bit(r, s)) { min = min + med + 1; len = len - (med + 1); }
        else len = med;
    }
    return min;
}

// One adaptive decision (compound.h:90-95 + chance.h:77-79).  CH: chance storage, TB: table.
template <typename CH, typename TB>
DEV int coder_bit(Rac &r, Stream &s, CH ch, int idx, TB table) {
    uint32_t c = ch[idx];
    int bit = rac_get(r, s, chance12(r.range, c));
    ch[idx] = table[c * 2 + bit];
    return bit;
}

// maniac/symbol.h:154-185
template <typename CH, typename TB>
DEV int read_symbol(Rac &r, Stream &s, CH ch, TB table, int min, int max) {
    if (min == max) return min;
    if (coder_bit(r, s, ch, CH_ZERO, table)) return 0;
    int sign;
    if (min < 0) { if (max > 0) sign = coder_bit(r, s, ch, CH_SIGN, table); else sign = 0; }
    else sign = 1;
    const int amax = sign ? max : -min;
    const int emax = ilog2u((uint32_t)amax);
    int e = 0;
    for (; e < emax; e++) if (coder_bit(r, s, ch, CH_EXP + e, table)) break;
    int have = 1 << e;
    for (int pos = e; pos > 0;) {
        pos--;
        int minabs1 = have | (1 << pos);
        if (minabs1 > amax) continue;
        if (coder_bit(r, s, ch, CH_MANT + pos, table)) have = minabs1;
    }
    return sign ? have : -have;
}
template <typename CH, typename TB>
DEV int read_symbol2(Rac &r, Stream &s, CH ch, TB table, int min, int max) {  // symbol.h:235-239
    if (min > 0) return read_symbol(r, s, ch, table, 0, max - min) + min;
    if (max < 0) return read_symbol(r, s, ch, table, min - max, 0) + max;
    return read_symbol(r, s, ch, table, min, max);
}

// maniac/symbol.h:115-138
DEV void symbol_chance_init(uint16_t *ch, int zero_chance) {
    uint32_t rp = 0x1000 - zero_chance;
    ch[CH_ZERO] = (uint16_t)zero_chance;
    ch[CH_SIGN] = 0x800;
    for (int i = 0; i < kMaxBitDepth - 1; i++) {
        if (rp < 0x100) rp = 0x100;
        if (rp > 0xf00) rp = 0xf00;
        ch[CH_EXP + i] = (uint16_t)(0x1000 - rp);
        rp = (rp * rp + 0x800) >> 12;
    }
    for (int i = 0; i < kMaxBitDepth; i++) ch[CH_MANT + i] = 1024;
}

// encoding/encoding.cpp:61-72
DEV bool check_bit_depth(int minv, int maxv, int predictor) {
    int maxav = iabs(maxv);
    if (-minv > maxav) maxav = -minv;
    if (predictor > 0 && maxv - minv > maxav) maxav = maxv - minv;
    if (predictor > 0 && iabs(minv - maxv) > maxav) maxav = iabs(minv - maxv);
    return ilog2u((uint32_t)maxav) + 1 <= kMaxBitDepth;
}

struct RefChan {  // one reference channel of the current group (context_predict.h:233-289)
    const int32_t *data;
    int32_t w, h, hshift, vshift;
};

struct Shared {
    uint16_t table[8192];        // pixel-coder transition table (cut 6, alpha 0x0d000000)
    uint16_t meta_ctx[3][32];    // three SimpleSymbolCoder contexts of the tree coder
    int32_t props[kMaxProps];
    int32_t lo[kMaxProps], hi[kMaxProps];
};

DEV void fill_plane(int32_t *plane, int64_t first, int64_t count, int v, int lane) {
    for (int64_t i = first + lane; i < first + count; i += 64) plane[i] = v;
}

}  // namespace

__global__ __launch_bounds__(64) void k_maniac_decode(DecodeParams P) {
    __shared__ Shared sh;
    const int img = blockIdx.x;
    const int lane = threadIdx.x;
    if (img >= P.n_images) return;

    // stage the pixel-coder transition table in LDS (16 KB, 128-bit loads)
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(P.tables + 8192);
        uint4 *dst = reinterpret_cast<uint4 *>(sh.table);
        for (int i = lane; i < 1024; i += 64) dst[i] = src[i];
    }
    __syncthreads();
    const uint16_t *tree_table = P.tables;  // cut 2, alpha 0xFFFFFFFF/19 (compound.h:262); only used while parsing trees

    const StreamJob job = P.jobs[img];
    Stream s;
    s.p = P.blobs + job.blob_off;
    s.size = job.blob_size;
    s.pos = job.data_start;
    s.limit = job.limit;
    s.eof_flag = 0;
    s.blob_mode = (int)(job.flags & 1u);

    int32_t *coef = P.coef + (int64_t)img * P.coef_stride;
    ChannelMeta *meta = P.meta + (int64_t)img * P.n_channels;
    uint8_t *scratch = P.scratch + (size_t)img * P.scratch_stride;
    Node *nodes = reinterpret_cast<Node *>(scratch);
    uint16_t *leaves = reinterpret_cast<uint16_t *>(scratch + P.leaves_off);
    Frame *stack = reinterpret_cast<Frame *>(scratch + P.stack_off);
    const ChannelGeom *geom = P.geom;
    const int nch = P.n_channels;
    int status = 0;

    // ---- fuif_decode channel loop: encoding.cpp:708-717 -------------------------------------
    for (int ci = 0; ci < nch; ci++) {
        if (!((s.limit == 0 || s.pos < s.limit) && !s_eof(s))) break;
        if (!geom[ci].w || !geom[ci].h) continue;

        // ---- fuif_decode_channel: encoding.cpp:259-429 --------------------------------------
        const int beginc = ci;
        if (s_limit_hit(s)) continue;
        int firstbyte = s_varint(s);
        if (s_limit_hit(s)) continue;
        const int endc = beginc + (firstbyte >> 4);
        const int compress = firstbyte & 1;
        const int predictor = (firstbyte & 14) >> 1;
        int global_minv = 1 - s_varint(s);
        if (s_limit_hit(s)) continue;
        if (global_minv == 1) global_minv = s_varint(s);
        if (s_limit_hit(s)) continue;
        const int global_maxv = global_minv + s_varint(s);
        if (s_limit_hit(s)) continue;
        if (endc >= nch || endc < beginc) { status |= ST_CORRUPT; break; }

        int firstrealc = beginc;
        bool fatal = false, early = false;
        for (int i = beginc; i <= endc; i++) {
            const ChannelGeom g = geom[i];
            if ((int64_t)g.w * g.h <= 0) continue;
            int minv = global_minv, maxv = global_maxv;
            if (endc > beginc && global_minv < global_maxv) {
                minv += s_varint(s);
                maxv = minv + s_varint(s);
            }
            int q = 1;
            if (minv == maxv) {
                fill_plane(coef + g.coef_off, 0, (int64_t)g.w * g.h, minv, lane);
                firstrealc++;
            }
            bool have_q = !(minv == 0 && maxv == 0);
            if (have_q) q = s_varint(s);
            if (lane == 0) { meta[i].minval = minv; meta[i].maxval = maxv; meta[i].q = q; meta[i].decoded = (minv == maxv) ? 1 : 0; }
            if (!have_q) continue;
            if (s_limit_hit(s)) {  // corrupt_or_truncated: encoding.cpp:209-219 (isEOF or limit => zero-fill, true)
                fill_plane(coef + g.coef_off, 0, (int64_t)g.w * g.h, 0, lane);
                if (lane == 0) meta[i].decoded = 1;
                status |= ST_TRUNCATED;
                early = true;
                break;
            }
            if (compress && !check_bit_depth(minv, maxv, predictor)) { fatal = true; break; }
        }
        __syncthreads();  // meta[] written by lane 0 is read below by every lane
        if (fatal) { status |= ST_UNSUPPORTED | ST_CORRUPT; break; }
        if (early) continue;
        if (firstrealc > endc) { ci = endc; continue; }

        // ---- init_properties: context_predict.h:67-120 --------------------------------------
        RefChan refs[kMaxRefs];
        int nrefs = 0;
        int nprops = 0;
        {
            int offset = 0;
            for (int j = beginc - 1; j >= 0 && offset < P.max_properties; j--) {
                const int jmin = meta[j].minval, jmax = meta[j].maxval;
                if (jmin == jmax) continue;
                if (geom[j].hshift < 0) continue;
                int mn = jmin; if (mn > 0) mn = 0;
                int mx = jmax; if (mx < 0) mx = 0;
                if (lane == 0) {
                    sh.lo[nprops] = 0; sh.hi[nprops] = iabs(mx > -mn ? mx : mn);
                    sh.lo[nprops + 1] = slog(mn); sh.hi[nprops + 1] = slog(mx);
                }
                nprops += 2; offset += 2;
                refs[nrefs].data = coef + geom[j].coef_off;
                refs[nrefs].w = geom[j].w; refs[nrefs].h = geom[j].h;
                refs[nrefs].hshift = geom[j].hshift; refs[nrefs].vshift = geom[j].vshift;
                nrefs++;
            }
            int mn = 0x7FFFFFFF, mx = (int)0x80000001, maxh = 0, maxw = 0;
            for (int j = beginc; j <= endc; j++) {
                const int jmin = meta[j].minval, jmax = meta[j].maxval;
                // note: zero-pixel channels keep their constructor range (0,0 for inserted
                // residual channels) in the reference; meta[] is zero-initialised likewise
                if (jmin < mn) mn = jmin;
                if (jmax > mx) mx = jmax;
                if (geom[j].h > maxh) maxh = geom[j].h;
                if (geom[j].w > maxw) maxw = geom[j].w;
            }
            if (mn > 0) mn = 0;
            if (mx < 0) mx = 0;
            const int amax = iabs(mn) > iabs(mx) ? iabs(mn) : iabs(mx);
            if (lane == 0) {
                int n = nprops;
                sh.lo[n] = 0; sh.hi[n] = amax; n++;
                sh.lo[n] = 0; sh.hi[n] = amax; n++;
                sh.lo[n] = slog(mn); sh.hi[n] = slog(mx); n++;
                sh.lo[n] = slog(mn); sh.hi[n] = slog(mx); n++;
                sh.lo[n] = 0; sh.hi[n] = maxh - 1; n++;
                sh.lo[n] = 0; sh.hi[n] = maxw - 1; n++;
                sh.lo[n] = mn + mn - mx; sh.hi[n] = mx + mx - mn; n++;
                sh.lo[n] = mn + mn - mx; sh.hi[n] = mx + mx - mn; n++;
                for (int k = 0; k < 5; k++) { sh.lo[n] = slog(mn - mx); sh.hi[n] = slog(mx - mn); n++; }
            }
            nprops += kNonRefProps;
        }
        const int nrefprops = nprops - kNonRefProps;

        int predictability = 2048;
        if (predictor == 0 && compress) {
            int rounded = s_varint(s);
            if (rounded < 1 || rounded > 127) {
                if (s_limit_hit(s)) {
                    const ChannelGeom g = geom[firstrealc];
                    fill_plane(coef + g.coef_off, 0, (int64_t)g.w * g.h, 0, lane);
                    if (lane == 0) meta[firstrealc].decoded = 1;
                    status |= ST_TRUNCATED;
                    continue;
                }
                status |= ST_CORRUPT;
                break;
            }
            predictability = rounded * 32;
        }

        Rac rac;
        rac_init(rac, s);

        if (!compress) {
            // uncompressed group: encoding.cpp:334-354
            for (int i = beginc; i <= endc; i++) {
                const ChannelGeom g = geom[i];
                const int minv = meta[i].minval, maxv = meta[i].maxval;
                if (minv == maxv) continue;
                int32_t *plane = coef + g.coef_off;
                const int zero = minv > 0 ? minv : (maxv < 0 ? maxv : 0);
                int y = 0;
                for (; y < g.h; y++) {
                    if (s_limit_hit(s)) break;
                    for (int x = 0; x < g.w; x++) {
                        int v = uniform_read(rac, s, minv, maxv - minv);
                        if (lane == 0) plane[(int64_t)y * g.w + x] = v;
                    }
                }
                if (y < g.h) { fill_plane(plane, (int64_t)y * g.w, (int64_t)(g.h - y) * g.w, zero, lane); status |= ST_TRUNCATED; }
                if (lane == 0) meta[i].decoded = 1;
                if (s_limit_hit(s)) break;
            }
            __syncthreads();
            ci = endc;
            continue;
        }

        // ---- MANIAC tree: compound.h:277-320 with an explicit stack -------------------------
        __syncthreads();  // sh.lo/hi
        for (int k = lane; k < 3 * 32; k += 64) sh.meta_ctx[k / 32][k % 32] = 0;
        __syncthreads();
        if (lane == 0) for (int k = 0; k < 3; k++) symbol_chance_init(sh.meta_ctx[k], 1024);
        __syncthreads();
        int tree_size = 1;
        bool tree_ok = true;
        {
            int pos = 0, depth = 0;
            while (true) {
                int p = read_symbol2(rac, s, sh.meta_ctx[0], tree_table, 0, nprops) - 1;
                if (p != -1) {
                    const int oldmin = sh.lo[p], oldmax = sh.hi[p];
                    if (oldmin >= oldmax) { tree_ok = false; break; }
                    const int splitval = read_symbol2(rac, s, sh.meta_ctx[2], tree_table, oldmin, oldmax - 1);
                    const int child = tree_size;
                    if (tree_size + 2 > P.max_nodes || depth >= kTreeStackDepth) { tree_ok = false; status |= ST_UNSUPPORTED; break; }
                    if (lane == 0) {
                        Node n; n.property = (int16_t)p; n.child = (uint16_t)child; n.splitval = splitval;
                        nodes[pos] = n;
                        Frame f; f.p = p; f.oldmin = oldmin; f.oldmax = oldmax; f.splitval = splitval; f.child = child; f.stage = 0;
                        stack[depth] = f;
                        sh.lo[p] = splitval + 1;
                    }
                    tree_size += 2;
                    depth++;
                    pos = child;
                    __syncthreads();
                    continue;
                }
                if (lane == 0) { Node n; n.property = -1; n.child = 0; n.splitval = 0; nodes[pos] = n; }
                // return to the nearest ancestor that still has its "<= splitval" branch to read
                bool done = false;
                while (true) {
                    if (depth == 0) { done = true; break; }
                    Frame f = stack[depth - 1];
                    if (f.stage == 0) {
                        if (lane == 0) { sh.lo[f.p] = f.oldmin; sh.hi[f.p] = f.splitval; stack[depth - 1].stage = 1; }
                        pos = f.child + 1;
                        break;
                    }
                    if (lane == 0) sh.hi[f.p] = f.oldmax;
                    depth--;
                }
                __syncthreads();
                if (done) break;
            }
        }
        __syncthreads();
        if (!tree_ok) {
            // corrupt_or_truncated(io, image.channel[beginc], ...): encoding.cpp:358
            if (s_limit_hit(s)) {
                const ChannelGeom g = geom[beginc];
                fill_plane(coef + g.coef_off, 0, (int64_t)g.w * g.h, 0, lane);
                if (lane == 0) meta[beginc].decoded = 1;
                status |= ST_TRUNCATED;
                continue;
            }
            status |= ST_CORRUPT;
            break;
        }

        // ---- FinalPropertySymbolCoder ctor: compound.h:213-225 ------------------------------
        const int nleaves = (tree_size + 1) / 2;
        {
            // leaf numbering in node-array order
            if (lane == 0) {
                int leaf_id = 0;
                for (int i = 0; i < tree_size; i++)
                    if (nodes[i].property == -1) { nodes[i].child = (uint16_t)leaf_id; leaf_id++; }
                symbol_chance_init(leaves, predictability);
                leaves[31] = 0;
            }
            __syncthreads();
            // replicate leaf 0 (64 bytes) into all leaves, one 32-bit word per lane-slot
            const uint32_t *l0 = reinterpret_cast<const uint32_t *>(leaves);
            uint32_t *lw = reinterpret_cast<uint32_t *>(leaves);
            const uint32_t mine = l0[lane & 15];
            for (int64_t i = 16 + lane; i < (int64_t)nleaves * 16; i += 64) lw[i] = mine;  // (i & 15) == (lane & 15)
            __syncthreads();
        }

        // ---- pixel loops: encoding.cpp:365-425 ----------------------------------------------
        for (int i = beginc; i <= endc; i++) {
            const ChannelGeom g = geom[i];
            const int minv = meta[i].minval, maxv = meta[i].maxval;
            if (minv == maxv) continue;
            int32_t *plane = coef + g.coef_off;
            const int zero = minv > 0 ? minv : (maxv < 0 ? maxv : 0);
            const int w = g.w, h = g.h;
            int y = 0;
            if (tree_size == 1 && predictor == 0 && zero == 0) {
                // fast track: encoding.cpp:371-383
                for (; y < h; y++) {
                    if (s_limit_hit(s)) break;
                    for (int x = 0; x < w; x++) {
                        int v = read_symbol(rac, s, leaves, sh.table, minv, maxv);
                        if (lane == 0) plane[(int64_t)y * w + x] = v;
                    }
                }
            } else {
                for (; y < h; y++) {
                    if (s_limit_hit(s)) break;
                    __syncthreads();  // previous row's stores (lane 0) become visible to the loads below
                    // reference rows for this y (context_predict.h:236-240)
                    const int32_t *refrow[kMaxRefs];
                    for (int k = 0; k < nrefs; k++) {
                        int ry = (y << g.vshift) >> refs[k].vshift;
                        if (ry >= refs[k].h) ry = refs[k].h - 1;
                        refrow[k] = refs[k].data + (int64_t)ry * refs[k].w;
                    }
                    const int32_t *row = plane + (int64_t)y * w;
                    const int32_t *row1 = row - w;        // y-1
                    const int32_t *row2 = row - 2 * (int64_t)w;  // y-2
                    int left = zero, leftleft = zero;
                    for (int x = 0; x < w; x++) {
                        // reference-channel properties: rx = min((x<<hshift)>>ref.hshift, ref.w-1)
                        // covers the three cases of context_predict.h:241-284
                        for (int k = 0; k < nrefs; k++) {
                            int rx = (x << g.hshift) >> refs[k].hshift;
                            if (rx >= refs[k].w) rx = refs[k].w - 1;
                            const int v = refrow[k][rx];
                            if (lane == 0) { sh.props[2 * k] = iabs(v); sh.props[2 * k + 1] = slog(v); }
                        }
                        // local neighbourhood: context_predict.h:126-133
                        const int l = x ? left : zero;
                        const int top = y ? row1[x] : zero;
                        const int topleft = (x && y) ? row1[x - 1] : l;
                        const int topright = (x + 1 < w && y) ? row1[x + 1] : top;
                        const int ll = x > 1 ? leftleft : l;
                        const int toptop = y > 1 ? row2[x] : top;
                        if (lane == 0) {
                            int32_t *p = sh.props + nrefprops;
                            p[0] = iabs(top); p[1] = iabs(l); p[2] = slog(top); p[3] = slog(l);
                            p[4] = y; p[5] = x;
                            p[6] = l + top - topleft; p[7] = topleft + topright - top;
                            p[8] = slog(l - topleft); p[9] = slog(topleft - top); p[10] = slog(top - topright);
                            p[11] = slog(top - toptop); p[12] = slog(l - ll);
                        }
                        int guess;
                        switch (predictor) {  // context_predict.h:157-166
                            case 0: guess = zero; break;
                            case 1: guess = (l + top) / 2; break;
                            case 2: guess = median3(l + top - topleft, l, top); break;
                            case 3: guess = l; break;
                            case 4: guess = top; break;
                            case 5: guess = (l + topleft + top + topright) / 4; break;
                            case 6: { int t = l + top - topleft; guess = t < minv ? minv : (t > maxv ? maxv : t); break; }
                            default: guess = median3(l + top - topleft, l, top); break;
                        }
                        __syncthreads();
                        const int mn = minv - guess, mx = maxv - guess;
                        int diff;
                        if (mn == mx) diff = mn;  // compound.h:228
                        else {
                            int pos = 0;  // find_leaf: compound.h:142-153
                            while (true) {
                                const Node n = nodes[pos];
                                if (n.property == -1) { pos = n.child; break; }
                                pos = (sh.props[n.property] > n.splitval) ? n.child : n.child + 1;
                            }
                            diff = read_symbol(rac, s, leaves + (int64_t)pos * kLeafStride, sh.table, mn, mx);
                        }
                        const int v = diff + guess;
                        if (lane == 0) plane[(int64_t)y * w + x] = v;
                        leftleft = l; left = v;
                    }
                }
            }
            if (y < h) { __syncthreads(); fill_plane(plane, (int64_t)y * w, (int64_t)(h - y) * w, zero, lane); status |= ST_TRUNCATED; }
            if (lane == 0) meta[i].decoded = 1;
            if (s_limit_hit(s)) break;
        }
        __syncthreads();
        ci = endc;
    }
    if (s_limit_hit(s)) status |= ST_TRUNCATED;
    // planes the stream never reached read as zeros in the reference (empty Channel::data,
    // image/image.h:82-85; zero-filled residuals, transform/squeeze.h:379-383)
    __syncthreads();
    for (int c = 0; c < nch; c++) {
        const ChannelGeom g = geom[c];
        if ((int64_t)g.w * g.h > 0 && meta[c].decoded == 0) fill_plane(coef + g.coef_off, 0, (int64_t)g.w * g.h, 0, lane);
    }
    if (lane == 0) { P.status[img] = status; P.consumed[img] = s.pos; }
}

void launch_maniac_decode(const DecodeParams &P, hipStream_t stream) {
    hipLaunchKernelGGL(k_maniac_decode, dim3(P.n_images), dim3(64), 0, stream, P);
}

}  // namespace fuifgpu
