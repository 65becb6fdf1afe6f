/* oracle/fuif_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the cloudinary/fuif DECODE path (entropy stage + inverse transforms)
 * with int32 planes.  It is the checker for the HIP path (tests/, smoke(), bench.py cpu_baseline)
 * and is never part of the product.  Every function cites the reference file:line it follows
 * (paths relative to /root/reference).  Parity is PINNED: tests/test_oracle_vs_ref.py compares it
 * plane-by-plane with the real reference compiled from source (oracle/_ref) and
 * tests/test_golden.py compares it with fixtures produced by the unmodified reference CLI.
 *
 * Scope: transforms YCbCr(0) YCoCg(1) ChromaSubsample(3) DCT(4) Quantize(5)
 * Squeeze(7), Palette(6), 2DMatch(8), Permute(9), Approximate(10).
 */
#include "fuif_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define FO_MAX_BIT_DEPTH 15          /* config.h:5 */
#define FO_MAX_FIRST_PREVIEW_SIZE 8  /* config.h:42 */
#define FO_NB_NONREF 13              /* encoding/context_predict.h:210 */
#define FO_MAX_PROPS 64
#define FO_MAX_NODES 65535           /* childID is uint16_t: maniac/compound.h:46 */

#define TR_YCBCR 0
#define TR_YCOCG 1
#define TR_SUBSAMPLE 3
#define TR_DCT 4
#define TR_QUANTIZE 5
#define TR_PALETTE 6
#define TR_SQUEEZE 7
#define TR_2DMATCH 8
#define TR_APPROXIMATE 10

#define CLAMPI(x, l, u) ((x) < (l) ? (l) : ((x) > (u) ? (u) : (x)))

/* ------------------------------------------------------------------------------------------- */
/* byte I/O: FileIO (fileio.h:33-79) / BlobReader (fileio.h:84-143) semantics                   */
typedef struct {
    const uint8_t *p;
    size_t n, pos;
    int kind;     /* 0 FileIO, 1 BlobReader */
    int eof_flag; /* FileIO: set by a read past the end (feof) */
} fo_io;

static int io_getc(fo_io *io) {
    if (io->pos >= io->n) { io->eof_flag = 1; return -1; }
    return io->p[io->pos++];
}
static int io_eof(const fo_io *io) { return io->kind == 0 ? io->eof_flag : (io->pos >= io->n); }
static size_t io_tell(const fo_io *io) { return io->pos; }

/* encoding/encoding.cpp:45-59 */
static int read_varint(fo_io *io) {
    uint32_t result = 0;
    int bytes_read = 0;
    while (bytes_read++ < 10) {
        int number = io_getc(io);
        if (number < 0) break;
        if (number < 128) return (int)(result + (uint32_t)number);
        number -= 128;
        result += (uint32_t)number;
        result <<= 7;
    }
    return -1;
}

/* maniac/util.h:34-37 */
static inline int ilog2u(uint32_t l) { return l == 0 ? 0 : 31 - __builtin_clz(l); }
/* encoding/context_predict.h:53-61 */
static inline int slog(int x) {
    if (x == 0) return 0;
    if (x > 0) return 32 - __builtin_clz((unsigned)x);
    return -(32 - __builtin_clz((unsigned)(-x)));
}
static inline int iabs(int x) { return x < 0 ? -x : x; }
/* util.h:9-23 */
static inline int median3(int a, int b, int c) {
    if (a < b) { if (b < c) return b; return a < c ? c : a; }
    if (a < c) return a;
    return b < c ? c : b;
}

/* ------------------------------------------------------------------------------------------- */
/* maniac/chance.cpp:31-65 build_table; table[i*2+bit] = next 12-bit chance                      */
void fo_build_table(uint16_t *t, uint32_t factor, int cut) {
    const int64_t one = 1LL << 32;
    const int size = 4096;
    unsigned max_p = 4096 - cut; /* maniac/chance.h:48-51 */
    int64_t p;
    unsigned last_p8, p8, i;
    memset(t, 0, sizeof(uint16_t) * size * 2);
    last_p8 = 0;
    p = one / 2;
    for (i = 0; i < (unsigned)size / 2; i++) {
        p8 = (unsigned)((size * p + one / 2) >> 32);
        if (p8 <= last_p8) p8 = last_p8 + 1;
        if (last_p8 && last_p8 < (unsigned)size && p8 <= max_p) t[last_p8 * 2 + 1] = (uint16_t)p8;
        p += ((one - p) * factor + one / 2) >> 32;
        last_p8 = p8;
    }
    for (i = size - max_p; i <= max_p; i++) {
        if (t[i * 2 + 1]) continue;
        p = ((int64_t)i * one + size / 2) / size;
        p += ((one - p) * factor + one / 2) >> 32;
        p8 = (unsigned)((size * p + one / 2) >> 32);
        if (p8 <= i) p8 = i + 1;
        if (p8 > max_p) p8 = max_p;
        t[i * 2 + 1] = (uint16_t)p8;
    }
    for (i = 1; i < (unsigned)size; i++) t[i * 2 + 0] = (uint16_t)(size - t[(size - i) * 2 + 1]);
}

/* maniac/symbol.h:72-139: layout [0]=zero [1]=sign [2..15]=exp[0..13] [16..30]=mant[0..14] */
#define CH_ZERO 0
#define CH_SIGN 1
#define CH_EXP 2
#define CH_MANT 16
#define CH_N 31
void fo_symbol_chance_init(uint16_t *ch, int zero_chance) {
    uint64_t rp = 0x1000 - zero_chance;
    ch[CH_ZERO] = (uint16_t)zero_chance;
    ch[CH_SIGN] = 0x800; /* maniac/chance.h:67-69 */
    for (int i = 0; i < FO_MAX_BIT_DEPTH - 1; i++) {
        if (rp < 0x100) rp = 0x100;
        if (rp > 0xf00) rp = 0xf00;
        ch[CH_EXP + i] = (uint16_t)(0x1000 - rp);
        rp = (rp * rp + 0x800) >> 12;
    }
    for (int i = 0; i < FO_MAX_BIT_DEPTH; i++) ch[CH_MANT + i] = 1024;
}

/* ------------------------------------------------------------------------------------------- */
/* maniac/rac.h:55-114 RacInput<RacConfig24>.  low is 64-bit like uint_fast32_t on x86-64 so that
 * the EOF garbage (low |= -1, rac.h:64-69,74) behaves identically.                              */
typedef struct {
    fo_io *io;
    uint64_t range, low;
    uint64_t decisions;
} fo_rac;

static inline void rac_input(fo_rac *r) {
    for (int k = 0; k < 2; k++) {
        if (r->range <= 0x10000) {
            r->low <<= 8;
            r->range <<= 8;
            r->low |= (uint64_t)(int64_t)io_getc(r->io);
        }
    }
}
static inline int rac_get(fo_rac *r, uint64_t chance) {
    r->decisions++;
    if (r->low >= r->range - chance) {
        r->low -= r->range - chance;
        r->range = chance;
        rac_input(r);
        return 1;
    } else {
        r->range -= chance;
        rac_input(r);
        return 0;
    }
}
static void rac_init(fo_rac *r, fo_io *io) {
    r->io = io; r->range = 1u << 24; r->low = 0; r->decisions = 0;
    for (int k = 0; k < 3; k++) { r->low <<= 8; r->low |= (uint64_t)(int64_t)io_getc(io); }
}
/* rac.h:43-52,107 */
static inline int rac_read_12bit(fo_rac *r, unsigned b12) { return rac_get(r, (r->range * b12 + 0x800) >> 12); }
/* rac.h:111 */
static inline int rac_read_bit(fo_rac *r) { return rac_get(r, r->range >> 1); }

/* maniac/symbol.h:44-57 */
static int uniform_read_int(fo_rac *r, int min, int len) {
    while (len != 0) {
        int med = len / 2;
        if (rac_read_bit(r)) { min = min + med + 1; len = len - (med + 1); }
        else len = med;
    }
    return min;
}

/* one adaptive binary decision: compound.h:90-95 / symbol.h:202-207 + chance.h:77-79 */
static inline int coder_read(fo_rac *r, uint16_t *ch, const uint16_t *table) {
    int bit = rac_read_12bit(r, *ch);
    *ch = table[(*ch) * 2 + bit];
    return bit;
}

static int g_stats;
static struct fo_stats_s { uint64_t sym, walked, steps, predepth, same_leaf, zero, nsign, edec, mdec, ehist[16], prehist[24], spec_exits, spec_inner, spec_hist[6], spec_round2, spec_hit, rounds_behind, leaf_sw, leaf_hit[3], wl_hit[2], wl_cand, dl_hit, dl_cand, r3_rounds, r3_hit[2]; } g_st;
/* maniac/symbol.h:154-185 reader<bits>(coder,min,max) */
static int read_symbol(fo_rac *r, uint16_t *ch, const uint16_t *table, int min, int max) {
    if (min == max) return min;
    if (coder_read(r, &ch[CH_ZERO], table)) { if (g_stats > 0) g_st.zero++; return 0; }
    int sign;
    if (min < 0) { if (max > 0) { sign = coder_read(r, &ch[CH_SIGN], table); if (g_stats > 0) g_st.nsign++; } else sign = 0; }
    else sign = 1;
    const int amax = sign ? max : -min;
    const int emax = ilog2u((uint32_t)amax);
    int e = 0;
    for (; e < emax; e++) { if (g_stats > 0) g_st.edec++; if (coder_read(r, &ch[CH_EXP + e], table)) break; }
    if (g_stats > 0) g_st.ehist[e & 15]++;
    int have = 1 << e;
    for (int pos = e; pos > 0;) {
        pos--;
        int minabs1 = have | (1 << pos);
        if (minabs1 > amax) continue;
        if (g_stats > 0) g_st.mdec++;
        if (coder_read(r, &ch[CH_MANT + pos], table)) have = minabs1;
    }
    return sign ? have : -have;
}
/* symbol.h:235-239 */
static int read_symbol2(fo_rac *r, uint16_t *ch, const uint16_t *table, int min, int max) {
    if (min > 0) return read_symbol(r, ch, table, 0, max - min) + min;
    if (max < 0) return read_symbol(r, ch, table, min - max, 0) + max;
    return read_symbol(r, ch, table, min, max);
}

/* ------------------------------------------------------------------------------------------- */
/* MANIAC tree: compound.h:41-56                                                                */
typedef struct { int16_t property; uint16_t childID; int32_t splitval; } fo_node;
typedef struct { fo_node *n; int size, cap; } fo_tree;

static int tree_push2(fo_tree *t) {
    if (t->size + 2 > FO_MAX_NODES) return 0;
    if (t->size + 2 > t->cap) { t->cap = t->cap * 2 + 16; t->n = (fo_node *)realloc(t->n, sizeof(fo_node) * t->cap); }
    for (int k = 0; k < 2; k++) { t->n[t->size].property = -1; t->n[t->size].childID = 0; t->n[t->size].splitval = 0; t->size++; }
    return 1;
}

typedef struct {
    fo_rac *rac;
    uint16_t ctx[3][CH_N]; /* three SimpleSymbolCoders, compound.h:263 */
    const uint16_t *table; /* cut 2, alpha 0xFFFFFFFF/19: compound.h:262 */
    int nprops;
    int lo[FO_MAX_PROPS], hi[FO_MAX_PROPS];
    int maxdepth;
} fo_meta;

/* compound.h:277-308 */
static int read_subtree(fo_meta *m, fo_tree *t, int pos, int depth) {
    int p = read_symbol2(m->rac, m->ctx[0], m->table, 0, m->nprops) - 1;
    t->n[pos].property = (int16_t)p;
    depth++;
    if (depth > m->maxdepth) m->maxdepth = depth;
    if (depth > 8192) return 0;
    if (p != -1) {
        int oldmin = m->lo[p], oldmax = m->hi[p];
        if (oldmin >= oldmax) return 0; /* "Invalid tree" */
        int splitval = read_symbol2(m->rac, m->ctx[2], m->table, oldmin, oldmax - 1);
        t->n[pos].splitval = splitval;
        int childID = t->size;
        t->n[pos].childID = (uint16_t)childID;
        if (!tree_push2(t)) return 0;
        m->lo[p] = splitval + 1;
        if (!read_subtree(m, t, childID, depth)) return 0;
        m->lo[p] = oldmin;
        m->hi[p] = splitval;
        if (!read_subtree(m, t, childID + 1, depth)) return 0;
        m->hi[p] = oldmax;
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------- */
/* containers                                                                                   */
static void ch_init(fo_channel *c) {
    memset(c, 0, sizeof(*c));
    c->q = 1; c->component = -1; /* image/image.h:66 */
}
/* image/image.h:68-72 */
static void ch_setzero(fo_channel *c) {
    if (c->minval > 0) c->zero = c->minval;
    else if (c->maxval < 0) c->zero = c->maxval;
    else c->zero = 0;
}
/* A stored sample is the reference's pixel_type = int16_t (image/image.h:35): what the entropy decoder stores -- decoded samples, the
 * fill of a constant plane, the `zero` a resize fills with -- is narrowed like an assignment to pixel_type narrows it.  No effect on
 * valid streams (check_bit_depth, encoding.cpp:61-72, caps compressed samples at 15 bits of magnitude); on damaged ones an uncompressed
 * group or a constant plane can name a larger value, and the product's coefficient slab holds int16 samples as well. */
static inline int32_t px(int v) { return (int32_t)(int16_t)v; }
/* image/image.h:73-79 resize(): data.resize(w*h, zero) keeps existing leading samples */
static void ch_materialize(fo_channel *c) {
    /* planes made by the Image constructor (image.h:117-122, data(iw*ih,0)) are kept virtual
     * (data==NULL, size>0) until something stores into them */
    if (!c->data) c->data = (int32_t *)calloc(c->size ? c->size : 1, sizeof(int32_t));
}
static void ch_resize(fo_channel *c) {
    size_t want = (size_t)c->w * (size_t)c->h;
    if (!c->data) {
        size_t keep = c->size < want ? c->size : want;
        c->data = (int32_t *)malloc(sizeof(int32_t) * (want ? want : 1));
        for (size_t i = 0; i < keep; i++) c->data[i] = 0;
        for (size_t i = keep; i < want; i++) c->data[i] = px(c->zero);
    } else if (want > c->size) {
        c->data = (int32_t *)realloc(c->data, sizeof(int32_t) * (want ? want : 1));
        for (size_t i = c->size; i < want; i++) c->data[i] = px(c->zero);
    }
    c->size = want;
}
static void ch_fill(fo_channel *c, int v) {
    size_t want = (size_t)c->w * (size_t)c->h;
    c->data = (int32_t *)realloc(c->data, sizeof(int32_t) * (want ? want : 1));
    for (size_t i = 0; i < want; i++) c->data[i] = px(v);
    c->size = want;
}
/* image/image.h:82-85 checked accessor (unsigned compare => negative indices also give zero) */
static inline int ch_value(const fo_channel *c, int r, int col) {
    size_t idx = (size_t)((int64_t)r * c->w + col);
    if (idx >= c->size) return c->zero;
    return c->data ? c->data[idx] : 0;
}
static void img_insert_channel(fo_image *img, int at, const fo_channel *c) {
    img->ch = (fo_channel *)realloc(img->ch, sizeof(fo_channel) * (img->nch + 1));
    memmove(&img->ch[at + 1], &img->ch[at], sizeof(fo_channel) * (img->nch - at));
    img->ch[at] = *c;
    img->nch++;
}
static void img_erase_channels(fo_image *img, int from, int count) {
    for (int i = from; i < from + count; i++) free(img->ch[i].data);
    memmove(&img->ch[from], &img->ch[from + count], sizeof(fo_channel) * (img->nch - from - count));
    img->nch -= count;
}

void fo_free(fo_image *img) {
    if (!img) return;
    for (int i = 0; i < img->nch; i++) free(img->ch[i].data);
    free(img->ch);
    for (int i = 0; i < img->ntr; i++) free(img->tr[i].params);
    free(img->tr);
    free(img->group_start); free(img->group_channel);
    free(img);
}

/* ------------------------------------------------------------------------------------------- */
/* meta transforms (geometry only)                                                              */

/* transform/squeeze.h:266-321 */
static void default_squeeze_parameters(fo_transform *t, const fo_image *img) {
    int nb = img->nb_channels, m = img->nb_meta_channels;
    int cap = 0, n = 0;
    int *p = NULL;
#define PUSH3(a, b, c) do { if (n + 3 > cap) { cap = cap * 2 + 12; p = (int *)realloc(p, sizeof(int) * cap); } p[n++] = (a); p[n++] = (b); p[n++] = (c); } while (0)
    int w = img->ch[m].w, h = img->ch[m].h;
    int wide = (w > h);
    if (nb > 2 && img->ch[m + 1].w == w && img->ch[m + 1].h == h) {
        PUSH3(1 + 2, m + 1, m + 2);
        PUSH3(0 + 2, m + 1, m + 2);
    }
    if (!wide) {
        if (h > FO_MAX_FIRST_PREVIEW_SIZE) { PUSH3(0, m, m + nb - 1); h = (h + 1) / 2; }
    }
    while (w > FO_MAX_FIRST_PREVIEW_SIZE || h > FO_MAX_FIRST_PREVIEW_SIZE) {
        if (w > FO_MAX_FIRST_PREVIEW_SIZE) { PUSH3(1, m, m + nb - 1); w = (w + 1) / 2; }
        if (h > FO_MAX_FIRST_PREVIEW_SIZE) { PUSH3(0, m, m + nb - 1); h = (h + 1) / 2; }
    }
#undef PUSH3
    free(t->params);
    t->params = p; t->nparams = n;
}

/* transform/squeeze.h:323-360 */
static int meta_squeeze(fo_image *img, fo_transform *t) {
    if (!t->nparams) default_squeeze_parameters(t, img);
    for (int i = 0; i + 2 < t->nparams; i += 3) {
        int horizontal = t->params[i] & 1;
        int in_place = !(t->params[i] & 2);
        int beginc = t->params[i + 1], endc = t->params[i + 2];
        int offset = in_place ? endc + 1 : img->nb_meta_channels + img->nb_channels;
        if (beginc < 0 || endc < beginc || endc >= img->nch || offset > img->nch) return 0;
        for (int c = beginc; c <= endc; c++) {
            fo_channel d; ch_init(&d);
            d.hcshift = img->ch[c].hcshift; d.vcshift = img->ch[c].vcshift; d.component = img->ch[c].component;
            if (horizontal) {
                int w = img->ch[c].w;
                img->ch[c].w = (w + 1) / 2; img->ch[c].hshift++; img->ch[c].hcshift++;
                d.w = w - (w + 1) / 2; d.h = img->ch[c].h;
            } else {
                int h = img->ch[c].h;
                img->ch[c].h = (h + 1) / 2; img->ch[c].vshift++; img->ch[c].vcshift++;
                d.h = h - (h + 1) / 2; d.w = img->ch[c].w;
            }
            d.hshift = img->ch[c].hshift; d.vshift = img->ch[c].vshift;
            int at = offset + c - beginc;
            if (at > img->nch) return 0;
            img_insert_channel(img, at, &d);
        }
    }
    return 1;
}

/* transform/dct.h:120-129 (the reference's own zig-zag variant) */
static const int fo_zigzag[64] = {
    0, 1, 4, 15, 16, 35, 36, 63, 2, 3, 5, 14, 17, 34, 37, 62, 8, 7, 6, 13, 18, 33, 38, 61,
    9, 10, 11, 12, 19, 32, 39, 60, 24, 23, 22, 21, 20, 31, 40, 59, 25, 26, 27, 28, 29, 30, 41, 58,
    48, 47, 46, 45, 44, 43, 42, 57, 49, 50, 51, 52, 53, 54, 55, 56};
/* transform/dct.h:159-171 */
static const int fo_dct_cshifts[64] = {3, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};

/* transform/dct.h:173-207: position p -> component p % nb, coefficient p / nb; ordering[c][k] = k*nb + c */

/* transform/dct.h:209-246 */
static int meta_dct(fo_image *img, fo_transform *t) {
    if (!t->nparams) {
        t->params = (int *)malloc(sizeof(int) * 2);
        t->params[0] = 0; t->params[1] = img->nb_channels - 1; t->nparams = 2;
    }
    if (t->nparams < 2) return 0;
    int beginc = img->nb_meta_channels + t->params[0];
    int endc = img->nb_meta_channels + t->params[1];
    int nb = endc - beginc + 1;
    if (beginc < 0 || nb < 1 || endc >= img->nch) return 0;
    for (int c = beginc; c <= endc; c++) {
        fo_channel *ch = &img->ch[c];
        ch->w = (ch->w + 7) / 8; ch->h = (ch->h + 7) / 8;
        ch->hshift += 3; ch->vshift += 3; ch->hcshift += 3; ch->vcshift += 3;
    }
    for (int i = nb; i < 64 * nb; i++) {
        fo_channel d; ch_init(&d);
        int c = beginc + (i % nb);
        int coeff = i / nb;
        d.w = img->ch[c].w; d.h = img->ch[c].h;
        d.hshift = img->ch[c].hshift; d.vshift = img->ch[c].vshift;
        d.hcshift = fo_dct_cshifts[coeff] + img->ch[c].hcshift - 3;
        d.vcshift = fo_dct_cshifts[coeff] + img->ch[c].vcshift - 3;
        d.component = img->ch[c].component;
        img_insert_channel(img, img->nch, &d);
    }
    return 1;
}

/* transform/subsample.h:33-69; returns malloc'd expanded parameter list */
static int *subsample_params(const fo_transform *t, int *n_out) {
    int n = t->nparams;
    int *p = (int *)malloc(sizeof(int) * (n + 4));
    memcpy(p, t->params, sizeof(int) * n);
    if (n == 1) {
        switch (p[0]) {
            case 0: p[0] = 1; p[1] = 2; p[2] = 2; p[3] = 2; n = 4; break;
            case 1: p[0] = 1; p[1] = 2; p[2] = 2; p[3] = 1; n = 4; break;
            case 2: p[0] = 1; p[1] = 2; p[2] = 1; p[3] = 2; n = 4; break;
            case 3: p[0] = 1; p[1] = 2; p[2] = 4; p[3] = 1; n = 4; break;
            default: break;
        }
    }
    if (n % 4) n = 0;
    *n_out = n;
    return p;
}
/* transform/subsample.h:135-157 */
static int meta_subsample(fo_image *img, const fo_transform *t) {
    int n; int *p = subsample_params(t, &n);
    int ok = 1;
    for (int i = 0; i < n && ok; i += 4) {
        int c1 = p[i], c2 = p[i + 1], srh = p[i + 2], srv = p[i + 3];
        if (c1 < 0 || c2 >= img->nch || srh < 1 || srv < 1 || srh > 8 || srv > 8) { ok = 0; break; }   /* (the asserts of :143-144 are compiled out of the reference's release build: 4:1:1 goes through) */
        for (int c = c1; c <= c2; c++) {
            img->ch[c].w = (img->ch[c].w + srh - 1) / srh;
            img->ch[c].h = (img->ch[c].h + srv - 1) / srv;
            img->ch[c].hshift += (srh == 1 ? 0 : 1);
            img->ch[c].vshift += (srv == 1 ? 0 : 1);
        }
    }
    free(p);
    return ok;
}

/* Channel(w,h,min,max): image/image.h:64-65 -- data(w*h,0) is kept virtual (data==NULL, size=w*h) */
static void ch_ctor(fo_channel *c, int w, int h, int minval, int maxval) {
    ch_init(c);
    c->w = w; c->h = h; c->minval = minval; c->maxval = maxval; ch_setzero(c);
    c->size = (size_t)w * (size_t)h; c->data = NULL;
}
/* transform/palette.h:76-96 */
static int meta_palette(fo_image *img, const fo_transform *t) {
    if (t->nparams != 3) return 0;
    int begin_c = img->nb_meta_channels + t->params[0];
    int end_c = img->nb_meta_channels + t->params[1];
    if (begin_c > end_c || end_c >= img->nch || begin_c < 0) return 0;
    int nb = end_c - begin_c + 1;
    int nb_colors = t->params[2];
    if (nb_colors < 0 || (int64_t)nb_colors * nb > ((int64_t)1 << 28)) return 0; /* the reference would try to allocate this */
    img->nb_meta_channels++;
    img->nb_channels -= nb - 1;
    img_erase_channels(img, begin_c + 1, nb - 1);
    fo_channel pch; ch_ctor(&pch, nb_colors, nb, 0, 1);
    pch.hshift = -1;
    img_insert_channel(img, 0, &pch);
    return 1;
}
/* transform/approximate.h:62-78 */
static int approx_q(const fo_transform *t, int c) {
    int k = c + 2 - t->params[0];
    return k < t->nparams ? t->params[k] : t->params[t->nparams - 1];
}
static int meta_approximate(fo_image *img, const fo_transform *t) {
    if (t->nparams < 3) return 0;
    int nb = t->params[1] - t->params[0] + 1;
    if (nb < 1 || t->params[0] < 0 || t->params[1] >= img->nch) return 0;
    for (int c = t->params[0]; c <= t->params[1]; c++) {
        if (!approx_q(t, c)) continue;
        fo_channel copy = img->ch[c];   /* image.channel.push_back(image.channel[c]) */
        if (copy.data) {
            copy.data = (int32_t *)malloc(sizeof(int32_t) * (copy.size ? copy.size : 1));
            memcpy(copy.data, img->ch[c].data, sizeof(int32_t) * copy.size);
        }
        img_insert_channel(img, img->nch, &copy);
    }
    return 1;
}

/* transform/2dmatch.h:115-121,179-194 */
static void match_default_params(const fo_image *img, int *p) { p[0] = 0; p[1] = img->nb_channels - 1; p[2] = 0; p[3] = 1000000; }
static int meta_match(fo_image *img, fo_transform *t) {
    if (!t->nparams) {
        t->params = (int *)realloc(t->params, sizeof(int) * 4);
        match_default_params(img, t->params); t->nparams = 4;
    }
    if (t->nparams < 3) return 0;
    int begin_c = img->nb_meta_channels + t->params[0], end_c = img->nb_meta_channels + t->params[1];
    if (begin_c > end_c || end_c >= img->nch || begin_c < 0) return 0;
    img->nb_meta_channels++;
    fo_channel mch; ch_ctor(&mch, img->ch[begin_c].w, img->ch[begin_c].h, 0, 1);
    img_insert_channel(img, 0, &mch);
    return 1;
}

/* transform/permute.h:56-84.  Two forms: the permutation as transform parameters (only channel METADATA moves here:
 * channel[m+c] = old channel[m+i]), or -- without parameters -- as the samples of a 1-row meta-channel inserted in front */
static int meta_permute(fo_image *img, const fo_transform *t) {
    int nb = img->nch - img->nb_meta_channels;
    if (t->nparams == 0) {
        img->nb_meta_channels++;
        fo_channel pch; ch_ctor(&pch, nb, 1, 0, nb - 1);
        pch.hshift = -1;
        img_insert_channel(img, 0, &pch);
        return 1;
    }
    if (t->nparams > nb) return 0;
    fo_channel *in = (fo_channel *)malloc(sizeof(fo_channel) * img->nch);
    memcpy(in, img->ch, sizeof(fo_channel) * img->nch);
    for (int i = 0; i < t->nparams; i++) {
        int c = t->params[i];
        if (c < 0 || c >= t->nparams) { free(in); return 0; }
        for (int j = 0; j < i; j++) if (t->params[i] == t->params[j]) { free(in); return 0; }
        img->ch[img->nb_meta_channels + c] = in[img->nb_meta_channels + i];
    }
    free(in);
    return 1;
}
/* encoding/encoding.cpp:576-596: right after the permutation meta-channel has been decoded, when Permute is the LAST transform
 * of the list and has no parameters, the metadata of the channels still to be decoded is put in coded order */
static int inv_permute_meta(fo_image *img) {
    const fo_channel *p = &img->ch[0];
    int nb = p->w;
    fo_channel *in = (fo_channel *)malloc(sizeof(fo_channel) * img->nch);
    memcpy(in, img->ch, sizeof(fo_channel) * img->nch);
    for (int i = 0; i < nb; i++) {
        int c = ch_value(p, 0, i);
        int bad = (c < 0 || c >= nb || img->nb_meta_channels + c >= img->nch);
        for (int j = 0; j < i && !bad; j++) if (ch_value(p, 0, j) == c) bad = 1;
        if (bad) { memcpy(img->ch, in, sizeof(fo_channel) * img->nch); free(in); img->error = 1; return 0; }
        img->ch[img->nb_meta_channels + c] = in[img->nb_meta_channels + i];
    }
    free(in);
    return 1;
}

/* transform/transform.h:85-102 */
static int tr_has_parameters(int id) {
    switch (id) { case 3: case 6: case 7: case 4: case 8: case 9: case 10: return 1; default: return 0; }
}
/* transform/transform.cpp:66-81; returns 1 ok, 0 corrupt, -1 unsupported transform */
static int meta_apply(fo_image *img, fo_transform *t) {
    switch (t->id) {
        case TR_YCBCR: case TR_YCOCG: case TR_QUANTIZE: return 1;
        case TR_SUBSAMPLE: return meta_subsample(img, t);
        case TR_DCT: return meta_dct(img, t);
        case TR_SQUEEZE: return meta_squeeze(img, t);
        case TR_PALETTE: return meta_palette(img, t);
        case TR_APPROXIMATE: return meta_approximate(img, t);
        case TR_2DMATCH: return meta_match(img, t);
        case 9: return meta_permute(img, t);
        default: return -1;
    }
}

/* ------------------------------------------------------------------------------------------- */
/* context modelling                                                                            */
typedef struct { int lo, hi; } fo_range;

/* encoding/context_predict.h:67-120 */
static int init_properties(fo_range *pr, const fo_image *img, int beginc, int endc, int max_properties) {
    int n = 0, offset = 0;
    for (int j = beginc - 1; j >= 0 && offset < max_properties; j--) {
        const fo_channel *c = &img->ch[j];
        if (c->minval == c->maxval) continue;
        if (c->hshift < 0) continue;
        int minval = c->minval; if (minval > 0) minval = 0;
        int maxval = c->maxval; if (maxval < 0) maxval = 0;
        pr[n].lo = 0; pr[n].hi = iabs(maxval > -minval ? maxval : minval); n++; offset++;
        pr[n].lo = slog(minval); pr[n].hi = slog(maxval); n++; offset++;
    }
    int minval = 0x7FFFFFFF, maxval = (int)0x80000001, maxh = 0, maxw = 0;
    for (int j = beginc; j <= endc; j++) {
        const fo_channel *c = &img->ch[j];
        if (c->minval < minval) minval = c->minval;
        if (c->maxval > maxval) maxval = c->maxval;
        if (c->h > maxh) maxh = c->h;
        if (c->w > maxw) maxw = c->w;
    }
    if (minval > 0) minval = 0;
    if (maxval < 0) maxval = 0;
    int amax = iabs(minval) > iabs(maxval) ? iabs(minval) : iabs(maxval);
    pr[n].lo = 0; pr[n].hi = amax; n++;
    pr[n].lo = 0; pr[n].hi = amax; n++;
    pr[n].lo = slog(minval); pr[n].hi = slog(maxval); n++;
    pr[n].lo = slog(minval); pr[n].hi = slog(maxval); n++;
    pr[n].lo = 0; pr[n].hi = maxh - 1; n++;
    pr[n].lo = 0; pr[n].hi = maxw - 1; n++;
    pr[n].lo = minval + minval - maxval; pr[n].hi = maxval + maxval - minval; n++;
    pr[n].lo = minval + minval - maxval; pr[n].hi = maxval + maxval - minval; n++;
    for (int k = 0; k < 5; k++) { pr[n].lo = slog(minval - maxval); pr[n].hi = slog(maxval - minval); n++; }
    return n;
}

/* encoding/context_predict.h:233-289; refs[x*nref + offset] */
static void precompute_references(const fo_channel *ch, int y, const fo_image *img, int i, int max_properties,
                                  int32_t *refs, int nref) {
    int offset = 0;
    int oy = y << ch->vshift;
    for (int j = i - 1; j >= 0 && offset < max_properties; j--) {
        const fo_channel *rc = &img->ch[j];
        if (rc->minval == rc->maxval) continue;
        if (rc->hshift < 0) continue;
        int ry = oy >> rc->vshift;
        if (ry >= rc->h) ry = rc->h - 1;
        if (!rc->data || rc->size < (size_t)rc->w * rc->h) { ch_materialize((fo_channel *)rc); if (rc->size < (size_t)rc->w * rc->h) ch_resize((fo_channel *)rc); }
        const int32_t *row = rc->data + (size_t)ry * rc->w;
        if (ch->hshift == rc->hshift && ch->w <= rc->w) {
            for (int x = 0; x < ch->w; x++) { int v = row[x]; refs[x * nref + offset] = iabs(v); refs[x * nref + offset + 1] = slog(v); }
        } else if (ch->hshift < rc->hshift) {
            /* ch->hshift is -1 for a meta-channel coded after ordinary channels (Approximate on a palette): the reference
             * shifts by -1 there, which its x86 build evaluates with the count masked to 31, i.e. stepsize 0 */
            int stepsize = (1 << rc->hshift) >> (ch->hshift & 31);
            int x = 0, rx = 0, v;
            for (; rx < rc->w - 1; rx++) {
                v = row[rx];
                for (int s = 0; s < stepsize; s++, x++)
                    if (x < ch->w) { refs[x * nref + offset] = iabs(v); refs[x * nref + offset + 1] = slog(v); }
            }
            v = row[rx];
            while (x < ch->w) { refs[x * nref + offset] = iabs(v); refs[x * nref + offset + 1] = slog(v); x++; }
        } else {
            for (int x = 0; x < ch->w; x++) {
                int ox = x << ch->hshift;
                int rx = ox >> rc->hshift;
                if (rx >= rc->w) rx = rc->w - 1;
                int v = row[rx];
                refs[x * nref + offset] = iabs(v); refs[x * nref + offset + 1] = slog(v);
            }
        }
        offset += 2;
    }
}

/* encoding/context_predict.h:124-168 (the no_edge_case variant :170-206 computes the same thing) */
static inline int predict_and_properties(int32_t *p, const fo_channel *ch, int x, int y, int predictor, int offset) {
    const int32_t *d = ch->data;
    const int w = ch->w;
    int left = (x ? d[(size_t)y * w + x - 1] : ch->zero);
    int top = (y ? d[(size_t)(y - 1) * w + x] : ch->zero);
    int topleft = (x && y ? d[(size_t)(y - 1) * w + x - 1] : left);
    int topright = (x + 1 < w && y ? d[(size_t)(y - 1) * w + x + 1] : top);
    int leftleft = (x > 1 ? d[(size_t)y * w + x - 2] : left);
    int toptop = (y > 1 ? d[(size_t)(y - 2) * w + x] : top);
    p[offset++] = iabs(top);
    p[offset++] = iabs(left);
    p[offset++] = slog(top);
    p[offset++] = slog(left);
    p[offset++] = y;
    p[offset++] = x;
    p[offset++] = left + top - topleft;
    p[offset++] = topleft + topright - top;
    p[offset++] = slog(left - topleft);
    p[offset++] = slog(topleft - top);
    p[offset++] = slog(top - topright);
    p[offset++] = slog(top - toptop);
    p[offset++] = slog(left - leftleft);
    switch (predictor) {
        case 0: return ch->zero;
        case 1: return (left + top) / 2;
        case 2: return median3(left + top - topleft, left, top);
        case 3: return left;
        case 4: return top;
        case 5: return (left + topleft + top + topright) / 4;
        case 6: return CLAMPI(left + top - topleft, ch->minval, ch->maxval);
        default: return median3(left + top - topleft, left, top);
    }
}

/* ------------------------------------------------------------------------------------------- */
/* encoding/encoding.cpp:61-72 */
static int check_bit_depth(int minv, int maxv, int predictor) {
    int maxav = iabs(maxv);
    if (-minv > maxav) maxav = -minv;
    if (predictor > 0 && maxv - minv > maxav) maxav = maxv - minv;
    if (predictor > 0 && iabs(minv - maxv) > maxav) maxav = iabs(minv - maxv);
    return ilog2u((uint32_t)maxav) + 1 <= FO_MAX_BIT_DEPTH;
}

#define LIMIT_HIT(io, btl) (io_eof(io) || ((btl) && io_tell(io) >= (btl)))

/* encoding/encoding.cpp:209-219 */
static int corrupt_or_truncated(fo_io *io, fo_channel *c, size_t btl) {
    if (LIMIT_HIT(io, btl)) { ch_fill(c, 0); return 1; }
    return 0;
}

/* FO_STATS: the walk of a pixel whose left neighbour is not decoded yet -- nodes that test a left-dependent property fork; how
 * many exits of the 6-level root supernode stay reachable (leaves above level 6 count as exits), and how many of them are inner
 * nodes (= a second-level supernode would have to be fetched) */
static int fo_left_dependent(int kl, int y) { return kl == 1 || kl == 3 || kl == 12 || (y ? (kl == 6 || kl == 8) : (kl == 7 || kl == 9)); }
static void fo_spec_walk(const fo_node *n, int pos, int depth, const int32_t *props, int nref, int y, int *exits, int *inner) {
    if (n[pos].property == -1) { (*exits)++; return; }
    if (depth == 6) { (*exits)++; (*inner)++; return; }
    if (fo_left_dependent(n[pos].property - nref, y)) {
        fo_spec_walk(n, n[pos].childID, depth + 1, props, nref, y, exits, inner);
        fo_spec_walk(n, n[pos].childID + 1, depth + 1, props, nref, y, exits, inner);
    } else fo_spec_walk(n, props[n[pos].property] > n[pos].splitval ? n[pos].childID : n[pos].childID + 1, depth + 1, props, nref, y, exits, inner);
}

/* the -DFUIF_SPEC_WALK policy of the HIP kernel, simulated: the inner exits reachable with an unknown left neighbour, in the order
 * the kernel takes them (exit lanes ascending = the "<=" branch first); the first two that are not in a slot yet go to the two LDS
 * slots alternately; the first pixel of a 32-pixel chunk is not speculated for */
static void fo_spec_list(const fo_node *n, int pos, int depth, const int32_t *props, int nref, int y, int *list, int *cnt, int cap) {
    if (n[pos].property == -1) return;
    if (depth == 6) { if (*cnt < cap) list[(*cnt)++] = pos; return; }
    if (fo_left_dependent(n[pos].property - nref, y)) {
        fo_spec_list(n, n[pos].childID + 1, depth + 1, props, nref, y, list, cnt, cap);
        fo_spec_list(n, n[pos].childID, depth + 1, props, nref, y, list, cnt, cap);
    } else fo_spec_list(n, props[n[pos].property] > n[pos].splitval ? n[pos].childID : n[pos].childID + 1, depth + 1, props, nref, y, list, cnt, cap);
}

/* root exits reachable with an unknown left neighbour, in kernel order, with their kind (leaf: its number; inner: its node) */
static void fo_spec_exits(const fo_node *n, int pos, int depth, const int32_t *props, int nref, int y, int *node, int *leaf, int *cnt, int cap) {
    if (*cnt >= cap) return;
    if (n[pos].property == -1) { node[*cnt] = -1; leaf[(*cnt)++] = n[pos].childID; return; }
    if (depth == 6) { node[*cnt] = pos; leaf[(*cnt)++] = -1; return; }
    if (fo_left_dependent(n[pos].property - nref, y)) {
        fo_spec_exits(n, n[pos].childID + 1, depth + 1, props, nref, y, node, leaf, cnt, cap);
        fo_spec_exits(n, n[pos].childID, depth + 1, props, nref, y, node, leaf, cnt, cap);
    } else fo_spec_exits(n, props[n[pos].property] > n[pos].splitval ? n[pos].childID : n[pos].childID + 1, depth + 1, props, nref, y, node, leaf, cnt, cap);
}
/* third-level supernodes reachable (unknown left) below the second-level supernode rooted at `pos` (depth 6): inner nodes at depth 12 */
static void fo_spec_third(const fo_node *n, int pos, int depth, const int32_t *props, int nref, int y, int *list, int *cnt, int cap) {
    if (n[pos].property == -1) return;
    if (depth == 12) { if (*cnt < cap) list[(*cnt)++] = pos; return; }
    if (fo_left_dependent(n[pos].property - nref, y)) {
        fo_spec_third(n, n[pos].childID + 1, depth + 1, props, nref, y, list, cnt, cap);
        fo_spec_third(n, n[pos].childID, depth + 1, props, nref, y, list, cnt, cap);
    } else fo_spec_third(n, props[n[pos].property] > n[pos].splitval ? n[pos].childID : n[pos].childID + 1, depth + 1, props, nref, y, list, cnt, cap);
}
/* second speculative step: the leaves reachable (unknown left) within the 6 levels below node `pos` -- what a speculative round on a
 * supernode that is already in LDS could name, so that their chances can be fetched before the pixel's own walk gets there */
static void fo_spec_leaves(const fo_node *n, int pos, int depth, int maxdepth, const int32_t *props, int nref, int y, int *list, int *cnt, int cap) {
    if (n[pos].property == -1) { if (*cnt < cap) list[(*cnt)++] = n[pos].childID; return; }
    if (depth == maxdepth) return;
    if (fo_left_dependent(n[pos].property - nref, y)) {
        fo_spec_leaves(n, n[pos].childID + 1, depth + 1, maxdepth, props, nref, y, list, cnt, cap);
        fo_spec_leaves(n, n[pos].childID, depth + 1, maxdepth, props, nref, y, list, cnt, cap);
    } else fo_spec_leaves(n, props[n[pos].property] > n[pos].splitval ? n[pos].childID : n[pos].childID + 1, depth + 1, maxdepth, props, nref, y, list, cnt, cap);
}

/* optional per-group stream statistics (FO_STATS=1, printed to stderr): what the HIP kernel's per-symbol phases see
 * (walk depth, depth of the first left-dependent test, leaf repeats, exponent lengths); not part of the decode semantics */
/* (g_stats / g_st are declared next to read_symbol) */

/* optional leaf-locality statistics (FO_LEAFSIM=1): hit rates of direct-mapped leaf caches, used to
 * size the LDS leaf cache of the HIP kernel; not part of the decode semantics */
static int g_leafsim = -1;
static uint64_t g_ls_access, g_ls_same, g_ls_hit[4];
static int g_ls_tags[4][1024];
static const int g_ls_sizes[4] = {64, 128, 256, 512};
static void leafsim_reset(void) { for (int k = 0; k < 4; k++) for (int i = 0; i < 1024; i++) g_ls_tags[k][i] = -1; }
static void leafsim_access(int id, int *prev) {
    g_ls_access++;
    if (id == *prev) { g_ls_same++; return; }
    *prev = id;
    for (int k = 0; k < 4; k++) { int slot = id % g_ls_sizes[k]; if (g_ls_tags[k][slot] == id) g_ls_hit[k]++; else g_ls_tags[k][slot] = id; }
}
/* supernode locality: which 6-level subtree (below the root one) each walk round enters; static = the first K in
 * breadth-first order stay resident (what the kernel does), lru = K most recently used */
#define SN_POL 5
static const int g_sn_k[SN_POL] = {2, 3, 4, 7, 12};
static uint64_t g_sn_rounds, g_sn_static[SN_POL], g_sn_lru[SN_POL];
static int g_sn_lru_tags[SN_POL][16];
static void snsim_reset(void) { for (int k = 0; k < SN_POL; k++) for (int i = 0; i < 16; i++) g_sn_lru_tags[k][i] = -1; }
static void snsim_access(int id) {   /* id = breadth-first rank of the supernode, 1.. */
    g_sn_rounds++;
    for (int k = 0; k < SN_POL; k++) {
        if (id <= g_sn_k[k]) g_sn_static[k]++;
        int *t = g_sn_lru_tags[k], n = g_sn_k[k], at = -1;
        for (int i = 0; i < n; i++) if (t[i] == id) { at = i; break; }
        if (at >= 0) g_sn_lru[k]++; else at = n - 1;
        for (int i = at; i > 0; i--) t[i] = t[i - 1];
        t[0] = id;
    }
}
static uint64_t g_depth_hist[32];
void fo_depth_hist(uint64_t *out32) { for (int i = 0; i < 32; i++) out32[i] = g_depth_hist[i]; }
void fo_snsim_report(uint64_t *out) { out[0] = g_sn_rounds; for (int k = 0; k < SN_POL; k++) { out[1 + k] = g_sn_static[k]; out[1 + SN_POL + k] = g_sn_lru[k]; } }
/* breadth-first rank of every node that roots a supernode (depth multiple of 6, inner node) */
static void sn_number(const fo_node *n, int size, int *rank) {
    int *queue = (int *)malloc(sizeof(int) * (size + 1)), qh = 0, qt = 0, next = 0;
    for (int i = 0; i < size; i++) rank[i] = -1;
    queue[qt++] = 0;
    while (qh < qt) {
        const int r = queue[qh++];
        rank[r] = next++;
        /* the inner nodes 6 levels below r, left to right */
        int level[64], cnt = 1; level[0] = r;
        for (int d = 0; d < 6; d++) {
            int nl[64], nc = 0;
            for (int i = 0; i < cnt; i++) if (n[level[i]].property != -1) { nl[nc++] = n[level[i]].childID; nl[nc++] = n[level[i]].childID + 1; }
            cnt = nc; for (int i = 0; i < nc; i++) level[i] = nl[i];
        }
        for (int i = 0; i < cnt; i++) if (n[level[i]].property != -1) queue[qt++] = level[i];
    }
    free(queue);
}
/* leaf slots: which 6-level subtree (supernode) a leaf hangs off and its order among that supernode's leaves (depth first);
 * sizes the "leaf slots next to their supernode" layout of the HIP kernel.  slot[leafID] = order, root_owned[leafID] = the
 * supernode is the root one */
static uint64_t g_slot_acc[6];   /* accesses: total, root-owned, slot < 4, < 8, < 16, supernodes' leaves > 8 */
void fo_slotsim_report(uint64_t *out6) { for (int k = 0; k < 6; k++) out6[k] = g_slot_acc[k]; }
static void slot_dfs(const fo_node *n, int pos, int depth, int *slot, int *next) {
    if (n[pos].property == -1) { slot[n[pos].childID] = (*next)++; return; }
    if (depth == 6) return;   /* an inner node 6 levels down roots its own supernode */
    slot_dfs(n, n[pos].childID, depth + 1, slot, next);
    slot_dfs(n, n[pos].childID + 1, depth + 1, slot, next);
}
static void slot_number(const fo_node *n, int size, const int *sn_rank, int *slot, char *root_owned) {
    for (int r = 0; r < size; r++) {
        if (sn_rank[r] < 0) continue;
        int next = 0, before;
        /* mark ownership by numbering, then flag the root's */
        before = 0; (void)before;
        slot_dfs(n, r, 0, slot, &next);
        if (r == 0) { /* leaves numbered so far belong to the root supernode */
            for (int i = 0; i < size; i++) if (n[i].property == -1 && slot[n[i].childID] >= 0) root_owned[n[i].childID] = 1;
        }
    }
}
void fo_leafsim_report(uint64_t *out6) { out6[0] = g_ls_access; out6[1] = g_ls_same; for (int k = 0; k < 4; k++) out6[2 + k] = g_ls_hit[k]; }
static void dfs_number(const fo_node *n, int pos, int *ids, int *next) {
    if (n[pos].property == -1) { ids[n[pos].childID] = (*next)++; return; }
    dfs_number(n, n[pos].childID, ids, next);
    dfs_number(n, n[pos].childID + 1, ids, next);
}

static uint16_t g_table_tree[8192], g_table_pixel[8192];
static int g_tables_ready = 0;
static void ensure_tables(void) {
    if (g_tables_ready) return;
    fo_build_table(g_table_tree, 0xFFFFFFFFu / 19, 2);  /* compound.h:262 */
    fo_build_table(g_table_pixel, 0x0d000000u, 6);      /* encoding.h:54-55 */
    g_tables_ready = 1;
}

/* encoding/encoding.cpp:259-429.  returns 1 = continue, 0 = fatal */
static int decode_channel_group(fo_io *io, fo_image *img, int *beginc_io, size_t btl) {
    int beginc = *beginc_io;
    if (LIMIT_HIT(io, btl)) return 1;
    int firstbyte = read_varint(io);
    if (LIMIT_HIT(io, btl)) return 1;
    int endc = beginc + (firstbyte >> 4);
    int compress = firstbyte & 1;
    int predictor = (firstbyte & 14) >> 1;
    int global_minv = 1 - read_varint(io);
    if (LIMIT_HIT(io, btl)) return 1;
    if (global_minv == 1) global_minv = read_varint(io);
    if (LIMIT_HIT(io, btl)) return 1;
    int global_maxv = global_minv + read_varint(io);
    if (LIMIT_HIT(io, btl)) return 1;
    if (endc >= img->nch || endc < beginc) return 0;

    int firstrealc = beginc;
    for (int i = beginc; i <= endc; i++) {
        fo_channel *c = &img->ch[i];
        if ((int64_t)c->w * c->h <= 0) continue;
        c->minval = global_minv;
        c->maxval = global_maxv;
        if (endc > beginc && global_minv < global_maxv) {
            c->minval += read_varint(io);
            c->maxval = c->minval + read_varint(io);
        }
        if (c->minval == c->maxval) { ch_fill(c, c->minval); firstrealc++; }
        if (c->minval == 0 && c->maxval == 0) continue;
        c->q = read_varint(io);
        if (LIMIT_HIT(io, btl)) return corrupt_or_truncated(io, c, btl);
        if (c->maxval < c->minval) return 0; /* corrupt varint; symbol.h:45 asserts len >= 0 */
        if (compress && !check_bit_depth(c->minval, c->maxval, predictor)) return 0;
    }
    if (firstrealc > endc) { *beginc_io = endc; return 1; }

    fo_range pr[FO_MAX_PROPS];
    int nprops = init_properties(pr, img, beginc, endc, img->max_properties);

    int predictability = 2048;
    if (predictor == 0 && compress) {
        int rounded = read_varint(io);
        if (rounded < 1 || rounded > 127) return corrupt_or_truncated(io, &img->ch[firstrealc], btl);
        predictability = rounded * 32;
    }

    fo_rac rac;
    rac_init(&rac, io);

    if (!compress) {
        for (int i = beginc; i <= endc; i++) {
            fo_channel *c = &img->ch[i];
            if (c->minval == c->maxval) continue;
            ch_setzero(c);
            ch_resize(c);
            for (int y = 0; y < c->h; y++) {
                if (LIMIT_HIT(io, btl)) break;
                for (int x = 0; x < c->w; x++) {
                    c->data[(size_t)y * c->w + x] = px(uniform_read_int(&rac, c->minval, c->maxval - c->minval));
                    img->stat_symbols++;
                }
            }
            if (LIMIT_HIT(io, btl)) break;
        }
        img->stat_rac_decisions += rac.decisions;
        *beginc_io = endc;
        return 1;
    }

    /* MANIAC tree: compound.h:309-320 */
    fo_tree tree; tree.n = NULL; tree.size = 0; tree.cap = 0;
    tree.cap = 16; tree.n = (fo_node *)malloc(sizeof(fo_node) * tree.cap);
    tree.n[0].property = -1; tree.n[0].childID = 0; tree.n[0].splitval = 0; tree.size = 1;
    fo_meta meta;
    meta.rac = &rac; meta.table = g_table_tree; meta.nprops = nprops; meta.maxdepth = 0;
    for (int k = 0; k < 3; k++) fo_symbol_chance_init(meta.ctx[k], 1024); /* ZERO_CHANCE symbol.h:67 */
    for (int k = 0; k < nprops; k++) { meta.lo[k] = pr[k].lo; meta.hi[k] = pr[k].hi; }
    /* FO_DUMP_TREES=path (analysis tooling, tools/supernode_packing.py): every group's context tree with the number of walks through each node */
    const char *dump_path = getenv("FO_DUMP_TREES");
    uint32_t *visits = NULL;
    if (!read_subtree(&meta, &tree, 0, 0)) {
        free(tree.n);
        img->stat_rac_decisions += rac.decisions;
        return corrupt_or_truncated(io, &img->ch[beginc], btl);
    }
    if (dump_path) visits = (uint32_t *)calloc((size_t)tree.size, sizeof(uint32_t));

    if (tree.size > img->stat_max_tree_nodes) img->stat_max_tree_nodes = tree.size;
    /* FinalPropertySymbolCoder ctor: compound.h:213-225 */
    int nleaves = (tree.size + 1) / 2;
    uint16_t *leaves = (uint16_t *)malloc(sizeof(uint16_t) * CH_N * nleaves);
    fo_symbol_chance_init(leaves, predictability);
    for (int l = 1; l < nleaves; l++) memcpy(leaves + (size_t)l * CH_N, leaves, sizeof(uint16_t) * CH_N);
    for (int i = 0, leafID = 0; i < tree.size; i++)
        if (tree.n[i].property == -1) { tree.n[i].childID = (uint16_t)leafID; leafID++; }

    int32_t props[FO_MAX_PROPS];
    memset(props, 0, sizeof(props));
    if (g_leafsim < 0) g_leafsim = getenv("FO_LEAFSIM") ? 1 : 0;
    if (g_stats <= 0 && g_stats != -2) g_stats = getenv("FO_STATS") ? 1 : -2;
    int st_prev_leaf = -1;
    int st_tag[2] = {-1, -1}, st_victim = 0;
    int st_ltag[3][8], st_lvict[3] = {0, 0, 0};
    int st_wtag[2][8], st_wvict[2] = {0, 0}, st_prev2_leaf = -1;
    int st_dtag[4] = {-1, -1, -1, -1}, st_dvict = 0;
    int st_t3[2][2] = {{-1, -1}, {-1, -1}}, st_t3v = 0;
    for (int b = 0; b < 2; b++) for (int q = 0; q < 8; q++) st_wtag[b][q] = -1;
    for (int b = 0; b < 3; b++) for (int q = 0; q < 8; q++) st_ltag[b][q] = -1;
    if (g_stats > 0) memset(&g_st, 0, sizeof(g_st));
    const uint64_t st_dec0 = rac.decisions;
    int *ls_ids = NULL, ls_prev = -1;
    int *sn_rank = NULL;
    int *lf_slot = NULL; char *lf_root = NULL;
    if (g_leafsim) { ls_ids = (int *)malloc(sizeof(int) * nleaves); int nx = 0; dfs_number(tree.n, 0, ls_ids, &nx); leafsim_reset();
                     sn_rank = (int *)malloc(sizeof(int) * tree.size); sn_number(tree.n, tree.size, sn_rank); snsim_reset();
                     lf_slot = (int *)malloc(sizeof(int) * nleaves); lf_root = (char *)calloc(nleaves, 1);
                     for (int l = 0; l < nleaves; l++) lf_slot[l] = -1;
                     slot_number(tree.n, tree.size, sn_rank, lf_slot, lf_root); }
    const int nref = nprops - FO_NB_NONREF;

    for (int i = beginc; i <= endc; i++) {
        fo_channel *c = &img->ch[i];
        if (c->minval == c->maxval) continue;
        ch_setzero(c);
        ch_resize(c);
        if (tree.size == 1 && predictor == 0 && c->zero == 0) {
            /* fast track: encoding.cpp:371-383 */
            for (int y = 0; y < c->h; y++) {
                if (LIMIT_HIT(io, btl)) break;
                for (int x = 0; x < c->w; x++) {
                    c->data[(size_t)y * c->w + x] = px(read_symbol(&rac, leaves, g_table_pixel, c->minval, c->maxval));
                    img->stat_symbols++;
                }
            }
        } else {
            int32_t *refs = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nref > 0 ? nref : 1) * (size_t)(c->w > 0 ? c->w : 1));
            /* FO_STATS: the pixel ABOVE as a predictor of this pixel's second-level supernode and leaf (a prefetch keyed on the previous row) */
            int *up_sn = (int *)malloc(sizeof(int) * (size_t)(c->w + 2)), *up_leaf = (int *)malloc(sizeof(int) * (size_t)(c->w + 2));
            for (int x = 0; x < c->w + 2; x++) { up_sn[x] = -1; up_leaf[x] = -1; }
            unsigned long long v_rounds = 0, v_hit = 0, v_hit3 = 0, v_leaf = 0, v_leaf_hit = 0, v_leaf_hit3 = 0;
            for (int y = 0; y < c->h; y++) {
                if (LIMIT_HIT(io, btl)) break;
                precompute_references(c, y, img, beginc, img->max_properties, refs, nref);
                for (int x = 0; x < c->w; x++) {
                    for (int k = 0; k < nref; k++) props[k] = refs[x * nref + k];
                    int guess = predict_and_properties(props, c, x, y, predictor, nref);
                    int mn = c->minval - guess, mx = c->maxval - guess;
                    int diff;
                    if (mn == mx) diff = mn; /* compound.h:228 */
                    else {
                        /* find_leaf: compound.h:142-153 */
                        int pos = 0;
                        int depth = 0, pre = -1;
                        if (g_stats > 0 && (x & 31)) {   /* what the kernel fetched for this pixel while the previous one was decoded */
                            int cl[4], cn = 0;
                            fo_spec_list(tree.n, 0, 0, props, nref, y, cl, &cn, 2);
                            {   /* dense configuration + FUIF_SPEC_WALK: leaf speculation only through candidates that are resident ALREADY (no waiting),
                                 * plus leaves hanging off the root; first two root exits, two leaves each, four slots, same skip / use-once rules */
                                int en[2], el[2], ec = 0, dl[4], dn = 0;
                                fo_spec_exits(tree.n, 0, 0, props, nref, y, en, el, &ec, 2);
                                for (int k = 0; k < ec; k++) {
                                    if (en[k] < 0) { if (dn < 4) dl[dn++] = el[k]; }
                                    else if (en[k] == st_tag[0] || en[k] == st_tag[1]) { int cap2 = dn + 2 > 4 ? 4 : dn + 2; fo_spec_leaves(tree.n, en[k], 6, 12, props, nref, y, dl, &dn, cap2); }
                                }
                                for (int k = 0; k < dn; k++) {
                                    if (dl[k] == st_prev_leaf || dl[k] == st_prev2_leaf) continue;
                                    g_st.dl_cand++;
                                    int have = 0;
                                    for (int q = 0; q < 4; q++) if (st_dtag[q] == dl[k]) have = 1;
                                    if (!have) { st_dtag[st_dvict] = dl[k]; st_dvict = (st_dvict + 1) & 3; }
                                }
                            }
                            {   /* a third (fourth) slot for third-level supernodes: a speculative round on every candidate that is resident already */
                                for (int k = 0; k < cn; k++) {
                                    if (cl[k] != st_tag[0] && cl[k] != st_tag[1]) continue;
                                    int t3[2], tn = 0;
                                    fo_spec_third(tree.n, cl[k], 6, props, nref, y, t3, &tn, 2);
                                    for (int q = 0; q < tn; q++) {
                                        if (t3[q] != st_t3[0][0]) st_t3[0][0] = (q == 0 && k == 0) ? t3[q] : st_t3[0][0];          /* one slot: the first candidate only */
                                        if (t3[q] != st_t3[1][0] && t3[q] != st_t3[1][1]) { st_t3[1][st_t3v] = t3[q]; st_t3v ^= 1; }   /* two slots, round robin */
                                    }
                                }
                            }
                            for (int k = 0; k < cn; k++)
                                if (cl[k] != st_tag[0] && cl[k] != st_tag[1]) { st_tag[st_victim] = cl[k]; st_victim ^= 1; }
                        }
                        if (g_stats > 0 && (x & 63)) {
                            /* -DFUIF_SPEC_LEAF policy of the WIDE configuration (second-level supernodes resident in LDS): the first two root
                             * exits; a leaf exit names its leaf, an inner exit the first two leaves a speculative round on it reaches; leaves
                             * equal to the current leaf or to the one just written back are skipped; 4 / 8 slots, round robin */
                            int en[2], el[2], ec = 0, wl[4], wn = 0;
                            fo_spec_exits(tree.n, 0, 0, props, nref, y, en, el, &ec, 2);
                            for (int k = 0; k < ec; k++) {
                                if (en[k] < 0) { if (wn < 4) wl[wn++] = el[k]; }
                                else { int cap2 = wn + 2 > 4 ? 4 : wn + 2; fo_spec_leaves(tree.n, en[k], 6, 12, props, nref, y, wl, &wn, cap2); }
                            }
                            for (int k = 0; k < wn; k++) {
                                if (wl[k] == st_prev_leaf || wl[k] == st_prev2_leaf) continue;
                                g_st.wl_cand++;
                                for (int b = 0; b < 2; b++) {
                                    const int slots = 4 << b;
                                    int have = 0;
                                    for (int q = 0; q < slots; q++) if (st_wtag[b][q] == wl[k]) have = 1;
                                    if (!have) { st_wtag[b][st_wvict[b]] = wl[k]; st_wvict[b] = (st_wvict[b] + 1) % slots; }
                                }
                            }
                        }
                        if (g_stats > 0 && (x & 31)) {
                            int cl[4], cn = 0;
                            fo_spec_list(tree.n, 0, 0, props, nref, y, cl, &cn, 2);
                            /* leaf speculation, three slot budgets (2 / 4 / 8 leaf slots, round robin): leaves that hang off the root supernode
                             * and leaves below the (up to two) candidate second-level supernodes, first come first served */
                            int ll[16], ln = 0;
                            fo_spec_leaves(tree.n, 0, 0, 6, props, nref, y, ll, &ln, 8);
                            for (int k = 0; k < cn && ln < 16; k++) fo_spec_leaves(tree.n, cl[k], 6, 12, props, nref, y, ll, &ln, ln + 4 > 16 ? 16 : ln + 4);
                            for (int b = 0; b < 3; b++) {
                                const int slots = 2 << b;
                                for (int k = 0; k < ln && k < slots; k++) {
                                    int have = 0;
                                    for (int q = 0; q < slots; q++) if (st_ltag[b][q] == ll[k]) have = 1;
                                    if (!have) { st_ltag[b][st_lvict[b]] = ll[k]; st_lvict[b] = (st_lvict[b] + 1) % slots; }
                                }
                            }
                        }
                        int my_sn = -1;
                        while (tree.n[pos].property != -1) {
                            if (g_stats > 0 && depth == 6) { my_sn = pos; v_rounds++; if (pos == up_sn[x + 1]) v_hit++; if (pos == up_sn[x + 1] || pos == up_sn[x] || pos == up_sn[x + 2]) v_hit3++; }
                            if (g_stats > 0 && depth == 6) { g_st.spec_round2++; if (pos == st_tag[0] || pos == st_tag[1]) g_st.spec_hit++; }
                            if (g_stats > 0 && depth && depth % 6 == 0) g_st.rounds_behind++;
                            if (g_stats > 0 && depth == 12) { g_st.r3_rounds++; if (pos == st_t3[0][0]) g_st.r3_hit[0]++; if (pos == st_t3[1][0] || pos == st_t3[1][1]) g_st.r3_hit[1]++; }
                            img->stat_tree_steps++;
                            if (visits) visits[pos]++;
                            if (g_stats > 0 && pre < 0) {
                                const int kl = tree.n[pos].property - nref;  /* local properties that read `left` */
                                if (kl == 1 || kl == 3 || kl == 12 || (y ? (kl == 6 || kl == 8) : (kl == 7 || kl == 9))) pre = depth;
                            }
                            if (g_leafsim && depth && depth % 6 == 0) snsim_access(sn_rank[pos]);
                            depth++;
                            if (props[tree.n[pos].property] > tree.n[pos].splitval) pos = tree.n[pos].childID;
                            else pos = tree.n[pos].childID + 1;
                        }
                        if (g_stats > 0 && (int)tree.n[pos].childID != st_prev_leaf) {
                            for (int b = 0; b < 2; b++) for (int q = 0; q < (4 << b); q++)
                                if (st_wtag[b][q] == (int)tree.n[pos].childID) { g_st.wl_hit[b]++; st_wtag[b][q] = -1; break; }   /* used: the copy is not valid any longer */
                            for (int q = 0; q < 4; q++) if (st_dtag[q] == (int)tree.n[pos].childID) { g_st.dl_hit++; st_dtag[q] = -1; break; }
                            g_st.leaf_sw++;
                            for (int b = 0; b < 3; b++) for (int q = 0; q < (2 << b); q++) if (st_ltag[b][q] == (int)tree.n[pos].childID) { g_st.leaf_hit[b]++; break; }
                        }
                        if (g_stats > 0) {
                            const int lf = (int)tree.n[pos].childID;
                            if (lf != st_prev_leaf) { v_leaf++; if (lf == up_leaf[x + 1]) v_leaf_hit++; if (lf == up_leaf[x + 1] || lf == up_leaf[x] || lf == up_leaf[x + 2]) v_leaf_hit3++; }
                            up_sn[x + 1] = my_sn; up_leaf[x + 1] = lf;
                        }
                        if (g_stats > 0) {
                            int se = 0, si = 0;
                            fo_spec_walk(tree.n, 0, 0, props, nref, y, &se, &si);
                            g_st.spec_exits += se; g_st.spec_inner += si;
                            g_st.spec_hist[se <= 1 ? 0 : se == 2 ? 1 : se <= 4 ? 2 : se <= 8 ? 3 : se <= 16 ? 4 : 5]++;
                            if (pre < 0) pre = depth;
                            g_st.walked++; g_st.steps += depth; g_st.predepth += pre; g_st.prehist[pre > 23 ? 23 : pre]++;
                            if ((int)tree.n[pos].childID == st_prev_leaf) g_st.same_leaf++;
                            if ((int)tree.n[pos].childID != st_prev_leaf) st_prev2_leaf = st_prev_leaf;
                            st_prev_leaf = tree.n[pos].childID;
                        }
                        if (g_leafsim) { leafsim_access(ls_ids[tree.n[pos].childID], &ls_prev); g_depth_hist[depth > 31 ? 31 : depth]++;
                                         const int lid = tree.n[pos].childID, sl = lf_slot[lid];
                                         g_slot_acc[0]++; if (lf_root[lid]) g_slot_acc[1]++; else { if (sl < 4) g_slot_acc[2]++; if (sl < 8) g_slot_acc[3]++; if (sl < 16) g_slot_acc[4]++; } }
                        if (visits) visits[pos]++;
                        diff = read_symbol(&rac, leaves + (size_t)tree.n[pos].childID * CH_N, g_table_pixel, mn, mx);
                    }
                    c->data[(size_t)y * c->w + x] = px(diff + guess);
                    img->stat_symbols++;
                }
            }
            free(refs);
            if (g_stats > 0 && v_rounds > 100000)
                fprintf(stderr, "  vertical predictor c%d: second-level supernode = the one of the pixel above: %.1f %% of %llu rounds (above or its neighbours: %.1f %%); leaf = the leaf above: %.1f %% of %llu switches (3 candidates: %.1f %%)\n",
                        i, 100.0 * v_hit / v_rounds, v_rounds, 100.0 * v_hit3 / v_rounds, 100.0 * v_leaf_hit / (v_leaf ? v_leaf : 1), v_leaf, 100.0 * v_leaf_hit3 / (v_leaf ? v_leaf : 1));
            free(up_sn); free(up_leaf);
        }
        if (LIMIT_HIT(io, btl)) break;
    }
    if (g_stats > 0) {
        fprintf(stderr, "group c%d-%d %dx%d pred %d tree %d leaves %d walked %llu depth %.2f predepth %.2f sameleaf %.3f zero %.3f sign %.3f edec %.2f mdec %.2f dec/sym %.2f\n  ehist",
                beginc, endc, img->ch[beginc].w, img->ch[beginc].h, predictor, tree.size, nleaves, (unsigned long long)g_st.walked,
                g_st.walked ? (double)g_st.steps / g_st.walked : 0.0, g_st.walked ? (double)g_st.predepth / g_st.walked : 0.0,
                g_st.walked ? (double)g_st.same_leaf / g_st.walked : 0.0, g_st.walked ? (double)g_st.zero / g_st.walked : 0.0,
                g_st.walked ? (double)g_st.nsign / g_st.walked : 0.0, g_st.walked ? (double)g_st.edec / g_st.walked : 0.0,
                g_st.walked ? (double)g_st.mdec / g_st.walked : 0.0, g_st.walked ? (double)(rac.decisions - st_dec0) / g_st.walked : 0.0);
        for (int k = 0; k < 12; k++) fprintf(stderr, " %.3f", g_st.walked ? (double)g_st.ehist[k] / g_st.walked : 0.0);
        fprintf(stderr, "\n  prehist");
        for (int k = 0; k < 16; k++) fprintf(stderr, " %.3f", g_st.walked ? (double)g_st.prehist[k] / g_st.walked : 0.0);
        fprintf(stderr, "\n  spec: reachable root exits %.2f (inner %.2f) per walk; 1 / 2 / 3-4 / 5-8 / 9-16 / more:", g_st.walked ? (double)g_st.spec_exits / g_st.walked : 0.0,
                g_st.walked ? (double)g_st.spec_inner / g_st.walked : 0.0);
        for (int k = 0; k < 6; k++) fprintf(stderr, " %.3f", g_st.walked ? (double)g_st.spec_hist[k] / g_st.walked : 0.0);
        fprintf(stderr, "\n  spec policy (2 slots, first 2 candidates): %.3f second-level rounds per walk, %.1f %% of them found in a slot; all rounds behind the root %llu, hits %llu\n",
                g_st.walked ? (double)g_st.spec_round2 / g_st.walked : 0.0, g_st.spec_round2 ? 100.0 * g_st.spec_hit / g_st.spec_round2 : 0.0,
                (unsigned long long)g_st.rounds_behind, (unsigned long long)g_st.spec_hit);
        fprintf(stderr, "  wide configuration, leaf speculation (first 2 root exits, 2 leaves each): %.2f candidates fetched per walk; leaf switches served from 4 / 8 slots: %.1f %% / %.1f %% (switches %llu, served from 4 slots %llu)\n",
                g_st.walked ? (double)g_st.wl_cand / g_st.walked : 0.0, g_st.leaf_sw ? 100.0 * g_st.wl_hit[0] / g_st.leaf_sw : 0.0, g_st.leaf_sw ? 100.0 * g_st.wl_hit[1] / g_st.leaf_sw : 0.0,
                (unsigned long long)g_st.leaf_sw, (unsigned long long)g_st.wl_hit[0]);
        fprintf(stderr, "  third-level rounds: %.3f per walk; found in an extra slot filled through resident candidates: %.1f %% (1 slot) / %.1f %% (2 slots)\n",
                g_st.walked ? (double)g_st.r3_rounds / g_st.walked : 0.0, g_st.r3_rounds ? 100.0 * g_st.r3_hit[0] / g_st.r3_rounds : 0.0, g_st.r3_rounds ? 100.0 * g_st.r3_hit[1] / g_st.r3_rounds : 0.0);
        fprintf(stderr, "  dense configuration, leaf speculation through candidates that are already resident: %.2f leaf fetches per walk, %.1f %% of the leaf switches served\n",
                g_st.walked ? (double)g_st.dl_cand / g_st.walked : 0.0, g_st.leaf_sw ? 100.0 * g_st.dl_hit / g_st.leaf_sw : 0.0);
        fprintf(stderr, "  leaf speculation: %.3f leaf switches per walk; found among the speculated leaves with 2 / 4 / 8 slots: %.1f %% / %.1f %% / %.1f %%\n",
                g_st.walked ? (double)g_st.leaf_sw / g_st.walked : 0.0, g_st.leaf_sw ? 100.0 * g_st.leaf_hit[0] / g_st.leaf_sw : 0.0,
                g_st.leaf_sw ? 100.0 * g_st.leaf_hit[1] / g_st.leaf_sw : 0.0, g_st.leaf_sw ? 100.0 * g_st.leaf_hit[2] / g_st.leaf_sw : 0.0);
    }
    img->stat_rac_decisions += rac.decisions;
    free(ls_ids); free(sn_rank); free(lf_slot); free(lf_root);
    if (visits) {
        FILE *df = fopen(dump_path, "ab");
        if (df) {
            const int32_t hdr[4] = {beginc, tree.size, nref, nprops};
            fwrite(hdr, 4, 4, df);
            for (int k = 0; k < tree.size; k++) { const int32_t rec[4] = {tree.n[k].property, tree.n[k].childID, tree.n[k].splitval, (int32_t)visits[k]}; fwrite(rec, 4, 4, df); }
            fclose(df);
        }
        free(visits);
    }
    free(leaves);
    free(tree.n);
    *beginc_io = endc;
    return 1;
}

/* ------------------------------------------------------------------------------------------- */
/* encoding/encoding.cpp:599-720 */
fo_image *fo_decode(const uint8_t *blob, size_t n, int preview, int io_kind, int *ok) {
    ensure_tables();
    fo_image *img = (fo_image *)calloc(1, sizeof(fo_image));
    img->error = 1; img->maxval = 255; img->nb_frames = 1; /* Image(): image.h:124 */
    *ok = 0;
    fo_io io; io.p = blob; io.n = n; io.pos = 0; io.kind = io_kind; io.eof_flag = 0;
    if (n < 4) return img;
    int multi = 0;
    if (!memcmp(blob, "FUAF", 4)) multi = 1;
    else if (memcmp(blob, "FUIF", 4)) return img;
    io.pos = 4;
    int nb_channels = read_varint(&io) - '0';
    int bit_depth = read_varint(&io) - '&';
    int w = read_varint(&io) + 1;
    int h = read_varint(&io) + 1;
    int nb_frames = 1;
    if (multi) {
        nb_frames = read_varint(&io) + 2;
        (void)read_varint(&io); /* den-1 */
        int numerator = read_varint(&io);
        if (numerator) for (int i = 1; i < nb_frames; i++) (void)read_varint(&io);
        (void)read_varint(&io); /* loops */
    }
    int colormodel = read_varint(&io);
    img->max_properties = read_varint(&io);
    if (nb_channels < 0 || nb_channels > 64 || bit_depth < 1 || bit_depth > 30 || w < 1 || h < 1 ||
        (int64_t)w * h > ((int64_t)1 << 31) - 1 || img->max_properties < 0 || img->max_properties > FO_MAX_PROPS - FO_NB_NONREF)
        return img; /* the reference would crash or allocate garbage here: corrupt header */
    /* Image(w,h,maxval,nb_channels,cm): image.h:117-122 (planes start zero-filled; kept lazily here) */
    img->w = w; img->h = h; img->minval = 0; img->maxval = (1 << bit_depth) - 1;
    img->nb_channels = img->real_nb_channels = nb_channels; img->nb_meta_channels = 0;
    img->colormodel = colormodel; img->nb_frames = nb_frames; img->error = 0;
    img->ch = (fo_channel *)malloc(sizeof(fo_channel) * (nb_channels ? nb_channels : 1));
    img->nch = nb_channels;
    for (int i = 0; i < nb_channels; i++) {
        fo_channel *c = &img->ch[i]; ch_init(c);
        c->w = w; c->h = h; c->minval = 0; c->maxval = img->maxval; c->component = i; ch_setzero(c);
        c->size = (size_t)w * h; c->data = NULL; /* virtual zeros, see ch_materialize */
    }
    if (nb_channels < 1) { *ok = 1; return img; }

    int rel = 0;
    for (int s = 0; s < 5; s++) { img->responsive_offsets[s] = read_varint(&io) + rel; rel = img->responsive_offsets[s]; }
    rel = (int)io_tell(&io);
    for (int s = 0; s < 5; s++) img->responsive_offsets[s] += rel;

    int nb_transforms = read_varint(&io);
    if (nb_transforms < 0 || nb_transforms > 1024) { img->error = 1; return img; }
    img->tr = (fo_transform *)calloc(nb_transforms ? nb_transforms : 1, sizeof(fo_transform));
    int unsupported = 0;
    for (int i = 0; i < nb_transforms; i++) {
        int v = read_varint(&io);
        if (v < 0) { img->error = 1; return img; }
        fo_transform *t = &img->tr[img->ntr];
        t->id = v & 0xf; t->nparams = 0; t->params = NULL;
        if (tr_has_parameters(t->id)) {
            int np = v >> 4;
            t->params = (int *)malloc(sizeof(int) * (np ? np : 1));
            for (int j = 0; j < np; j++) t->params[j] = read_varint(&io);
            t->nparams = np;
        }
        img->ntr++;
        int r = meta_apply(img, t);
        if (r == 0) { img->error = 1; return img; }
        if (r < 0) unsupported = 1;
    }
    if (unsupported) { img->error = 1; *ok = -1; return img; }

    size_t btl = 0;
    if (preview >= 0 && preview < 5) btl = (size_t)img->responsive_offsets[preview];
    for (int i = 0; i < img->nch; i++) {
        if ((preview < 0 || io_tell(&io) < btl) && !io_eof(&io)) {
            if (!img->ch[i].w || !img->ch[i].h) continue;
            if (img->ngroups == img->groups_cap) {
                img->groups_cap = img->groups_cap ? img->groups_cap * 2 : 64;
                img->group_start = (uint32_t *)realloc(img->group_start, sizeof(uint32_t) * img->groups_cap);
                img->group_channel = (int32_t *)realloc(img->group_channel, sizeof(int32_t) * img->groups_cap);
            }
            img->group_start[img->ngroups] = (uint32_t)io_tell(&io); img->group_channel[img->ngroups] = i; img->ngroups++;
            const int was_first = (i == 0);
            if (!decode_channel_group(&io, img, &i, btl)) { img->bytes_consumed = io_tell(&io); return img; }
            /* encoding.cpp:712 */
            if (was_first && img->ntr > 0 && img->tr[img->ntr - 1].id == 9 && img->tr[img->ntr - 1].nparams == 0) inv_permute_meta(img);
        } else break;
    }
    img->bytes_consumed = io_tell(&io);
    *ok = 1;
    return img;
}

/* ------------------------------------------------------------------------------------------- */
/* inverse transforms                                                                           */

/* transform/squeeze.h:61-77 */
int fo_smooth_tendency(int B, int a, int n) {
    int diff = 0;
    if (B >= a && a >= n) {
        diff = (4 * B - 3 * n - a + 6) / 12;
        if (diff - (diff & 1) > 2 * (B - a)) diff = 2 * (B - a) + 1;
        if (diff + (diff & 1) > 2 * (a - n)) diff = 2 * (a - n);
    } else if (B <= a && a <= n) {
        diff = (4 * B - 3 * n - a - 6) / 12;
        if (diff + (diff & 1) < 2 * (B - a)) diff = 2 * (B - a) - 1;
        if (diff - (diff & 1) < 2 * (a - n)) diff = 2 * (a - n);
    }
    return diff;
}

static void chout_from(fo_channel *o, const fo_channel *in, int w, int h) {
    ch_init(o);
    o->w = w; o->h = h; o->minval = in->minval; o->maxval = in->maxval; o->q = in->q;
    o->hshift = in->hshift; o->vshift = in->vshift; o->hcshift = in->hcshift; o->vcshift = in->vcshift;
    o->component = in->component;
    ch_setzero(o);
    size_t want = (size_t)w * h;
    o->data = (int32_t *)calloc(want ? want : 1, sizeof(int32_t));
    o->size = want;
}
/* checked store, image.h:84-85 (out-of-range stores land in `zero`, i.e. are dropped) */
static inline void ch_store(fo_channel *c, int r, int col, int v) {
    size_t idx = (size_t)((int64_t)r * c->w + col);
    if (idx < c->size) { if (!c->data) ch_materialize(c); c->data[idx] = v; } else c->zero = v;
}

/* transform/squeeze.h:81-132 */
static void inv_hsqueeze(fo_image *img, int c, int rc) {
    fo_channel *chin = &img->ch[c];
    const fo_channel *res = &img->ch[rc];
    fo_channel out;
    chout_from(&out, chin, chin->w + res->w, chin->h);
    out.hshift = chin->hshift - 1; out.hcshift = chin->hcshift - 1;
    for (int y = 0; y < chin->h; y++) {
        int avg = ch_value(chin, y, 0);
        int next_avg = (1 < chin->w ? ch_value(chin, y, 1) : avg);
        int tendency = fo_smooth_tendency(avg, avg, next_avg);
        int diff = ch_value(res, y, 0) + tendency;
        int A = ((avg << 1) + diff + (diff > 0 ? -(diff & 1) : (diff & 1))) >> 1;
        int B = A - diff;
        ch_store(&out, y, 0, A);
        ch_store(&out, y, 1, B);
        for (int x = 1; x < res->w; x++) {
            int dmt = ch_value(res, y, x);
            avg = ch_value(chin, y, x);
            next_avg = (x + 1 < chin->w ? ch_value(chin, y, x + 1) : avg);
            int left = out.data[(size_t)y * out.w + (x << 1) - 1];
            tendency = fo_smooth_tendency(left, avg, next_avg);
            diff = dmt + tendency;
            A = ((avg << 1) + diff + (diff > 0 ? -(diff & 1) : (diff & 1))) >> 1;
            ch_store(&out, y, x << 1, A);
            B = A - diff;
            ch_store(&out, y, (x << 1) + 1, B);
        }
        if (out.w & 1) ch_store(&out, y, out.w - 1, ch_value(chin, y, chin->w - 1));
    }
    free(chin->data);
    *chin = out;
}

/* transform/squeeze.h:173-224 */
static void inv_vsqueeze(fo_image *img, int c, int rc) {
    fo_channel *chin = &img->ch[c];
    const fo_channel *res = &img->ch[rc];
    fo_channel out;
    chout_from(&out, chin, chin->w, chin->h + res->h);
    out.vshift = chin->vshift - 1; out.vcshift = chin->vcshift - 1;
    for (int x = 0; x < chin->w; x++) {
        int dmt = ch_value(res, 0, x);
        int avg = ch_value(chin, 0, x);
        int next_avg = avg;
        if (1 < chin->h) next_avg = ch_value(chin, 1, x);
        int tendency = fo_smooth_tendency(avg, avg, next_avg);
        int diff = dmt + tendency;
        int A = ((avg << 1) + diff + (diff > 0 ? -(diff & 1) : (diff & 1))) >> 1;
        ch_store(&out, 0, x, A);
        int B = A - diff;
        ch_store(&out, 1, x, B);
    }
    for (int y = 1; y < res->h; y++) {
        for (int x = 0; x < chin->w; x++) {
            int dmt = ch_value(res, y, x);
            int avg = ch_value(chin, y, x);
            int next_avg = avg;
            if (y + 1 < chin->h) next_avg = ch_value(chin, y + 1, x);
            int top = out.data[(size_t)((y << 1) - 1) * out.w + x];
            int tendency = fo_smooth_tendency(top, avg, next_avg);
            int diff = dmt + tendency;
            int A = ((avg << 1) + diff + (diff > 0 ? -(diff & 1) : (diff & 1))) >> 1;
            ch_store(&out, y << 1, x, A);
            int B = A - diff;
            ch_store(&out, (y << 1) + 1, x, B);
        }
    }
    if (out.h & 1) {
        int y = chin->h - 1;
        for (int x = 0; x < chin->w; x++) ch_store(&out, y << 1, x, ch_value(chin, y, x));
    }
    free(chin->data);
    *chin = out;
}

/* transform/squeeze.h:363-388 (inverse branch) */
static int inv_squeeze(fo_image *img, const fo_transform *t) {
    for (int i = t->nparams - 3; i >= 0; i -= 3) {
        int horizontal = t->params[i] & 1;
        int in_place = !(t->params[i] & 2);
        int beginc = t->params[i + 1], endc = t->params[i + 2];
        int offset = in_place ? endc + 1 : img->nb_meta_channels + img->nb_channels;
        if (beginc < 0 || endc < beginc || offset + (endc - beginc) >= img->nch) return 0;
        for (int c = beginc; c <= endc; c++) {
            fo_channel *res = &img->ch[offset + c - beginc];
            if (res->size == 0) ch_resize(res); /* zero-fill missing residuals :379-383 */
            if (horizontal) inv_hsqueeze(img, c, offset + c - beginc);
            else inv_vsqueeze(img, c, offset + c - beginc);
        }
        img_erase_channels(img, offset, endc - beginc + 1);
    }
    return 1;
}

/* transform/ycocg.h:33-63 */
static int inv_ycocg(fo_image *img) {
    int m = img->nb_meta_channels;
    if (img->nb_channels < 3) return 0;
    fo_channel *c0 = &img->ch[m], *c1 = &img->ch[m + 1], *c2 = &img->ch[m + 2];
    int w = c0->w, h = c0->h;
    if (c1->w < w || c1->h < h || c2->w < w || c2->h < h) return 0;
    const int maxval = img->maxval;
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            int Y = CLAMPI(ch_value(c0, y, x), 0, maxval);
            int Co = ch_value(c1, y, x);
            int Cg = ch_value(c2, y, x);
            int G = CLAMPI(Y - ((-Cg) >> 1), 0, maxval);
            int B = CLAMPI(Y + ((1 - Cg) >> 1) - (Co >> 1), 0, maxval);
            int R = CLAMPI(Co + B, 0, maxval);
            ch_store(c0, y, x, R);
            ch_store(c1, y, x, G);
            ch_store(c2, y, x, B);
        }
    }
    return 1;
}

/* transform/ycbcr.h:33-63: float operands, double arithmetic, truncating store */
static int inv_ycbcr(fo_image *img) {
    if (img->nch < 3) return 0;
    fo_channel *c0 = &img->ch[0], *c1 = &img->ch[1], *c2 = &img->ch[2];
    int w = c0->w, h = c0->h;
    if (c1->w < w || c1->h < h || c2->w < w || c2->h < h) return 0;
    float half = (float)((img->maxval + 1) / 2);
    const int mn = img->minval, mx = img->maxval;
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            float yy = (float)ch_value(c0, y, x);
            float cb = (float)ch_value(c1, y, x) - half;
            float cr = (float)ch_value(c2, y, x) - half;
            double r = yy + 1.402 * cr + 0.5;
            double g = yy - 0.344136 * cb - 0.714136 * cr + 0.5;
            double b = yy + 1.772 * cb + 0.5;
            ch_store(c0, y, x, (int)(r < mn ? mn : (r > mx ? mx : r)));
            ch_store(c1, y, x, (int)(g < mn ? mn : (g > mx ? mx : g)));
            ch_store(c2, y, x, (int)(b < mn ? mn : (b > mx ? mx : b)));
        }
    }
    return 1;
}

/* transform/quantize.h:32-49 */
static int inv_quantize(fo_image *img) {
    for (int c = img->nb_meta_channels; c < img->nch; c++) {
        fo_channel *ch = &img->ch[c];
        if (ch->size == 0) continue;
        int q = ch->q;
        if (q == 1) continue;
        for (int y = 0; y < ch->h; y++)
            for (int x = 0; x < ch->w; x++) ch_store(ch, y, x, ch_value(ch, y, x) * q);
        ch->minval *= q; ch->maxval *= q; ch->q = 1;
    }
    return 1;
}

/* transform/dct.h:60-77 (values exactly as printed in the reference) */
static const double kDCT[64] = {
    0.3535533906, 0.3535533906, 0.3535533906, 0.3535533906, 0.3535533906, 0.3535533906, 0.3535533906, 0.3535533906,
    0.4903926402, 0.4157348062, 0.2777851165, 0.0975451610, -0.0975451610, -0.2777851165, -0.4157348062, -0.4903926402,
    0.4619397663, 0.1913417162, -0.1913417162, -0.4619397663, -0.4619397663, -0.1913417162, 0.1913417162, 0.4619397663,
    0.4157348062, -0.0975451610, -0.4903926402, -0.2777851165, 0.2777851165, 0.4903926402, 0.0975451610, -0.4157348062,
    0.3535533906, -0.3535533906, -0.3535533906, 0.3535533906, 0.3535533906, -0.3535533906, -0.3535533906, 0.3535533906,
    0.2777851165, -0.4903926402, 0.0975451610, 0.4157348062, -0.4157348062, -0.0975451610, 0.4903926402, -0.2777851165,
    0.1913417162, -0.4619397663, 0.4619397663, -0.1913417162, -0.1913417162, 0.4619397663, -0.4619397663, 0.1913417162,
    0.0975451610, -0.2777851165, 0.4157348062, -0.4903926402, 0.4903926402, -0.4157348062, 0.2777851165, -0.0975451610,
};
/* transform/dct.h:88-107: columns first, then rows; sequential += in double, no FMA */
void fo_idct8x8(double *block) {
    double tmp[64];
    for (int x = 0; x < 8; x++)
        for (int o = 0; o < 8; o++) {
            double acc = 0.0;
            for (int u = 0; u < 8; u++) acc += kDCT[8 * u + o] * block[u * 8 + x];
            tmp[o * 8 + x] = acc;
        }
    for (int y = 0; y < 8; y++)
        for (int o = 0; o < 8; o++) {
            double acc = 0.0;
            for (int u = 0; u < 8; u++) acc += kDCT[8 * u + o] * tmp[8 * y + u];
            block[8 * y + o] = acc;
        }
}

/* transform/dct.h:249-296 */
static int inv_dct(fo_image *img, fo_transform *t) {
    if (!t->nparams) {
        t->params = (int *)malloc(sizeof(int) * 2);
        t->params[0] = 0; t->params[1] = img->nb_channels - 1; t->nparams = 2;
    }
    int beginc = img->nb_meta_channels + t->params[0];
    int endc = img->nb_meta_channels + t->params[1];
    int nb = endc - beginc + 1;
    int offset = img->nch - 63 * nb;
    if (offset <= endc || nb < 1 || beginc < 0) return 0;
    for (int c = beginc; c <= endc; c++) {
        fo_channel *dc = &img->ch[c];
        int bw = img->ch[c - beginc + offset].w, bh = img->ch[c - beginc + offset].h;
        if (dc->w < bw) bw = dc->w;
        if (dc->h < bh) bh = dc->h;
        fo_channel out; ch_init(&out);
        out.w = bw * 8; out.h = bh * 8; out.minval = 0; out.maxval = 0; out.zero = 0;
        out.size = (size_t)out.w * out.h;
        out.data = (int32_t *)calloc(out.size ? out.size : 1, sizeof(int32_t));
        out.component = dc->component;
        out.hshift = dc->hshift - 3; out.vshift = dc->vshift - 3;
        out.hcshift = dc->hcshift - 3; out.vcshift = dc->hcshift - 3; /* sic: dct.h:280 */
        float DCoffset = (float)((img->maxval + 1.0) * 4.0);
        for (int by = 0; by < bh; by++)
            for (int bx = 0; bx < bw; bx++) {
                double block[64];
                block[0] = (double)((float)ch_value(dc, by, bx) + DCoffset);
                for (int i = 1; i < 64; i++) {
                    /* ordering[comp][k] = k*nb + comp (dct.h:173-207) */
                    const fo_channel *ac = &img->ch[offset - nb + fo_zigzag[i] * nb + (c - beginc)];
                    block[i] = (double)ch_value(ac, by, bx);
                }
                fo_idct8x8(block);
                for (int y = 0; y < 8; y++)
                    for (int x = 0; x < 8; x++) out.data[(size_t)(by * 8 + y) * out.w + bx * 8 + x] = (int32_t)round(block[y * 8 + x]);
            }
        free(dc->data);
        *dc = out;
    }
    img_erase_channels(img, offset, nb * 63);
    return 1;
}

/* transform/subsample.h:73-127 */
static int inv_subsample(fo_image *img, const fo_transform *t) {
    int n; int *p = subsample_params(t, &n);
    for (int i = 0; i < n; i += 4) {
        int c1 = p[i], c2 = p[i + 1], srh = p[i + 2], srv = p[i + 3];
        for (int c = c1; c <= c2 && c < img->nch; c++) {
            fo_channel *in = &img->ch[c];
            int ow = in->w, oh = in->h;
            if (ow >= img->ch[img->nb_meta_channels].w && oh >= img->ch[img->nb_meta_channels].h) continue;
            fo_channel out; ch_init(&out);
            out.w = ow * srh; out.h = oh * srv; out.minval = in->minval; out.maxval = in->maxval; ch_setzero(&out);
            out.size = (size_t)out.w * out.h;
            out.data = (int32_t *)calloc(out.size ? out.size : 1, sizeof(int32_t));
            if (srv <= 2 && srh <= 2) {
                if (srh == 2) {
                    for (int y = 0; y < oh; y++)
                        for (int x = 0; x < ow; x++) {
                            ch_store(&out, y * srv, x * srh, (3 * ch_value(in, y, x) + ch_value(in, y, (x ? x - 1 : 0)) + 1) >> 2);
                            ch_store(&out, y * srv, x * srh + 1, (3 * ch_value(in, y, x) + ch_value(in, y, (x + 1 < ow ? x + 1 : x)) + 2) >> 2);
                        }
                } else {
                    for (int y = 0; y < oh; y++)
                        for (int x = 0; x < ow; x++) ch_store(&out, y * srv, x, ch_value(in, y, x));
                }
                if (srv == 2) {
                    fo_channel orig = out;
                    orig.data = (int32_t *)malloc(sizeof(int32_t) * (out.size ? out.size : 1));
                    memcpy(orig.data, out.data, sizeof(int32_t) * out.size);
                    for (int y = 0; y < oh; y++)
                        for (int x = 0; x < ow * srh; x++) {
                            ch_store(&out, y * srv, x, (3 * ch_value(&orig, y * srv, x) + ch_value(&orig, (y ? (y - 1) * srv : 0), x) + 1) >> 2);
                            ch_store(&out, y * srv + 1, x, (3 * ch_value(&orig, y * srv, x) + ch_value(&orig, (y + 1 < oh ? (y + 1) * srv : y * srv), x) + 2) >> 2);
                        }
                    free(orig.data);
                }
            } else {
                for (int y = 0; y < oh * srv; y++)
                    for (int x = 0; x < ow * srh; x++) ch_store(&out, y, x, ch_value(in, y / srv, x / srh));
            }
            free(in->data);
            *in = out;
        }
    }
    free(p);
    return 1;
}

/* transform/transform.cpp:48-63 (inverse) */
/* transform/palette.h:32-68 */
static int inv_palette(fo_image *img, const fo_transform *t) {
    if (img->nb_meta_channels < 1 || t->nparams != 3) return 0;
    int nb = img->ch[0].h;
    int c0 = img->nb_meta_channels + t->params[0];
    if (c0 >= img->nch) return 0;
    int w = img->ch[c0].w, h = img->ch[c0].h;
    for (int i = 1; i < nb; i++) {
        /* palette.h:52-55: every new channel is inserted at c0+1 and THEN channel[c0+i] is labelled, so the
         * label lands on the channel inserted first each time: only position c0+nb-1 ends up with a component */
        fo_channel n; ch_ctor(&n, w, h, 0, 1);
        ch_materialize(&n);
        img_insert_channel(img, c0 + 1, &n);
        img->ch[c0 + i].component = t->params[0] + i;
    }
    const fo_channel *pal = &img->ch[0];
    fo_channel *idx = &img->ch[c0];
    ch_materialize(idx);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int index = ch_value(idx, y, x);
            index = CLAMPI(index, 0, pal->w - 1);
            for (int c = 0; c < nb; c++) {
                fo_channel *o = &img->ch[c0 + c];
                size_t at = (size_t)y * o->w + x;
                if (at < o->size) o->data[at] = ch_value(pal, c, index);   /* value(r,c) = ...: out-of-range stores hit `zero` */
            }
        }
    img->nb_channels += nb - 1;
    img->nb_meta_channels--;
    img_erase_channels(img, 0, 1);
    return 1;
}
/* transform/approximate.h:32-60 */
static int inv_approximate(fo_image *img, const fo_transform *t) {
    int beginc = t->params[0], endc = t->params[1];
    int offset = img->nch - (endc - beginc + 1);
    for (int c = beginc; c <= endc; c++) if (!approx_q(t, c)) offset++;
    if (beginc < 0 || endc < beginc || offset <= endc || offset > img->nch) return 0;
    int i = 0;
    for (int c = beginc; c <= endc; c++) {
        int q = approx_q(t, c) + 1;
        if (q == 1) continue;
        fo_channel *ch = &img->ch[c];
        const fo_channel *chr = &img->ch[offset + i];
        i++;
        int have = chr->size != 0;
        if (have) ch->q = chr->q;
        ch_materialize(ch);
        for (int y = 0; y < ch->h; y++)
            for (int x = 0; x < ch->w; x++) {
                size_t at = (size_t)y * ch->w + x;
                if (at >= ch->size) continue;
                ch->data[at] = ch->data[at] * q + (have ? ch_value(chr, y, x) : 0);
            }
    }
    img_erase_channels(img, offset, img->nch - offset);
    return 1;
}

/* transform/2dmatch.h:50-78 */
static void match_offset(int code, int *xo, int *yo) {
    int layer = 0, size = 4;
    while (code > size) { code -= size; layer++; size += 4; }
    if (layer & 1) {
        if (code <= layer) { *xo = 1 + layer; *yo = -code; }
        else if (code <= 3 + 3 * layer) { *xo = 2 + 2 * layer - code; *yo = -1 - layer; }
        else { *xo = -1 - layer; *yo = -4 - 4 * layer + code; }
    } else {
        if (code <= 1 + layer) { *xo = -1 - layer; *yo = 1 - code; }
        else if (code <= 4 + 3 * layer) { *xo = -3 - 2 * layer + code; *yo = -1 - layer; }
        else { *xo = 1 + layer; *yo = -5 - 4 * layer + code; }
    }
}
/* Channel::value(r,c) on a plane being rewritten in place: image.h:82-85 (linear index check only) */
static inline int32_t *ch_slot(fo_channel *c, int r, int col, int32_t *zero_slot) {
    size_t idx = (size_t)((int64_t)r * c->w + col);
    if (idx >= c->size) { *zero_slot = c->zero; return zero_slot; }
    return &c->data[idx];
}
/* transform/2dmatch.h:123-177 */
static int inv_match(fo_image *img, fo_transform *t) {
    if (img->nb_meta_channels < 1) return 0;
    int dflt[4]; const int *p = t->params; int np = t->nparams;
    if (!np) { match_default_params(img, dflt); p = dflt; np = 4; }
    if (np < 3) return 0;
    fo_channel *m = &img->ch[0];
    int c0 = img->nb_meta_channels + p[0], cn = img->nb_meta_channels + p[1];
    if (c0 >= img->nch || cn >= img->nch || c0 < 0) return 0;
    int softmatch = p[2];
    int w = img->ch[c0].w, h = img->ch[c0].h;
    for (int c = c0; c <= cn; c++) ch_materialize(&img->ch[c]);
    int fh = h / img->nb_frames;
    int offsetcode = 2 * fh * fh + (fh & 1);
    if (m->q != 1 && m->q != offsetcode) return 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int z = ch_value(m, y, x);
            if (!z) continue;
            int xo = 0, yo;
            if (m->q == 1) {
                if (z < 0 || z > m->maxval) return 0;   /* offsets_table[z] out of range in the reference */
                match_offset(z, &xo, &yo);
            } else yo = -z * fh;
            for (int c = c0; c <= cn; c++) {
                fo_channel *ch = &img->ch[c];
                int32_t zs_a, zs_b;
                int32_t *dst = ch_slot(ch, y, x, &zs_a);
                int32_t src = *ch_slot(ch, y + yo, x + xo, &zs_b);
                if (softmatch) *dst += src; else *dst = src;
            }
        }
    img->nb_meta_channels--;
    img_erase_channels(img, 0, 1);
    return 1;
}

/* transform/permute.h:31-54: channel[m+i] = old channel[m+c_i], c from the parameters or from the meta-channel (then dropped) */
static int inv_permute(fo_image *img, const fo_transform *t) {
    const int use_channel = (t->nparams == 0);
    if (use_channel && (img->nb_meta_channels < 1 || img->nch < 1)) return 0;
    const int n = use_channel ? img->ch[0].w : t->nparams;
    const int m = img->nb_meta_channels;
    if (m + n > img->nch) return 0;
    fo_channel *tmp = (fo_channel *)malloc(sizeof(fo_channel) * img->nch);
    memcpy(tmp, img->ch, sizeof(fo_channel) * img->nch);
    char *used = (char *)calloc(n ? n : 1, 1);
    for (int i = 0; i < n; i++) {
        int c = use_channel ? ch_value(&img->ch[0], 0, i) : t->params[i];
        if (c < 0 || c >= n || used[c]) { memcpy(img->ch, tmp, sizeof(fo_channel) * img->nch); free(tmp); free(used); return 0; }  /* the reference indexes blindly */
        used[c] = 1;
        img->ch[m + i] = tmp[m + c];
    }
    free(tmp); free(used);
    if (use_channel) { img->nb_meta_channels--; img_erase_channels(img, 0, 1); }
    return 1;
}

static int tr_apply_inverse(fo_image *img, fo_transform *t) {
    switch (t->id) {
        case 9: return inv_permute(img, t);
        case TR_YCBCR: return inv_ycbcr(img);
        case TR_SUBSAMPLE: return inv_subsample(img, t);
        case TR_DCT: return inv_dct(img, t);
        case TR_QUANTIZE: return inv_quantize(img);
        case TR_YCOCG: return inv_ycocg(img);
        case TR_SQUEEZE: return inv_squeeze(img, t);
        case TR_PALETTE: return inv_palette(img, t);
        case TR_APPROXIMATE: return inv_approximate(img, t);
        case TR_2DMATCH: return inv_match(img, t);
        default: return 0;
    }
}

/* image/image.cpp:94-115 */
int fo_undo_transforms(fo_image *img, int keep) {
    while (img->ntr > keep) {
        fo_transform *t = &img->tr[img->ntr - 1];
        if (!tr_apply_inverse(img, t)) { img->error = 1; return 0; }
        free(t->params); t->params = NULL;
        img->ntr--;
    }
    if (!keep) {
        for (int i = 0; i < img->nch; i++) {
            fo_channel *c = &img->ch[i];
            if (!c->data) continue; /* virtual zeros: clamp(0) = 0 since Image::minval = 0 */
            for (size_t j = 0; j < c->size; j++) c->data[j] = CLAMPI(c->data[j], img->minval, img->maxval);
        }
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------- */
void fo_image_info(fo_image *img, int32_t *out) {
    out[0] = img->w; out[1] = img->h; out[2] = img->minval; out[3] = img->maxval;
    out[4] = img->nb_channels; out[5] = img->real_nb_channels; out[6] = img->nb_meta_channels;
    out[7] = img->nch; out[8] = img->ntr; out[9] = img->error;
}
void fo_channel_info(fo_image *img, int c, int32_t *out) {
    const fo_channel *ch = &img->ch[c];
    out[0] = ch->w; out[1] = ch->h; out[2] = ch->minval; out[3] = ch->maxval; out[4] = ch->q;
    out[5] = ch->hshift; out[6] = ch->vshift; out[7] = ch->hcshift; out[8] = ch->vcshift;
    out[9] = ch->component; out[10] = ch->zero; out[11] = (int32_t)ch->size;
}
void fo_channel_data(fo_image *img, int c, int32_t *out) {
    const fo_channel *ch = &img->ch[c];
    if (ch->data) memcpy(out, ch->data, sizeof(int32_t) * ch->size);
    else memset(out, 0, sizeof(int32_t) * ch->size);
}
void fo_transform_info(fo_image *img, int t, int32_t *out, int cap) {
    const fo_transform *tr = &img->tr[t];
    out[0] = tr->id; out[1] = tr->nparams;
    for (int i = 0; i < tr->nparams && i + 2 < cap; i++) out[i + 2] = tr->params[i];
}
int fo_groups(fo_image *img, int32_t *first_channel, uint32_t *start, int cap) {
    for (int g = 0; g < img->ngroups && g < cap; g++) { first_channel[g] = img->group_channel[g]; start[g] = img->group_start[g]; }
    return img->ngroups;
}
int fo_max_tree_nodes(fo_image *img) { return img->stat_max_tree_nodes; }
void fo_stats(fo_image *img, uint64_t *out) {
    out[0] = img->stat_symbols; out[1] = img->stat_rac_decisions; out[2] = img->stat_tree_steps; out[3] = img->bytes_consumed;
}

/* ------------------------------------------------------------------------------------------- */
/* known-answer drivers                                                 */
int fo_kat_simple_symbols(const uint8_t *buf, size_t n, int count, int min, int max, int32_t *out, int *pos) {
    ensure_tables();
    fo_io io = {buf, n, 0, 1, 0};
    fo_rac rac; rac_init(&rac, &io);
    uint16_t ctx[CH_N]; fo_symbol_chance_init(ctx, 1024);
    for (int i = 0; i < count; i++) out[i] = read_symbol(&rac, ctx, g_table_tree, min, max);
    *pos = (int)io.pos;
    return 1;
}
int fo_kat_uniform_symbols(const uint8_t *buf, size_t n, int count, int min, int len, int32_t *out, int *pos) {
    fo_io io = {buf, n, 0, 1, 0};
    fo_rac rac; rac_init(&rac, &io);
    for (int i = 0; i < count; i++) out[i] = uniform_read_int(&rac, min, len);
    *pos = (int)io.pos;
    return 1;
}
int fo_kat_final_symbols(const uint8_t *buf, size_t n, int count, int zero_chance, int min, int max, int32_t *out, int *pos) {
    ensure_tables();
    fo_io io = {buf, n, 0, 1, 0};
    fo_rac rac; rac_init(&rac, &io);
    uint16_t leaf[CH_N]; fo_symbol_chance_init(leaf, zero_chance);
    for (int i = 0; i < count; i++) out[i] = read_symbol(&rac, leaf, g_table_pixel, min, max);
    *pos = (int)io.pos;
    return 1;
}
int fo_kat_read_bits(const uint8_t *buf, size_t n, int count, int32_t *out) {
    fo_io io = {buf, n, 0, 1, 0};
    fo_rac rac; rac_init(&rac, &io);
    for (int i = 0; i < count; i++) out[i] = rac_read_bit(&rac);
    return 1;
}

/* ------------------------------------------------------------------------------------------- */
/* single inverse transforms on raw planes: the checkers of the fuifgpu_inv_* / fuifgpu_idct8x8 / fuifgpu_upsample  */
/* entry points (tests/test_gpu_transform_exports.py).  Each builds the little Image the reference's               */
/* Transform::apply(image, true) (transform/transform.cpp:48-63) would be handed and runs the restatement above.   */
static void kat_plane(fo_channel *c, const int32_t *data, int w, int h, int minval, int maxval) {
    ch_init(c);
    c->w = w; c->h = h; c->minval = minval; c->maxval = maxval; ch_setzero(c);
    c->size = (size_t)w * h;
    c->data = (int32_t *)malloc(sizeof(int32_t) * (c->size ? c->size : 1));
    if (data) memcpy(c->data, data, sizeof(int32_t) * c->size);
}
static fo_image *kat_image(int nch, int maxval) {
    fo_image *img = (fo_image *)calloc(1, sizeof(fo_image));
    img->ch = (fo_channel *)calloc(nch, sizeof(fo_channel));
    img->nch = nch; img->nb_channels = nch; img->real_nb_channels = nch; img->maxval = maxval; img->minval = 0; img->nb_frames = 1;
    return img;
}
/* transform/squeeze.h:81-132 (horizontal) / :173-224 (vertical): out is (aw+rw) x ah resp. aw x (ah+rh) */
int fo_kat_inv_squeeze(int horizontal, const int32_t *avg, int aw, int ah, const int32_t *res, int rw, int rh, int32_t *out) {
    fo_image *img = kat_image(2, 255);
    kat_plane(&img->ch[0], avg, aw, ah, -32768, 32767);
    kat_plane(&img->ch[1], res, rw, rh, -32768, 32767);
    img->ch[0].hshift = img->ch[0].vshift = img->ch[0].hcshift = img->ch[0].vcshift = 1;
    if (horizontal) inv_hsqueeze(img, 0, 1); else inv_vsqueeze(img, 0, 1);
    memcpy(out, img->ch[0].data, sizeof(int32_t) * (size_t)img->ch[0].w * img->ch[0].h);
    fo_free(img);
    return 1;
}
/* transform/ycocg.h:33-63 / transform/ycbcr.h:33-63, in place on three w x h planes */
int fo_kat_inv_color(int ycbcr, int32_t *c0, int32_t *c1, int32_t *c2, int w, int h, int minval, int maxval) {
    fo_image *img = kat_image(3, maxval);
    img->minval = minval;
    int32_t *p[3] = {c0, c1, c2};
    for (int k = 0; k < 3; k++) kat_plane(&img->ch[k], p[k], w, h, -32768, 32767);
    int ok = ycbcr ? inv_ycbcr(img) : inv_ycocg(img);
    for (int k = 0; k < 3; k++) memcpy(p[k], img->ch[k].data, sizeof(int32_t) * (size_t)w * h);
    fo_free(img);
    return ok;
}
/* transform/dct.h:249-296 for one component: planes64 = the 64 coefficient planes in CHANNEL order (DC, then the AC
 * planes as meta_dct lays them out: plane k holds zig-zag rank k), each bw x bh; out = 8bw x 8bh */
int fo_kat_inv_dct(const int32_t *planes64, int bw, int bh, int maxval, int32_t *out) {
    fo_image *img = kat_image(64, maxval);
    for (int k = 0; k < 64; k++) kat_plane(&img->ch[k], planes64 + (size_t)k * bw * bh, bw, bh, -32768, 32767);
    img->nb_channels = 1;
    fo_transform t; t.id = 4; t.nparams = 0; t.params = NULL;
    int ok = inv_dct(img, &t);
    if (ok) memcpy(out, img->ch[0].data, sizeof(int32_t) * (size_t)bw * 8 * bh * 8);
    free(t.params);
    fo_free(img);
    return ok;
}
/* transform/2dmatch.h:123-177 on raw planes: match w x h (its Channel::q and ::maxval given), planes = n_planes x (w x h) rewritten in place;
 * returns 0 where the reference returns false */
int fo_kat_inv_match(const int32_t *match, int w, int h, int32_t *planes, int n_planes, int softmatch, int q, int maxval, int nb_frames) {
    fo_image *img = kat_image(1 + n_planes, 255);
    img->nb_meta_channels = 1; img->nb_channels = n_planes; img->nb_frames = nb_frames;
    kat_plane(&img->ch[0], match, w, h, 0, maxval);
    img->ch[0].q = q;
    for (int k = 0; k < n_planes; k++) kat_plane(&img->ch[1 + k], planes + (size_t)k * w * h, w, h, -32768, 32767);
    fo_transform t; t.id = TR_2DMATCH; t.nparams = 4;
    t.params = (int *)malloc(sizeof(int) * 4);
    t.params[0] = 0; t.params[1] = n_planes - 1; t.params[2] = softmatch; t.params[3] = 1000000;
    int ok = inv_match(img, &t);
    if (ok) for (int k = 0; k < n_planes; k++) memcpy(planes + (size_t)k * w * h, img->ch[k].data, sizeof(int32_t) * (size_t)w * h);
    free(t.params);
    fo_free(img);
    return ok;
}
void fo_kat_zigzag(int32_t *out64) { for (int i = 0; i < 64; i++) out64[i] = fo_zigzag[i]; }
/* transform/subsample.h:73-127 for one plane that is smaller than the first (luma) plane; srh, srv in {1,2} */
int fo_kat_upsample(const int32_t *in, int w, int h, int srh, int srv, int32_t *out) {
    fo_image *img = kat_image(2, 255);
    kat_plane(&img->ch[0], NULL, w * srh, h * srv, 0, 255);
    memset(img->ch[0].data, 0, sizeof(int32_t) * img->ch[0].size);
    kat_plane(&img->ch[1], in, w, h, -32768, 32767);
    fo_transform t; t.id = 3; t.nparams = 4;
    t.params = (int *)malloc(sizeof(int) * 4);
    t.params[0] = 1; t.params[1] = 1; t.params[2] = srh; t.params[3] = srv;
    int ok = inv_subsample(img, &t);
    if (ok) memcpy(out, img->ch[1].data, sizeof(int32_t) * (size_t)w * srh * h * srv);
    free(t.params);
    fo_free(img);
    return ok;
}
