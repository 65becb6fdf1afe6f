/* oracle/fuif_oracle.h -- TEST INFRASTRUCTURE ONLY: CPU restatement of the FUIF decode path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 * The product (fuif_amd/, libfuifgpu.so) never links, imports or calls it.
 */
#ifndef FUIF_ORACLE_H
#define FUIF_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference: image/image.h:54-91 (Channel), planes widened to int32 (image/image.h:39-45 variant) */
typedef struct {
    int32_t *data;
    size_t size; /* number of samples held (0 = never decoded), mirrors data.size() */
    int w, h;
    int minval, maxval, zero;
    int q;
    int hshift, vshift, hcshift, vcshift;
    int component;
} fo_channel;

/* reference: transform/transform.h:77-106 */
typedef struct {
    int id;
    int nparams;
    int *params;
} fo_transform;

/* reference: image/image.h:98-129 */
typedef struct {
    fo_channel *ch;
    int nch;
    fo_transform *tr;
    int ntr;
    int w, h, minval, maxval;
    int nb_channels, real_nb_channels, nb_meta_channels, colormodel;
    int nb_frames;
    int max_properties;
    int responsive_offsets[5];
    int error;
    /* statistics (not in the reference): */
    uint64_t stat_symbols, stat_rac_decisions, stat_tree_steps;
    int stat_max_tree_nodes;            /* largest MANIAC tree of any channel group (nodes) */
    size_t bytes_consumed;
    /* byte offset / first channel of every channel group, in stream order (test aid for the group index) */
    uint32_t *group_start; int32_t *group_channel; int ngroups, groups_cap;
} fo_image;

/* io_kind 0: FileIO semantics (feof only after a failed read; what the CLI uses, fileio.h:55-63)
 * io_kind 1: BlobReader semantics (EOF as soon as the cursor reaches the end, fileio.h:100-102) */
fo_image *fo_decode(const uint8_t *blob, size_t n, int preview, int io_kind, int *ok);
int fo_undo_transforms(fo_image *img, int keep);
void fo_free(fo_image *img);

void fo_image_info(fo_image *img, int32_t *out10);
void fo_channel_info(fo_image *img, int c, int32_t *out12);
void fo_channel_data(fo_image *img, int c, int32_t *out);
void fo_transform_info(fo_image *img, int t, int32_t *out, int cap);
void fo_stats(fo_image *img, uint64_t *out4);
int fo_max_tree_nodes(fo_image *img);
int fo_groups(fo_image *img, int32_t *first_channel, uint32_t *start, int cap);

/* known-answer helpers for unit tests */
void fo_build_table(uint16_t *table8192, uint32_t alpha, int cut);
void fo_symbol_chance_init(uint16_t *ch31, int zero_chance);
int fo_smooth_tendency(int B, int a, int n);
void fo_idct8x8(double *block64);
int fo_kat_simple_symbols(const uint8_t *buf, size_t n, int count, int min, int max, int32_t *out, int *pos);
int fo_kat_uniform_symbols(const uint8_t *buf, size_t n, int count, int min, int len, int32_t *out, int *pos);
int fo_kat_final_symbols(const uint8_t *buf, size_t n, int count, int zero_chance, int min, int max, int32_t *out, int *pos);
int fo_kat_read_bits(const uint8_t *buf, size_t n, int count, int32_t *out);
/* single inverse transforms on raw planes (checkers of the fuifgpu_inv_* / fuifgpu_idct8x8 / fuifgpu_upsample entry points) */
int fo_kat_inv_squeeze(int horizontal, const int32_t *avg, int aw, int ah, const int32_t *res, int rw, int rh, int32_t *out);
int fo_kat_inv_color(int ycbcr, int32_t *c0, int32_t *c1, int32_t *c2, int w, int h, int minval, int maxval);
int fo_kat_inv_dct(const int32_t *planes64, int bw, int bh, int maxval, int32_t *out);
int fo_kat_inv_match(const int32_t *match, int w, int h, int32_t *planes, int n_planes, int softmatch, int q, int maxval, int nb_frames);
void fo_kat_zigzag(int32_t *out64);
int fo_kat_upsample(const int32_t *in, int w, int h, int srh, int srv, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif
