// fuif_amd/csrc/index.cpp -- the group index ("FGIX" trailer), SURVEY.md §8(f) rank 1.
//
// A FUIF stream is a chain of channel groups, each with its own range coder, whose boundaries are
// byte aligned but implicit (a group ends where its coder stopped reading; the encoder knows the
// positions, encoding/encoding.cpp:525-527,542, but does not store them).  The index stores them:
// with it every group of an image can be handed to its own wavefront.
//
//   <FUIF stream exactly as the reference writes it> <payload> <u32 LE payload length> "FGIX"
//   payload = varint 1 (version) ; varint n ; n x { varint channel delta ; varint byte-offset delta }
//
// (same big-endian base-128 varints as the stream itself, encoding.cpp:32-59; deltas are against
// the previous entry, the first against channel 0 / byte 0.)  The reference decoder never reads past
// the last group (encoding.cpp:708-717), so an indexed file decodes unchanged with the unmodified
// reference CLI; a file without the trailer is decoded one wavefront per image as before.
#include <cstdlib>
#include <cstring>

#include "../../include/fuifgpu.h"
#include "fuifgpu_internal.h"

namespace fuifgpu {

namespace {
void put_varint(std::vector<uint8_t> &b, uint64_t v) {
    uint8_t tmp[10];
    int n = 0;
    tmp[n++] = (uint8_t)(v & 127);
    v >>= 7;
    while (v) { tmp[n++] = (uint8_t)(128 | (v & 127)); v >>= 7; }
    while (n) b.push_back(tmp[--n]);
}
bool get_varint(const uint8_t *p, size_t n, size_t &pos, uint64_t &out) {
    uint64_t r = 0;
    for (int k = 0; k < 10; k++) {
        if (pos >= n) return false;
        const uint8_t c = p[pos++];
        if (r >> 57) return false;   // ten groups carry 70 bits: the value must fit 64
        r = (r << 7) | (c & 127);
        if (c < 128) { out = r; return true; }
    }
    return false;
}
}  // namespace

void build_index_trailer(const std::vector<GroupEntry> &groups, std::vector<uint8_t> &out) {
    std::vector<uint8_t> payload;
    put_varint(payload, 1);
    put_varint(payload, groups.size());
    uint32_t pc = 0, ps = 0;
    for (const GroupEntry &g : groups) {
        put_varint(payload, (uint32_t)g.first_channel - pc);
        put_varint(payload, g.start - ps);
        pc = (uint32_t)g.first_channel; ps = g.start;
    }
    out = payload;
    const uint32_t len = (uint32_t)payload.size();
    for (int k = 0; k < 4; k++) out.push_back((uint8_t)(len >> (8 * k)));
    out.push_back('F'); out.push_back('G'); out.push_back('I'); out.push_back('X');
}

// Returns true and fills `groups` only for a trailer that is consistent with the stream it sits
// behind: first group at `data_start`, channels and offsets strictly ascending, every offset inside
// the stream part.  *stream_end = first byte of the trailer.
bool parse_index_trailer(const uint8_t *blob, size_t n, size_t data_start, int nch, std::vector<GroupEntry> &groups, size_t *stream_end) {
    groups.clear();
    if (n < 8 + 3 || memcmp(blob + n - 4, "FGIX", 4) != 0) return false;
    uint32_t len = 0;
    for (int k = 0; k < 4; k++) len |= (uint32_t)blob[n - 8 + k] << (8 * k);
    if (len < 2 || (size_t)len + 8 > n) return false;
    const size_t begin = n - 8 - len;
    const uint8_t *p = blob + begin;
    size_t pos = 0;
    uint64_t version = 0, count = 0;
    if (!get_varint(p, len, pos, version) || version != 1) return false;
    if (!get_varint(p, len, pos, count) || count < 1 || count > (uint64_t)nch) return false;
    uint64_t c = 0, s = 0;
    for (uint64_t g = 0; g < count; g++) {
        uint64_t dc = 0, ds = 0;
        if (!get_varint(p, len, pos, dc) || !get_varint(p, len, pos, ds)) return false;
        if (g && (dc == 0 || ds == 0)) return false;              // strictly ascending channels and offsets
        if (dc >= (uint64_t)nch || ds >= begin) return false;     // bounded before the sums: a 70-bit varint cannot wrap them
        c += dc; s += ds;
        if (c >= (uint64_t)nch || s >= begin) return false;
        groups.push_back(GroupEntry{(uint32_t)s, (int32_t)c});
    }
    if (pos != len || groups[0].start != data_start) { groups.clear(); return false; }
    if (stream_end) *stream_end = begin;
    return true;
}

}  // namespace fuifgpu

using namespace fuifgpu;

extern "C" {

int fuifgpu_index_parse(const uint8_t *blob, size_t size, int32_t *first_channel, uint32_t *start, int cap, int *n_groups) {
    if (!blob || !n_groups) return FUIFGPU_E_ARG;
    *n_groups = 0;
    Plan plan;
    int r = parse_and_plan(blob, size, plan);
    if (r != FUIFGPU_OK) return r;
    std::vector<GroupEntry> groups;
    if (!parse_index_trailer(blob, size, plan.data_start, (int)plan.coded.size(), groups, nullptr)) return FUIFGPU_OK;
    *n_groups = (int)groups.size();
    for (int g = 0; g < (int)groups.size() && g < cap; g++) {
        if (first_channel) first_channel[g] = groups[g].first_channel;
        if (start) start[g] = groups[g].start;
    }
    return FUIFGPU_OK;
}

int fuifgpu_index_append(const uint8_t *blob, size_t size, const int32_t *first_channel, const uint32_t *start, int n_groups,
                         uint8_t **blob_out, size_t *size_out) {
    if (!blob || !first_channel || !start || n_groups < 1 || !blob_out || !size_out) return FUIFGPU_E_ARG;
    Plan plan;
    int r = parse_and_plan(blob, size, plan);
    if (r != FUIFGPU_OK) return r;
    std::vector<GroupEntry> old;
    size_t stream_end = size;
    parse_index_trailer(blob, size, plan.data_start, (int)plan.coded.size(), old, &stream_end);  // an existing trailer is replaced
    std::vector<GroupEntry> groups(n_groups);
    for (int g = 0; g < n_groups; g++) {
        groups[g] = GroupEntry{start[g], first_channel[g]};
        const bool ascending = g == 0 || (first_channel[g] > first_channel[g - 1] && start[g] > start[g - 1]);
        if (!ascending || first_channel[g] < 0 || first_channel[g] >= (int)plan.coded.size() || start[g] >= stream_end) return FUIFGPU_E_ARG;
    }
    if (groups[0].start != plan.data_start) return FUIFGPU_E_ARG;
    std::vector<uint8_t> trailer;
    build_index_trailer(groups, trailer);
    uint8_t *out = (uint8_t *)malloc(stream_end + trailer.size());
    if (!out) return FUIFGPU_E_NOMEM;
    memcpy(out, blob, stream_end);
    memcpy(out + stream_end, trailer.data(), trailer.size());
    *blob_out = out;
    *size_out = stream_end + trailer.size();
    return FUIFGPU_OK;
}

}  // extern "C"
