// C ABI, host-only part (no HIP): host arithmetic, PEX trees, index lifetime on the host, the index's meta block.
// (Compiled into libfloxer_amd.so and, with the other HIP-free sources, into the sanitizer build of tests/sanitize.)
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "flx_fm_core.hpp"

namespace flx {
const char* last_error_cstr();
}
using namespace flx;

// an index made by flx_index_meta_import holds sizes and sequence bounds only (its arrays live in another rank's HBM image)
#define FLX_NEED_ARRAYS(index, who)                                                                                                   \
    do {                                                                                                                              \
        if (!index_has_arrays(*(index)->host)) { set_error(std::string(who) + ": this index holds no arrays (flx_index_meta_import)"); return FLX_ERR_INVALID; } \
    } while (0)

extern "C" {

const char* flx_last_error(void) { return last_error_cstr(); }
const char* flx_version(void) { return "floxer_amd 0.1.0 (gfx950)"; }

uint64_t flx_ceil_div(uint64_t a, uint64_t b) { return ceil_div(a, b); }
uint64_t flx_floating_point_error_aware_ceil(double value) { return fp_aware_ceil(value); }
int32_t flx_saturate_value_to_int32_max(uint64_t value) { return saturate_i32(value); }
void flx_chars_to_rank_sequence(const char* chars, uint64_t n, uint8_t* out) {
    // (a table: the CLI's reader encodes 160 MB of bases per batch, and a call into a switch per character was half of its time)
    static const struct Table { uint8_t rank[256]; Table() { for (int c = 0; c < 256; ++c) rank[c] = char_to_rank((char)c); } } table;
    for (uint64_t i = 0; i < n; ++i) out[i] = table.rank[(uint8_t)chars[i]];
}
void flx_reverse_complement_rank(const uint8_t* ranks, uint64_t n, uint8_t* out) { reverse_complement(ranks, n, out); }

int flx_pex_tree_build(uint64_t query_length, uint64_t query_num_errors, uint64_t leaf_max_num_errors, int bottom_up,
                       flx_pex_node* nodes, uint64_t capacity, uint64_t* n_inner, uint64_t* n_leaves) {
    if (!n_inner || !n_leaves) { set_error("flx_pex_tree_build: null argument"); return FLX_ERR_INVALID; }
    if (query_length == 0 || query_length > SCH_POS_MASK || query_num_errors >= query_length) { set_error("flx_pex_tree_build: invalid length / errors"); return FLX_ERR_INVALID; }
    PexTree const t = build_pex_tree(query_length, query_num_errors, leaf_max_num_errors, bottom_up != 0);
    *n_inner = t.inner.size();
    *n_leaves = t.leaves.size();
    if (t.inner.size() + t.leaves.size() > capacity || !nodes) { set_error("node buffer too small"); return FLX_ERR_CAPACITY; }
    if (!t.inner.empty()) memcpy(nodes, t.inner.data(), t.inner.size() * sizeof(flx_pex_node));
    memcpy(nodes + t.inner.size(), t.leaves.data(), t.leaves.size() * sizeof(flx_pex_node));
    return FLX_OK;
}

// ---------------------------------------------------------------- index
int flx_index_build(const uint8_t* concat, const uint64_t* lens, uint32_t n_refs, flx_index** out) {
    if (!concat || !lens || !out || n_refs == 0) { set_error("flx_index_build: null argument or no reference"); return FLX_ERR_INVALID; }
    HostIndex* h = build_host_index(concat, lens, n_refs);
    if (!h) return FLX_ERR_INVALID;
    *out = new flx_index{h};
    return FLX_OK;
}
int flx_index_save(const flx_index* index, const char* path) {
    if (!index || !path) { set_error("flx_index_save: null argument"); return FLX_ERR_INVALID; }
    FLX_NEED_ARRAYS(index, "flx_index_save");
    return save_host_index(*index->host, path);
}
int flx_index_load(const char* path, flx_index** out) {
    if (!path || !out) { set_error("flx_index_load: null argument"); return FLX_ERR_INVALID; }
    HostIndex* h = load_host_index(path);
    if (!h) return FLX_ERR_IO;
    *out = new flx_index{h};
    return FLX_OK;
}
void flx_index_free(flx_index* index) {
    if (!index) return;
    delete index->host;
    delete index;
}
uint64_t flx_index_text_length(const flx_index* index) { return index ? index->host->n : 0; }
uint32_t flx_index_num_references(const flx_index* index) { return index ? (uint32_t)index->host->seq_len.size() : 0; }
uint64_t flx_index_device_bytes(const flx_index* index) {
    if (!index) return 0;
    HostIndex const& h = *index->host;
    // (from the sizes, so that an index without arrays reports what its image takes as well)
    u64 const nb = h.n / OCC_BLOCK_POS + 1;
    return 2 * nb * sizeof(OccBlock) + h.n * 4 + h.n + 2 * TEXT_PAD + (((u64)1 << (2 * KMER_Q)) * 3) * 4;
}
uint64_t flx_index_derived_device_bytes(const flx_index* index) {
    if (!index || index->host->n == 0) return 0;
    u64 const n = index->host->n;
    u32 const k = filter_k_default(n);
    return n * 4 + filter_words(k) * 8;
}
int flx_index_matches_reference(const flx_index* index, const uint8_t* concat, const uint64_t* lens, uint32_t n_refs) {
    if (!index || !concat || !lens) { set_error("flx_index_matches_reference: null argument"); return FLX_ERR_INVALID; }
    FLX_NEED_ARRAYS(index, "flx_index_matches_reference");
    HostIndex const& h = *index->host;
    if (h.seq_len.size() != n_refs) { set_error("the index holds " + std::to_string(h.seq_len.size()) + " sequences, the reference " + std::to_string(n_refs)); return FLX_ERR_INVALID; }
    uint64_t off = 0;
    for (uint32_t r = 0; r < n_refs; ++r) {
        if (h.seq_len[r] != lens[r]) { set_error("sequence " + std::to_string(r) + " has another length in the index"); return FLX_ERR_INVALID; }
        if (lens[r] && memcmp(h.text.data() + h.seq_start[r], concat + off, lens[r]) != 0) { set_error("sequence " + std::to_string(r) + " differs from the indexed text"); return FLX_ERR_INVALID; }
        off += lens[r];
    }
    return FLX_OK;
}
int flx_index_copy_sa(const flx_index* index, uint64_t* out) {
    if (!index || !out) { set_error("null argument"); return FLX_ERR_INVALID; }
    FLX_NEED_ARRAYS(index, "flx_index_copy_sa");
    for (size_t i = 0; i < index->host->sa.size(); ++i) out[i] = index->host->sa[i];
    return FLX_OK;
}
int flx_index_copy_sa_u32(const flx_index* index, uint32_t* out) {
    if (!index || !out) { set_error("null argument"); return FLX_ERR_INVALID; }
    FLX_NEED_ARRAYS(index, "flx_index_copy_sa_u32");
    memcpy(out, index->host->sa.data(), index->host->sa.size() * 4);
    return FLX_OK;
}
int flx_index_copy_bwt(const flx_index* index, int reversed, uint8_t* out) {
    if (!index || !out) { set_error("null argument"); return FLX_ERR_INVALID; }
    FLX_NEED_ARRAYS(index, "flx_index_copy_bwt");
    auto const& b = index->host->bwt[reversed ? 1 : 0];
    if (b.size() != index->host->n) { set_error("flx_index_copy_bwt: this index was loaded without its BWTs"); return FLX_ERR_INVALID; }
    memcpy(out, b.data(), b.size());
    return FLX_OK;
}

// meta block: magic, n, C[7], n_refs, seq_start[], seq_len[]
int flx_index_meta_export(const flx_index* index, uint8_t* buf, uint64_t* len) {
    if (!index || !len) { set_error("flx_index_meta_export: null argument"); return FLX_ERR_INVALID; }
    HostIndex const& H = *index->host;
    u64 const n_refs = H.seq_len.size(), need = 8 * (1 + 1 + 7 + 1 + 2 * n_refs);
    u64 const cap = *len;
    *len = need;
    if (!buf || cap < need) { set_error("meta buffer too small"); return FLX_ERR_CAPACITY; }
    u64* w = reinterpret_cast<u64*>(buf);
    *w++ = 0x314154454D584C46ull;                 // "FLXMETA1"
    *w++ = H.n;
    for (int c = 0; c < 7; ++c) *w++ = H.C[c];
    *w++ = n_refs;
    for (u64 r = 0; r < n_refs; ++r) *w++ = H.seq_start[r];
    for (u64 r = 0; r < n_refs; ++r) *w++ = H.seq_len[r];
    return FLX_OK;
}
int flx_index_meta_import(const uint8_t* buf, uint64_t len, flx_index** out) {
    if (!buf || !out || len < 80) { set_error("flx_index_meta_import: null argument or short buffer"); return FLX_ERR_INVALID; }
    const u64* w = reinterpret_cast<const u64*>(buf);
    if (w[0] != 0x314154454D584C46ull) { set_error("not an index meta block"); return FLX_ERR_INVALID; }
    auto h = std::make_unique<HostIndex>();
    h->n = w[1];
    for (int c = 0; c < 7; ++c) h->C[c] = w[2 + c];
    u64 const n_refs = w[9];
    if (h->n == 0 || h->n >= ((u64)1 << 32) || n_refs == 0 || n_refs > (len / 8 - 10) / 2) { set_error("index meta block is corrupt"); return FLX_ERR_INVALID; }
    h->seq_start.assign(w + 10, w + 10 + n_refs);
    h->seq_len.assign(w + 10 + n_refs, w + 10 + 2 * n_refs);
    *out = new flx_index{h.release()};
    return FLX_OK;
}


}  // extern "C"
