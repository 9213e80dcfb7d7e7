// ORACLE — TEST INFRASTRUCTURE ONLY. Flat C entry points over floxer_oracle.cpp so that tests/ and bench.py's
// cpu_baseline leg can drive the CPU restatement through ctypes. Never linked into the product.
#include "floxer_oracle.hpp"

#include <chrono>
#include <cstring>
#include <memory>

using namespace orc;

namespace {
struct index_handle {
    std::vector<std::vector<uint8_t>> refs;
    fm_index idx;
};
struct run_handle { run_output out; double seconds = 0; };

params make_params(const double* pv) {
    // pv layout (doubles, so one array carries everything):
    // 0 error_probability (<0 -> unset) 1 query_num_errors 2 seed_errors 3 hard 4 soft 5 group_order 6 choice 7 erase
    // 8 seed_step 9 bottom_up 10 interval_opt 11 extra_ratio 12 direct_full 13 anchors_per_task 14 without_cigar 15 align_algo
    params p;
    p.error_probability = pv[0];
    p.query_num_errors = (uint64_t)pv[1];
    p.seed_errors = (uint64_t)pv[2];
    p.search.max_num_anchors_hard = (uint64_t)pv[3];
    p.search.max_num_anchors_soft = (uint64_t)pv[4];
    p.search.anchor_group_order = (int)pv[5];
    p.search.anchor_choice_strategy = (int)pv[6];
    p.search.erase_useless_anchors = pv[7] != 0;
    p.seed_sampling_step = (uint64_t)pv[8];
    p.bottom_up = pv[9] != 0;
    p.interval_optimization = pv[10] != 0;
    p.extra_verification_ratio = pv[11];
    p.direct_full = pv[12] != 0;
    p.anchors_per_task = (uint64_t)pv[13];
    p.without_cigar = pv[14] != 0;
    p.align_algo = (int)pv[15];
    return p;
}
}  // namespace

extern "C" {

// ---- math
uint64_t orc_ceil_div(uint64_t a, uint64_t b) { return ceil_div(a, b); }
uint64_t orc_fp_ceil(double v) { return floating_point_error_aware_ceil(v); }
int32_t orc_saturate_i32(uint64_t v) { return saturate_value_to_int32_max(v); }

// ---- input
void orc_chars_to_ranks(const char* s, uint64_t n, uint8_t* out) {
    auto v = chars_to_rank_sequence(s, n);
    memcpy(out, v.data(), n);
}
void orc_revcomp(const uint8_t* s, uint64_t n, uint8_t* out) {
    auto v = reverse_complement_rank(std::vector<uint8_t>(s, s + n));
    memcpy(out, v.data(), n);
}

// ---- pex: nodes written as rows {parent_id, from, to, errors}; inner nodes first, then leaves
int orc_pex_build(uint64_t len, uint64_t k, uint64_t s, int bottom_up, uint64_t* nodes, uint64_t cap, uint64_t* n_inner,
                  uint64_t* n_leaves) {
    pex_tree t(len, k, s, bottom_up != 0);
    *n_inner = t.inner_nodes.size();
    *n_leaves = t.leaves.size();
    if (t.inner_nodes.size() + t.leaves.size() > cap) return 1;
    uint64_t r = 0;
    for (auto const* vec : {&t.inner_nodes, &t.leaves})
        for (auto const& nd : *vec) {
            nodes[r * 4 + 0] = nd.parent_id; nodes[r * 4 + 1] = nd.from; nodes[r * 4 + 2] = nd.to; nodes[r * 4 + 3] = nd.num_errors;
            ++r;
        }
    return 0;
}

// ---- index
void* orc_index_build(const uint8_t* pool, const uint64_t* lens, uint32_t n_refs, uint32_t sampling) {
    auto h = std::make_unique<index_handle>();
    uint64_t off = 0;
    for (uint32_t i = 0; i < n_refs; ++i) { h->refs.emplace_back(pool + off, pool + off + lens[i]); off += lens[i]; }
    try { h->idx = build_index(h->refs, sampling); } catch (...) { return nullptr; }
    return h.release();
}
// the same with the suffix array (u32) and the two BWTs given as data (see import_index)
void* orc_index_import(const uint8_t* pool, const uint64_t* lens, uint32_t n_refs, uint32_t sampling, const uint32_t* sa,
                       const uint8_t* bwt, const uint8_t* bwt_rev) {
    auto h = std::make_unique<index_handle>();
    uint64_t off = 0;
    for (uint32_t i = 0; i < n_refs; ++i) { h->refs.emplace_back(pool + off, pool + off + lens[i]); off += lens[i]; }
    try { h->idx = import_index(h->refs, sampling, sa, bwt, bwt_rev); } catch (...) { return nullptr; }
    return h.release();
}
void orc_index_free(void* h) { delete (index_handle*)h; }
uint64_t orc_index_size(void* h) { return ((index_handle*)h)->idx.n; }
void orc_index_sa(void* h, int64_t* out) { auto& i = ((index_handle*)h)->idx; for (uint64_t r = 0; r < i.n; ++r) out[r] = (int64_t)i.sa[r]; }
void orc_index_bwt(void* h, int rev, uint8_t* out) {
    auto& i = ((index_handle*)h)->idx;
    memcpy(out, (rev ? i.bwt_rev : i.bwt).data(), i.n);
}
void orc_index_locate(void* h, uint64_t row, uint64_t* seq_id, uint64_t* pos) { ((index_handle*)h)->idx.locate(row, *seq_id, *pos); }

// ---- raw search_n emission for one seed: rows {lb, len, errors}
int64_t orc_search_groups(void* h, const uint8_t* seq, uint64_t len, uint32_t k, uint64_t n, uint64_t* out, uint64_t cap,
                          uint64_t* counters) {
    std::vector<anchor_group> g;
    search_counters c;
    search_n(((index_handle*)h)->idx, seq, len, k, n, g, &c);
    if (counters) { counters[0] = c.n_extend_all; counters[1] = c.n_extend_one; }
    if (g.size() > cap) return -(int64_t)g.size();
    for (size_t i = 0; i < g.size(); ++i) { out[i * 3] = g[i].cur.lb; out[i * 3 + 1] = g[i].cur.len; out[i * 3 + 2] = g[i].num_errors; }
    return (int64_t)g.size();
}

// ---- search_seeds on explicit seeds: seeds rows {offset into pool, len, errors, leaf_index}; cfg {hard, soft, order, choice, erase}
// anchors rows {seed_row, leaf, ref_id, pos, errors} in anchor_iterator order; stats rows {useful, raw, excluded_soft, fully_excluded}
int64_t orc_search_seeds(void* h, const uint8_t* pool, const uint64_t* seeds, uint64_t n_seeds, const uint64_t* cfg,
                         uint64_t* anchors, uint64_t cap, uint64_t* stats) {
    std::vector<seed_t> sv;
    for (uint64_t i = 0; i < n_seeds; ++i)
        sv.push_back(seed_t{pool + seeds[i * 4], seeds[i * 4 + 1], seeds[i * 4 + 2], seeds[i * 4], seeds[i * 4 + 3]});
    search_config c;
    c.max_num_anchors_hard = cfg[0]; c.max_num_anchors_soft = cfg[1]; c.anchor_group_order = (int)cfg[2];
    c.anchor_choice_strategy = (int)cfg[3]; c.erase_useless_anchors = cfg[4] != 0;
    auto res = search_seeds(((index_handle*)h)->idx, sv, c, nullptr);
    uint64_t n = 0;
    for (uint64_t i = 0; i < n_seeds; ++i) {
        if (stats) {
            stats[i * 4] = res[i].num_kept_useful_anchors; stats[i * 4 + 1] = res[i].num_kept_raw_anchors;
            stats[i * 4 + 2] = res[i].num_excluded_raw_anchors_by_soft_cap; stats[i * 4 + 3] = res[i].anchors_by_reference.empty();
        }
        for (auto const& by_ref : res[i].anchors_by_reference)
            for (auto const& a : by_ref) {
                if (n < cap) {
                    anchors[n * 5] = i; anchors[n * 5 + 1] = a.pex_leaf_index; anchors[n * 5 + 2] = a.reference_id;
                    anchors[n * 5 + 3] = a.reference_position; anchors[n * 5 + 4] = a.num_errors;
                }
                ++n;
            }
    }
    return (int64_t)n;
}

// erase_useless_anchors on one reference's anchors: rows {pos, errors}; returns kept count, rows rewritten in place
uint64_t orc_erase_useless(uint64_t* rows, uint64_t n) {
    std::vector<std::vector<anchor_t>> v(1);
    for (uint64_t i = 0; i < n; ++i) v[0].push_back(anchor_t{0, 0, rows[i * 2], rows[i * 2 + 1]});
    uint64_t kept = erase_useless_anchors(v);
    for (uint64_t i = 0; i < v[0].size(); ++i) { rows[i * 2] = v[0][i].reference_position; rows[i * 2 + 1] = v[0][i].num_errors; }
    return kept;
}

// ---- alignment::align; returns 1 if an alignment exists. cigar_len in/out (capacity in, length out)
int orc_align(const uint8_t* ref, uint64_t n, const uint8_t* query, uint64_t m, uint64_t k, int mode, int algo, uint64_t* nm,
              uint64_t* begin, uint32_t* cigar, uint64_t* cigar_len) {
    align_result r = align(ref, n, query, m, k, mode, algo, nullptr);
    if (!r.exists) { if (cigar_len) *cigar_len = 0; return 0; }
    *nm = r.num_errors;
    *begin = r.begin;
    if (cigar_len) {
        uint64_t const cap = *cigar_len;
        *cigar_len = r.cigar.size();
        if (cigar && r.cigar.size() <= cap) memcpy(cigar, r.cigar.data(), r.cigar.size() * 4);
    }
    return 1;
}

// ---- intervals
int orc_relationship(uint64_t s1, uint64_t e1, uint64_t s2, uint64_t e2) { return relationship_with({s1, e1}, {s2, e2}); }
void orc_trim(uint64_t s, uint64_t e, uint64_t amount, uint64_t* os, uint64_t* oe) {
    auto t = trim_from_both_sides({s, e}, amount);
    *os = t.start; *oe = t.end;
}
void* orc_intervals_new(int active) { auto* v = new verified_intervals(); v->active = active != 0; return v; }
void orc_intervals_free(void* v) { delete (verified_intervals*)v; }
void orc_intervals_insert(void* v, uint64_t s, uint64_t e) { ((verified_intervals*)v)->insert({s, e}); }
int orc_intervals_contains(void* v, uint64_t s, uint64_t e) { return ((verified_intervals*)v)->contains({s, e}); }
uint64_t orc_intervals_count(void* v) { return ((verified_intervals*)v)->ivs.size(); }

// ---- verification
void orc_span(uint64_t anchor_pos, uint64_t node_from, uint64_t node_to, uint64_t node_errors, uint64_t leaf_from,
              uint64_t reflen, double ratio, uint64_t* out3) {
    pex_node nd{0, node_from, node_to, node_errors};
    auto sc = compute_reference_span_start_and_length(anchor_pos, nd, leaf_from, reflen, ratio);
    out3[0] = sc.offset; out3[1] = sc.length; out3[2] = sc.extra;
}

// query_verifier::verify for one anchor on an explicit tree (len,k,s,bottom_up); appends alignments rows {start, nm, reverse,
// cigar_off, cigar_len}; intervals handle persists across calls (may be null -> fresh deactivated cache)
int64_t orc_verify_anchor(uint64_t qlen, uint64_t k, uint64_t s, int bottom_up, uint64_t leaf_index, uint64_t anchor_pos,
                          uint64_t anchor_errors, const uint8_t* query, int reverse, const uint8_t* reference, uint64_t reflen,
                          const double* pv, void* intervals, uint64_t* rows, uint64_t cap, uint32_t* cigars, uint64_t cigar_cap) {
    params p = make_params(pv);
    pex_tree tree(qlen, k, s, bottom_up != 0);
    anchor_t a{leaf_index, 0, anchor_pos, anchor_errors};
    verified_intervals local; local.active = false;
    verified_intervals& ivs = intervals ? *(verified_intervals*)intervals : local;
    std::vector<query_alignment> out;
    verify_anchor(tree, a, query, reverse != 0, reference, reflen, p, ivs, out, nullptr);
    uint64_t coff = 0;
    for (size_t i = 0; i < out.size() && i < cap; ++i) {
        rows[i * 5] = out[i].start_in_reference; rows[i * 5 + 1] = out[i].num_errors; rows[i * 5 + 2] = out[i].reverse;
        rows[i * 5 + 3] = coff; rows[i * 5 + 4] = out[i].cigar.size();
        if (coff + out[i].cigar.size() <= cigar_cap) memcpy(cigars + coff, out[i].cigar.data(), out[i].cigar.size() * 4);
        coff += out[i].cigar.size();
    }
    return (int64_t)out.size();
}

// ---- whole path
void* orc_run(void* h, const uint8_t* read_pool, const uint64_t* read_offsets, uint64_t n_reads, const double* pv, uint32_t threads) {
    auto* ih = (index_handle*)h;
    std::vector<std::vector<uint8_t>> reads;
    for (uint64_t i = 0; i < n_reads; ++i) reads.emplace_back(read_pool + read_offsets[i], read_pool + read_offsets[i + 1]);
    auto rh = std::make_unique<run_handle>();
    auto t0 = std::chrono::steady_clock::now();
    rh->out = align_reads(ih->idx, ih->refs, reads, make_params(pv), threads);
    rh->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rh.release();
}
void orc_run_free(void* r) { delete (run_handle*)r; }
double orc_run_seconds(void* r) { return ((run_handle*)r)->seconds; }
uint64_t orc_run_num_records(void* r) { return ((run_handle*)r)->out.records.size(); }
uint64_t orc_run_num_cigar_words(void* r) { return ((run_handle*)r)->out.cigars.size(); }
// records rows (int64): {read_index, flag, ref_id, pos, nm, cigar_off, cigar_len}
void orc_run_get(void* r, int64_t* rows, uint32_t* cigars, uint8_t* skipped) {
    auto& o = ((run_handle*)r)->out;
    for (size_t i = 0; i < o.records.size(); ++i) {
        auto const& x = o.records[i];
        rows[i * 7] = (int64_t)x.read_index; rows[i * 7 + 1] = x.flag; rows[i * 7 + 2] = x.ref_id; rows[i * 7 + 3] = x.pos;
        rows[i * 7 + 4] = x.nm; rows[i * 7 + 5] = (int64_t)x.cigar_off; rows[i * 7 + 6] = (int64_t)x.cigar_len;
    }
    if (cigars && !o.cigars.empty()) memcpy(cigars, o.cigars.data(), o.cigars.size() * 4);
    if (skipped && !o.skipped.empty()) memcpy(skipped, o.skipped.data(), o.skipped.size());
}
// counters: {extend_all, extend_one, locates, lf_steps, inner_jobs, root_jobs, inner_word_steps, root_word_steps, ref_query_bytes}
void orc_run_counters(void* r, uint64_t* c) {
    auto& o = ((run_handle*)r)->out;
    c[0] = o.sc.n_extend_all; c[1] = o.sc.n_extend_one; c[2] = o.sc.n_locates; c[3] = o.sc.n_lf_steps;
    c[4] = o.vc.inner_jobs; c[5] = o.vc.root_jobs; c[6] = o.vc.inner_word_steps; c[7] = o.vc.root_word_steps; c[8] = o.vc.ref_query_bytes;
}

// the raw values behind histogram `id` (run_statistics, floxer_oracle.hpp): returns their number, copies them when out != null;
// id == N_STAT_LISTS: one value, the number of completely excluded queries
uint64_t orc_run_stat_values(void* r, uint32_t id, uint64_t* out) {
    auto& st = ((run_handle*)r)->out.st;
    if (id == (uint32_t)N_STAT_LISTS) { if (out) out[0] = st.completely_excluded_queries; return 1; }
    if (id > (uint32_t)N_STAT_LISTS) return 0;
    auto const& v = st.values[id];
    if (out && !v.empty()) memcpy(out, v.data(), v.size() * 8);
    return v.size();
}

}  // extern "C"
