// ORACLE — TEST INFRASTRUCTURE ONLY (see floxer_oracle.hpp for the parity status header).
// CPU restatement of floxer's hot path; every function cites the reference file:line it follows.
#include "floxer_oracle.hpp"

#include <algorithm>
#include <atomic>
#include <cassert>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <numeric>
#include <queue>
#include <set>
#include <stdexcept>
#include <thread>

namespace orc {

// ================================================================ math (include/math.hpp:10-27)
int32_t saturate_value_to_int32_max(uint64_t value) {
    if (value > (uint64_t)std::numeric_limits<int32_t>::max()) return std::numeric_limits<int32_t>::max();
    return (int32_t)value;
}
uint64_t ceil_div(uint64_t a, uint64_t b) { return (a % b) ? a / b + 1 : a / b; }
uint64_t floating_point_error_aware_ceil(double value) {
    static constexpr double epsilon = 0.000000001;
    return (uint64_t)(std::ceil(value - epsilon) + epsilon);
}

// ================================================================ input (input.cpp:161-176, ivs::d_dna5)
uint8_t char_to_rank(char c) {
    switch (c) {
        case '$': return 0;
        case 'A': case 'a': return 1;
        case 'C': case 'c': return 2;
        case 'G': case 'g': return 3;
        case 'T': case 't': case 'U': case 'u': return 4;   // input_test.cpp:24 ("'U' becomes 4")
        default: return 5;                                    // N and every invalid char (input.cpp:168-173)
    }
}
char rank_to_char(uint8_t r) { static const char t[] = "$ACGTN"; return r < 6 ? t[r] : 'N'; }
std::vector<uint8_t> chars_to_rank_sequence(const char* s, size_t n) {
    std::vector<uint8_t> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = char_to_rank(s[i]);
    return out;
}
std::vector<uint8_t> reverse_complement_rank(const std::vector<uint8_t>& seq) {
    static const uint8_t comp[6] = {0, 4, 3, 2, 1, 5};
    std::vector<uint8_t> out(seq.size());
    for (size_t i = 0; i < seq.size(); ++i) out[i] = comp[seq[seq.size() - 1 - i] % 6];
    return out;
}
std::string extract_record_id(const std::string& tag) { return tag.substr(0, tag.find(' ')); }

// ================================================================ PEX tree (pex.cpp:84-256)
pex_tree::pex_tree(uint64_t total_len, uint64_t query_num_errors, uint64_t leaf_max, bool bottom_up)
    : no_error_seed_length(total_len / (query_num_errors + 1)), leaf_max_num_errors(leaf_max) {
    if (!bottom_up) add_nodes_recursive(1, total_len, query_num_errors, pex_node::null_id);   // pex.cpp:90-96
    else add_nodes_bottom_up(total_len, query_num_errors);
}

void pex_tree::add_nodes_recursive(uint64_t from1, uint64_t to1, uint64_t num_errors, uint64_t parent_id) {
    uint64_t const num_leafs_left = ceil_div(num_errors + 1, 2);                               // pex.cpp:117
    pex_node const curr{parent_id, from1 - 1, to1 - 1, num_errors};
    if (num_errors <= leaf_max_num_errors) { leaves.push_back(curr); return; }                 // pex.cpp:126
    uint64_t const curr_id = inner_nodes.size();
    inner_nodes.push_back(curr);
    uint64_t const split = from1 + num_leafs_left * no_error_seed_length;                      // pex.cpp:136
    uint64_t const e_left = (num_leafs_left * num_errors) / (num_errors + 1);                  // pex.cpp:140
    uint64_t const e_right = ((num_errors + 1 - num_leafs_left) * num_errors) / (num_errors + 1);
    add_nodes_recursive(from1, split - 1, e_left, curr_id);
    add_nodes_recursive(split, to1, e_right, curr_id);
}

void pex_tree::add_nodes_bottom_up(uint64_t total_len, uint64_t query_num_errors) {           // pex.cpp:158-213
    uint64_t const base_leaf_weight = leaf_max_num_errors + 1;
    uint64_t const num_desired_leaves = ceil_div(query_num_errors + 1, base_leaf_weight);
    if (num_desired_leaves == 1) {
        leaves.push_back(pex_node{pex_node::null_id, 0, total_len - 1, query_num_errors});
        return;
    }
    // create_leaves, pex.cpp:215-239
    uint64_t const base_len = total_len / num_desired_leaves, rem = total_len % num_desired_leaves;
    uint64_t start = 0;
    for (uint64_t i = 0; i < num_desired_leaves; ++i) {
        uint64_t const len = i < rem ? base_len + 1 : base_len;
        leaves.push_back(pex_node{0, start, start + len - 1, leaf_max_num_errors});
        start += len;
    }
    inner_nodes.reserve(num_desired_leaves);
    inner_nodes.emplace_back();   // slot for the root at index 0
    // create_parent_node, pex.cpp:241-256
    auto make_parent = [](pex_node* children, size_t count, uint64_t parent_id) {
        uint64_t err = 0;
        for (size_t i = 0; i < count; ++i) { children[i].parent_id = parent_id; err += children[i].num_errors; }
        return pex_node{0, children[0].from, children[count - 1].to, err + count - 1};
    };
    pex_node* level = leaves.data();
    size_t level_size = leaves.size();
    while (level_size > 3) {
        for (size_t i = 0; i < level_size; i += 2) {
            size_t const remaining = level_size - i;
            if (remaining == 1) break;
            size_t const nchild = (remaining == 3) ? 3 : 2;
            uint64_t const new_id = inner_nodes.size();
            pex_node parent = make_parent(level + i, nchild, new_id);
            inner_nodes.push_back(parent);
        }
        level_size = level_size / 2;
        level = inner_nodes.data() + (inner_nodes.size() - level_size);
    }
    inner_nodes.front() = make_parent(level, level_size, 0);
    inner_nodes.front().parent_id = pex_node::null_id;
}

std::vector<seed_t> generate_seeds(const pex_tree& tree, const uint8_t* query, uint64_t step) {   // pex.cpp:258-277
    std::vector<seed_t> seeds;
    for (uint64_t i = 0; i < tree.leaves.size(); i += step) {
        auto const& leaf = tree.leaves[i];
        seeds.push_back(seed_t{query + leaf.from, leaf.length(), leaf.num_errors, leaf.from, i});
    }
    return seeds;
}

// ================================================================ FM index
// [3P-UNVERIFIED] fmindex-collection BiFMIndex(Sequences, samplingRate, threads): each sequence is followed by
// samplingRate - (len % samplingRate) zero sentinels (>= 1), SA = plain suffix array of that byte string, BWT of the text and
// of the reversed text, sampled SA keeps rows whose text position % samplingRate == 0, locate() LF-walks to a sampled row.
// (call sites: floxer.cpp:93-97, search.cpp:253, 284)
static std::vector<int64_t> build_suffix_array(const std::vector<uint8_t>& t) {
    // prefix doubling with bucket refinement; a suffix that is a prefix of another sorts first (plain SA order)
    int64_t const n = (int64_t)t.size();
    std::vector<int64_t> sa(n), rnk(n), tmp(n);
    if (n == 0) return sa;
    int const H0 = 8;
    std::vector<uint32_t> key(n);
    for (int64_t i = 0; i < n; ++i) {
        uint32_t k = 0;
        for (int j = 0; j < H0; ++j) k = (k << 3) | (i + j < n ? (uint32_t)t[i + j] + 1u : 0u);
        key[i] = k;
    }
    std::iota(sa.begin(), sa.end(), 0);
    std::sort(sa.begin(), sa.end(), [&](int64_t a, int64_t b) { return key[a] < key[b]; });
    rnk[sa[0]] = 0;
    for (int64_t i = 1; i < n; ++i) rnk[sa[i]] = key[sa[i]] == key[sa[i - 1]] ? rnk[sa[i - 1]] : i;
    for (int64_t h = H0; h < n; h *= 2) {
        bool all_unique = true;
        auto second = [&](int64_t p) { return p + h < n ? rnk[p + h] : (int64_t)-1; };
        int64_t i = 0;
        while (i < n) {
            int64_t j = i + 1;
            while (j < n && rnk[sa[j]] == rnk[sa[i]]) ++j;
            if (j - i > 1) {
                all_unique = false;
                std::sort(sa.begin() + i, sa.begin() + j, [&](int64_t a, int64_t b) { return second(a) < second(b); });
            }
            i = j;
        }
        if (all_unique) break;
        // recompute ranks (rank = index of first element of its group)
        i = 0;
        while (i < n) {
            int64_t j = i + 1;
            while (j < n && rnk[sa[j]] == rnk[sa[i]]) ++j;
            // within [i,j) split by second key
            int64_t g = i;
            tmp[sa[i]] = i;
            for (int64_t x = i + 1; x < j; ++x) {
                if (second(sa[x]) != second(sa[x - 1])) g = x;
                tmp[sa[x]] = g;
            }
            i = j;
        }
        rnk.swap(tmp);
    }
    return sa;
}

static void build_occ(const std::vector<uint8_t>& bwt, std::vector<uint32_t>& cp) {
    uint64_t const n = bwt.size();
    cp.assign((n / 64 + 1) * 6, 0);
    uint32_t cnt[6] = {0, 0, 0, 0, 0, 0};
    for (uint64_t i = 0; i < n; ++i) {
        if (i % 64 == 0) for (int c = 0; c < 6; ++c) cp[(i / 64) * 6 + c] = cnt[c];
        cnt[bwt[i]]++;
    }
    if (n % 64 == 0) for (int c = 0; c < 6; ++c) cp[(n / 64) * 6 + c] = cnt[c];
}

static void lay_out_text(fm_index& idx, const std::vector<std::vector<uint8_t>>& refs, uint32_t sampling) {
    idx.sampling = sampling;
    uint64_t total = 0;
    for (auto const& r : refs) total += r.size() + (sampling - (r.size() % sampling));
    idx.text.assign(total, 0);
    uint64_t at = 0;
    for (auto const& r : refs) {
        idx.seq_start.push_back(at);
        idx.seq_len.push_back(r.size());
        std::copy(r.begin(), r.end(), idx.text.begin() + at);
        at += r.size() + (sampling - (r.size() % sampling));         // delimiters: zero sentinels up to the next multiple (>= 1)
    }
    idx.n = idx.text.size();
    if (idx.n >= (uint64_t)1 << 32) throw std::runtime_error("oracle index limited to < 2^32 symbols");
}

static void finish_index(fm_index& idx) {
    uint64_t cnt[6] = {0, 0, 0, 0, 0, 0};
    for (auto c : idx.text) cnt[c]++;
    idx.C[0] = 0;
    for (int c = 0; c < 6; ++c) idx.C[c + 1] = idx.C[c] + cnt[c];
    build_occ(idx.bwt, idx.occ_cp);
    build_occ(idx.bwt_rev, idx.occ_rev_cp);
}

fm_index build_index(const std::vector<std::vector<uint8_t>>& refs, uint32_t sampling) {
    fm_index idx;
    lay_out_text(idx, refs, sampling);
    {
        auto const sa = build_suffix_array(idx.text);
        idx.sa.assign(sa.begin(), sa.end());
    }
    idx.bwt.resize(idx.n);
    for (uint64_t i = 0; i < idx.n; ++i) idx.bwt[i] = idx.text[(idx.sa[i] + idx.n - 1) % idx.n];
    {
        std::vector<uint8_t> rev(idx.text.rbegin(), idx.text.rend());
        auto sa_rev = build_suffix_array(rev);
        idx.bwt_rev.resize(idx.n);
        for (uint64_t i = 0; i < idx.n; ++i) idx.bwt_rev[i] = rev[(sa_rev[i] + idx.n - 1) % idx.n];
    }
    finish_index(idx);
    return idx;
}

fm_index import_index(const std::vector<std::vector<uint8_t>>& refs, uint32_t sampling, const uint32_t* sa, const uint8_t* bwt,
                      const uint8_t* bwt_rev) {
    fm_index idx;
    lay_out_text(idx, refs, sampling);
    uint64_t const n = idx.n;
    idx.sa.assign(sa, sa + n);
    idx.bwt.assign(bwt, bwt + n);
    idx.bwt_rev.assign(bwt_rev, bwt_rev + n);
    // consistency on a sample of rows: neighbouring suffixes in order, BWT = symbol in front of the suffix, and the two BWTs are
    // permutations of the same text
    auto suffix_less_eq = [&](uint64_t a, uint64_t b) {
        for (uint64_t d = 0; d < 4096; ++d) {
            if (a + d >= n) return true;
            if (b + d >= n) return false;
            if (idx.text[a + d] != idx.text[b + d]) return idx.text[a + d] < idx.text[b + d];
        }
        return true;
    };
    uint64_t const step = std::max<uint64_t>(1, n / 100000);
    for (uint64_t i = 0; i < n; i += step) {
        if (idx.sa[i] >= n) throw std::runtime_error("imported suffix array: entry outside the text");
        if (i + 1 < n && !suffix_less_eq(idx.sa[i], idx.sa[i + 1])) throw std::runtime_error("imported suffix array: rows out of order");
        if (idx.bwt[i] != idx.text[(idx.sa[i] + n - 1) % n]) throw std::runtime_error("imported BWT does not match the suffix array");
    }
    uint64_t c0[8] = {0}, c1[8] = {0}, c2[8] = {0};
    for (uint64_t i = 0; i < n; ++i) { c0[idx.text[i] & 7]++; c1[idx.bwt[i] & 7]++; c2[idx.bwt_rev[i] & 7]++; }
    for (int c = 0; c < 8; ++c) if (c0[c] != c1[c] || c0[c] != c2[c]) throw std::runtime_error("imported BWT is not a permutation of the text");
    finish_index(idx);
    return idx;
}

uint64_t fm_index::occ(bool rev, uint8_t c, uint64_t i) const {
    auto const& b = rev ? bwt_rev : bwt;
    auto const& cp = rev ? occ_rev_cp : occ_cp;
    uint64_t r = cp[(i / 64) * 6 + c];
    for (uint64_t x = (i / 64) * 64; x < i; ++x) r += (b[x] == c);
    return r;
}
void fm_index::all_occ(bool rev, uint64_t i, uint64_t out[6]) const {
    auto const& b = rev ? bwt_rev : bwt;
    auto const& cp = rev ? occ_rev_cp : occ_cp;
    for (int c = 0; c < 6; ++c) out[c] = cp[(i / 64) * 6 + c];
    for (uint64_t x = (i / 64) * 64; x < i; ++x) out[b[x]]++;
}
void fm_index::locate(uint64_t row, uint64_t& seq_id, uint64_t& pos, uint64_t* lf_steps) const {
    uint64_t steps = 0;
    while (sa[row] % sampling != 0) {                    // csa.value(idx) is empty for unsampled rows
        uint8_t const c = bwt[row];
        row = C[c] + occ(false, c, row);                 // idx = occ.rank(idx, occ.symbol(idx))
        ++steps;
    }
    uint64_t const p = (uint64_t)sa[row];
    size_t const s = std::upper_bound(seq_start.begin(), seq_start.end(), p) - seq_start.begin() - 1;
    seq_id = s;
    pos = p - seq_start[s] + steps;
    if (lf_steps) *lf_steps += steps;
}

// ================================================================ search schemes
// [3P-UNVERIFIED] search_schemes::generator::optimum(0,K): Kianfar et al. optimum schemes, 0-based parts.
std::vector<search_def> optimum_scheme(uint32_t k) {
    switch (k) {
        case 0: return {{{0}, {0}, {0}}};
        case 1: return {{{0, 1}, {0, 0}, {0, 1}}, {{1, 0}, {0, 1}, {0, 1}}};
        case 2: return {{{0, 1, 2, 3}, {0, 0, 1, 1}, {0, 0, 2, 2}},
                        {{2, 1, 0, 3}, {0, 0, 0, 0}, {0, 1, 1, 2}},
                        {{3, 2, 1, 0}, {0, 0, 0, 2}, {0, 1, 2, 2}}};
        case 3:  // [3P-UNVERIFIED: complete and non-redundant (checked in tests/test_oracle_pins.py); order of the
                 //  searches and the exact L/U rows are a recollection of Kianfar et al.'s K=3, P=5 table]
            return {{{0, 1, 2, 3, 4}, {0, 0, 0, 0, 0}, {0, 0, 3, 3, 3}},
                    {{2, 1, 0, 3, 4}, {0, 0, 1, 1, 1}, {0, 1, 1, 2, 3}},
                    {{3, 2, 1, 0, 4}, {0, 0, 0, 2, 2}, {0, 1, 2, 2, 3}},
                    {{4, 3, 2, 1, 0}, {0, 0, 0, 0, 3}, {0, 2, 2, 3, 3}}};
        default: throw std::runtime_error("optimum scheme only for k <= 3");
    }
}


// [3P-UNVERIFIED] search_schemes::expand(scheme, len): part p (by position) gets len/P characters, the first len%P
// parts one more; pi is expanded character-wise in search direction; U is repeated per character; L applies only at
// the last character of a part (earlier characters inherit the previous part's lower bound).
std::vector<search_def> expand_scheme(const std::vector<search_def>& scheme, uint64_t len) {
    std::vector<search_def> out;
    for (auto const& s : scheme) {
        uint64_t const P = s.pi.size();
        if (len < P) return {};                          // not expandable -> search_n sees an empty scheme
        std::vector<uint64_t> counts(P, len / P), starts(P, 0);
        for (uint64_t i = 0; i < len % P; ++i) counts[i] += 1;
        for (uint64_t i = 1; i < P; ++i) starts[i] = starts[i - 1] + counts[i - 1];
        search_def e;
        for (uint64_t i = 0; i < P; ++i) {
            uint32_t const part = s.pi[i];
            bool const right = (i == 0) ? true : (s.pi[i - 1] < s.pi[i]);
            for (uint64_t j = 0; j < counts[part]; ++j) {
                e.pi.push_back((uint32_t)(right ? starts[part] + j : starts[part] + counts[part] - 1 - j));
                e.u.push_back(s.u[i]);
                e.l.push_back(j + 1 == counts[part] ? s.l[i] : (i > 0 ? s.l[i - 1] : 0));
            }
        }
        out.push_back(std::move(e));
    }
    return out;
}

// ================================================================ search_ng21 (fmindex-collection/search/SearchNg21.h)
// [3P-UNVERIFIED] depth-first search over one expanded search with edit operations. Side-info flags ('M' match,
// 'S' substitution, 'D' deletion = extra text symbol, 'I' insertion = unmatched query symbol) per extension side
// forbid I directly after D/S and D directly after I/S, and hits whose outermost operation on a side is D or S are not
// reported. Child order: match; for each symbol 1..5 {deletion, substitution}; insertion.
namespace {

struct ng21 {
    const fm_index& idx;
    const search_def& s;
    const uint8_t* query;
    uint64_t len;
    std::function<bool(cursor const&, uint64_t)> delegate;
    search_counters* ctr;

    bool is_right(uint64_t part) const { return part == 0 || s.pi[part - 1] < s.pi[part]; }

    // BiFMIndexCursor::extendRight()/extendLeft() for all symbols
    void extend_all(cursor const& cur, bool right, cursor out[6]) const {
        if (ctr) ctr->n_extend_all++;
        uint64_t a[6], b[6];
        if (right) {
            idx.all_occ(true, cur.lb_rev, a);
            idx.all_occ(true, cur.lb_rev + cur.len, b);
            uint64_t acc = cur.lb;
            for (int c = 0; c < 6; ++c) {
                out[c].lb_rev = idx.C[c] + a[c];
                out[c].len = b[c] - a[c];
                out[c].lb = acc;
                acc += out[c].len;
            }
        } else {
            idx.all_occ(false, cur.lb, a);
            idx.all_occ(false, cur.lb + cur.len, b);
            uint64_t acc = cur.lb_rev;
            for (int c = 0; c < 6; ++c) {
                out[c].lb = idx.C[c] + a[c];
                out[c].len = b[c] - a[c];
                out[c].lb_rev = acc;
                acc += out[c].len;
            }
        }
    }
    cursor extend_one(cursor const& cur, bool right, uint8_t sym) const {
        if (ctr) ctr->n_extend_one++;
        cursor all[6];
        search_counters* saved = ctr;
        const_cast<ng21*>(this)->ctr = nullptr;
        extend_all(cur, right, all);
        const_cast<ng21*>(this)->ctr = saved;
        return all[sym];
    }

    bool search_part(cursor const& cur, uint64_t e, uint64_t part, char LInfo, char RInfo) const {
        if (cur.len == 0) return false;
        if (part == len) {
            if ((LInfo == 'M' || LInfo == 'I') && (RInfo == 'M' || RInfo == 'I')) {
                if (s.l[part - 1] <= e && e <= s.u[part - 1]) return delegate(cur, e);
            }
            return false;
        }
        if (e > s.u[part]) return false;

        bool const mismatch_allowed = s.l[part] <= e + 1 && e + 1 <= s.u[part];
        bool const match_allowed = s.l[part] <= e && e <= s.u[part];
        bool const right = is_right(part);
        char const TInfo = right ? RInfo : LInfo;
        bool const deletion = TInfo == 'M' || TInfo == 'D';
        bool const insertion = TInfo == 'M' || TInfo == 'I';
        auto L = [&](char c) { return right ? LInfo : c; };
        auto R = [&](char c) { return right ? c : RInfo; };
        uint8_t const next_symb = query[s.pi[part]];

        if (mismatch_allowed) {
            cursor cursors[6];
            extend_all(cur, right, cursors);
            if (match_allowed) {
                if (search_part(cursors[next_symb], e, part + 1, L('M'), R('M'))) return true;
            }
            for (uint8_t i = 1; i < 6; ++i) {
                if (deletion) {
                    if (search_part(cursors[i], e + 1, part, L('D'), R('D'))) return true;   // extra symbol in the text
                }
                if (i == next_symb) continue;
                if (search_part(cursors[i], e + 1, part + 1, L('S'), R('S'))) return true;   // substitution
            }
            if (insertion) {
                if (search_part(cur, e + 1, part + 1, L('I'), R('I'))) return true;           // query symbol unmatched
            }
        } else if (match_allowed) {
            cursor const nc = extend_one(cur, right, next_symb);
            return search_part(nc, e, part + 1, L('M'), R('M'));
        }
        return false;
    }
};

}  // namespace

// search_ng21::search_n: report cursors until n hits are collected; the last cursor is truncated (search.cpp:173-188)
void search_n(const fm_index& idx, const uint8_t* query, uint64_t len, uint32_t k, uint64_t n,
              std::vector<anchor_group>& out, search_counters* ctr) {
    auto const scheme = expand_scheme(optimum_scheme(k), len);
    if (scheme.empty()) return;
    uint64_t ct = 0;
    auto delegate = [&](cursor const& cur, uint64_t e) {
        if (ct + cur.len > n) {
            cursor c2 = cur;
            c2.len = n - ct;
            out.push_back(anchor_group{c2, e});
            ct = n;
        } else {
            out.push_back(anchor_group{cur, e});
            ct += cur.len;
        }
        return ct == n;
    };
    for (auto const& s : scheme) {
        ng21 search{idx, s, query, len, delegate, ctr};
        if (search.search_part(cursor{0, 0, idx.n}, 0, 0, 'M', 'M')) break;
    }
}

// ================================================================ search.cpp
static bool is_better_than(anchor_t const& a, anchor_t const& other) {                        // search.cpp:38-44
    uint64_t const d = a.reference_position < other.reference_position ? other.reference_position - a.reference_position
                                                                       : a.reference_position - other.reference_position;
    return a.num_errors <= other.num_errors && d <= other.num_errors - a.num_errors;
}

uint64_t erase_useless_anchors(std::vector<std::vector<anchor_t>>& anchors_by_reference) {   // search.cpp:352-389
    uint64_t kept = 0;
    for (auto& v : anchors_by_reference) {
        if (v.empty()) continue;
        std::sort(v.begin(), v.end(),
                  [](anchor_t const& x, anchor_t const& y) { return x.reference_position < y.reference_position; });
        for (size_t cur = 0; cur < v.size() - 1;) {
            size_t other = cur + 1;
            while (other < v.size() && is_better_than(v[cur], v[other])) {
                v[other].num_errors = erase_marker;
                ++other;
            }
            if (other < v.size() && is_better_than(v[other], v[cur])) v[cur].num_errors = erase_marker;
            cur = other;
        }
        v.erase(std::remove_if(v.begin(), v.end(), [](anchor_t const& a) { return a.num_errors == erase_marker; }), v.end());
        kept += v.size();
    }
    return kept;
}

std::vector<anchors_of_seed> search_seeds(const fm_index& idx, const std::vector<seed_t>& seeds, const search_config& cfg,
                                          search_counters* ctr) {                             // search.cpp:143-324
    std::vector<anchors_of_seed> result;
    size_t const num_refs = idx.seq_len.size();
    for (auto const& seed : seeds) {
        std::vector<anchor_group> groups;
        uint64_t const n = cfg.anchor_choice_strategy == CHOICE_FIRST_REPORTED
                               ? cfg.max_num_anchors_soft
                               : std::max(cfg.max_num_anchors_hard, cfg.max_num_anchors_hard + 1);
        search_n(idx, seed.seq, seed.len, (uint32_t)seed.num_errors, n, groups, ctr);
        uint64_t total_raw = 0;
        for (auto const& g : groups) total_raw += g.cur.len;

        if (total_raw > cfg.max_num_anchors_hard && cfg.anchor_choice_strategy != CHOICE_FIRST_REPORTED) {
            result.emplace_back();                                                             // search.cpp:190-202
            continue;
        }
        switch (cfg.anchor_group_order) {                                                      // search.cpp:204-229
            case ORDER_COUNT_FIRST:
                std::sort(groups.begin(), groups.end(), [](anchor_group const& g1, anchor_group const& g2) {
                    if (g1.cur.len != g2.cur.len) return g1.cur.len < g2.cur.len;
                    return g1.num_errors < g2.num_errors;
                });
                break;
            case ORDER_ERRORS_FIRST:   // reproduced literally, search.cpp:215-222
                std::sort(groups.begin(), groups.end(), [](anchor_group const& g1, anchor_group const& g2) {
                    if (g1.num_errors != g2.num_errors) return g1.cur.len < g2.cur.len;
                    return g1.num_errors < g2.num_errors;
                });
                break;
            default: break;
        }
        uint64_t kept_raw = 0;
        std::vector<std::vector<anchor_t>> by_ref(num_refs);
        auto locate_into = [&](uint64_t row, uint64_t errors) {
            uint64_t rid, pos;
            if (ctr) ctr->n_locates++;
            idx.locate(row, rid, pos, ctr ? &ctr->n_lf_steps : nullptr);
            by_ref[rid].push_back(anchor_t{seed.pex_leaf_index, rid, pos, errors});
            ++kept_raw;
        };
        if (cfg.anchor_choice_strategy == CHOICE_ROUND_ROBIN) {                                // search.cpp:239-272
            std::set<size_t> remaining;
            for (size_t i = 0; i < groups.size(); ++i) remaining.insert(i);
            auto it = remaining.begin();
            uint64_t round = 0;
            while (kept_raw != cfg.max_num_anchors_soft && !remaining.empty()) {
                auto const& g = groups[*it];
                locate_into(g.cur.lb + round, g.num_errors);
                auto prev = it;
                ++it;
                if (g.cur.len == round + 1) remaining.erase(prev);
                if (it == remaining.end()) { it = remaining.begin(); ++round; }
            }
        } else {                                                                               // search.cpp:273-299
            size_t gi = 0;
            while (kept_raw != cfg.max_num_anchors_soft && gi < groups.size()) {
                auto const& g = groups[gi];
                for (uint64_t row = g.cur.lb; row < g.cur.lb + g.cur.len; ++row) {
                    locate_into(row, g.num_errors);
                    if (kept_raw == cfg.max_num_anchors_soft) break;
                }
                ++gi;
            }
        }
        anchors_of_seed a;
        a.num_excluded_raw_anchors_by_soft_cap = total_raw - kept_raw;
        a.num_kept_raw_anchors = kept_raw;
        a.num_kept_useful_anchors = kept_raw;
        if (cfg.erase_useless_anchors) a.num_kept_useful_anchors = erase_useless_anchors(by_ref);
        a.anchors_by_reference = std::move(by_ref);
        result.push_back(std::move(a));
    }
    return result;
}

// ================================================================ alignment
// alignment.cpp:83-181 configures seqan3::align_pairwise as: unit-cost edit distance, sequence1 = reference window with
// free leading+trailing gaps, sequence2 = query without free gaps, min_score = -k.
// [3P-UNVERIFIED seqan3 edit_distance_unbanded] best end column = last column with the minimal last-row score
// (update on score <= best; pinned by floxer_whole_program_via_cli_test.cpp:75-84); trace priority up (I) > left (D) >
// diagonal (=/X) (up > diagonal pinned ibid. :70-84; left > diagonal and up > left are recollection).
std::string cigar_to_string(const uint32_t* c, uint64_t n) {
    static const char ops[] = "MIDNSHP=X";
    std::string s;
    for (uint64_t i = 0; i < n; ++i) { s += std::to_string(c[i] >> 4); s += ops[c[i] & 15]; }
    return s;
}

namespace {

enum : uint32_t { OP_I = 1, OP_D = 2, OP_EQ = 7, OP_X = 8 };

void push_op_reversed(std::vector<uint32_t>& rev_ops, uint32_t op) {
    if (!rev_ops.empty() && (rev_ops.back() & 15) == op) rev_ops.back() += 16;
    else rev_ops.push_back((1u << 4) | op);
}

align_result align_dp(const uint8_t* ref, uint64_t n, const uint8_t* query, uint64_t m, uint64_t k, bool want_trace,
                      align_counters* ctr) {
    // D[i][j]: i query rows, j reference columns
    std::vector<uint32_t> D((m + 1) * (n + 1));
    auto at = [&](uint64_t i, uint64_t j) -> uint32_t& { return D[i * (n + 1) + j]; };
    for (uint64_t j = 0; j <= n; ++j) at(0, j) = 0;
    for (uint64_t i = 1; i <= m; ++i) {
        at(i, 0) = (uint32_t)i;
        for (uint64_t j = 1; j <= n; ++j) {
            uint32_t const d = at(i - 1, j - 1) + (query[i - 1] != ref[j - 1] ? 1u : 0u);
            at(i, j) = std::min(d, std::min(at(i - 1, j) + 1, at(i, j - 1) + 1));
        }
    }
    if (ctr) ctr->cells += m * n;
    align_result r;
    uint32_t best = at(m, 0);
    uint64_t best_col = 0;
    for (uint64_t j = 1; j <= n; ++j) if (at(m, j) <= best) { best = at(m, j); best_col = j; }
    if (best > k) return r;
    r.exists = true;
    r.num_errors = best;
    r.begin = best_col;     // end column; callers use it for MODE_WITHOUT_CIGAR
    if (!want_trace) return r;
    std::vector<uint32_t> rev;
    uint64_t i = m, j = best_col;
    while (i > 0) {
        bool const up = at(i, j) == at(i - 1, j) + 1;
        bool const left = j > 0 && at(i, j) == at(i, j - 1) + 1;
        if (up) { push_op_reversed(rev, OP_I); --i; }
        else if (left) { push_op_reversed(rev, OP_D); --j; }
        else { push_op_reversed(rev, query[i - 1] == ref[j - 1] ? OP_EQ : OP_X); --i; --j; }
    }
    r.begin = j;
    r.cigar.assign(rev.rbegin(), rev.rend());
    return r;
}

align_result align_myers(const uint8_t* ref, uint64_t n, const uint8_t* query, uint64_t m, uint64_t k, bool want_trace,
                         align_counters* ctr) {
    uint64_t const W = (m + 63) / 64;
    std::vector<uint64_t> peq(8 * W, 0);
    for (uint64_t i = 0; i < m; ++i) peq[(query[i] & 7) * W + i / 64] |= uint64_t(1) << (i % 64);
    std::vector<uint64_t> vp(W, ~uint64_t(0)), vn(W, 0);
    std::vector<uint64_t> tr_hp, tr_vp, tr_db;           // per column j>=1: W words each
    if (want_trace) { tr_hp.resize(n * W); tr_vp.resize(n * W); tr_db.resize(n * W); }
    uint64_t const last_bit = uint64_t(1) << ((m - 1) % 64);
    uint64_t score = m, best = m, best_col = 0;
    for (uint64_t j = 1; j <= n; ++j) {
        const uint64_t* eqs = &peq[(ref[j - 1] & 7) * W];
        uint64_t carry_d0 = 0, carry_hp = 0, carry_hn = 0;   // semi-global: row 0 has horizontal delta 0
        for (uint64_t w = 0; w < W; ++w) {
            uint64_t const eq = eqs[w], pv = vp[w], mv = vn[w];
            uint64_t const x = eq | mv;
            uint64_t const a = x & pv;
            uint64_t const t1 = pv + a;
            uint64_t const t = t1 + carry_d0;
            uint64_t const cout = (t1 < pv) | (t < t1);
            uint64_t const d0 = (t ^ pv) | x;
            uint64_t const hn = pv & d0;
            uint64_t const hp = mv | ~(pv | d0);
            carry_d0 = cout;
            uint64_t const xh = (hp << 1) | carry_hp;
            uint64_t const nvn = xh & d0;
            uint64_t const nvp = (hn << 1) | ~(xh | d0) | carry_hn;
            carry_hp = hp >> 63;
            carry_hn = hn >> 63;
            vn[w] = nvn;
            vp[w] = nvp;
            if (want_trace) {
                tr_hp[(j - 1) * W + w] = hp;
                tr_vp[(j - 1) * W + w] = nvp;
                tr_db[(j - 1) * W + w] = ~(eq ^ d0);
            }
            if (w == W - 1) {
                if (hp & last_bit) ++score;
                else if (hn & last_bit) --score;
            }
        }
        if (score <= best) { best = score; best_col = j; }
    }
    if (ctr) ctr->word_steps += n * W;
    align_result r;
    if (best > k) return r;
    r.exists = true;
    r.num_errors = best;
    r.begin = best_col;
    if (!want_trace) return r;
    std::vector<uint32_t> rev;
    uint64_t i = m, j = best_col;
    while (i > 0) {
        uint64_t const w = (i - 1) / 64, b = (i - 1) % 64;
        bool up, left, diag;
        if (j == 0) { up = true; left = diag = false; }
        else {
            up = (tr_vp[(j - 1) * W + w] >> b) & 1;
            left = (tr_hp[(j - 1) * W + w] >> b) & 1;
            diag = (tr_db[(j - 1) * W + w] >> b) & 1;
        }
        (void)diag;
        if (up) { push_op_reversed(rev, OP_I); --i; }
        else if (left) { push_op_reversed(rev, OP_D); --j; }
        else { push_op_reversed(rev, query[i - 1] == ref[j - 1] ? OP_EQ : OP_X); --i; --j; }
    }
    r.begin = j;
    r.cigar.assign(rev.rbegin(), rev.rend());
    return r;
}

}  // namespace

align_result align(const uint8_t* ref, uint64_t n, const uint8_t* query, uint64_t m, uint64_t k, int mode, int algo,
                   align_counters* ctr) {
    auto run = [&](const uint8_t* r, const uint8_t* q, bool trace) {
        return algo == 0 ? align_dp(r, n, q, m, k, trace, ctr) : align_myers(r, n, q, m, k, trace, ctr);
    };
    if (mode == MODE_EXISTS) {                                                                 // alignment.cpp:98-112
        align_result r = run(ref, query, false);
        r.begin = 0;
        return r;
    }
    if (mode == MODE_WITHOUT_CIGAR) {                                                          // alignment.cpp:115-145
        std::vector<uint8_t> rr(ref, ref + n), rq(query, query + m);
        std::reverse(rr.begin(), rr.end());
        std::reverse(rq.begin(), rq.end());
        align_result r = run(rr.data(), rq.data(), false);
        if (r.exists) r.begin = n - r.begin;     // reference.size() - sequence1_end_position
        return r;
    }
    return run(ref, query, true);                                                              // alignment.cpp:147-180
}

// ================================================================ intervals (intervals.cpp:26-127)
int relationship_with(half_open_interval a, half_open_interval o) {
    if (a.start > o.end) return 0;                                  // completely_above
    if (a.end < o.start) return 1;                                  // completely_below
    if (a.start == o.start && a.end == o.end) return 3;             // equal
    if (a.start <= o.start && a.end >= o.end) return 2;             // contains
    if (a.start >= o.start && a.end <= o.end) return 4;             // inside
    if (a.start > o.start && a.start <= o.end) return 5;            // overlapping_or_touching_above
    return 6;                                                       // overlapping_or_touching_below
}
half_open_interval trim_from_both_sides(half_open_interval a, uint64_t amount) {
    uint64_t const new_end = std::max(a.start + 1, amount > a.end ? 0 : a.end - amount);
    uint64_t const new_start = std::min(new_end - 1, a.start + amount);
    return {new_start, new_end};
}
void verified_intervals::insert(half_open_interval iv) {
    if (!active || contains(iv)) return;
    ivs.push_back(iv);
}
bool verified_intervals::contains(half_open_interval target) const {
    if (!active) return false;
    for (auto const& e : ivs) {
        // overlap_find_all on closed intervals [low, high]
        if (e.start <= target.end && target.start <= e.end) {
            int const rel = relationship_with(e, target);
            if (rel == 3 || rel == 2) return true;
        }
    }
    return false;
}

// ================================================================ verification (verification.cpp)
span_config compute_reference_span_start_and_length(uint64_t anchor_pos, const pex_node& node, uint64_t leaf_from,
                                                    uint64_t full_reference_length, double extra_ratio) {   // :157-184
    uint64_t const base = node.length() + 2 * node.num_errors + 1;
    uint64_t const extra = floating_point_error_aware_ceil(base * extra_ratio);
    int64_t const start_signed = (int64_t)anchor_pos - (int64_t)(leaf_from - node.from) - (int64_t)node.num_errors - (int64_t)extra;
    uint64_t const start = start_signed >= 0 ? (uint64_t)start_signed : 0;
    uint64_t const length = std::min(base + 2 * extra, full_reference_length - start);
    return span_config{start, length, extra};
}

namespace {

struct verifier {
    const pex_tree& tree;
    const anchor_t& anchor;
    const pex_node& leaf;
    const uint8_t* query;
    bool reverse;
    const uint8_t* reference;
    uint64_t reference_len;
    const params& p;
    verified_intervals& ivs;
    std::vector<query_alignment>& out;
    verify_counters* ctr;
    run_statistics* st;

    span_config root_span() const {
        return compute_reference_span_start_and_length(anchor.reference_position, tree.root(), leaf.from, reference_len,
                                                       p.extra_verification_ratio);
    }
    bool root_was_already_verified() const {                                                   // verification.cpp:119-136
        auto const sc = root_span();
        auto const target = trim_from_both_sides({sc.offset, sc.offset + sc.length}, sc.extra);
        if (ivs.contains(target)) {
            if (st) st->values[13].push_back(sc.length);                                        // verification.cpp:130
            return true;
        }
        return false;
    }
    bool try_align(const pex_node& node, span_config sc) {                                     // verification.cpp:186-245
        int mode = MODE_EXISTS;
        if (node.is_root()) mode = p.without_cigar ? MODE_WITHOUT_CIGAR : MODE_WITH_CIGAR;
        align_counters ac;
        align_result r = align(reference + sc.offset, sc.length, query + node.from, node.length(), node.num_errors, mode,
                               p.align_algo, &ac);
        if (ctr) {
            uint64_t const ws = sc.length * ((node.length() + 63) / 64);
            if (node.is_root()) { ctr->root_jobs++; ctr->root_word_steps += ws; }
            else { ctr->inner_jobs++; ctr->inner_word_steps += ws; }
            ctr->ref_query_bytes += sc.length + node.length();
        }
        if (st) st->values[node.is_root() ? 12 : 11].push_back(sc.length);                      // verification.cpp:238-242
        if (r.exists && mode != MODE_EXISTS)
            out.push_back(query_alignment{sc.offset + r.begin, r.num_errors, reverse, std::move(r.cigar)});
        return r.exists;
    }
    void direct_full() {                                                                       // verification.cpp:23-42
        if (root_was_already_verified()) return;
        auto const sc = root_span();
        try_align(tree.root(), sc);
        ivs.insert({sc.offset, sc.offset + sc.length});
    }
    void hierarchical() {                                                                      // verification.cpp:44-117
        if (root_was_already_verified()) return;
        auto const rsc = root_span();
        if (leaf.is_root()) {
            try_align(leaf, rsc);
            ivs.insert({rsc.offset, rsc.offset + rsc.length});
            return;
        }
        const pex_node* node = &tree.inner_nodes.at(leaf.parent_id);
        while (true) {
            auto const sc = compute_reference_span_start_and_length(anchor.reference_position, *node, leaf.from,
                                                                    reference_len, node->is_root() ? p.extra_verification_ratio : 0.0);
            if (sc.length > 512 && root_was_already_verified()) return;
            bool const exists = try_align(*node, sc);
            if (node->is_root()) ivs.insert({sc.offset, sc.offset + sc.length});
            if (!exists || node->is_root()) break;
            node = &tree.inner_nodes.at(node->parent_id);
        }
    }
};

}  // namespace

void verify_anchor(const pex_tree& tree, const anchor_t& anchor, const uint8_t* query, bool reverse, const uint8_t* reference,
                   uint64_t reference_len, const params& p, verified_intervals& ivs, std::vector<query_alignment>& out,
                   verify_counters* ctr, run_statistics* st) {
    verifier v{tree, anchor, tree.leaves.at(anchor.pex_leaf_index), query, reverse, reference, reference_len, p, ivs, out, ctr, st};
    if (p.direct_full) v.direct_full();
    else v.hierarchical();
}

// ================================================================ whole path (parallelization.cpp, output.cpp), --threads 1 order
namespace {

struct anchor_package { std::vector<anchor_t> anchors; bool reverse; };

void append_anchor_packages(const std::vector<anchors_of_seed>& res, std::vector<anchor_package>& out, uint64_t per_package,
                            bool reverse) {                                                    // search.cpp:78-141
    anchor_package pkg{{}, reverse};
    for (auto const& s : res)
        for (auto const& by_ref : s.anchors_by_reference)
            for (auto const& a : by_ref) {
                pkg.anchors.push_back(a);
                if (pkg.anchors.size() == per_package) { out.push_back(std::move(pkg)); pkg = anchor_package{{}, reverse}; }
            }
    if (!pkg.anchors.empty()) out.push_back(std::move(pkg));
}

// BS::thread_pool 4.1.0 keeps tasks in a std::priority_queue ordered by priority only; with one worker the verification
// tasks (high priority) of a read all run before the next search task (low priority), in heap order (parallelization.cpp:131-148, 291)
struct pr_task { int priority; int id; bool operator<(pr_task const& o) const { return priority < o.priority; } };

std::vector<int> single_thread_task_order(int n_packages) {
    std::priority_queue<pr_task> q;
    for (int i = 0; i < n_packages; ++i) q.push(pr_task{16383, i});
    q.push(pr_task{-16384, -1});
    std::vector<int> order;
    while (!q.empty()) {
        pr_task t = q.top();
        q.pop();
        if (t.id < 0) break;
        order.push_back(t.id);
    }
    return order;
}

struct read_result {
    bool skipped = false;
    std::vector<record> records;        // cigar_off relative to own pool
    std::vector<uint32_t> cigars;
    search_counters sc;
    verify_counters vc;
    run_statistics st;
};

read_result align_one_read(const fm_index& idx, const std::vector<std::vector<uint8_t>>& refs, const std::vector<uint8_t>& read,
                           uint64_t read_index, const params& p) {
    read_result rr;
    uint64_t const len = read.size();
    // input.cpp:95-129
    if (len == 0 || len > 100000) { rr.skipped = true; return rr; }
    uint64_t const k = p.error_probability >= 0 ? floating_point_error_aware_ceil(len * p.error_probability) : p.query_num_errors;
    if (len <= k || k < p.seed_errors) { rr.skipped = true; return rr; }

    auto const rc = reverse_complement_rank(read);
    pex_tree const tree(len, k, p.seed_errors, p.bottom_up);                                   // parallelization.cpp:91-92
    auto const fwd_seeds = generate_seeds(tree, read.data(), p.seed_sampling_step);
    auto const rev_seeds = generate_seeds(tree, rc.data(), p.seed_sampling_step);
    auto const fwd_res = search_seeds(idx, fwd_seeds, p.search, &rr.sc);
    auto const rev_res = search_seeds(idx, rev_seeds, p.search, &rr.sc);
    std::vector<anchor_package> packages;                                                      // parallelization.cpp:14-43
    append_anchor_packages(fwd_res, packages, p.anchors_per_task, false);
    append_anchor_packages(rev_res, packages, p.anchors_per_task, true);
    {   // parallelization.cpp:107-110 -> statistics.cpp:269-285 (seeds), :367-419 (search results)
        run_statistics& st = rr.st;
        st.values[0].push_back(len);
        st.values[3].push_back(fwd_seeds.size() + rev_seeds.size());
        for (auto const* seeds : {&fwd_seeds, &rev_seeds})
            for (auto const& sd : *seeds) { st.values[2].push_back(sd.num_errors); st.values[1].push_back(sd.len); }
        uint64_t fully_excluded = 0, kept = 0, by_soft = 0, by_erase = 0;
        bool all_excluded = true;
        for (auto const* res : {&fwd_res, &rev_res})
            for (auto const& a : *res) {
                if (a.num_kept_useful_anchors == 0) { ++fully_excluded; continue; }
                all_excluded = false;
                kept += a.num_kept_useful_anchors;
                st.values[8].push_back(a.num_kept_useful_anchors);
                by_soft += a.num_excluded_raw_anchors_by_soft_cap;
                st.values[9].push_back(a.num_excluded_raw_anchors_by_soft_cap);
                uint64_t const erased = a.num_kept_raw_anchors - a.num_kept_useful_anchors;
                by_erase += erased;
                st.values[10].push_back(erased);
            }
        st.values[4].push_back(fully_excluded);
        st.values[5].push_back(kept);
        st.values[6].push_back(by_soft);
        st.values[7].push_back(by_erase);
        if (all_excluded) ++st.completely_excluded_queries;
    }

    size_t const nref = refs.size();
    std::vector<verified_intervals> iv_fwd(nref), iv_rev(nref);
    for (auto& v : iv_fwd) v.active = p.interval_optimization;
    for (auto& v : iv_rev) v.active = p.interval_optimization;
    std::vector<std::vector<query_alignment>> all(nref);
    for (int pid : single_thread_task_order((int)packages.size())) {                           // parallelization.cpp:230-260
        auto const& pkg = packages[pid];
        std::vector<std::vector<query_alignment>> mine(nref);
        for (auto const& a : pkg.anchors) {
            auto& ivs = (pkg.reverse ? iv_rev : iv_fwd)[a.reference_id];
            verify_anchor(tree, a, pkg.reverse ? rc.data() : read.data(), pkg.reverse, refs[a.reference_id].data(),
                          refs[a.reference_id].size(), p, ivs, mine[a.reference_id], &rr.vc, &rr.st);
        }
        for (size_t r = 0; r < nref; ++r)
            for (auto& al : mine[r]) all[r].push_back(std::move(al));
    }
    // output.cpp:49-108
    bool have_best = false;
    uint64_t best = 0;
    for (auto const& v : all) for (auto const& al : v) if (!have_best || al.num_errors < best) { best = al.num_errors; have_best = true; }
    bool primary_written = false;
    for (size_t r = 0; r < nref; ++r)
        for (auto const& al : all[r]) {
            uint32_t flag = al.reverse ? 16u : 0u;
            bool const primary = !primary_written && best == al.num_errors;
            if (primary) primary_written = true;
            else flag |= 256u;
            record rec{read_index, flag, (int64_t)r, saturate_value_to_int32_max(al.start_in_reference), (uint32_t)al.num_errors,
                       rr.cigars.size(), al.cigar.size()};
            rr.cigars.insert(rr.cigars.end(), al.cigar.begin(), al.cigar.end());
            rr.records.push_back(rec);
        }
    if (!primary_written) rr.records.push_back(record{read_index, 4u, -1, 0, 0, 0, 0});
    {   // parallelization.cpp:262-269
        uint64_t n_al = 0;
        for (auto const& v : all) { n_al += v.size(); for (auto const& al : v) rr.st.values[15].push_back(al.num_errors); }
        rr.st.values[14].push_back(n_al);
    }
    return rr;
}

}  // namespace

run_output align_reads(const fm_index& idx, const std::vector<std::vector<uint8_t>>& refs,
                       const std::vector<std::vector<uint8_t>>& reads, const params& p, unsigned threads) {
    std::vector<read_result> per_read(reads.size());
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (size_t i; (i = next.fetch_add(1)) < reads.size();) per_read[i] = align_one_read(idx, refs, reads[i], i, p);
    };
    if (threads <= 1) worker();
    else {
        std::vector<std::thread> ts;
        for (unsigned t = 0; t < threads; ++t) ts.emplace_back(worker);
        for (auto& t : ts) t.join();
    }
    run_output out;
    out.skipped.resize(reads.size());
    for (size_t i = 0; i < reads.size(); ++i) {
        auto& rr = per_read[i];
        out.skipped[i] = rr.skipped;
        uint64_t const base = out.cigars.size();
        out.cigars.insert(out.cigars.end(), rr.cigars.begin(), rr.cigars.end());
        for (auto rec : rr.records) { rec.cigar_off += base; out.records.push_back(rec); }
        out.sc.n_extend_all += rr.sc.n_extend_all; out.sc.n_extend_one += rr.sc.n_extend_one;
        out.sc.n_locates += rr.sc.n_locates; out.sc.n_lf_steps += rr.sc.n_lf_steps;
        out.vc.inner_jobs += rr.vc.inner_jobs; out.vc.root_jobs += rr.vc.root_jobs;
        out.vc.inner_word_steps += rr.vc.inner_word_steps; out.vc.root_word_steps += rr.vc.root_word_steps;
        out.vc.ref_query_bytes += rr.vc.ref_query_bytes;
        out.st.merge(rr.st);
    }
    return out;
}

}  // namespace orc
