// K1 (approximate FM search, search.cpp:173-188 -> search_ng21::search_n) as per-lane state machines. One lane serves one seed
// (fm_step) or one single-row subtree of a seed (tx_step); the wave-level parts (seed hand-out, slot reservation for the
// output) live in flx_search.hip. Everything here is plain integer code over plain pointers, compiled for the device by hipcc
// and for the host by the CPU check of the test suite (tests/fm_core_check.cpp), which runs the same code seed by seed against
// the oracle. The product path has no host use of it.
//
// What the walk computes is the reference's: the DFS of search_ng21 over the expanded optimum scheme with edit operations, every
// hit a cursor {lb, len} with its error count. How it gets there differs in three ways, none of which changes the set of hits:
//
//  (1) error children first, match child last (a tail call): at most `errors` frames alive, the stack lives in LDS; every hit
//      carries a key that restores the reference's emission order (round 2).
//  (2) presence filter. A child that has spent the last error its scheme positions allow can only continue by exact matches
//      ("forced" positions). Its string after those positions is then known in advance: a piece of the seed with one edit at the
//      junction. If that string does not occur in the text the child's interval runs empty before the forced run ends and the
//      child reports nothing, so it need not be walked. Whether a string of K symbols occurs is one bit of a table with 4^K
//      bits (index->filter, built from the text when the context is made); strings of K-3 .. K-1 symbols are looked up as the
//      4 / 16 / 64 neighbouring bits of their left extensions. On a 3.1 Gb text two thirds of all cursor extensions of the
//      reference walk are such chains that die (measured with class counters, profiles/r03_k1_classes.txt); each costs one
//      64-bit load here instead of 2-7 rank pairs. The filter has no false negatives (a string that occurs always has its bit
//      set), false positives just walk on as before.
//  (3) text mode. Once an interval has one row the remaining DFS below it is a comparison of seed symbols with text symbols at a
//      known place: SA[row] gives the place, every child interval has one row or none, and the final lb of a hit is ISA[start of
//      its string]. Such subtrees are queued (fm_step emits an item) and walked by tx_step against the text itself: no rank
//      query at all, one SA read per subtree and one ISA read per hit. 30 % of the reference walk's extensions on the same text.
#pragma once

#include "flx_internal.hpp"

namespace flx {

// ------------------------------------------------------------------------------------------------ small helpers
FLX_HD inline u32 fm_popc(u32 v) { return (u32)__builtin_popcount(v); }

// ------------------------------------------------------------------------------------------------ rank queries (OccBlock)
struct alignas(16) FmU4 { u32 x, y, z, w; };

// r[c] = number of symbol c in bwt[0, pos) for c = 0..4
FLX_HD inline void fm_rank5(const OccBlock* __restrict__ tab, u32 pos, u32 r[5]) {
    const FmU4* __restrict__ q = reinterpret_cast<const FmU4*>(tab + (pos >> 5));
    FmU4 const a = q[0], b = q[1];
    u32 const mask = (1u << (pos & 31u)) - 1u;
    u32 const p0 = b.y, p1 = b.z, p2 = b.w;
    u32 const n2 = ~p2 & mask;
    r[0] = a.x + fm_popc(n2 & ~(p1 | p0));
    r[1] = a.y + fm_popc(n2 & ~p1 & p0);
    r[2] = a.z + fm_popc(n2 & p1 & ~p0);
    r[3] = a.w + fm_popc(n2 & p1 & p0);
    r[4] = b.x + fm_popc(p2 & mask & ~(p1 | p0));
}

// both ends of the interval [lo, lo + nlen): cl[c] = rows of the child of symbol c (c = 0..5), ab[c] = its lower bound on the
// extended side (symbol 0, the sequence delimiter, is only ever a match child: a read holding the character '$', input.cpp:165-176)
FLX_HD inline void fm_extend_all(const DevIndex& idx, const OccBlock* __restrict__ tab, u32 lo, u32 nlen, u32 ab[6], u32 cl[6]) {
    u32 ra[5], rb[5];
    fm_rank5(tab, lo, ra);
    fm_rank5(tab, lo + nlen, rb);
    u32 sum_a = 0, sum_l = 0;
#pragma unroll
    for (u32 c = 0; c < 5; ++c) { cl[c] = rb[c] - ra[c]; sum_a += ra[c]; sum_l += cl[c]; }
    cl[5] = nlen - sum_l;
    ab[0] = ra[0];                                                    // C[0] = 0
#pragma unroll
    for (u32 c = 1; c < 5; ++c) ab[c] = idx.C[c] + ra[c];
    ab[5] = idx.C[5] + (lo - sum_a);
}

// ------------------------------------------------------------------------------------------------ scheme entries
// low word: sch_pack (flx_internal.hpp); high word: bits 0..14 end of the run of entries around this one that share its upper
// bound and its direction (first entry behind the run), bits 15..28 lowest seed position among the entries before this one
FLX_HD inline u32 sch_run_end(u64 e) { return (u32)(e >> 32) & 0x7FFFu; }
FLX_HD inline u32 sch_lo(u64 e) { return (u32)(e >> 47) & 0x3FFFu; }
FLX_HD inline u32 sch_lower(u32 s) { return (s >> 20) & 7u; }
FLX_HD inline u32 sch_upper(u32 s) { return (s >> 23) & 7u; }
FLX_HD inline u32 sch_right(u32 s) { return (s >> 26) & 1u; }

// ------------------------------------------------------------------------------------------------ 2-bit packed sequence pool
// symbol i of the pool at bits 2 (i % 16) of word (i + PACK_FRONT) / 16: A, C, G, T -> 0..3 (anything else packs as (rank - 1) & 3;
// seeds of reads that hold such symbols do not use the filter)
constexpr u32 PACK_FRONT = 32;
FLX_HD inline u64 pack_words_for(u64 pool_len) { return (pool_len + PACK_FRONT) / 16 + 4; }
// the 32 symbols starting at pool position g (g >= -PACK_FRONT)
FLX_HD inline u64 pack_extract(const u32* __restrict__ qpack, i64 g) {
    u64 const at = (u64)(g + (i64)PACK_FRONT);
    const u32* __restrict__ w = qpack + (at >> 4);
    u32 const sh = (u32)(at & 15u) * 2u;
    u64 const lo = (u64)w[0] | ((u64)w[1] << 32);
    return sh ? (lo >> sh) | ((u64)w[2] << (64u - sh)) : lo;
}
FLX_HD inline u64 low_syms(u64 v, u32 n) { return n >= 32u ? v : v & ((1ull << (2u * n)) - 1ull); }
// word w of the packed form of seq[0, len)
FLX_HD inline u32 pack_word(const u8* __restrict__ seq, u64 len, u64 w) {
    i64 const first = (i64)w * 16 - (i64)PACK_FRONT;
    u32 v = 0;
#pragma unroll
    for (u32 j = 0; j < 16; ++j) {
        i64 const p = first + (i64)j;
        u32 const c = (p >= 0 && (u64)p < len) ? seq[p] : 1u;
        v |= ((c - 1u) & 3u) << (2u * j);
    }
    return v;
}

// ------------------------------------------------------------------------------------------------ presence filter
// bit (sum of (symbol_i - 1) * 4^i, i = 0 the leftmost symbol) of the table is set when the K-symbol string occurs in the text.
// A string S of T < K symbols is asked for as "some left extension of S occurs": the 4^(K-T) consecutive bits from code(S) << 2(K-T).
// For that to hold for occurrences whose left neighbours are not A, C, G, T (sequence starts, delimiters, N) the builder sets
// all left extensions of such an occurrence (filter_add below).
struct FilterQuery { u64 word; u64 mask; };
FLX_HD inline FilterQuery filter_query(u32 K, u64 code, u32 T) {        // code: T symbols, K - 3 <= T <= K
    u32 const u = K - T;
    u64 const start = code << (2u * u);
    u32 const cnt = 1u << (2u * u);                                    // 1, 4, 16, 64 bits
    u64 const m = cnt >= 64u ? ~0ull : (((1ull << cnt) - 1ull) << (start & 63ull));
    return FilterQuery{start >> 6, m};
}
constexpr u32 FILTER_MAX_K = 19, FILTER_MIN_K = 8;
FLX_HD inline u64 filter_words(u32 K) { return ((1ull << (2u * K)) + 63ull) / 64ull; }
// K for a text of n symbols: two symbols more than the text needs to tell its positions apart (a K-mer drawn at random is present
// with probability n / 4^K <= 1/16)
inline u32 filter_k_default(u64 n) {
    u32 bits = 0;
    while ((1ull << bits) < n && bits < 63) ++bits;
    u32 const k = (bits + 1) / 2 + 2;
    return k < FILTER_MIN_K ? FILTER_MIN_K : k > FILTER_MAX_K ? FILTER_MAX_K : k;
}
// shortest string worth asking for: one that a random text of this length holds with probability <= 1/2, and at most 3 short of K
inline u32 filter_tmin_for(u64 n, u32 k) {
    u32 t = 1;
    while (t < k && (1ull << (2 * t)) < 2 * n) ++t;
    u32 const least = k >= 3 ? k - 3 : 1u;
    return t > least ? t : least;
}

// The builder's unit of work: the windows that end at text positions [q0, q1). run = length of the run of A/C/G/T that ends at the
// current position (capped at K), code = the last K symbols. set(word, mask) ors into the table.
// set_m(word, mask) ors into the mirrored table (rightmost symbol lowest: full windows only), see FmFilterKind::mirrored
template <class SET, class SETM>
FLX_HD inline void filter_add_range(const u8* __restrict__ text, i64 n, i64 q0, i64 q1, u32 K, u32 tmin, SET&& set, SETM&& set_m) {
    u64 code = 0, code_m = 0;
    u32 run = 0;
    u64 const kmask = K >= 32u ? ~0ull : ((1ull << (2u * K)) - 1ull);
    for (i64 q = q0 - (i64)K + 1; q < q1; ++q) {
        u32 const c = (q >= 0 && q < n) ? text[q] : 0u;
        if (c >= 1u && c <= 4u) {
            code = ((code >> 2) | ((u64)(c - 1u) << (2u * (K - 1u)))) & kmask;
            code_m = ((code_m << 2) | (u64)(c - 1u)) & kmask;
            run = run < K ? run + 1u : K;
        } else { run = 0; code = 0; code_m = 0; }
        if (q < q0) continue;
        if (run >= K) { set(code >> 6, 1ull << (code & 63ull)); set_m(code_m >> 6, 1ull << (code_m & 63ull)); }
        else if (run >= tmin && K - run <= 3u) {
            // the run's symbols are the top `run` symbols of code; every left extension of them counts as present
            u32 const u = K - run;
            u64 const start = (code >> (2u * u)) << (2u * u);
            u32 const cnt = 1u << (2u * u);
            set(start >> 6, cnt >= 64u ? ~0ull : (((1ull << cnt) - 1ull) << (start & 63ull)));
        }
    }
}

template <class SET>
FLX_HD inline void filter_add_range(const u8* __restrict__ text, i64 n, i64 q0, i64 q1, u32 K, u32 tmin, SET&& set) {
    filter_add_range(text, n, q0, q1, K, tmin, set, [](u64, u64) {});
}
// the first cnt 2-bit symbols of v in reverse order
FLX_HD inline u64 rev_syms(u64 v, u32 cnt) {
    v = ((v >> 2) & 0x3333333333333333ull) | ((v & 0x3333333333333333ull) << 2);
    v = ((v >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((v & 0x0F0F0F0F0F0F0F0Full) << 4);
    v = __builtin_bswap64(v);
    return cnt >= 32u ? v : cnt == 0u ? 0ull : v >> (2u * (32u - cnt));
}

// ------------------------------------------------------------------------------------------------ state words, keys
// frame state word: x:14 | e:3 | linfo:2 | rinfo:2 | next_sym:3 | right:1 | dlen+4:3  (dlen = deletions - insertions so far)
enum : u32 { FM_INFO_M = 0, FM_INFO_I = 1, FM_INFO_D = 2, FM_INFO_S = 3 };
FLX_HD inline u32 fst_pack(u32 x, u32 e, u32 li, u32 ri, u32 sym, u32 right, u32 dl) {
    return x | (e << 14) | (li << 17) | (ri << 19) | (sym << 21) | (right << 24) | (dl << 25);
}
FLX_HD inline u32 fst_x(u32 s) { return s & 0x3FFFu; }
FLX_HD inline u32 fst_e(u32 s) { return (s >> 14) & 7u; }
FLX_HD inline u32 fst_li(u32 s) { return (s >> 17) & 3u; }
FLX_HD inline u32 fst_ri(u32 s) { return (s >> 19) & 3u; }
FLX_HD inline u32 fst_sym(u32 s) { return (s >> 21) & 7u; }
FLX_HD inline u32 fst_right(u32 s) { return (s >> 24) & 1u; }
FLX_HD inline u32 fst_dl(u32 s) { return (s >> 25) & 7u; }
// a queued subtree (DevHit used as the record: seed = launch position of the seed, lb = the row, len = this word, key = the node's key)
FLX_HD inline u32 item_pack(u32 x, u32 e, u32 li, u32 ri, u32 srch, u32 dl) { return x | (e << 14) | (li << 17) | (ri << 19) | (srch << 21) | (dl << 25); }
FLX_HD inline u32 item_srch(u32 s) { return (s >> 21) & 7u; }

constexpr u32 FMK_BITS = 18;                // per error edge: (0x3FFF - x) << 4 | child index
constexpr u32 FMK_MAX_X = 0x3FFFu;
FLX_HD inline u64 fm_key_edge(u64 pkey, u32 px, u32 pe, u32 ci) {
    return pkey | ((u64)(((FMK_MAX_X - px) << 4) | ci) << (FMK_BITS * (2u - pe)));
}

// the children of a branching node that exist: bit 0 match, bits 2c-1 / 2c deletion / substitution of symbol c, bit 11 insertion
FLX_HD inline u32 fm_child_mask6(const u32 cl[6], u32 next_sym, bool match_allowed, bool deletion, bool insertion) {
    u32 mask = 0;
#pragma unroll
    for (u32 c = 1; c < 6; ++c) {
        if (cl[c] > 0u) {
            if (deletion) mask |= 1u << (2u * c - 1u);
            if (c != next_sym) mask |= 1u << (2u * c);
            else if (match_allowed) mask |= 1u;
        }
    }
    if (next_sym == 0u && match_allowed && cl[0] > 0u) mask |= 1u;      // a '$' of the query matches a sequence delimiter
    if (insertion) mask |= 1u << 11;
    return mask;
}

// ------------------------------------------------------------------------------------------------ the lanes
enum : u32 { FM_OUT_NONE = 0, FM_OUT_HIT = 1, FM_OUT_ITEM = 2 };
enum : u32 { SEED_HAS_DELIM = 1, SEED_NOT_ACGT = 2 };      // DevSeed::flags: the seed's read holds a symbol 0 / a symbol outside 1..4

struct FmConst {                            // the same for every lane of a launch
    DevIndex idx;
    const u8* seq;                          // sequence pool, one byte per symbol
    const u32* qpack;                       // its 2-bit form (null: no filter)
    const u64* scheme;
    const DevSeed* seeds;                   // the launch's seed records (a lane keeps a seed's launch position and reads what it needs rarely from here)
    u32 max_hits;
    u32 levels;                             // frames a lane may hold
    u32 text_min_remain;                    // a one-row node is queued when at least this many scheme entries remain (0: never)
    u32 use_filter;                         // 0: no filter; 1: one look per child; 2: a second, shifted look at the children that pass
};

constexpr u32 FM_FRAME_WORDS = 18;          // oth[1..5], end, abs[1..5], lb, lb_rev, state, mask, key lo, key hi, abs[0]
constexpr u32 TX_FRAME_WORDS = 6;           // state, mask, pL, pR, key lo, key hi

// A lane's registers are what bounds the number of resident search waves (and what the compiler spilled to scratch memory in round 3:
// 72 B per lane, written and re-read through HBM in every DFS step), so the state is packed:
//   ws (the seed and the search): len:14 | flags:2 @14 | searches:3 @16 | search:3 @19 | l_last:3 @22 | u_last:3 @25 | stolen:1 @28
//   wn (the node): x:14 | e:3 @14 | linfo:2 @17 | rinfo:2 @19 | depth:3 @21 | need_child:1 @24 | dlen+4:3 @25 | out:2 @28 | busy:1 @30 |
//      in_search:1 @31  - bits 0..20 and 25..27 are where a frame's state word and a queued subtree's word have the same fields
// What a step produces (out = FM_OUT_HIT / FM_OUT_ITEM) is read from the node's own registers by the caller before the next step:
// hit = {nlb, nlen (rows, capped), e, nkey}; item = {nlb, item word, nkey}.
struct FmLane {
    u32 pos = 0;                            // launch position of the seed (C.seeds[pos])
    u64 qoff = 0;
    u32 exo = 0;                            // first scheme entry of the current search (C.scheme + exo)
    u32 ws = 0, wn = 0;
    u32 ct = 0;
    u32 nlb = 0, nlbr = 0, nlen = 0;
    u64 nkey = 0;
    u32 n_ext = 0, n_lookup = 0, n_pruned = 0, n_prefix_kills = 0;     // (the last two only count in the STATS form of fm_step)

    FLX_HD u32 len() const { return ws & 0x3FFFu; }
    FLX_HD u32 flags() const { return (ws >> 14) & 3u; }
    FLX_HD u32 num_searches() const { return (ws >> 16) & 7u; }
    FLX_HD u32 srch() const { return (ws >> 19) & 7u; }
    FLX_HD u32 l_last() const { return (ws >> 22) & 7u; }
    FLX_HD u32 u_last() const { return (ws >> 25) & 7u; }
    FLX_HD bool stolen() const { return (ws >> 28) & 1u; }
    FLX_HD u32 nx() const { return wn & 0x3FFFu; }
    FLX_HD u32 ne() const { return (wn >> 14) & 7u; }
    FLX_HD u32 nli() const { return (wn >> 17) & 3u; }
    FLX_HD u32 nri() const { return (wn >> 19) & 3u; }
    FLX_HD u32 depth() const { return (wn >> 21) & 7u; }
    FLX_HD bool need_child() const { return (wn >> 24) & 1u; }
    FLX_HD u32 ndl() const { return (wn >> 25) & 7u; }
    FLX_HD u32 out() const { return (wn >> 28) & 3u; }
    FLX_HD bool busy() const { return (wn >> 30) & 1u; }
    FLX_HD bool in_search() const { return wn >> 31; }
    FLX_HD bool overflow() const { return out() == 3u; }                // (frames ran out: the lane stops, the launch is repeated with the ordered kernel)
    FLX_HD void clear_out() { wn &= ~(3u << 28); }
    FLX_HD u32 item_word() const { return (wn & 0x0E1FFFFFu) | (srch() << 21); }    // = item_pack(x, e, li, ri, search, dlen)
    FLX_HD const u64* ex(FmConst const& C) const { return C.scheme + exo; }
};
constexpr u32 WN_NODE_MASK = 0x0E1FFFFFu;                              // x, e, linfo, rinfo, dlen
constexpr u32 WN_DEPTH1 = 1u << 21, WN_NEED_CHILD = 1u << 24, WN_BUSY = 1u << 30, WN_IN_SEARCH = 1u << 31, WN_OUT_SHIFT = 28;
FLX_HD inline u32 wn_node(u32 x, u32 e, u32 li, u32 ri, u32 dl) { return x | (e << 14) | (li << 17) | (ri << 19) | (dl << 25); }

FLX_HD inline void fm_take_seed(FmConst const& C, FmLane& L, DevSeed const& seed, u32 pos) {
    L.pos = pos;
    L.qoff = seed.seq_off;
    L.exo = seed.scheme_off;
    L.ws = seed.length | ((seed.flags & 3u) << 14) | ((seed.frames_searches >> 24) << 16);
    L.ct = 0;
    L.wn = WN_BUSY;
}

// what the filter needs from memory besides the table itself, asked for before the rank queries of the node so that both are in
// flight together: the scheme entry behind the node's and the seed's symbols on both sides of the junction (32 each)
struct FmFilterPre { u64 e1, xs, xf; };
FLX_HD inline FmFilterPre fm_filter_prefetch(FmConst const& C, FmLane const& L, u32 x, u64 e64, u32 right) {
    FmFilterPre P;
    P.e1 = x + 1u < L.len() ? L.ex(C)[x + 1u] : 0ull;
    u32 const a = sch_lo(e64);
    i64 const g0 = (i64)L.qoff;
    //   right: xs = seed[b-31 .. b] with b = a + x - 1 the string's last position, xf = seed[b+1 ...]
    //   left:  xs = seed[a ...], xf = seed[a-32 .. a-1]
    P.xs = pack_extract(C.qpack, right ? g0 + (i64)(a + x) - 32 : g0 + (i64)a);
    P.xf = pack_extract(C.qpack, right ? g0 + (i64)(a + x) : g0 + (i64)a - 32);
    return P;
}
// one kind of child (0 substitution, 1 deletion, 2 insertion): the code of its string without the junction symbol, where that symbol
// goes, how far the code is shifted up (unknown symbols in front of a string shorter than K), and the symbols to ask for
// mirrored: the lookup goes to the mirrored table (rightmost symbol of the window lowest). The four substitution (or deletion) children of a
// node differ in the junction symbol only; when that symbol sits in the code's low six bits they share one 64-bit word = one memory request.
// In the plain code that is so for rightward extension in the narrow look (two string symbols, the junction, the forced run); for leftward
// extension the same window has the junction near the top of the plain code (nine lines per node: two thirds of the filter walk's requests
// in round 3's counters) and near the bottom of the mirrored one.
struct FmFilterKind { u64 base; u32 csh, ush, want; bool mirrored; };
// wide: the window with as many symbols of the string as fit (second look at a child that passed the first, whose window holds as
// many forced symbols as fit: the two share about half of their symbols)
FLX_HD inline FmFilterKind fm_filter_kind(u32 kind, u32 K, u32 tmin, u32 x, u32 right, u32 F, u32 want, u64 xs, u64 xf, bool wide, bool have_mirrored) {
    FmFilterKind Q{0ull, 0u, 0u, 0u, false};
    u32 const jn = kind == 2u ? 0u : 1u;                             // a junction symbol c (the insertion has none: the position is skipped)
    u32 h, r;
    if (!wide) {
        u32 const keep = x < 2u ? x : 2u;                           // string symbols a window always holds
        h = F < K - jn - keep ? F : K - jn - keep;
        r = x < K - jn - h ? x : K - jn - h;
    } else {
        r = x < K - jn - 1u ? x : K - jn - 1u;
        h = F < K - jn - r ? F : K - jn - r;
    }
    u32 const t = h + jn + r;
    if (h < 1u || t < tmin || K - t > 3u || (kind == 2u && r < 1u)) return Q;
    u32 const skip = kind == 1u ? 0u : 2u;                           // the forced symbols of a deletion child start at the position itself
    u64 str, forced;
    if (right) { str = r ? xs >> (2u * (32u - r)) : 0ull; forced = low_syms(xf >> skip, h); }
    else { str = low_syms(xs, r); forced = (xf << skip) >> (2u * (32u - h)); }
    // codes: leftmost symbol lowest. right: string | c | forced; left: forced | c | string
    Q.ush = 2u * (K - t);
    Q.want = want;
    if (!right && have_mirrored && jn && t == K) {
        // window left to right: forced (h) | c | string (r); mirrored code: the string's last symbol lowest, c at symbol r
        Q.base = rev_syms(str, r) | (rev_syms(forced, h) << (2u * (r + 1u)));
        Q.csh = 2u * r;
        Q.mirrored = true;
        return Q;
    }
    Q.base = right ? str | (forced << (2u * (r + jn))) : forced | (str << (2u * (h + jn)));
    Q.csh = 2u * (right ? r : h);
    return Q;
}
// The presence filter at a branching node without errors so far: the node's string is seed[a, a + x). Children that have no error
// left at the positions that follow are dropped from the mask when the string they are bound to reach does not occur in the text.
// Two looks: one window of K symbols around the junction per child, then a second, shifted window for the children that passed
// (a K-mer drawn at random is present with probability n / 4^K, 4.5 % at hg38 size: two windows leave 0.2 % of the wrong children).
template <bool STATS>
FLX_HD inline u32 fm_filter_children(FmConst const& C, FmLane& L, u32 x, u64 e64, u32 right, u32 mask, FmFilterPre const& P) {
    u32 const K = C.idx.filter_k, tmin = C.idx.filter_tmin;
    // forced positions behind the children: F1 for the children that move on to entry x + 1 (substitution, insertion), F0 for
    // the deletion children, which stay at entry x
    u32 F1 = 0, F0 = 0;
    if (x + 1u < L.len() && sch_upper((u32)P.e1) == 1u && sch_right((u32)P.e1) == right) F1 = sch_run_end(P.e1) - (x + 1u);
    if (sch_upper((u32)e64) == 1u) F0 = sch_run_end(e64) - x;
    if ((F1 | F0) == 0u) return mask;
    const u64* __restrict__ bits = C.idx.filter;
    bool const have_m = C.idx.filter_m != nullptr;
    // symbols asked for per kind, bit 2(c-1): symbol c
    u32 want_s = (mask >> 2) & 0x55u, want_d = (mask >> 1) & 0x55u, want_i = (mask >> 11) & 1u;
#pragma unroll 1
    for (u32 look = 0; look < C.use_filter; ++look) {
        FmFilterKind const qs = fm_filter_kind(0u, K, tmin, x, right, F1, want_s, P.xs, P.xf, look != 0u, have_m);
        FmFilterKind const qd = fm_filter_kind(1u, K, tmin, x, right, F0, want_d, P.xs, P.xf, look != 0u, have_m);
        FmFilterKind const qi = fm_filter_kind(2u, K, tmin, x, right, F1, want_i, P.xs, P.xf, look != 0u, false);
        const u64* __restrict__ bits_s = qs.mirrored ? C.idx.filter_m : bits;
        const u64* __restrict__ bits_d = qd.mirrored ? C.idx.filter_m : bits;
        if ((qs.want | qd.want | qi.want) == 0u) break;
        // a lookup is the 64-bit word that holds the string's bit, or the 4 / 16 / 64 bits of its left extensions; all of them are
        // asked for before the first is looked at
        u64 ws[4], wd[4];
#pragma unroll
        for (u32 c = 0; c < 4u; ++c) {
            ws[c] = ((qs.want >> (2u * c)) & 1u) ? bits_s[((qs.base | ((u64)c << qs.csh)) << qs.ush) >> 6] : ~0ull;
            wd[c] = ((qd.want >> (2u * c)) & 1u) ? bits_d[((qd.base | ((u64)c << qd.csh)) << qd.ush) >> 6] : ~0ull;
        }
        u64 const wi = qi.want ? bits[(qi.base << qi.ush) >> 6] : ~0ull;
        u64 const span_s = qs.ush >= 6u ? ~0ull : (1ull << (1u << qs.ush)) - 1ull;
        u64 const span_d = qd.ush >= 6u ? ~0ull : (1ull << (1u << qd.ush)) - 1ull;
        u64 const span_i = qi.ush >= 6u ? ~0ull : (1ull << (1u << qi.ush)) - 1ull;
        u32 drop_s = 0, drop_d = 0, drop_i = 0;
#pragma unroll
        for (u32 c = 0; c < 4u; ++c) {
            u32 const at_s = (u32)((qs.base | ((u64)c << qs.csh)) << qs.ush) & 63u;
            u32 const at_d = (u32)((qd.base | ((u64)c << qd.csh)) << qd.ush) & 63u;
            if (((ws[c] >> at_s) & span_s) == 0ull) drop_s |= 1u << (2u * c);
            if (((wd[c] >> at_d) & span_d) == 0ull) drop_d |= 1u << (2u * c);
        }
        if (((wi >> ((u32)(qi.base << qi.ush) & 63u)) & span_i) == 0ull) drop_i = 1u;
        // (counted in 64-bit words asked for: the children of one kind share a word when the junction symbol is within its low six bits)
        L.n_lookup += (qs.want ? (qs.csh + qs.ush <= 4u ? 1u : fm_popc(qs.want)) : 0u) + (qd.want ? (qd.csh + qd.ush <= 4u ? 1u : fm_popc(qd.want)) : 0u) + qi.want;
        if (STATS) L.n_pruned += fm_popc(drop_s) + fm_popc(drop_d) + drop_i;
        mask &= ~((drop_s << 2) | (drop_d << 1) | (drop_i << 11));
        // the second look: only the children the first one was asked about and let pass
        want_s = qs.want & ~drop_s; want_d = qd.want & ~drop_d; want_i = qi.want & ~drop_i;
        if ((want_s | want_d | want_i) == 0u) break;
    }
    return mask;
}

// start of the current search of the seed: the root cursor, or the cursor of the seed's first KMER_Q symbols when the search begins
// with an exact, rightward part that long and free of N. false: the search finds nothing. Sets nlb, nlbr, nlen; returns x in *x0.
template <bool STATS>
FLX_HD inline bool fm_begin_search(FmConst const& C, FmLane& L, u32* x0) {
    DevIndex const& idx = C.idx;
    L.nlb = 0; L.nlbr = 0; L.nlen = idx.n; *x0 = 0;
    const u64* __restrict__ ex = L.ex(C);
    u32 const len = L.len();
    u64 const e0 = ex[0];
    if (len >= KMER_Q && (((u32)ex[KMER_Q - 1] >> 27) & 1u)) {
        u32 const p0 = (u32)e0 & SCH_POS_MASK;
        // presence of the exact prefix (its first K symbols): most searches of a read with errors end here
        if (C.use_filter && !(L.flags() & SEED_NOT_ACGT)) {
            u32 const run = sch_run_end(e0);                            // entries 0 .. run-1: exact, rightward, consecutive positions
            u32 const K = idx.filter_k;
            u32 const T = run < K ? run : K;
            if (T >= idx.filter_tmin && K - T <= 3u) {
                u64 const code = low_syms(pack_extract(C.qpack, (i64)L.qoff + (i64)p0), T);
                FilterQuery const fq = filter_query(K, code, T);
                ++L.n_lookup;
                if ((idx.filter[fq.word] & fq.mask) == 0ull) { if (STATS) ++L.n_prefix_kills; return false; }
            }
        }
        u32 w[2];
        __builtin_memcpy(w, C.seq + L.qoff + p0, 8);                     // eight ranks, first character in the low byte
        u32 const t0 = w[0] - 0x01010101u, t1 = w[1] - 0x01010101u;      // A,C,G,T -> 0..3; anything else leaves bits 2..7 set
        if (((t0 | t1) & 0xFCFCFCFCu) == 0u) {
            // gather the four 2-bit fields of a word, first character most significant: b0<<6 | b1<<4 | b2<<2 | b3
            u32 const code = (((t0 * 0x40100401u) >> 24) << 8) | ((t1 * 0x40100401u) >> 24);
            const u32* __restrict__ e = idx.kmer + 3u * code;
            L.nlb = e[0]; L.nlbr = e[1]; L.nlen = e[2];
            *x0 = KMER_Q;
            if (L.nlen == 0) return false;
        }
    }
    return true;
}

// the search after the current one (or the end of the seed)
FLX_HD inline void fm_next_search(FmLane& L) {
    u32 const s = L.srch() + 1u;
    if (s >= L.num_searches() || L.stolen()) { L.wn = 0; return; }      // (a lane that works on children taken from another lane ends with them)
    L.ws += 1u << 19;
    L.exo += L.len();
    L.wn = WN_BUSY;
}

// One DFS step of a busy lane. FR: u32& fr(level, word), the lane's frames.
template <bool STATS, class FR>
FLX_HD inline void fm_step(FmConst const& C, FmLane& L, FR&& fr) {
    DevIndex const& idx = C.idx;
    u32 const len = L.len();
    const u64* __restrict__ ex = L.ex(C);
    if (!L.in_search()) {
        if (L.srch() >= L.num_searches()) { L.wn = 0; return; }
        u32 const last_entry = (u32)ex[len - 1];
        L.ws = (L.ws & ~(63u << 22)) | (sch_lower(last_entry) << 22) | (sch_upper(last_entry) << 25);
        L.nkey = (u64)L.srch() << (3u * FMK_BITS);
        u32 x0;
        if (!fm_begin_search<STATS>(C, L, &x0)) { fm_next_search(L); return; }
        L.wn = WN_BUSY | WN_IN_SEARCH | wn_node(x0, 0u, FM_INFO_M, FM_INFO_M, 4u);
    }

    // ---- the next child of the top frame becomes the node: children that cost an error first, the match child last
    if (L.need_child()) {
        u32 const depth = L.depth();
        if (depth == 0u) { fm_next_search(L); return; }                      // search exhausted
        u32 const lv = depth - 1u;
        u32 const mask = fr(lv, 14);
        u32 const st = fr(lv, 13);
        u32 const costly = mask & ~1u;
        u32 const ci = costly ? (u32)__builtin_ctz(costly) : 0u;
        u32 const rest = mask & ~(1u << ci);
        u32 wn = (L.wn & ~(WN_NODE_MASK | WN_NEED_CHILD));                   // depth, busy, in_search stay
        if (rest) fr(lv, 14) = rest;
        else wn -= WN_DEPTH1;                                                // the last child of a frame is a tail call: the frame is gone
        u32 const right = fst_right(st);
        u32 const px = fst_x(st), pe = fst_e(st);
        u32 info, sym, dl = fst_dl(st), cx, ce;
        if (ci == 0) { sym = fst_sym(st); cx = px + 1; ce = pe; info = FM_INFO_M; }
        else if (ci == 11) { sym = 1; cx = px + 1; ce = pe + 1; info = FM_INFO_I; dl -= 1u; }
        else {
            sym = (ci + 1) >> 1;
            bool const del = ci & 1u;
            cx = del ? px : px + 1;
            ce = pe + 1;
            info = del ? FM_INFO_D : FM_INFO_S;
            dl += del ? 1u : 0u;
        }
        // sym is 1..5 for every child but the match of a '$' (symbol 0: its cursor starts where the node's does)
        u32 const p_lb = fr(lv, 11), p_lbr = fr(lv, 12);
        u32 const c_oth = sym ? fr(lv, sym - 1u) : (right ? p_lb : p_lbr), c_end = fr(lv, sym), c_abs = fr(lv, sym ? 5u + sym : 17u);
        u64 const pkey = (u64)fr(lv, 15) | ((u64)fr(lv, 16) << 32);
        if (ci == 11) { L.nlb = p_lb; L.nlbr = p_lbr; L.nlen = fr(lv, 5) - (right ? p_lb : p_lbr); }
        else { L.nlen = c_end - c_oth; L.nlb = right ? c_oth : c_abs; L.nlbr = right ? c_abs : c_oth; }
        L.wn = wn | wn_node(cx, ce, right ? fst_li(st) : info, right ? info : fst_ri(st), dl);
        L.nkey = ci ? fm_key_edge(pkey, px, pe, ci) : pkey;
    }

    // ---- inspect node (nlb, nlbr, nlen, nx, ne, nli, nri); nlen > 0 by construction
    u32 const nx = L.nx(), ne = L.ne();
    if (nx == len) {
        u32 const nli = L.nli(), nri = L.nri();
        bool const ok_l = nli == FM_INFO_M || nli == FM_INFO_I, ok_r = nri == FM_INFO_M || nri == FM_INFO_I;
        L.wn |= WN_NEED_CHILD;
        if (ok_l && ok_r && L.l_last() <= ne && ne <= L.u_last()) {
            u32 rep = L.nlen;
            if (L.ct + rep > C.max_hits) rep = C.max_hits - L.ct;     // more rows than the caller wants to know of
            L.ct += rep;
            L.nlen = rep;
            L.wn |= FM_OUT_HIT << WN_OUT_SHIFT;
            if (L.ct == C.max_hits) L.wn &= ~(WN_BUSY | WN_IN_SEARCH);  // the seed has too many rows: its other hits do not matter
        }
        return;
    }
    u64 const e64 = ex[nx];
    u32 const sch = (u32)e64;
    u32 const lower = sch_lower(sch), upper = sch_upper(sch), right = sch_right(sch);
    if (ne > upper) { L.wn |= WN_NEED_CHILD; return; }
    bool const mismatch_allowed = lower <= ne + 1 && ne + 1 <= upper;
    bool const match_allowed = lower <= ne && ne <= upper;
    if (!mismatch_allowed && !match_allowed) { L.wn |= WN_NEED_CHILD; return; }

    // ---- one row: what is below this node is a comparison with the text at SA[row] (tx_step)
    u32 const flags = L.flags();
    if (L.nlen == 1u && C.text_min_remain && len - nx >= C.text_min_remain && !(flags & SEED_HAS_DELIM)) {
        L.wn |= WN_NEED_CHILD | (FM_OUT_ITEM << WN_OUT_SHIFT);
        return;
    }

    bool const filtered = mismatch_allowed && C.use_filter && ne == 0u && nx >= 1u && !(flags & SEED_NOT_ACGT);
    FmFilterPre pre{0ull, 0ull, 0ull};
    if (filtered) pre = fm_filter_prefetch(C, L, nx, e64, right);
    u32 const next_sym = C.seq[L.qoff + (sch & SCH_POS_MASK)];
    u32 const lo = right ? L.nlbr : L.nlb, other = right ? L.nlb : L.nlbr;
    u32 ab[6], cl[6];
    fm_extend_all(idx, idx.occ[right], lo, L.nlen, ab, cl);
    ++L.n_ext;

    if (mismatch_allowed) {
        // this node branches: its frame goes on top of the frames of the error edges taken so far (at most `ne` of them)
        u32 const tinfo = right ? L.nri() : L.nli();
        u32 mask = fm_child_mask6(cl, next_sym, match_allowed, tinfo == FM_INFO_M || tinfo == FM_INFO_D, tinfo == FM_INFO_M || tinfo == FM_INFO_I);
        if (mask == 0u) { L.wn |= WN_NEED_CHILD; return; }
        u32 const lv = L.depth();
        if (lv >= C.levels) { L.wn = 3u << WN_OUT_SHIFT; return; }           // overflow: not busy any more
        // (the frame is written before the filter is asked: the child bounds need not stay in registers across its lookups; a frame
        // whose children the filter drops is simply not kept)
        u32 const o1 = other + cl[0], o2 = o1 + cl[1], o3 = o2 + cl[2], o4 = o3 + cl[3], o5 = o4 + cl[4];
        fr(lv, 0) = o1; fr(lv, 1) = o2; fr(lv, 2) = o3; fr(lv, 3) = o4; fr(lv, 4) = o5; fr(lv, 5) = o5 + cl[5];
        fr(lv, 6) = ab[1]; fr(lv, 7) = ab[2]; fr(lv, 8) = ab[3]; fr(lv, 9) = ab[4]; fr(lv, 10) = ab[5];
        fr(lv, 11) = L.nlb; fr(lv, 12) = L.nlbr;
        fr(lv, 13) = (L.wn & WN_NODE_MASK) | (next_sym << 21) | (right << 24);      // = fst_pack(x, e, li, ri, next_sym, right, dlen)
        fr(lv, 15) = (u32)L.nkey; fr(lv, 16) = (u32)(L.nkey >> 32);
        fr(lv, 17) = ab[0];
        if (filtered && (mask & ~1u)) {
            mask = fm_filter_children<STATS>(C, L, nx, e64, right, mask, pre);
            if (mask == 0u) { L.wn |= WN_NEED_CHILD; return; }
        }
        fr(lv, 14) = mask;
        L.wn += WN_DEPTH1;
        L.wn |= WN_NEED_CHILD;
    } else {
        // only an exact extension is possible: continue in place (no frame)
        if (next_sym > 5u) { L.wn |= WN_NEED_CHILD; return; }
        u32 clen = cl[0], cabs = ab[0], coth = other;
#pragma unroll
        for (u32 c = 1; c < 6; ++c) {
            coth += c <= next_sym ? cl[c - 1u] : 0u;
            bool const take = c == next_sym;
            clen = take ? cl[c] : clen;
            cabs = take ? ab[c] : cabs;
        }
        if (clen == 0) { L.wn |= WN_NEED_CHILD; return; }
        L.nlb = right ? coth : cabs;
        L.nlbr = right ? cabs : coth;
        // x + 1; the extended side's info becomes "match"
        L.wn = (L.wn & ~(right ? 3u << 19 : 3u << 17)) + 1u;
        L.nlen = clen;
    }
}

// ------------------------------------------------------------------------------------------------ text mode
// A lane walks the subtree below a one-row node: its string occupies text[pL, pR]; every extension reads the text symbol next to it.
struct TxLane {
    u32 sid = 0, len = 0;
    u64 qoff = 0;
    const u64* ex = nullptr;
    bool busy = false, need_child = false;
    u32 pL = 0, pR = 0, nx = 0, ne = 0, nli = 0, nri = 0;
    u64 nkey = 0;
    u32 depth = 0;
    u32 out = FM_OUT_NONE, out_lb = 0, out_e = 0;
    u64 out_key = 0;
    u32 n_nodes = 0;
    bool overflow = false;
};

FLX_HD inline void tx_take_item(FmConst const& C, TxLane& L, DevHit const& item, DevSeed const& seed) {
    u32 const st = item.len;
    L.pL = C.idx.sa[item.lb];
    L.sid = seed.id;
    L.len = seed.length;
    L.qoff = seed.seq_off;
    L.ex = C.scheme + seed.scheme_off + (u64)item_srch(st) * seed.length;
    L.nx = fst_x(st); L.ne = fst_e(st); L.nli = fst_li(st); L.nri = fst_ri(st);
    L.pR = L.pL + L.nx + fst_dl(st) - 4u - 1u;          // the string has nx + (deletions - insertions) text symbols
    L.nkey = item.key;
    L.depth = 0;
    L.need_child = false;
    L.busy = true;
}

// How a text-mode lane reads symbols. The plain form goes to memory for every symbol (text and sequence pool, one byte each); the kernel's
// form reads 4-bit copies of the seed and of the text around the subtree's string from LDS (fm_search_text_kernel).
struct TxPlainAccess {
    const u8* text; const u8* q;
    FLX_HD u32 text_at(i64 pos) const { return text[pos]; }
    FLX_HD u32 q_at(u32 qp) const { return q[qp]; }
    // seed symbols qp, qp + 1, ... against text symbols tpos, tpos + 1, ...: how many of the first `run` agree (an A, C, G, T or N on both sides)
    FLX_HD u32 run_right(u32 qp, i64 tpos, u32 run) const {
        u32 i = 0;
        for (; i < run; ++i) { u32 const c = q[qp + i]; if (c != text[tpos + (i64)i] || c - 1u >= 5u) break; }
        return i;
    }
    // the same leftwards: seed symbols qp, qp - 1, ... against text symbols tpos, tpos - 1, ...
    FLX_HD u32 run_left(u32 qp, i64 tpos, u32 run) const {
        u32 i = 0;
        for (; i < run; ++i) { u32 const c = q[qp - i]; if (c != text[tpos - (i64)i] || c - 1u >= 5u) break; }
        return i;
    }
};

// AC: text_at(absolute text position), q_at(seed position), run_right / run_left (a forced run compared in one call)
template <class FR, class AC>
FLX_HD inline void tx_step(FmConst const& C, TxLane& L, FR&& fr, AC const& ac) {
    ++L.n_nodes;
    if (L.need_child) {
        if (L.depth == 0u) { L.busy = false; return; }                       // subtree exhausted
        u32 const lv = L.depth - 1u;
        u32 const mask = fr(lv, 1);
        u32 const st = fr(lv, 0);
        u32 const costly = mask & ~1u;
        u32 const ci = costly ? (u32)__builtin_ctz(costly) : 0u;
        u32 const rest = mask & ~(1u << ci);
        if (rest) fr(lv, 1) = rest;
        else --L.depth;
        u32 const right = fst_right(st);
        u32 const px = fst_x(st), pe = fst_e(st);
        u32 info, grow = 1;
        if (ci == 0) { L.nx = px + 1; L.ne = pe; info = FM_INFO_M; }
        else if (ci == 11) { L.nx = px + 1; L.ne = pe + 1; info = FM_INFO_I; grow = 0; }
        else {
            bool const del = ci & 1u;
            L.nx = del ? px : px + 1;
            L.ne = pe + 1;
            info = del ? FM_INFO_D : FM_INFO_S;
        }
        u32 const fL = fr(lv, 2), fR = fr(lv, 3);
        L.pL = right ? fL : fL - grow;
        L.pR = right ? fR + grow : fR;
        u64 const pkey = (u64)fr(lv, 4) | ((u64)fr(lv, 5) << 32);
        L.nli = right ? fst_li(st) : info;
        L.nri = right ? info : fst_ri(st);
        L.nkey = ci ? fm_key_edge(pkey, px, pe, ci) : pkey;
        L.need_child = false;
    }
    if (L.nx == L.len) {
        bool const ok_l = L.nli == FM_INFO_M || L.nli == FM_INFO_I, ok_r = L.nri == FM_INFO_M || L.nri == FM_INFO_I;
        if (ok_l && ok_r) {
            u32 const last_entry = (u32)L.ex[L.len - 1];              // (read here: few nodes get this far)
            if (sch_lower(last_entry) <= L.ne && L.ne <= sch_upper(last_entry)) {
                L.out = FM_OUT_HIT; L.out_lb = C.idx.isa[L.pL]; L.out_e = L.ne; L.out_key = L.nkey;
            }
        }
        L.need_child = true;
        return;
    }
    u64 const e64 = L.ex[L.nx];
    u32 const sch = (u32)e64;
    u32 const lower = sch_lower(sch), upper = sch_upper(sch), right = sch_right(sch);
    if (L.ne > upper) { L.need_child = true; return; }
    bool const mismatch_allowed = lower <= L.ne + 1 && L.ne + 1 <= upper;
    bool const match_allowed = lower <= L.ne && L.ne <= upper;
    if (!mismatch_allowed && !match_allowed) { L.need_child = true; return; }
    u32 const qp = sch & SCH_POS_MASK;
    if (!mismatch_allowed) {
        // a run of forced positions: seed and text symbol by symbol up to the end of the run (entries of one direction and one
        // upper bound = ne; their lower bounds cannot exceed it)
        u32 const run = sch_run_end(e64) - L.nx;
        u32 const i = right ? ac.run_right(qp, (i64)L.pR + 1, run) : ac.run_left(qp, (i64)L.pL - 1, run);
        if (i < run) { L.need_child = true; return; }
        if (right) { L.pR += run; L.nri = FM_INFO_M; } else { L.pL -= run; L.nli = FM_INFO_M; }
        L.nx += run;
        return;
    }
    u32 const next_sym = ac.q_at(qp);
    u32 const tc = right ? ac.text_at((i64)L.pR + 1) : ac.text_at((i64)L.pL - 1);
    u32 const tinfo = right ? L.nri : L.nli;
    bool const deletion = tinfo == FM_INFO_M || tinfo == FM_INFO_D, insertion = tinfo == FM_INFO_M || tinfo == FM_INFO_I;
    u32 mask = 0;
    if (tc - 1u < 5u) {
        if (deletion) mask |= 1u << (2u * tc - 1u);
        if (tc != next_sym) mask |= 1u << (2u * tc);
        else if (match_allowed) mask |= 1u;
    }
    if (insertion) mask |= 1u << 11;
    if (mask == 0u) { L.need_child = true; return; }
    if (L.depth >= C.levels) { L.overflow = true; L.busy = false; return; }
    u32 const lv = L.depth;
    fr(lv, 0) = fst_pack(L.nx, L.ne, L.nli, L.nri, 0u, right, 0u);
    fr(lv, 1) = mask;
    fr(lv, 2) = L.pL; fr(lv, 3) = L.pR;
    fr(lv, 4) = (u32)L.nkey; fr(lv, 5) = (u32)(L.nkey >> 32);
    ++L.depth;
    L.need_child = true;
}

template <class FR>
FLX_HD inline void tx_step(FmConst const& C, TxLane& L, FR&& fr) {
    tx_step(C, L, fr, TxPlainAccess{C.idx.text, C.seq + L.qoff});
}

}  // namespace flx
