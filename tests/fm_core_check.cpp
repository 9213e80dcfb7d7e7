// CPU check of the K1 lane logic (floxer_amd/csrc/flx_fm_core.hpp): the same fm_step / tx_step code the HIP kernels run, driven seed by
// seed on the host over the product's host-built index, against the oracle's search_n (emission order, duplicates included).
// Test infrastructure: built and run by tests/test_host_cpu.py::test_fm_core_matches_oracle; nothing of the product calls it.
//
//   fm_core_check [n_seeds] [rng seed]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../floxer_amd/csrc/flx_fm_core.hpp"
#include "../oracle/floxer_oracle.hpp"

using namespace flx;

// the one device entry point the host sources reference; never reached here (hip_device < 0)
int DeviceApi::index_arrays(int, const u8*, u64, u32*, u8*, u8*, OccBlock*, OccBlock*) { return 1; }

namespace {

struct Hit { u32 lb, len, e; u64 key; };

struct Mode { const char* name; bool filter, text; u32 k_override; bool mirrored = true; };

struct Harness {
    HostIndex* H = nullptr;
    std::vector<u8> padded;            // TEXT_PAD zeros, the text, TEXT_PAD zeros
    std::vector<u32> isa;
    std::vector<u64> filter, filter_m;
    DevIndex didx{};
    u64 lookups = 0, pruned = 0, ext = 0, items = 0, tx_nodes = 0, kills = 0;

    void build(const std::vector<std::vector<u8>>& refs, u32 k_override) {
        std::vector<u8> concat;
        std::vector<u64> lens;
        for (auto const& r : refs) { concat.insert(concat.end(), r.begin(), r.end()); lens.push_back(r.size()); }
        H = build_host_index(concat.data(), lens.data(), (u32)lens.size(), -1);
        if (!H) { fprintf(stderr, "index build failed\n"); exit(2); }
        u64 const n = H->n;
        padded.assign(n + 2 * TEXT_PAD + 16, 0);
        memcpy(padded.data() + TEXT_PAD, H->text.data(), n);
        isa.resize(n);
        for (u64 i = 0; i < n; ++i) isa[H->sa[i]] = (u32)i;
        didx.occ[0] = H->occ[0].data();
        didx.occ[1] = H->occ[1].data();
        didx.sa = H->sa.data();
        didx.text = padded.data() + TEXT_PAD;
        didx.kmer = H->kmer_table.data();
        didx.isa = isa.data();
        for (int c = 0; c < 7; ++c) didx.C[c] = (u32)H->C[c];
        didx.n = (u32)n;
        set_filter(k_override);
    }
    bool mirrored = true;              // the mirrored table is there (leftward lookups go to it)
    void set_filter(u32 k_override) {
        u64 const n = H->n;
        u32 const K = k_override ? k_override : filter_k_default(n);
        didx.filter_k = K;
        didx.filter_tmin = filter_tmin_for(n, K);
        filter.assign(filter_words(K), 0);
        filter_m.assign(filter_words(K), 0);
        // (in pieces, like the kernel's threads)
        for (i64 q0 = 0; q0 < (i64)n; q0 += 64)
            filter_add_range(didx.text, (i64)n, q0, std::min<i64>((i64)n, q0 + 64), K, didx.filter_tmin, [&](u64 w, u64 m) { filter[w] |= m; },
                             [&](u64 w, u64 m) { filter_m[w] |= m; });
        didx.filter = filter.data();
        didx.filter_m = mirrored ? filter_m.data() : nullptr;
    }

    // hits of one seed in emission order
    std::vector<Hit> search(const std::vector<u8>& pool, const std::vector<u32>& qpack, u64 off, u32 len, u32 k, Mode const& mode, u64 max_hits) {
        std::vector<Hit> hits;
        auto const scheme = expanded_scheme(k, len);
        if (scheme.empty()) return hits;
        u32 flags = 0;
        for (u32 j = 0; j < len; ++j) { if (pool[off + j] == 0) flags |= SEED_HAS_DELIM; if (pool[off + j] - 1u > 3u) flags |= SEED_NOT_ACGT; }
        DevSeed seed{};
        seed.seq_off = off; seed.length = len; seed.scheme_off = 0; seed.frames_searches = (u32)(scheme.size() / len) << 24; seed.id = 7; seed.flags = flags;
        FmConst C{};
        C.idx = didx;
        C.seq = pool.data();
        C.qpack = qpack.data();
        C.scheme = scheme.data();
        C.max_hits = (u32)std::min<u64>(max_hits, 0xFFFFFFF0u);
        C.levels = std::max(1u, k);
        C.use_filter = mode.filter ? 2u : 0u;
        C.text_min_remain = mode.text ? 2u : 0u;
        std::vector<u32> frames((size_t)C.levels * FM_FRAME_WORDS);
        auto fr = [&](u32 level, u32 word) -> u32& { return frames[level * FM_FRAME_WORDS + word]; };
        std::vector<DevHit> queued;
        FmLane L;
        fm_take_seed(C, L, seed, 0);
        u64 guard = 0;
        C.seeds = &seed;
        while (L.busy()) {
            fm_step<true>(C, L, fr);
            // (what a step produced is read from the node's registers before the next step, as the kernel does)
            if (L.out() == FM_OUT_HIT) hits.push_back(Hit{L.nlb, L.nlen, L.ne(), L.nkey});
            else if (L.out() == FM_OUT_ITEM) queued.push_back(DevHit{0u, L.nlb, L.item_word(), 0u, L.nkey});
            if (L.overflow()) { fprintf(stderr, "frame overflow\n"); exit(2); }
            L.clear_out();
            if (++guard > (1ull << 32)) { fprintf(stderr, "fm_step does not terminate\n"); exit(2); }
        }
        lookups += L.n_lookup; pruned += L.n_pruned; ext += L.n_ext; items += queued.size(); kills += L.n_prefix_kills;
        std::vector<u32> tframes((size_t)C.levels * TX_FRAME_WORDS);
        auto tfr = [&](u32 level, u32 word) -> u32& { return tframes[level * TX_FRAME_WORDS + word]; };
        for (auto const& it : queued) {
            TxLane T;
            tx_take_item(C, T, it, seed);
            while (T.busy) {
                tx_step(C, T, tfr);
                if (T.out == FM_OUT_HIT) hits.push_back(Hit{T.out_lb, 1u, T.out_e, T.out_key});
                T.out = FM_OUT_NONE;
                if (++guard > (1ull << 32)) { fprintf(stderr, "tx_step does not terminate\n"); exit(2); }
            }
            if (T.overflow) { fprintf(stderr, "text frame overflow\n"); exit(2); }
            tx_nodes += T.n_nodes;
        }
        std::stable_sort(hits.begin(), hits.end(), [](Hit const& a, Hit const& b) { return a.key < b.key; });
        return hits;
    }
};

std::vector<u8> random_bases(std::mt19937_64& rng, size_t n) {
    std::vector<u8> v(n);
    for (auto& c : v) c = (u8)(1 + rng() % 4);
    return v;
}

// one pass: a reference set, seeds cut from it, every mode against the oracle. read_like: a plain random reference and seeds shaped
// like the leaves of a long read (the benchmark's shape) instead of the mix of special cases
int run(u32 n_seeds, u64 rng_seed, bool read_like) {
    std::mt19937_64 rng(rng_seed);
    // references: random sequence with a homopolymer, a repeat across sequences, tandem repeats, a run of N, a short sequence
    std::vector<std::vector<u8>> refs;
    refs.push_back(random_bases(rng, read_like ? 1000000 : 60000));
    refs.push_back(random_bases(rng, 20000));
    refs.push_back(random_bases(rng, 777));
    refs.push_back(random_bases(rng, 9));
    if (!read_like) {
    for (size_t i = 1000; i < 1400; ++i) refs[0][i] = 1;
    for (size_t i = 0; i < 3000; ++i) refs[1][500 + i] = refs[0][20000 + i];
    for (size_t i = 0; i < 600; ++i) refs[0][30000 + i] = refs[0][30000 + i % 7];
    for (size_t i = 0; i < 2000; ++i) refs[1][8000 + i] = refs[1][8000 + i % 31];
    for (size_t i = 100; i < 110; ++i) refs[2][i] = 5;
    }
    // a diverged repeat family: 40 copies of a 300-symbol unit with 5 % substitutions
    if (!read_like) {
        auto unit = random_bases(rng, 300);
        for (int c = 0; c < 40; ++c) {
            size_t const at = 40000 + (size_t)c * 450;
            for (size_t i = 0; i < 300; ++i) refs[0][at + i] = (rng() % 100 < 5) ? (u8)(1 + rng() % 4) : unit[i];
        }
    }
    Harness hs;
    hs.build(refs, 0);
    orc::fm_index const oidx = orc::build_index(refs, 4);
    if (oidx.n != hs.H->n) { fprintf(stderr, "oracle and product disagree on the text length\n"); return 1; }

    // seeds: substrings with edits, unrelated strings, poly-A, seeds with N, seeds at sequence ends
    std::vector<u8> pool;
    struct S { u64 off; u32 len, k; };
    std::vector<S> seeds;
    for (u32 i = 0; read_like && i < n_seeds; ++i) {
        // a leaf of a 10-kb read at 8 % errors: 36 symbols with two errors allowed, or 72 with one, every symbol edited with probability 0.08
        auto const& r = refs[rng() % 2];
        u32 const k = i % 4 == 0 ? 1u : 2u;
        u32 const L = k == 1 ? 72u : 36u + (u32)(rng() % 3);
        size_t const st = rng() % (r.size() - L);
        std::vector<u8> s;
        for (u32 j = 0; j < L; ++j) {
            if (rng() % 100 < 8) {
                switch (rng() % 3) {
                    case 0: s.push_back((u8)(1 + (r[st + j] - 1 + 1 + rng() % 3) % 4)); break;
                    case 1: break;
                    default: s.push_back(r[st + j]); s.push_back((u8)(1 + rng() % 4)); break;
                }
            } else s.push_back(r[st + j]);
        }
        if (s.size() < 8) continue;
        seeds.push_back(S{pool.size(), (u32)s.size(), k});
        pool.insert(pool.end(), s.begin(), s.end());
    }
    for (u32 i = 0; !read_like && i < n_seeds; ++i) {
        auto const& r = refs[rng() % 3];
        u32 L = (u32)(8 + rng() % 70);
        if (i % 11 == 0) L = (u32)(60 + rng() % 200);
        if (L + 2 > r.size()) L = (u32)r.size() - 2;
        size_t st = rng() % (r.size() - L);
        if (i % 13 == 0) st = 0;
        if (i % 19 == 0) st = r.size() - L;
        std::vector<u8> s(r.begin() + st, r.begin() + st + L);
        u32 const k = (u32)(rng() % 4);
        u32 const n_edits = (u32)(rng() % (k + 2));
        for (u32 e = 0; e < n_edits && s.size() > 6; ++e) {
            size_t const p = rng() % s.size();
            switch (rng() % 3) {
                case 0: s[p] = (u8)(1 + rng() % 4); break;
                case 1: s.erase(s.begin() + p); break;
                default: s.insert(s.begin() + p, (u8)(1 + rng() % 4)); break;
            }
        }
        if (i % 17 == 0) s = random_bases(rng, s.size());
        if (i % 29 == 0) std::fill(s.begin(), s.end(), (u8)1);
        if (i % 37 == 0) s[rng() % s.size()] = 5;
        if (i % 41 == 0) s[rng() % s.size()] = 0;
        seeds.push_back(S{pool.size(), (u32)s.size(), k});
        pool.insert(pool.end(), s.begin(), s.end());
        // (seeds lie back to back in the pool, as the leaves of a read do)
    }
    pool.resize(pool.size() + 64, 0);
    std::vector<u32> qpack(pack_words_for(pool.size()));
    for (u64 w = 0; w < qpack.size(); ++w) qpack[w] = pack_word(pool.data(), pool.size(), w);

    u32 const kd = filter_k_default(hs.H->n);
    Mode const modes[] = {{"rank queries only", false, false, 0}, {"filter", true, false, 0}, {"text", false, true, 0}, {"filter + text", true, true, 0},
                          {"filter + text, K - 1", true, true, kd - 1}, {"filter + text, K + 2", true, true, kd + 2}, {"filter + text, K = 8", true, true, 8},
                          {"filter + text, plain table only", true, true, 0, false}, {"filter, plain table only", true, false, 0, false}};
    int failures = 0;
    for (Mode const& mode : modes) {
        hs.mirrored = mode.mirrored;
        hs.set_filter(mode.k_override);
        hs.lookups = hs.pruned = hs.ext = hs.items = hs.tx_nodes = hs.kills = 0;
        u64 n_hits = 0, oracle_ext = 0;
        for (u64 cap : {(u64)1 << 40, (u64)501}) {
            for (size_t i = 0; i < seeds.size(); ++i) {
                S const& s = seeds[i];
                std::vector<orc::anchor_group> exp;
                orc::search_counters ctr;
                orc::search_n(oidx, pool.data() + s.off, s.len, s.k, cap == 501 ? 1ull << 40 : cap, exp, &ctr);
                oracle_ext += ctr.n_extend_all + ctr.n_extend_one;
                auto got = hs.search(pool, qpack, s.off, s.len, s.k, mode, cap);
                if (cap == 501) {
                    // capped run: the seed stops once it has seen cap rows; only seeds under the cap must come out whole
                    u64 total = 0;
                    for (auto const& g : exp) total += g.cur.len;
                    if (total >= cap) continue;
                }
                n_hits += got.size();
                bool ok = got.size() == exp.size();
                for (size_t j = 0; ok && j < got.size(); ++j)
                    ok = got[j].lb == exp[j].cur.lb && got[j].len == exp[j].cur.len && got[j].e == exp[j].num_errors;
                if (!ok) {
                    if (++failures <= 5) {
                        fprintf(stderr, "MISMATCH mode '%s' seed %zu (len %u, k %u): %zu hits, oracle %zu\n", mode.name, i, s.len, s.k, got.size(), exp.size());
                        for (size_t j = 0; j < std::max(got.size(), exp.size()) && j < 12; ++j) {
                            if (j < got.size()) fprintf(stderr, "   got (%u,%u,%u)", got[j].lb, got[j].len, got[j].e); else fprintf(stderr, "   got -");
                            if (j < exp.size()) fprintf(stderr, "   exp (%llu,%llu,%llu)\n", (unsigned long long)exp[j].cur.lb, (unsigned long long)exp[j].cur.len, (unsigned long long)exp[j].num_errors);
                            else fprintf(stderr, "   exp -\n");
                        }
                    }
                }
            }
        }
        printf("mode %-24s K %2u tmin %2u: hits %llu, rank pairs %llu (oracle extensions %llu), filter words asked %llu, children dropped %llu, prefix kills %llu, "
               "subtrees %llu text steps %llu\n", mode.name, hs.didx.filter_k, hs.didx.filter_tmin, (unsigned long long)n_hits, (unsigned long long)hs.ext,
               (unsigned long long)oracle_ext, (unsigned long long)hs.lookups, (unsigned long long)hs.pruned, (unsigned long long)hs.kills,
               (unsigned long long)hs.items, (unsigned long long)hs.tx_nodes);
    }
    delete hs.H;
    return failures;
}

}  // namespace

int main(int argc, char** argv) {
    u32 const n_seeds = argc > 1 ? (u32)atoi(argv[1]) : 600;
    u64 const rng_seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
    int failures = run(n_seeds, rng_seed, false);
    failures += run(n_seeds * 4, rng_seed + 1, true);
    if (failures) { printf("fm_core_check FAILED: %d mismatches\n", failures); return 1; }
    printf("fm_core_check ok\n");
    return 0;
}
