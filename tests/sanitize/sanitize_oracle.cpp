// AddressSanitizer / UBSan driver for the CPU oracle (test infrastructure): index build and import, search on mutated / degenerate
// seeds, alignment in all modes and both algorithms on ragged sizes, the whole path with and without -I, -w, -d, -b.
#include <cstdio>
#include <cstdint>
#include <random>
#include <vector>

#include "../../oracle/floxer_oracle.hpp"

using namespace orc;

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "sanitize_oracle: check failed: %s (line %d)\n", #x, __LINE__); return 1; } } while (0)

int main() {
    std::mt19937_64 rng(5);
    auto rnd = [&](uint64_t n) { return (uint64_t)(rng() % n); };
    std::vector<std::vector<uint8_t>> refs(3);
    for (size_t r = 0; r < 3; ++r) { refs[r].resize(r == 2 ? 37 : 9000 + 501 * r); for (auto& c : refs[r]) c = (uint8_t)(1 + rnd(4)); }
    for (size_t i = 100; i < 160; ++i) refs[0][i] = 1;           // homopolymer
    for (size_t i = 300; i < 320; ++i) refs[1][i] = 5;           // N run
    fm_index const idx = build_index(refs, 4);
    std::vector<uint32_t> sa(idx.sa.begin(), idx.sa.end());
    fm_index const imp = import_index(refs, 4, sa.data(), idx.bwt.data(), idx.bwt_rev.data());
    CHECK(imp.n == idx.n && imp.occ_cp == idx.occ_cp && imp.occ_rev_cp == idx.occ_rev_cp);
    // ---- search_n on seeds of all kinds
    for (int it = 0; it < 300; ++it) {
        auto const& g = refs[rnd(2)];
        uint64_t const len = 1 + rnd(50), at = rnd(g.size() - len);
        std::vector<uint8_t> seed(g.begin() + at, g.begin() + at + len);
        if (it % 3 == 0 && len > 2) seed[rnd(len)] = (uint8_t)rnd(6);          // incl. '$' and N
        if (it % 7 == 0) for (auto& c : seed) c = 1;
        std::vector<anchor_group> out;
        search_counters ctr;
        search_n(idx, seed.data(), len, (uint32_t)rnd(4), it % 2 ? 501 : 50, out, &ctr);
        for (auto const& a : out) CHECK(a.cur.lb + a.cur.len <= idx.n);
    }
    // ---- align: ragged sizes, all modes, both algorithms agree
    for (int it = 0; it < 200; ++it) {
        uint64_t const m = 1 + rnd(200), n = rnd(400), k = rnd(m + 1);
        std::vector<uint8_t> q(m), r(n);
        for (auto& c : q) c = (uint8_t)(1 + rnd(4));
        for (auto& c : r) c = (uint8_t)(1 + rnd(4));
        if (n >= m && it % 2) for (uint64_t i = 0; i < m; ++i) r[(n - m) / 2 + i] = q[i];
        for (int mode = 0; mode < 3; ++mode) {
            align_result const a = align(r.data(), n, q.data(), m, k, mode, 0), b = align(r.data(), n, q.data(), m, k, mode, 1);
            CHECK(a.exists == b.exists);
            if (a.exists) CHECK(a.num_errors == b.num_errors && a.begin == b.begin && a.cigar == b.cigar);
        }
    }
    // ---- whole path
    std::vector<std::vector<uint8_t>> reads;
    for (int i = 0; i < 12; ++i) {
        auto const& g = refs[rnd(2)];
        uint64_t const len = 300 + rnd(500), at = rnd(g.size() - len);
        std::vector<uint8_t> rd(g.begin() + at, g.begin() + at + len);
        for (uint64_t e = 0; e < len / 25; ++e) rd[rnd(len)] = (uint8_t)(1 + rnd(4));
        reads.push_back(rd);
    }
    reads.push_back({});
    reads.push_back({1, 2, 3});
    for (int variant = 0; variant < 5; ++variant) {
        params p;
        p.error_probability = 0.06;
        p.interval_optimization = variant == 1;
        p.without_cigar = variant == 2;
        p.direct_full = variant == 3;
        p.bottom_up = variant == 4;
        run_output const out = align_reads(idx, refs, reads, p, 2);
        CHECK(out.skipped.size() == reads.size());
        for (auto const& r : out.records) CHECK(r.cigar_off + r.cigar_len <= out.cigars.size());
    }
    printf("sanitize_oracle ok\n");
    return 0;
}
