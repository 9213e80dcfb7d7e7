// Synthetic genome + long-read generator with the semantics of the reference's simulator
// (src/main/simulated_dataset.cpp:30-49 create_genome, :81-223 create_and_write_reads), multi-threaded, for the benchmark and the
// scale tests (1 M x 10 kb reads are 10 GB; the per-read Python generator of floxer_amd/simulate.py takes minutes for that).
//   genome: i.i.d. uniform over {A,C,G,T} (ranks 1..4)
//   read:   substring of base_len at a uniform start on a uniform chromosome; exactly floor(rate * base_len) distinct positions
//           mutated, kind uniform over {mismatch (always a different base, :75-79), insertion (base kept, a uniform base inserted
//           after it, :148-150), deletion}; forward strand, then reverse-complemented with probability revcomp_fraction
//           (documented deviation: the reference's simulator only emits forward reads)
// The reference seeds std::mt19937 and draws through std::uniform_int_distribution / std::sample, which are not reproducible
// across standard libraries; this generator is xoshiro256** seeded per read (splitmix64 of seed and read index), so its output
// does not depend on the number of threads.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "flx_internal.hpp"

namespace {
using namespace flx;

struct Rng {
    u64 s[4];
    static u64 splitmix(u64& x) {
        u64 z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    Rng(u64 seed, u64 stream) {
        u64 x = seed ^ (stream * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull);
        for (auto& v : s) v = splitmix(x);
    }
    static u64 rotl(u64 x, int k) { return (x << k) | (x >> (64 - k)); }
    u64 next() {
        u64 const result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return result;
    }
    u64 below(u64 n) {                       // uniform in [0, n), unbiased (Lemire's multiply-and-reject)
        unsigned __int128 m = (unsigned __int128)next() * n;
        if ((u64)m < n) {
            u64 const t = (0 - n) % n;
            while ((u64)m < t) m = (unsigned __int128)next() * n;
        }
        return (u64)(m >> 64);
    }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

unsigned sim_threads() {
    unsigned n = std::thread::hardware_concurrency();
    if (const char* env = getenv("FLX_SIM_THREADS")) n = (unsigned)strtoul(env, nullptr, 10);
    return std::max(1u, std::min(n, 64u));
}

template <class F>
void parallel_ranges(u64 n, F&& body) {     // body(first, last) on disjoint ranges
    unsigned const t = (unsigned)std::min<u64>(sim_threads(), std::max<u64>(1, n));
    if (t <= 1) { body((u64)0, n); return; }
    std::vector<std::thread> threads;
    for (unsigned i = 0; i < t; ++i) threads.emplace_back([&, i] { body(n * i / t, n * (i + 1) / t); });
    for (auto& th : threads) th.join();
}

struct ReadPlan { u32 chrom; u8 reverse; u64 start; };

// the mutations of read `id`: kind[p] for p in [0, base_len) (0 none, 1 mismatch, 2 insertion, 3 deletion), new base in base[p]
struct Mutator {
    u32 base_len, num_errors;
    std::vector<u8> kind, base;
    std::vector<u32> touched;
    Mutator(u32 bl, u32 ne) : base_len(bl), num_errors(ne), kind(bl, 0), base(bl, 0) { touched.reserve(ne); }
    ReadPlan plan(Rng& rng, const u64* chrom_lens, u32 n_chrom, double revcomp_fraction, i64& delta) {
        for (u32 p : touched) kind[p] = 0;
        touched.clear();
        ReadPlan pl;
        pl.chrom = (u32)rng.below(n_chrom);
        pl.start = rng.below(chrom_lens[pl.chrom] - base_len);          // [0, chromosome_length - base_len - 1], :96
        delta = 0;
        while (touched.size() < num_errors) {                           // distinct positions (std::sample, :118-126)
            u32 const p = (u32)rng.below(base_len);
            if (kind[p]) continue;
            u32 const k = (u32)rng.below(3);
            kind[p] = (u8)(k + 1);
            base[p] = (u8)(k == 0 ? rng.below(3) : rng.below(4));
            delta += k == 1 ? 1 : k == 2 ? -1 : 0;
            touched.push_back(p);
        }
        pl.reverse = rng.unit() < revcomp_fraction ? 1 : 0;
        return pl;
    }
    void emit(const u8* origin, ReadPlan const& pl, u8* out, u64 out_len) const {
        static const u8 comp[6] = {0, 4, 3, 2, 1, 5};
        u64 w = 0;
        auto put = [&](u8 r) { if (pl.reverse) out[out_len - 1 - w] = comp[r]; else out[w] = r; ++w; };
        for (u32 p = 0; p < base_len; ++p) {
            u8 const o = origin[p];
            switch (kind[p]) {
                case 0: put(o); break;
                case 1: { u8 const g = base[p], orank = (u8)(o - 1); put((u8)((g >= orank ? g + 1 : g) + 1)); break; }   // choose_distinct_rank
                case 2: put(o); put((u8)(base[p] + 1)); break;
                default: break;                                          // deletion
            }
        }
    }
};

}  // namespace

extern "C" int flx_sim_genome(uint64_t length, uint64_t seed, uint8_t* out) {
    if (!out && length) { set_error("flx_sim_genome: null output"); return FLX_ERR_INVALID; }
    constexpr u64 BLOCK = 1 << 20;           // one generator per block: the result does not depend on the thread count
    u64 const n_blocks = (length + BLOCK - 1) / BLOCK;
    parallel_ranges(n_blocks, [&](u64 b0, u64 b1) {
        for (u64 b = b0; b < b1; ++b) {
            Rng rng(seed, b);
            u64 const first = b * BLOCK, last = std::min(length, first + BLOCK);
            u64 i = first;
            for (; i + 32 <= last; i += 32) {
                u64 const r = rng.next();
                for (u32 j = 0; j < 32; ++j) out[i + j] = (u8)(((r >> (2 * j)) & 3u) + 1u);
            }
            if (i < last) {
                u64 const r = rng.next();
                for (u32 j = 0; i < last; ++i, ++j) out[i] = (u8)(((r >> (2 * j)) & 3u) + 1u);
            }
        }
    });
    return FLX_OK;
}

// A repeat-rich reference at genome scale: what a uniform random text lacks and a human genome has. Per 256-kb block (its own
// generator stream, so the result does not depend on the thread count) stretches of unique sequence alternate with elements drawn
// by their share of the bases:
//   interspersed families (the consensus sequences are a function of the seed alone): one of 300 bp with copies 2-15 % diverged
//   (Alu-like, 11 % of the bases), one of 6 kb whose copies are 5'-truncated to 200 bp .. 6 kb and 2-20 % diverged (L1-like,
//   18 %), six more of 250 .. 1500 bp (12 %); a copy is the consensus or its reverse complement with substitutions (80 % of the
//   edits), deletions and insertions;
//   tandem repeats: units of 1-6 bp in arrays of 20 .. 300 bp and units of 10-60 bp in arrays of 5 .. 100 copies (3 %);
//   two-letter low-complexity stretches of 100 .. 1500 bp (1 %);
//   runs of N of 100 .. 20000 bp (0.3 %).
// A second pass copies segments of 5 .. 50 kb to other places with 1-3 % divergence (segmental duplications, 5 % of the bases).
// About half of the bases stay unique. floxer's caps (-M 500 / -m 50, search.cpp:190-272) exist because of such sequence.
extern "C" int flx_sim_genome_repeats(uint64_t length, uint64_t seed, uint8_t* out) {
    if (!out && length) { set_error("flx_sim_genome_repeats: null output"); return FLX_ERR_INVALID; }
    static const u8 comp[6] = {0, 4, 3, 2, 1, 5};
    struct Family { std::vector<u8> cons; double share; u32 min_len; double div_lo, div_hi; bool truncate; };
    std::vector<Family> fams;
    {
        Rng rng(seed, 0xFA111E5ull);
        auto make = [&](u32 len, double share, u32 min_len, double lo, double hi, bool trunc) {
            Family f{std::vector<u8>(len), share, min_len, lo, hi, trunc};
            for (auto& c : f.cons) c = (u8)(1 + rng.below(4));
            fams.push_back(std::move(f));
        };
        make(300, 0.11, 300, 0.02, 0.15, false);
        make(6000, 0.18, 200, 0.02, 0.20, true);
        for (u32 i = 0; i < 6; ++i) make((u32)(250 + rng.below(1251)), 0.02, 100, 0.03, 0.25, true);
    }
    double const share_tandem = 0.03, share_low = 0.01, share_n = 0.003;
    // elements per base of each kind = share / mean length; the unique stretch between two elements has the mean length that
    // leaves the unique share
    std::vector<double> rate;
    for (auto const& f : fams) rate.push_back(f.share / (f.truncate ? 0.5 * (f.min_len + f.cons.size()) : (double)f.cons.size()));
    rate.push_back(share_tandem / 400.0);
    rate.push_back(share_low / 800.0);
    rate.push_back(share_n / 10050.0);
    double rate_sum = 0, repeat_share = share_tandem + share_low + share_n;
    for (double r : rate) rate_sum += r;
    for (auto const& f : fams) repeat_share += f.share;
    double const mean_unique = (1.0 - repeat_share) / rate_sum;
    constexpr u64 BLOCK = 256 << 10;
    u64 const n_blocks = (length + BLOCK - 1) / BLOCK;
    parallel_ranges(n_blocks, [&](u64 b0, u64 b1) {
        std::vector<u8> piece;
        for (u64 b = b0; b < b1; ++b) {
            Rng rng(seed, b + 1);
            u64 at = b * BLOCK;
            u64 const last = std::min(length, at + BLOCK);
            auto put = [&](u8 c) { if (at < last) out[at++] = c; };
            while (at < last) {
                // unique stretch: geometric with the mean above (at least 20)
                u64 const ulen = 20 + (u64)(-std::log(1.0 - rng.unit()) * mean_unique);
                for (u64 i = 0; i < ulen && at < last; ++i) put((u8)(1 + (rng.next() >> 62)));
                if (at >= last) break;
                double pick = rng.unit() * rate_sum;
                size_t kind = 0;
                while (kind + 1 < rate.size() && pick >= rate[kind]) { pick -= rate[kind]; ++kind; }
                if (kind < fams.size()) {
                    Family const& f = fams[kind];
                    u32 const clen = f.truncate ? (u32)(f.min_len + rng.below(f.cons.size() - f.min_len + 1)) : (u32)f.cons.size();
                    u32 const first = (u32)f.cons.size() - clen;                      // 5'-truncated: the copy keeps the consensus' end
                    double const div = f.div_lo + rng.unit() * (f.div_hi - f.div_lo);
                    bool const rc = rng.next() & 1;
                    piece.clear();
                    for (u32 i = 0; i < clen; ++i) {
                        u8 const c = f.cons[first + i];
                        if (rng.unit() < div) {
                            u64 const t = rng.below(10);
                            if (t < 8) piece.push_back((u8)(1 + (c - 1 + 1 + rng.below(3)) % 4));
                            else if (t == 8) continue;
                            else { piece.push_back(c); piece.push_back((u8)(1 + rng.below(4))); }
                        } else piece.push_back(c);
                    }
                    if (rc) for (size_t i = piece.size(); i-- > 0;) put(comp[piece[i]]);
                    else for (u8 c : piece) put(c);
                } else if (kind == fams.size()) {                                   // tandem repeat
                    bool const micro = rng.next() & 1;
                    u32 const ulen = micro ? (u32)(1 + rng.below(6)) : (u32)(10 + rng.below(51));
                    u64 const total = micro ? 20 + rng.below(281) : (u64)ulen * (5 + rng.below(96));
                    u8 unit[64];
                    for (u32 i = 0; i < ulen; ++i) unit[i] = (u8)(1 + rng.below(4));
                    for (u64 i = 0; i < total; ++i) put(rng.unit() < 0.01 ? (u8)(1 + rng.below(4)) : unit[i % ulen]);
                } else if (kind == fams.size() + 1) {                               // two-letter low complexity
                    u8 const x = (u8)(1 + rng.below(4)), y = (u8)(1 + rng.below(4));
                    u64 const total = 100 + rng.below(1401);
                    for (u64 i = 0; i < total; ++i) put((rng.next() & 1) ? x : y);
                } else {                                                            // run of N
                    u64 const total = 100 + rng.below(19901);
                    for (u64 i = 0; i < total; ++i) put(5);
                }
            }
        }
    });
    // segmental duplications, one after the other (a copy may be copied again)
    if (length > 200000) {
        Rng rng(seed, 0x5E6D09ull);
        u64 copied = 0;
        while (copied < length / 20) {
            u64 const len = 5000 + rng.below(45001);
            u64 const from = rng.below(length - len), to = rng.below(length - len);
            if (from + len > to && to + len > from) continue;                        // overlapping: draw again
            double const div = 0.01 + rng.unit() * 0.02;
            for (u64 i = 0; i < len; ++i) {
                u8 c = out[from + i];
                if (c >= 1 && c <= 4 && rng.unit() < div) c = (u8)(1 + (c - 1 + 1 + rng.below(3)) % 4);
                out[to + i] = c;
            }
            copied += len;
        }
    }
    return FLX_OK;
}

extern "C" int flx_sim_reads(const uint8_t* genome_concat, const uint64_t* chrom_lens, uint32_t n_chrom, uint64_t n_reads,
                             uint32_t base_len, double error_rate, double revcomp_fraction, uint64_t seed, uint8_t* out_pool,
                             uint64_t pool_capacity, uint64_t* out_offsets, uint32_t* out_chrom, uint64_t* out_pos,
                             uint8_t* out_reverse) {
    if (!genome_concat || !chrom_lens || n_chrom == 0 || !out_offsets || (n_reads && !out_pool)) { set_error("flx_sim_reads: null argument"); return FLX_ERR_INVALID; }
    if (base_len == 0 || error_rate < 0 || error_rate >= 1) { set_error("flx_sim_reads: invalid read length / error rate"); return FLX_ERR_INVALID; }
    std::vector<u64> chrom_start(n_chrom);
    u64 total = 0;
    for (u32 c = 0; c < n_chrom; ++c) {
        if (chrom_lens[c] <= base_len) { set_error("flx_sim_reads: a chromosome is not longer than the reads"); return FLX_ERR_INVALID; }
        chrom_start[c] = total;
        total += chrom_lens[c];
    }
    u32 const num_errors = (u32)(error_rate * base_len);                // size_t num_errors = error_rate * base_read_length, :90
    // pass 1: lengths (the plan of a read is a function of (seed, read index) only)
    std::vector<u64> lens(n_reads);
    parallel_ranges(n_reads, [&](u64 r0, u64 r1) {
        Mutator m(base_len, num_errors);
        for (u64 r = r0; r < r1; ++r) {
            Rng rng(seed, r);
            i64 delta = 0;
            (void)m.plan(rng, chrom_lens, n_chrom, revcomp_fraction, delta);
            lens[r] = (u64)((i64)base_len + delta);
        }
    });
    out_offsets[0] = 0;
    for (u64 r = 0; r < n_reads; ++r) out_offsets[r + 1] = out_offsets[r] + lens[r];
    if (out_offsets[n_reads] > pool_capacity) { set_error("flx_sim_reads: pool too small"); return FLX_ERR_CAPACITY; }
    // pass 2: sequences
    parallel_ranges(n_reads, [&](u64 r0, u64 r1) {
        Mutator m(base_len, num_errors);
        for (u64 r = r0; r < r1; ++r) {
            Rng rng(seed, r);
            i64 delta = 0;
            ReadPlan const pl = m.plan(rng, chrom_lens, n_chrom, revcomp_fraction, delta);
            m.emit(genome_concat + chrom_start[pl.chrom] + pl.start, pl, out_pool + out_offsets[r], lens[r]);
            if (out_chrom) out_chrom[r] = pl.chrom;
            if (out_pos) out_pos[r] = pl.start;
            if (out_reverse) out_reverse[r] = pl.reverse;
        }
    });
    return FLX_OK;
}
