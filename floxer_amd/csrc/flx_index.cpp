// Host-side construction of the bidirectional FM index in its HBM layout.
// Replaces fmindex(refs, sampling_rate, threads) (floxer.cpp:93-97; fmindex.hpp:7-10 = BiFMIndex<EprV2_16<6>>).
//
// Layout decisions (MI355X-first, not the reference's EPR layout):
//  * occurrence tables as 32-byte blocks over 32 BWT positions (absolute u32 counts + 3 bit-planes): a rank query is one
//    32-byte block, 2 B per text symbol and direction (hg38: 6.2 GB per direction);
//  * the FULL suffix array as u32 (4 B per symbol, 12.4 GB for hg38 out of 288 GB HBM) instead of a sampled one:
//    locate() becomes one 4-byte gather with no LF walk. The result is identical to BiFMIndex::locate because both return
//    SA[row] split into (sequence, offset).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <type_traits>

#include "flx_internal.hpp"

namespace flx {

namespace {

// ---------------------------------------------------------------- SA-IS (Nong, Zhang, Chan 2009), own implementation
// s[n-1] must be the unique smallest symbol 0.
template <class Idx, class Sym>
struct Sais {
    static constexpr Idx EMPTY = (Idx)-1;
    const Sym* s;
    Idx* SA;
    Idx n;
    Idx K;     // symbols in [0, K]
    std::vector<uint8_t> t;   // 1 = S-type
    std::vector<Idx> bkt;

    bool is_s(Idx i) const { return (t[i >> 3] >> (i & 7)) & 1; }
    void set_s(Idx i, bool v) { if (v) t[i >> 3] |= (uint8_t)(1u << (i & 7)); else t[i >> 3] &= (uint8_t)~(1u << (i & 7)); }
    bool is_lms(Idx i) const { return i > 0 && is_s(i) && !is_s(i - 1); }

    void get_buckets(bool end) {
        std::fill(bkt.begin(), bkt.end(), 0);
        for (Idx i = 0; i < n; ++i) bkt[s[i]]++;
        Idx sum = 0;
        for (Idx i = 0; i <= K; ++i) { sum += bkt[i]; bkt[i] = end ? sum : sum - bkt[i]; }
    }
    void induce_l() {
        get_buckets(false);
        for (Idx i = 0; i < n; ++i) {
            Idx const v = SA[i];
            if (v != EMPTY && v > 0) { Idx const j = v - 1; if (!is_s(j)) SA[bkt[s[j]]++] = j; }
        }
    }
    void induce_s() {
        get_buckets(true);
        for (Idx i = n; i-- > 0;) {
            Idx const v = SA[i];
            if (v != EMPTY && v > 0) { Idx const j = v - 1; if (is_s(j)) SA[--bkt[s[j]]] = j; }
        }
    }

    void run() {
        t.assign((size_t)n / 8 + 1, 0);
        bkt.assign((size_t)K + 1, 0);
        if (n == 1) { SA[0] = 0; return; }
        set_s(n - 1, true);
        set_s(n - 2, false);
        for (Idx i = n - 2; i-- > 0;) set_s(i, s[i] < s[i + 1] || (s[i] == s[i + 1] && is_s(i + 1)));
        // stage 1: sort LMS substrings
        get_buckets(true);
        for (Idx i = 0; i < n; ++i) SA[i] = EMPTY;
        for (Idx i = 1; i < n; ++i) if (is_lms(i)) SA[--bkt[s[i]]] = i;
        induce_l();
        induce_s();
        Idx n1 = 0;
        for (Idx i = 0; i < n; ++i) if (SA[i] != EMPTY && is_lms(SA[i])) SA[n1++] = SA[i];
        for (Idx i = n1; i < n; ++i) SA[i] = EMPTY;
        Idx name = 0, prev = EMPTY;
        for (Idx i = 0; i < n1; ++i) {
            Idx const pos = SA[i];
            bool diff = false;
            for (Idx d = 0; d < n; ++d) {
                if (prev == EMPTY || s[pos + d] != s[prev + d] || is_s(pos + d) != is_s(prev + d)) { diff = true; break; }
                if (d > 0 && (is_lms(pos + d) || is_lms(prev + d))) break;
            }
            if (diff) { ++name; prev = pos; }
            SA[n1 + pos / 2] = name - 1;
        }
        for (Idx i = n, j = n; i-- > n1;) if (SA[i] != EMPTY) SA[--j] = SA[i];
        // stage 2: solve the reduced problem
        Idx* SA1 = SA;
        Idx* s1 = SA + n - n1;
        if (name < n1) {
            Sais<Idx, Idx> sub{s1, SA1, n1, name - 1, {}, {}};
            sub.run();
        } else {
            for (Idx i = 0; i < n1; ++i) SA1[s1[i]] = i;
        }
        // stage 3: induce the final order
        get_buckets(true);
        for (Idx i = 1, j = 0; i < n; ++i) if (is_lms(i)) s1[j++] = i;
        for (Idx i = 0; i < n1; ++i) SA1[i] = s1[SA1[i]];
        for (Idx i = n1; i < n; ++i) SA[i] = EMPTY;
        for (Idx i = n1; i-- > 0;) {
            Idx const j = SA[i];
            SA[i] = EMPTY;
            SA[--bkt[s[j]]] = j;
        }
        induce_l();
        induce_s();
    }
};

// suffix array of `text` in plain order (a suffix that is a prefix of another sorts first)
std::vector<u32> suffix_array(const std::vector<u8>& text) {
    u64 const n = text.size();
    std::vector<u8> shifted(n + 1);
    for (u64 i = 0; i < n; ++i) shifted[i] = (u8)(text[i] + 1);
    shifted[n] = 0;
    std::vector<u32> out(n);
    if (n + 1 < ((u64)1 << 31)) {
        std::vector<int32_t> sa(n + 1);
        Sais<int32_t, u8> w{shifted.data(), sa.data(), (int32_t)(n + 1), 7, {}, {}};
        w.run();
        for (u64 i = 0; i < n; ++i) out[i] = (u32)sa[i + 1];     // sa[0] is the virtual sentinel
    } else {
        std::vector<int64_t> sa(n + 1);
        Sais<int64_t, u8> w{shifted.data(), sa.data(), (int64_t)(n + 1), 7, {}, {}};
        w.run();
        for (u64 i = 0; i < n; ++i) out[i] = (u32)sa[i + 1];
    }
    return out;
}

void build_occ_blocks(const std::vector<u8>& bwt, std::vector<OccBlock>& blocks) {
    u64 const n = bwt.size();
    u64 const nb = n / OCC_BLOCK_POS + 1;
    blocks.assign(nb, OccBlock{});
    u32 cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (u64 b = 0; b < nb; ++b) {
        OccBlock& blk = blocks[b];
        for (int c = 0; c < 5; ++c) blk.w[c] = cnt[c];
        u32 p0 = 0, p1 = 0, p2 = 0;
        for (u32 k = 0; k < OCC_BLOCK_POS; ++k) {
            u64 const pos = b * OCC_BLOCK_POS + k;
            u8 const sym = pos < n ? bwt[pos] : 7;
            p0 |= (u32)(sym & 1) << k;
            p1 |= (u32)((sym >> 1) & 1) << k;
            p2 |= (u32)((sym >> 2) & 1) << k;
            if (pos < n) cnt[sym]++;
        }
        blk.w[5] = p0; blk.w[6] = p1; blk.w[7] = p2;
    }
}

// rank of all symbols at position i on the host (same block layout the device reads)
void host_rank_all(const std::vector<OccBlock>& tab, u64 i, u64 out[6]) {
    OccBlock const& b = tab[i / OCC_BLOCK_POS];
    u32 const off = (u32)(i % OCC_BLOCK_POS);
    for (int c = 0; c < 5; ++c) out[c] = b.w[c];
    for (u32 k = 0; k < off; ++k) {
        u32 const sym = ((b.w[5] >> k) & 1) | (((b.w[6] >> k) & 1) << 1) | (((b.w[7] >> k) & 1) << 2);
        if (sym < 5) out[sym]++;
    }
    out[5] = i - (out[0] + out[1] + out[2] + out[3] + out[4]);
}

// cursor of every KMER_Q-mer, built level by level with BiFMIndexCursor::extendRight semantics
void build_kmer_table(HostIndex& idx) {
    struct Cur { u64 lb, lb_rev, len; };
    std::vector<Cur> level{Cur{0, 0, idx.n}};
    for (u32 d = 0; d < KMER_Q; ++d) {
        std::vector<Cur> next(level.size() * 4);
        for (size_t i = 0; i < level.size(); ++i) {
            Cur const& c = level[i];
            u64 a[6] = {0, 0, 0, 0, 0, 0}, b[6] = {0, 0, 0, 0, 0, 0};
            if (c.len) { host_rank_all(idx.occ[1], c.lb_rev, a); host_rank_all(idx.occ[1], c.lb_rev + c.len, b); }
            u64 acc = c.lb + (b[0] - a[0]);
            for (int sym = 1; sym <= 4; ++sym) {
                u64 const len = b[sym] - a[sym];
                next[i * 4 + (sym - 1)] = Cur{acc, idx.C[sym] + a[sym], len};
                acc += len;
            }
        }
        level.swap(next);
    }
    idx.kmer_table.resize(level.size() * 3);
    for (size_t i = 0; i < level.size(); ++i) {
        idx.kmer_table[i * 3] = (u32)level[i].lb;
        idx.kmer_table[i * 3 + 1] = (u32)level[i].lb_rev;
        idx.kmer_table[i * 3 + 2] = (u32)level[i].len;
    }
}

}  // namespace

HostIndex* build_host_index(const u8* concat, const u64* lens, u32 n_refs, int hip_device) {
    auto idx = std::make_unique<HostIndex>();
    constexpr u64 sampling = 4;    // floxer.cpp:92: padding keeps every sequence a multiple of the reference's sampling rate
    u64 total = 0;
    for (u32 r = 0; r < n_refs; ++r) total += lens[r] + (sampling - lens[r] % sampling);
    if (total == 0) { set_error("empty reference"); return nullptr; }
    if (total >= ((u64)1 << 32) - 512) { set_error("text of 2^32 symbols or more is not supported by this build"); return nullptr; }
    idx->text.assign(total, 0);
    u64 off = 0, at = 0;
    for (u32 r = 0; r < n_refs; ++r) {
        idx->seq_start.push_back(at);
        idx->seq_len.push_back(lens[r]);
        memcpy(idx->text.data() + at, concat + off, lens[r]);
        off += lens[r];
        at += lens[r] + (sampling - lens[r] % sampling);
    }
    idx->n = total;
    u64 const n = idx->n;
    u64 cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u8 seen = 0;
    for (u8 c : idx->text) { cnt[c & 7]++; seen |= c; }
    if ((seen & 0xF8) || cnt[6] || cnt[7]) { set_error("reference rank > 5"); return nullptr; }
    for (int c = 0; c < 6; ++c) idx->C[c + 1] = idx->C[c] + cnt[c];
    if (hip_device >= 0) {
        // suffix arrays (prefix doubling with radix sorts), BWTs and occurrence tables on the device
        idx->sa.resize(n);
        idx->bwt[0].resize(n);
        idx->bwt[1].resize(n);
        u64 const nb = n / OCC_BLOCK_POS + 1;
        idx->occ[0].resize(nb);
        idx->occ[1].resize(nb);
        int const e = DeviceApi::index_arrays(hip_device, idx->text.data(), n, idx->sa.data(), idx->bwt[0].data(), idx->bwt[1].data(),
                                              idx->occ[0].data(), idx->occ[1].data());
        if (e) { set_error("index construction on the device failed (HIP error " + std::to_string(e) + ")"); return nullptr; }
    } else {
        idx->sa = suffix_array(idx->text);
        idx->bwt[0].resize(n);
        for (u64 i = 0; i < n; ++i) idx->bwt[0][i] = idx->text[(idx->sa[i] + n - 1) % n];
        {
            std::vector<u8> rev(idx->text.rbegin(), idx->text.rend());
            std::vector<u32> sa_rev = suffix_array(rev);
            idx->bwt[1].resize(n);
            for (u64 i = 0; i < n; ++i) idx->bwt[1][i] = rev[(sa_rev[i] + n - 1) % n];
        }
        build_occ_blocks(idx->bwt[0], idx->occ[0]);
        build_occ_blocks(idx->bwt[1], idx->occ[1]);
    }
    build_kmer_table(*idx);
    return idx.release();
}

// ---------------------------------------------------------------- own index file format (replaces the cereal archive)
namespace {
constexpr char MAGIC[8] = {'F', 'L', 'X', 'I', 'D', 'X', '0', '5'};
template <class T> bool wr(FILE* f, const std::vector<T>& v) {
    u64 const n = v.size();
    return fwrite(&n, 8, 1, f) == 1 && (n == 0 || fwrite(v.data(), sizeof(T), n, f) == n);
}
template <class T> bool rd(FILE* f, std::vector<T>& v) {
    u64 n = 0;
    if (fread(&n, 8, 1, f) != 1) return false;
    v.resize(n);
    return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}
}  // namespace

int save_host_index(const HostIndex& idx, const char* path) {
    FILE* f = fopen(path, "wb");
    if (!f) { set_error(std::string("cannot open index file for writing: ") + path); return FLX_ERR_IO; }
    bool ok = fwrite(MAGIC, 8, 1, f) == 1 && fwrite(&idx.n, 8, 1, f) == 1 && fwrite(idx.C, 8, 7, f) == 7 && wr(f, idx.text) &&
              wr(f, idx.seq_start) && wr(f, idx.seq_len) && wr(f, idx.sa) && wr(f, idx.occ[0]) && wr(f, idx.occ[1]) &&
              wr(f, idx.bwt[0]) && wr(f, idx.bwt[1]) && wr(f, idx.kmer_table);
    ok = (fclose(f) == 0) && ok;
    if (!ok) { set_error("short write while saving the index"); return FLX_ERR_IO; }
    return FLX_OK;
}

HostIndex* load_host_index(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { set_error(std::string("cannot open index file: ") + path); return nullptr; }
    // the length words of the file are not trusted: a bogus one must not make resize() throw across the C ABI
    struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{f};
    u64 file_bytes = 0;
    if (fseek(f, 0, SEEK_END) == 0) { long const e = ftell(f); if (e > 0) file_bytes = (u64)e; rewind(f); }
    auto idx = std::make_unique<HostIndex>();
    auto rd_checked = [&](auto& v, u64 expected) {
        using T = typename std::remove_reference<decltype(v)>::type::value_type;
        u64 n = 0;
        if (fread(&n, 8, 1, f) != 1 || n != expected || n * sizeof(T) > file_bytes) return false;
        try { v.resize(n); } catch (std::exception const&) { return false; }
        return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
    };
    char magic[8];
    bool ok = fread(magic, 8, 1, f) == 1 && memcmp(magic, MAGIC, 8) == 0 && fread(&idx->n, 8, 1, f) == 1 && fread(idx->C, 8, 7, f) == 7;
    u64 const n = idx->n;
    ok = ok && n > 0 && n < ((u64)1 << 32) && n <= file_bytes && rd_checked(idx->text, n);
    if (ok) {
        u64 n_refs = 0;                                   // seq_start: its own length word says how many references there are
        long const at = ftell(f);
        ok = fread(&n_refs, 8, 1, f) == 1 && n_refs >= 1 && n_refs <= n && fseek(f, at, SEEK_SET) == 0 &&
             rd_checked(idx->seq_start, n_refs) && rd_checked(idx->seq_len, n_refs);
    }
    u64 const nb = n / OCC_BLOCK_POS + 1;
    ok = ok && rd_checked(idx->sa, n) && rd_checked(idx->occ[0], nb) && rd_checked(idx->occ[1], nb) && rd_checked(idx->bwt[0], n) &&
         rd_checked(idx->bwt[1], n) && rd_checked(idx->kmer_table, ((u64)1 << (2 * KMER_Q)) * 3);
    if (ok) {
        // sequences lie inside the text, in order, with their sentinel padding; C is a prefix sum that ends at n
        u64 prev_end = 0;
        for (size_t r = 0; ok && r < idx->seq_start.size(); ++r) {
            ok = idx->seq_start[r] >= prev_end && idx->seq_len[r] <= n && idx->seq_start[r] + idx->seq_len[r] < n + 1;
            prev_end = idx->seq_start[r] + idx->seq_len[r];
        }
        ok = ok && idx->C[0] == 0 && idx->C[6] == n;
        for (int c = 0; ok && c < 6; ++c) ok = idx->C[c] <= idx->C[c + 1];
    }
    if (!ok) { set_error("index file is corrupt, truncated or of another version"); return nullptr; }
    return idx.release();
}

}  // namespace flx
