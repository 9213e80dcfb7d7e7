// Host-side construction of the bidirectional FM index in its HBM layout.
// Replaces fmindex(refs, sampling_rate, threads) (floxer.cpp:93-97; fmindex.hpp:7-10 = BiFMIndex<EprV2_16<6>>).
//
// Layout decisions (MI355X-first, not the reference's EPR layout):
//  * occurrence tables as 128-byte blocks over 256 BWT positions (absolute u32 counts + 3 bit-planes): a rank query is one
//    128-byte line, 0.5 B per text symbol and direction (hg38: 1.6 GB per direction instead of the reference's ~5 GB);
//  * the FULL suffix array as u32 (4 B per symbol, 12.4 GB for hg38 out of 288 GB HBM) instead of a sampled one:
//    locate() becomes one 4-byte gather with no LF walk. The result is identical to BiFMIndex::locate because both return
//    SA[row] split into (sequence, offset).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>

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
    u64 const nb = n / 256 + 1;
    blocks.assign(nb, OccBlock{});
    u32 cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (u64 b = 0; b < nb; ++b) {
        OccBlock& blk = blocks[b];
        for (int c = 0; c < 6; ++c) blk.w[(c >> 1) * 8 + (c & 1) * 4 + 3] = cnt[c];
        for (int j = 0; j < 8; ++j) {                   // 32-position chunk j lives in quarter j/2, half j%2
            u32 p0 = 0, p1 = 0, p2 = 0;
            for (int k = 0; k < 32; ++k) {
                u64 const pos = b * 256 + (u64)j * 32 + k;
                u8 const sym = pos < n ? bwt[pos] : 7;
                p0 |= (u32)(sym & 1) << k;
                p1 |= (u32)((sym >> 1) & 1) << k;
                p2 |= (u32)((sym >> 2) & 1) << k;
                if (pos < n) cnt[sym]++;
            }
            u32* q = &blk.w[(j >> 1) * 8 + (j & 1) * 4];
            q[0] = p0; q[1] = p1; q[2] = p2;
        }
    }
}

// rank of all symbols at position i on the host (same block layout the device reads)
void host_rank_all(const std::vector<OccBlock>& tab, u64 i, u64 out[6]) {
    OccBlock const& b = tab[i >> 8];
    u32 const off = (u32)(i & 255);
    for (int c = 0; c < 6; ++c) out[c] = b.w[(c >> 1) * 8 + (c & 1) * 4 + 3];
    for (u32 j = 0; j < 8; ++j) {
        if (off <= j * 32) break;
        u32 const take = std::min<u32>(32, off - j * 32);
        u32 const mask = take == 32 ? ~0u : ((1u << take) - 1u);
        const u32* q = &b.w[(j >> 1) * 8 + (j & 1) * 4];
        for (u32 k = 0; k < 32; ++k) {
            if (!((mask >> k) & 1)) continue;
            u32 const sym = ((q[0] >> k) & 1) | (((q[1] >> k) & 1) << 1) | (((q[2] >> k) & 1) << 2);
            if (sym < 6) out[sym]++;
        }
    }
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
    u64 off = 0;
    for (u32 r = 0; r < n_refs; ++r) {
        idx->seq_start.push_back(idx->text.size());
        idx->seq_len.push_back(lens[r]);
        for (u64 i = 0; i < lens[r]; ++i) {
            u8 const c = concat[off + i];
            if (c > 5) { set_error("reference rank > 5"); return nullptr; }
            idx->text.push_back(c);
        }
        off += lens[r];
        idx->text.resize(idx->text.size() + (sampling - lens[r] % sampling), 0);
    }
    idx->n = idx->text.size();
    if (idx->n == 0) { set_error("empty reference"); return nullptr; }
    if (idx->n >= ((u64)1 << 32) - 512) { set_error("text of 2^32 symbols or more is not supported by this build"); return nullptr; }
    u64 const n = idx->n;
    auto sa_of = [&](const std::vector<u8>& t, std::vector<u32>& out) {
        if (hip_device < 0) { out = suffix_array(t); return true; }
        out.resize(t.size());
        int const e = DeviceApi::suffix_array(hip_device, t.data(), t.size(), out.data());
        if (e) { set_error("suffix array construction on the device failed (HIP error " + std::to_string(e) + ")"); return false; }
        return true;
    };
    if (!sa_of(idx->text, idx->sa)) return nullptr;
    idx->bwt[0].resize(n);
    for (u64 i = 0; i < n; ++i) idx->bwt[0][i] = idx->text[(idx->sa[i] + n - 1) % n];
    {
        std::vector<u8> rev(idx->text.rbegin(), idx->text.rend());
        std::vector<u32> sa_rev;
        if (!sa_of(rev, sa_rev)) return nullptr;
        idx->bwt[1].resize(n);
        for (u64 i = 0; i < n; ++i) idx->bwt[1][i] = rev[(sa_rev[i] + n - 1) % n];
    }
    u64 cnt[6] = {0, 0, 0, 0, 0, 0};
    for (u8 c : idx->text) cnt[c]++;
    for (int c = 0; c < 6; ++c) idx->C[c + 1] = idx->C[c] + cnt[c];
    build_occ_blocks(idx->bwt[0], idx->occ[0]);
    build_occ_blocks(idx->bwt[1], idx->occ[1]);
    build_kmer_table(*idx);
    return idx.release();
}

// ---------------------------------------------------------------- own index file format (replaces the cereal archive)
namespace {
constexpr char MAGIC[8] = {'F', 'L', 'X', 'I', 'D', 'X', '0', '3'};
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
    auto idx = std::make_unique<HostIndex>();
    char magic[8];
    bool ok = fread(magic, 8, 1, f) == 1 && memcmp(magic, MAGIC, 8) == 0 && fread(&idx->n, 8, 1, f) == 1 &&
              fread(idx->C, 8, 7, f) == 7 && rd(f, idx->text) && rd(f, idx->seq_start) && rd(f, idx->seq_len) && rd(f, idx->sa) &&
              rd(f, idx->occ[0]) && rd(f, idx->occ[1]) && rd(f, idx->bwt[0]) && rd(f, idx->bwt[1]) && rd(f, idx->kmer_table);
    fclose(f);
    if (!ok || idx->text.size() != idx->n || idx->sa.size() != idx->n) { set_error("index file is corrupt or of another version"); return nullptr; }
    return idx.release();
}

}  // namespace flx
