// SAM / BAM record writer with the reference's record conventions (output.cpp:49-108, 197-212; seqan3::sam_file_output):
//   header @HD VN:1.6 + one @SQ per reference; MAPQ 255; RNEXT * / PNEXT 0 / TLEN 0; NM tag on mapped records;
//   the primary record carries the forward read + qualities, secondary records have empty SEQ/QUAL;
//   unmapped records: flag 4, RNAME *, no tags. Format chosen by extension (output.hpp:33-38). BGZF via zlib.
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "flx_internal.hpp"

using namespace flx;

struct flx_sam_writer {
    FILE* f = nullptr;
    bool bam = false;
    std::vector<std::string> ref_ids;
    std::vector<uint64_t> ref_lens;
    std::vector<uint8_t> pending;    // BAM: uncompressed bytes that do not fill a BGZF block yet
    unsigned threads = 1;            // record formatting and BGZF compression run on this many threads (flx_sam_set_threads)
    bool failed = false;
};

namespace {

// FLX_WRITER_PROFILE=1: seconds the writer's threads spent formatting records, deflating and on checksums (summed over the threads), on
// stderr when the file is closed
std::atomic<uint64_t> g_ns_format{0}, g_ns_deflate{0}, g_ns_crc{0};
bool writer_profile() { static bool const v = getenv("FLX_WRITER_PROFILE") != nullptr; return v; }
inline uint64_t prof_ns() { return writer_profile() ? (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0; }

constexpr size_t BGZF_BLOCK = 0xff00;
constexpr size_t BGZF_MAX_OUT = 0x10000 + 64;

// Without --interval-optimization a 10-kb read yields some forty records with a 1600-operation CIGAR each: 6 KB of BAM per
// record, and deflating them is what the end-to-end rate of the CLI is made of. Level 1 by default (FLX_BGZF_LEVEL = 0..9
// overrides): the records differ from zlib's default level by ~15 % in size and by 4x in time (scripts/cli_throughput.sh).
int bgzf_level() {
    static int const v = [] { const char* e = getenv("FLX_BGZF_LEVEL"); int const x = e ? atoi(e) : 1; return x < 0 ? 0 : x > 9 ? 9 : x; }();
    return v;
}

// A deflate stream per I/O thread, made once and reset per block: deflateInit2 allocates and clears a quarter of a megabyte, which
// for 64-KB blocks at level 1 costs as much as the compression and, with all threads in the allocator at once, made eight threads
// slower than one.
struct BlockDeflater {
    z_stream zs;
    bool ready = false;
    ~BlockDeflater() { if (ready) deflateEnd(&zs); }
    z_stream* get() {
        if (ready) return deflateReset(&zs) == Z_OK ? &zs : nullptr;
        memset(&zs, 0, sizeof(zs));
        if (deflateInit2(&zs, bgzf_level(), Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return nullptr;
        ready = true;
        return &zs;
    }
};

// ---- A deflate stream of literals only: one dynamic-Huffman block per BGZF block, no string matching (FLX_BGZF_FAST=1). BAM records of
// long reads are CIGAR arrays (32-bit words whose upper bytes are zero), 4-bit packed bases and qualities; an order-0 code runs at 570 MB/s
// per thread against zlib level 1's 100, and the writer's deflate threads are what bounds the CLI with -I (46.7 k -> 61.6 k reads/s,
// profiles/r03_cli_throughput_fast_bgzf.txt). It gives up the repeats zlib finds between CIGAR words, though: files are half as large
// again (5.6 against 3.7 GB for 2.85 M records with 1600-operation CIGARs), so with default flags, where the output volume is what
// costs, it gains little on a disk that writes 0.6 GB/s. Off by default (FLX_BGZF_FAST=1; see lz_deflate below for the default).
struct BitSink {
    uint8_t* p;
    uint64_t acc = 0;
    unsigned n = 0;                                        // bits held (< 32 between calls)
    explicit BitSink(uint8_t* out) : p(out) {}
    inline void put(uint32_t bits, unsigned count) {       // count <= 32
        acc |= (uint64_t)bits << n;
        n += count;
        if (n >= 32) { memcpy(p, &acc, 4); p += 4; acc >>= 32; n -= 32; }
    }
    uint8_t* finish() { while (n > 0) { *p++ = (uint8_t)acc; acc >>= 8; n = n > 8 ? n - 8 : 0; } return p; }
};
inline uint32_t bit_reverse(uint32_t v, unsigned bits) { uint32_t r = 0; for (unsigned i = 0; i < bits; ++i) { r = (r << 1) | (v & 1u); v >>= 1; } return r; }

// code lengths (<= 15) of an optimal prefix code for freq[0..n): Huffman by the two-queue method over the used symbols, then lengths
// beyond the limit moved down (the longest codes lengthen one another until the Kraft sum fits). At least two symbols are used.
void huffman_lengths(const uint32_t* freq, unsigned n, uint8_t* len) {
    constexpr unsigned MAXL = 15;
    struct Node { uint64_t w; int left, right; };
    unsigned order[320];
    unsigned m = 0;
    for (unsigned i = 0; i < n; ++i) { len[i] = 0; if (freq[i]) order[m++] = i; }
    std::sort(order, order + m, [&](unsigned a, unsigned b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
    Node nodes[640];
    for (unsigned i = 0; i < m; ++i) nodes[i] = Node{freq[order[i]], -1, -1};
    unsigned leaf = 0, inner = m, made = m;
    auto take = [&]() -> int {                             // the lighter of the next leaf and the next inner node
        if (leaf < m && (inner >= made || nodes[leaf].w <= nodes[inner].w)) return (int)leaf++;
        return (int)inner++;
    };
    while (m - leaf + (made - inner) > 1) {
        int const a = take(), b = take();
        nodes[made] = Node{nodes[a].w + nodes[b].w, a, b};
        ++made;
    }
    // depths: parents come after their children, so one pass from the root down
    unsigned depth[640];
    depth[made - 1] = 0;
    for (unsigned i = made; i-- > m;) { depth[nodes[i].left] = depth[i] + 1; depth[nodes[i].right] = depth[i] + 1; }
    unsigned count[64] = {0};
    for (unsigned i = 0; i < m; ++i) ++count[std::min(depth[i], 63u)];
    // enforce the limit: everything deeper than MAXL comes up to MAXL, then codes are lengthened until the Kraft sum is 2^MAXL
    for (unsigned d = MAXL + 1; d < 64; ++d) { count[MAXL] += count[d]; count[d] = 0; }
    uint64_t total = 0;
    for (unsigned d = 1; d <= MAXL; ++d) total += (uint64_t)count[d] << (MAXL - d);
    while (total > (1ull << MAXL)) {
        --count[MAXL];
        for (unsigned d = MAXL - 1; d > 0; --d) if (count[d]) { --count[d]; count[d + 1] += 2; break; }
        --total;
    }
    // the rarest symbols get the longest codes (order is by ascending frequency)
    unsigned at = 0;
    for (unsigned d = MAXL; d > 0; --d) for (unsigned c = 0; c < count[d]; ++c) len[order[at++]] = (uint8_t)d;
}

// raw deflate stream for data[0, n), n <= BGZF_BLOCK, into out (capacity >= n + 600); returns its length
size_t literal_deflate(const uint8_t* data, size_t n, uint8_t* out) {
    if (n == 0) { out[0] = 0x03; out[1] = 0x00; return 2; }          // an empty fixed-Huffman block
    uint32_t f4[4][256];
    memset(f4, 0, sizeof(f4));
    size_t i = 0;
    for (; i + 4 <= n; i += 4) { ++f4[0][data[i]]; ++f4[1][data[i + 1]]; ++f4[2][data[i + 2]]; ++f4[3][data[i + 3]]; }
    for (; i < n; ++i) ++f4[0][data[i]];
    uint32_t freq[257];
    for (unsigned b = 0; b < 256; ++b) freq[b] = f4[0][b] + f4[1][b] + f4[2][b] + f4[3][b];
    freq[256] = 1;                                                    // end of block
    uint8_t len[257];
    huffman_lengths(freq, 257, len);
    uint64_t bits = 3 + 5 + 5 + 4 + 19 * 3 + 258 * 4;
    for (unsigned b = 0; b < 257; ++b) bits += (uint64_t)freq[b] * len[b];
    if ((bits + 7) / 8 >= n + 5) {                                    // does not pay: a stored block
        out[0] = 0x01;
        uint16_t const l = (uint16_t)n, nl = (uint16_t)~l;
        memcpy(out + 1, &l, 2);
        memcpy(out + 3, &nl, 2);
        memcpy(out + 5, data, n);
        return n + 5;
    }
    // canonical codes, stored bit-reversed (deflate packs Huffman codes most significant bit first into a least-significant-bit-first stream)
    uint32_t code[257];
    {
        unsigned bl_count[16] = {0}, next[16];
        for (unsigned b = 0; b < 257; ++b) ++bl_count[len[b]];
        bl_count[0] = 0;
        unsigned c = 0;
        for (unsigned d = 1; d <= 15; ++d) { c = (c + bl_count[d - 1]) << 1; next[d] = c; }
        for (unsigned b = 0; b < 257; ++b) code[b] = len[b] ? bit_reverse(next[len[b]]++, len[b]) : 0;
    }
    BitSink bs(out);
    bs.put(1, 1);                                                     // BFINAL
    bs.put(2, 2);                                                     // BTYPE = dynamic Huffman
    bs.put(0, 5);                                                     // HLIT: 257 literal / length codes
    bs.put(0, 5);                                                     // HDIST: 1 distance code (of zero bits: no distances)
    bs.put(15, 4);                                                    // HCLEN: all 19 code length codes
    // the code length code: lengths 0..15 get 4 bits each (a complete code), the run symbols 16..18 are not used
    static const uint8_t cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    for (unsigned k = 0; k < 19; ++k) bs.put(cl_order[k] >= 16 ? 0u : 4u, 3);
    for (unsigned b = 0; b < 257; ++b) bs.put(bit_reverse(len[b], 4), 4);
    bs.put(bit_reverse(0, 4), 4);                                     // the one distance code: length 0
    uint64_t packed[256];                                             // code | length << 32
    for (unsigned b = 0; b < 256; ++b) packed[b] = code[b] | ((uint64_t)len[b] << 32);
    for (size_t k = 0; k < n; ++k) { uint64_t const e = packed[data[k]]; bs.put((uint32_t)e, (unsigned)(e >> 32)); }
    bs.put(code[256], len[256]);
    return (size_t)(bs.finish() - out);
}

// ---- A deflate stream with string matching of the cheap kind (the default): long repeats only - every eighth position enters a hash
// table of sixteen-byte strings, every position is looked up, a hit is extended both ways eight bytes at a time and taken from 24 bytes
// on - then one dynamic-Huffman block. What it is for: without
// --interval-optimization the windows of one locus all take their union's alignment, so a read's forty records carry the same 6-8 KB
// CIGAR array one after the other - a back-reference of a few bytes instead of 8 KB of literals (zlib level 1 finds the same repeats
// at 100 MB/s per thread, and the writer's deflate threads were what bounded the CLI at default flags: 4.9 k reads/s, 12.1 of 13.4 s).
// Inside a CIGAR array (words (length << 4 | op), three zero bytes in four) it finds nothing and costs a hash per byte: the order-0 code
// of the literals does the rest. FLX_BGZF_ZLIB=1 selects zlib, FLX_BGZF_FAST=1 the literal-only encoder.
enum BgzfEncoder { ENC_LZ = 0, ENC_ZLIB = 1, ENC_LITERAL = 2 };
BgzfEncoder bgzf_encoder() {
    static BgzfEncoder const v = getenv("FLX_BGZF_ZLIB") ? ENC_ZLIB : getenv("FLX_BGZF_FAST") ? ENC_LITERAL : ENC_LZ;
    return v;
}
struct LzToken { uint16_t lit_run; uint16_t len; uint16_t dist; };    // lit_run literals, then a match of len (3..258; 0: none) at dist
// deflate's length and distance codes (RFC 1951, 3.2.5)
struct LenCode { uint16_t code; uint8_t extra_bits; uint16_t base; };
inline void length_code(unsigned len, unsigned& code, unsigned& ebits, unsigned& eval) {
    static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t bits[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    unsigned c = 28;
    if (len < 258) { c = 0; while (c + 1 < 28 && base[c + 1] <= len) ++c; }
    code = 257 + c; ebits = bits[c]; eval = len - base[c];
}
inline void distance_code(unsigned dist, unsigned& code, unsigned& ebits, unsigned& eval) {
    static const uint16_t base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t bits[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    unsigned c = 0;
    while (c + 1 < 30 && base[c + 1] <= dist) ++c;
    code = c; ebits = bits[c]; eval = dist - base[c];
}
void canonical_codes(const uint8_t* len, unsigned n, uint32_t* code) {
    unsigned bl_count[16] = {0}, next[16];
    for (unsigned b = 0; b < n; ++b) ++bl_count[len[b]];
    bl_count[0] = 0;
    unsigned c = 0;
    for (unsigned d = 1; d <= 15; ++d) { c = (c + bl_count[d - 1]) << 1; next[d] = c; }
    for (unsigned b = 0; b < n; ++b) code[b] = len[b] ? bit_reverse(next[len[b]]++, len[b]) : 0;
}

// What the caller knows about a stretch of the block. dist > 0, a repeat: data[pos, pos + len) == data[pos - dist, pos - dist + len) (the
// CIGAR array of a record that shares it with the record before); checked before it is used, a wrong one is ignored. dist == 0, nothing to
// find: the stretch repeats nothing in front of it and nothing behind it will want it (a CIGAR array seen for the first time in this block: the
// records that share it are hinted themselves): it is coded as literals without a look at the table. Sorted by pos, no overlaps.
struct LzHint { uint32_t pos, len, dist; };

// raw deflate stream for data[0, n), n <= BGZF_BLOCK, into out (capacity >= n + 1200); returns its length
size_t lz_deflate(const uint8_t* data, size_t n, uint8_t* out, const LzHint* hints = nullptr, size_t n_hints = 0) {
    if (n < 64) return literal_deflate(data, n, out);
    // ---- parse
    constexpr unsigned HASH_BITS = 13;
    uint16_t table[1u << HASH_BITS];                      // position + 1 of the last occurrence of a hash (0: none); positions < 65536
    memset(table, 0, sizeof(table));
    thread_local std::vector<LzToken> tokens;
    tokens.clear();
    uint32_t lfreq[286], dfreq[30];
    memset(lfreq, 0, sizeof(lfreq));
    memset(dfreq, 0, sizeof(dfreq));
    auto read64 = [&](size_t i) { uint64_t v; memcpy(&v, data + i, 8); return v; };
    size_t i = 0, lit_start = 0;
    size_t const last = n - 16;                            // positions from which sixteen bytes can be read
    auto flush_literals = [&](size_t upto) {
        size_t run = upto - lit_start;
        while (run > 65535) { tokens.push_back(LzToken{65535, 0, 0}); run -= 65535; }
        return (uint16_t)run;
    };
    // (sixteen bytes per table entry: two CIGAR words repeat all over an array, four rarely do - the table keeps one position per hash, and
    // it has to be the one in the record before)
    auto hash8 = [&](size_t k) { return (uint32_t)(((read64(k) * 0x9E3779B97F4A7C15ull) ^ (read64(k + 8) * 0xC2B2AE3D27D4EB4Full)) >> (64 - HASH_BITS)); };
    constexpr size_t MIN_MATCH = 32;                       // shorter repeats (the zero bytes of CIGAR words) are left to the entropy code
    size_t matched = 0;                                    // bytes covered by matches so far
    size_t next_insert = 0;                                // every eighth position enters the table: a repeat of MIN_MATCH bytes holds two of them
    auto emit_match = [&](size_t at, size_t len, unsigned dist) {
        uint16_t const run = flush_literals(at);
        // a long match goes out as pieces of at most 258 bytes (never leaving a piece shorter than 3)
        unsigned dcode, deb, dev;
        distance_code(dist, dcode, deb, dev);
        size_t left = len;
        bool first = true;
        while (left) {
            size_t const piece = left > 258 ? (left - 258 < 3 ? 255 : 258) : left;
            tokens.push_back(LzToken{first ? run : (uint16_t)0, (uint16_t)piece, (uint16_t)dist});
            unsigned code, eb, ev;
            length_code((unsigned)piece, code, eb, ev);
            ++lfreq[code];
            ++dfreq[dcode];
            left -= piece;
            first = false;
        }
        lit_start = at + len;
        matched += len;
    };
    size_t hint_at = 0;
    while (i <= last) {
        // a hinted repeat that starts here (or that the scan has walked into): taken whole, and nothing of it enters the table - the record
        // after it is hinted too, or finds the literal copy at the start of the block
        while (hint_at < n_hints && (size_t)hints[hint_at].pos + hints[hint_at].len <= i) ++hint_at;
        if (hint_at < n_hints && hints[hint_at].pos <= i) {
            LzHint const& h = hints[hint_at];
            size_t const end = std::min<size_t>((size_t)h.pos + h.len, n);
            if (h.dist == 0) {
                i = end;
                next_insert = (i + 7) & ~(size_t)7;
                ++hint_at;
                continue;
            }
            if (h.dist >= 1 && h.dist <= 32768 && h.dist <= i && end - i >= MIN_MATCH && memcmp(data + i, data + i - h.dist, end - i) == 0) {
                emit_match(i, end - i, h.dist);
                // (a position in 64 still enters the table: a repeat nobody announced - the same array again after another one - finds this copy,
                // and is extended back to its start from wherever it is hit)
                for (size_t k = (i + 63) & ~(size_t)63; k < end && k <= last; k += 64) table[hash8(k)] = (uint16_t)(k + 1);
                i = end;
                next_insert = (i + 7) & ~(size_t)7;
                ++hint_at;
                continue;
            }
            ++hint_at;
        }
        while (next_insert < i) { if (next_insert <= last) table[hash8(next_insert)] = (uint16_t)(next_insert + 1); next_insert += 8; }
        // (looked up before it enters the table itself: records whose size is a multiple of eight repeat at exactly the table's positions)
        uint32_t const h = hash8(i);
        size_t const cand1 = table[h];
        if (next_insert == i) { table[h] = (uint16_t)(i + 1); next_insert += 8; }
        size_t c = cand1 ? cand1 - 1 : 0;
        // (four words of a CIGAR array do repeat here and there: a candidate has to hold for 32 bytes before it is looked at any closer)
        if (cand1 && c < i && i - c <= 32768 && i + 32 <= n && read64(c) == read64(i) && read64(c + 8) == read64(i + 8) && read64(c + 16) == read64(i + 16) &&
            read64(c + 24) == read64(i + 24)) {
            size_t len = 32;
            while (i + len + 8 <= n && read64(c + len) == read64(i + len)) len += 8;
            while (i + len < n && data[c + len] == data[i + len]) ++len;
            size_t at = i;
            while (at > lit_start && c > 0 && data[c - 1] == data[at - 1]) { --c; --at; ++len; }      // the match began before the table's position
            if (len >= MIN_MATCH) {
                emit_match(at, len, (unsigned)(at - c));
                i = at + len;
                continue;
            }
        }
        // (every position is looked up: only every eighth is in the table, and a stride would miss the repeats whose distance it does not divide)
        ++i;
        if (i == 16384 && matched < 4096) return literal_deflate(data, n, out);      // a quarter of the block and little to show: records that share nothing (-I)
    }
    uint16_t const tail_run = flush_literals(n);
    // literal frequencies: everything outside the matches
    {
        // (four tables: three of four CIGAR bytes are zero, and one counter incremented byte after byte is a chain of store-to-load forwards)
        uint32_t h4[4][256];
        memset(h4, 0, sizeof(h4));
        auto count = [&](const uint8_t* p, size_t len) {
            size_t k = 0;
            for (; k + 4 <= len; k += 4) { ++h4[0][p[k]]; ++h4[1][p[k + 1]]; ++h4[2][p[k + 2]]; ++h4[3][p[k + 3]]; }
            for (; k < len; ++k) ++h4[0][p[k]];
        };
        size_t pos = 0;
        for (auto const& t : tokens) {
            count(data + pos, t.lit_run);
            pos += (size_t)t.lit_run + t.len;
        }
        count(data + pos, tail_run);
        for (unsigned b = 0; b < 256; ++b) lfreq[b] += h4[0][b] + h4[1][b] + h4[2][b] + h4[3][b];
    }
    lfreq[256] = 1;
    bool any_dist = false;
    for (unsigned d = 0; d < 30; ++d) any_dist |= dfreq[d] != 0;
    if (!any_dist) return literal_deflate(data, n, out);
    // (a Huffman code needs two symbols)
    { unsigned used = 0; for (unsigned d = 0; d < 30; ++d) used += dfreq[d] != 0; if (used == 1) { for (unsigned d = 0; d < 30; ++d) if (!dfreq[d]) { dfreq[d] = 1; break; } } }
    uint8_t llen[286], dlen[30];
    huffman_lengths(lfreq, 286, llen);
    huffman_lengths(dfreq, 30, dlen);
    uint64_t bits = 3 + 5 + 5 + 4 + 19 * 3 + (286 + 30) * 4;
    for (unsigned b = 0; b < 286; ++b) bits += (uint64_t)lfreq[b] * llen[b];
    for (unsigned d = 0; d < 30; ++d) bits += (uint64_t)dfreq[d] * (dlen[d] + 13);      // (extra bits: an upper bound)
    for (unsigned c = 265; c < 285; ++c) bits += (uint64_t)lfreq[c] * 5;
    if ((bits + 7) / 8 >= n + 5) return literal_deflate(data, n, out);                  // (it falls back to a stored block itself)
    uint32_t lcode[286], dcode[30];
    canonical_codes(llen, 286, lcode);
    canonical_codes(dlen, 30, dcode);
    BitSink bs(out);
    bs.put(1, 1);                                                     // BFINAL
    bs.put(2, 2);                                                     // BTYPE = dynamic Huffman
    bs.put(286 - 257, 5);                                             // HLIT
    bs.put(30 - 1, 5);                                                // HDIST
    bs.put(15, 4);                                                    // HCLEN: all 19 code length codes
    static const uint8_t cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    for (unsigned k = 0; k < 19; ++k) bs.put(cl_order[k] >= 16 ? 0u : 4u, 3);
    for (unsigned b = 0; b < 286; ++b) bs.put(bit_reverse(llen[b], 4), 4);
    for (unsigned d = 0; d < 30; ++d) bs.put(bit_reverse(dlen[d], 4), 4);
    uint64_t packed[256];
    for (unsigned b = 0; b < 256; ++b) packed[b] = lcode[b] | ((uint64_t)llen[b] << 32);
    size_t pos = 0;
    for (auto const& t : tokens) {
        for (size_t k = 0; k < t.lit_run; ++k) { uint64_t const e = packed[data[pos + k]]; bs.put((uint32_t)e, (unsigned)(e >> 32)); }
        pos += t.lit_run;
        if (t.len) {
            unsigned code, eb, ev;
            length_code(t.len, code, eb, ev);
            bs.put(lcode[code], llen[code]);
            if (eb) bs.put(ev, eb);
            distance_code(t.dist, code, eb, ev);
            bs.put(dcode[code], dlen[code]);
            if (eb) bs.put(ev, eb);
            pos += t.len;
        }
    }
    for (size_t k = 0; k < tail_run; ++k) { uint64_t const e = packed[data[pos + k]]; bs.put((uint32_t)e, (unsigned)(e >> 32)); }
    bs.put(lcode[256], llen[256]);
    return (size_t)(bs.finish() - out);
}

// ---- CRC-32 of a block by carry-less multiplication (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ
// Instruction", Intel 2009; bit-reflected form for the gzip polynomial): four 128-bit lanes are folded 64 bytes ahead per step
// (x^(512+64) and x^512 mod P), the lanes folded into one (x^(128+64), x^128), then 128 -> 64 -> 32 bits (x^96, x^64; Barrett with
// floor(x^64 / P) and P). zlib 1.2.11's table code runs at 0.9 GB/s and was a third of a writer thread's time at default flags.
#if defined(__x86_64__)
#define FLX_CLMUL __attribute__((target("pclmul,sse4.1")))
FLX_CLMUL inline __m128i clmul_load(const uint8_t* p) { return _mm_loadu_si128(reinterpret_cast<const __m128i*>(p)); }
FLX_CLMUL inline __m128i clmul_fold(__m128i x, __m128i k, __m128i next) {
    return _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x, k, 0x00), _mm_clmulepi64_si128(x, k, 0x11)), next);
}
FLX_CLMUL uint32_t crc32_clmul(const uint8_t* buf, size_t len, uint32_t state) {      // len >= 64, a multiple of 16; state: the inverted CRC
    __m128i const k_512 = _mm_set_epi64x(0x01c6e41596, 0x0154442bd4);        // low: x^(512+64) mod P, high: x^512 mod P
    __m128i const k_128 = _mm_set_epi64x(0x00ccaa009e, 0x01751997d0);        // low: x^(128+64) mod P, high: x^128 mod P
    __m128i const k_64 = _mm_set_epi64x(0, 0x0163cd6124);                    // x^64 mod P
    __m128i const k_poly = _mm_set_epi64x(0x01f7011641, 0x01db710641);       // low: P, high: floor(x^64 / P)
#define load clmul_load
#define fold clmul_fold
    __m128i x1 = _mm_xor_si128(load(buf), _mm_cvtsi32_si128((int)state)), x2 = load(buf + 16), x3 = load(buf + 32), x4 = load(buf + 48);
    buf += 64; len -= 64;
    while (len >= 64) {
        x1 = fold(x1, k_512, load(buf));
        x2 = fold(x2, k_512, load(buf + 16));
        x3 = fold(x3, k_512, load(buf + 32));
        x4 = fold(x4, k_512, load(buf + 48));
        buf += 64; len -= 64;
    }
    x1 = fold(x1, k_128, x2);
    x1 = fold(x1, k_128, x3);
    x1 = fold(x1, k_128, x4);
    while (len >= 16) { x1 = fold(x1, k_128, load(buf)); buf += 16; len -= 16; }
    __m128i const mask32 = _mm_setr_epi32(~0, 0, ~0, 0);
    __m128i t = _mm_clmulepi64_si128(x1, k_128, 0x10);                       // 128 -> 96 bits
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
    t = _mm_srli_si128(x1, 4);                                               // 96 -> 64
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), k_64, 0x00), t);
    t = _mm_and_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), k_poly, 0x10), mask32);      // Barrett
    t = _mm_clmulepi64_si128(t, k_poly, 0x00);
    return (uint32_t)_mm_extract_epi32(_mm_xor_si128(x1, t), 1);
#undef load
#undef fold
}
#endif
uint32_t block_crc32(const uint8_t* data, size_t len) {
#if defined(__x86_64__)
    static bool const have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1") && !getenv("FLX_CRC_ZLIB");
    if (have && len >= 64) {
        size_t const head = len & ~(size_t)15;
        uint32_t const c = ~crc32_clmul(data, head, 0xFFFFFFFFu);
        return head == len ? c : (uint32_t)crc32(c, data + head, (uInt)(len - head));
    }
#endif
    return (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)len);
}

// one BGZF block (a gzip member with the BC extra field) for `len` <= BGZF_BLOCK bytes; returns its size, 0 on failure.
// hints: repeats the caller knows of (lz_deflate)
size_t bgzf_compress_block(const uint8_t* data, size_t len, uint8_t* out, const LzHint* hints = nullptr, size_t n_hints = 0) {
    size_t clen;
    BgzfEncoder const enc = bgzf_encoder();
    uint64_t const t_deflate = prof_ns();
    if (enc == ENC_LZ) clen = lz_deflate(data, len, out + 18, hints, n_hints);
    else if (enc == ENC_LITERAL) clen = literal_deflate(data, len, out + 18);
    else {
        thread_local BlockDeflater deflater;
        z_stream* const zp = deflater.get();
        if (!zp) return 0;
        z_stream& zs = *zp;
        zs.next_in = const_cast<Bytef*>(data);
        zs.avail_in = (uInt)len;
        zs.next_out = out + 18;
        zs.avail_out = (uInt)(BGZF_MAX_OUT - 18 - 8);
        int const rc = deflate(&zs, Z_FINISH);
        clen = zs.total_out;
        if (rc != Z_STREAM_END) return 0;
    }
    size_t const bsize = clen + 18 + 8;
    static const uint8_t hdr[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
    memcpy(out, hdr, 16);
    out[16] = (uint8_t)((bsize - 1) & 0xff);
    out[17] = (uint8_t)((bsize - 1) >> 8);
    uint64_t const t_crc = prof_ns();
    uint32_t const crc = block_crc32(data, len);
    if (writer_profile()) { g_ns_deflate += t_crc - t_deflate; g_ns_crc += prof_ns() - t_crc; }
    uint32_t const isize = (uint32_t)len;
    memcpy(out + 18 + clen, &crc, 4);
    memcpy(out + 18 + clen + 4, &isize, 4);
    return bsize;
}

template <class F>
void io_parallel(size_t n, unsigned threads, F&& body) {          // body(i) for i in [0, n), each index once
    unsigned const t = (unsigned)std::min<size_t>(threads, n);
    if (t <= 1) { for (size_t i = 0; i < n; ++i) body(i); return; }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (unsigned k = 0; k < t; ++k) pool.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < n;) body(i); });
    for (auto& th : pool) th.join();
}

// the full blocks of w->pending (all of it with `all`), compressed in parallel, written in order
bool bgzf_flush(flx_sam_writer* w, bool all) {
    size_t const n_full = w->pending.size() / BGZF_BLOCK, tail = w->pending.size() - n_full * BGZF_BLOCK;
    size_t const n_blocks = n_full + ((all && tail) ? 1 : 0);
    if (n_blocks == 0) return true;
    std::unique_ptr<uint8_t[]> const out(new uint8_t[n_blocks * BGZF_MAX_OUT]);      // (not cleared: every block is written before it is read)
    std::vector<size_t> sizes(n_blocks, 0);
    io_parallel(n_blocks, w->threads, [&](size_t b) {
        size_t const len = b < n_full ? BGZF_BLOCK : tail;
        sizes[b] = bgzf_compress_block(w->pending.data() + b * BGZF_BLOCK, len, out.get() + b * BGZF_MAX_OUT);
    });
    bool ok = true;
    for (size_t b = 0; b < n_blocks && ok; ++b) ok = sizes[b] != 0 && fwrite(out.get() + b * BGZF_MAX_OUT, 1, sizes[b], w->f) == sizes[b];
    size_t const used = all ? w->pending.size() : n_full * BGZF_BLOCK;
    w->pending.erase(w->pending.begin(), w->pending.begin() + (long)used);
    return ok;
}

bool emit(flx_sam_writer* w, const void* data, size_t len) {
    if (!w->bam) return fwrite(data, 1, len, w->f) == len;
    const uint8_t* p = (const uint8_t*)data;
    w->pending.insert(w->pending.end(), p, p + len);
    return bgzf_flush(w, false);
}

template <class T> void put(std::vector<uint8_t>& v, T x) { const uint8_t* p = (const uint8_t*)&x; v.insert(v.end(), p, p + sizeof(T)); }

int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

std::string header_text(flx_sam_writer const* w) {
    std::string h = "@HD\tVN:1.6\n";
    for (size_t i = 0; i < w->ref_ids.size(); ++i) h += "@SQ\tSN:" + w->ref_ids[i] + "\tLN:" + std::to_string(w->ref_lens[i]) + "\n";
    return h;
}

}  // namespace

extern "C" int flx_sam_open(const char* path, const char* const* ref_ids, const uint64_t* ref_lens, uint32_t n_refs, flx_sam_writer** out) {
    if (!path || !out || (n_refs && (!ref_ids || !ref_lens))) { set_error("flx_sam_open: null argument"); return FLX_ERR_INVALID; }
    std::string const p(path);
    bool bam;
    if (p.size() >= 4 && p.compare(p.size() - 4, 4, ".bam") == 0) bam = true;
    else if (p.size() >= 4 && p.compare(p.size() - 4, 4, ".sam") == 0) bam = false;
    else { set_error("output file must end in .sam or .bam (floxer_cli.cpp:258)"); return FLX_ERR_INVALID; }
    FILE* f = fopen(path, "wb");
    if (!f) { set_error(std::string("cannot open output file: ") + path); return FLX_ERR_IO; }
    auto* w = new flx_sam_writer();
    w->f = f;
    w->bam = bam;
    for (uint32_t i = 0; i < n_refs; ++i) { w->ref_ids.emplace_back(ref_ids[i]); w->ref_lens.push_back(ref_lens[i]); }
    std::string const text = header_text(w);
    bool ok;
    if (!bam) ok = emit(w, text.data(), text.size());
    else {
        std::vector<uint8_t> h;
        h.insert(h.end(), {'B', 'A', 'M', 1});
        put<int32_t>(h, (int32_t)text.size());
        h.insert(h.end(), text.begin(), text.end());
        put<int32_t>(h, (int32_t)n_refs);
        for (uint32_t i = 0; i < n_refs; ++i) {
            put<int32_t>(h, (int32_t)w->ref_ids[i].size() + 1);
            h.insert(h.end(), w->ref_ids[i].begin(), w->ref_ids[i].end());
            h.push_back(0);
            put<int32_t>(h, (int32_t)w->ref_lens[i]);
        }
        ok = emit(w, h.data(), h.size());
    }
    if (!ok) { fclose(f); delete w; set_error("write error"); return FLX_ERR_IO; }
    *out = w;
    return FLX_OK;
}

namespace {
// one record as SAM text or as a BAM record, appended to `out`; false: the record cannot be represented (error set)
// cigar_at (BAM): where in `out` the record's CIGAR array starts (SIZE_MAX: it has none of its own)
bool format_record(flx_sam_writer const* w, flx_record const& r, const char* const* read_ids, const uint8_t* read_pool,
                   const uint64_t* read_offsets, const char* const* quals, const uint32_t* cigar_words, std::vector<uint8_t>& out, std::string& err,
                   size_t* cigar_at = nullptr) {
    static const char ops[] = "MIDNSHP=X";
    const char* id = read_ids[r.read_index];
    bool const unmapped = (r.flag & 4u) != 0;
    bool const with_seq = unmapped || !(r.flag & 256u);        // primary or unmapped carry SEQ/QUAL (output.cpp:69-72, 97-105)
    const uint8_t* seq = read_pool + read_offsets[r.read_index];
    uint64_t const slen = with_seq ? read_offsets[r.read_index + 1] - read_offsets[r.read_index] : 0;
    const char* qual = (with_seq && quals) ? quals[r.read_index] : nullptr;
    const uint32_t* cig = cigar_words ? cigar_words + r.cigar_offset : nullptr;
    if (!w->bam) {
        char num[24];
        auto app = [&](const char* p, size_t n) { out.insert(out.end(), p, p + n); };
        auto app_num = [&](long long v) {                        // (a CIGAR is a thousand numbers per record)
            unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
            char* e = num + sizeof(num), * p = e;
            do { *--p = (char)('0' + u % 10); u /= 10; } while (u);
            if (v < 0) *--p = '-';
            app(p, (size_t)(e - p));
        };
        app(id, strlen(id)); out.push_back('\t'); app_num(r.flag); out.push_back('\t');
        if (unmapped) out.push_back('*'); else app(w->ref_ids[(size_t)r.reference_id].data(), w->ref_ids[(size_t)r.reference_id].size());
        out.push_back('\t');
        app_num((long long)r.position + 1);                      // seqan3 writes ref_offset + 1 (also for the 0 floxer passes when unmapped)
        app("\t255\t", 5);
        if (r.cigar_length == 0) out.push_back('*');
        else for (uint32_t c = 0; c < r.cigar_length; ++c) { app_num(cig[c] >> 4); out.push_back((uint8_t)ops[cig[c] & 15]); }
        app("\t*\t0\t0\t", 7);
        if (slen == 0) out.push_back('*');
        else { size_t const at = out.size(); out.resize(at + slen); for (uint64_t b = 0; b < slen; ++b) out[at + b] = (uint8_t)rank_to_char(seq[b]); }
        out.push_back('\t');
        if (slen == 0 || !qual || !*qual) out.push_back('*');
        else app(qual, slen);
        if (!unmapped) { app("\tNM:i:", 6); app_num(r.num_errors); }
        out.push_back('\n');
        return true;
    }
    size_t const l_name = strlen(id) + 1;
    if (l_name > 255) { err = std::string("read name longer than 254 characters cannot be written to BAM: ") + id; return false; }
    // (the forty records of a read at one locus share one CIGAR array: its span is summed once)
    thread_local const uint32_t* span_of = nullptr;
    thread_local uint32_t span_len = 0;
    thread_local int64_t span = 0;
    if (span_of != cig || span_len != r.cigar_length || !cig) {
        span = 0;
        for (uint32_t c = 0; c < r.cigar_length; ++c) {
            uint32_t const op = cig[c] & 15, len = cig[c] >> 4;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += len;
        }
        span_of = cig; span_len = r.cigar_length;
    }
    int64_t const ref_span = span;
    // more than 65535 operations do not fit n_cigar_op: the record carries kSmN and the real CIGAR in the CG:B,I tag (SAM spec 4.2.2)
    bool const long_cigar = r.cigar_length > 65535u;
    size_t const start = out.size();
    auto put32 = [&](int32_t x) { const uint8_t* p = (const uint8_t*)&x; out.insert(out.end(), p, p + 4); };
    auto put16 = [&](uint16_t x) { const uint8_t* p = (const uint8_t*)&x; out.insert(out.end(), p, p + 2); };
    int32_t const pos = r.position;
    put32(0);                                                  // block_size, patched below
    put32(unmapped ? -1 : r.reference_id);
    put32(pos);
    out.push_back((uint8_t)l_name);
    out.push_back(255);
    put16((uint16_t)reg2bin(pos, pos + (ref_span ? ref_span : 1)));
    put16((uint16_t)(long_cigar ? 2u : r.cigar_length));
    put16((uint16_t)r.flag);
    put32((int32_t)slen);
    put32(-1);
    put32(-1);
    put32(0);
    out.insert(out.end(), id, id + l_name);
    // (the placeholder's S length is the stored SEQ length, 0 for a secondary record written without SEQ: that is what htslib's
    // bam_tag2cigar and seqan3 compare it with)
    if (cigar_at) *cigar_at = SIZE_MAX;
    if (long_cigar) { put32((int32_t)(((uint32_t)slen << 4) | 4u)); put32((int32_t)(((uint32_t)ref_span << 4) | 3u)); }
    else {
        size_t const at = out.size();
        if (r.cigar_length) out.insert(out.end(), reinterpret_cast<const uint8_t*>(cig), reinterpret_cast<const uint8_t*>(cig + r.cigar_length));
        if (cigar_at && r.cigar_length) *cigar_at = at;
    }
    if (slen) {
        static const uint8_t nib[6] = {15, 1, 2, 4, 8, 15};   // =ACMGRSVTWYHKDBN codes for $ACGTN
        size_t const at = out.size(), packed = (size_t)((slen + 1) / 2);
        out.resize(at + packed + slen);
        uint8_t* const sp = out.data() + at;
        for (uint64_t b = 0; b + 1 < slen; b += 2) sp[b >> 1] = (uint8_t)(nib[seq[b] < 6 ? seq[b] : 5] << 4 | nib[seq[b + 1] < 6 ? seq[b + 1] : 5]);
        if (slen & 1) sp[slen >> 1] = (uint8_t)(nib[seq[slen - 1] < 6 ? seq[slen - 1] : 5] << 4);
        uint8_t* const qp = sp + packed;
        if (qual && *qual) for (uint64_t b = 0; b < slen; ++b) qp[b] = (uint8_t)(qual[b] - 33);
        else memset(qp, 0xff, slen);
    }
    if (!unmapped) {
        out.push_back('N'); out.push_back('M');
        if (r.num_errors < 256) { out.push_back('C'); out.push_back((uint8_t)r.num_errors); }
        else if (r.num_errors < 65536) { out.push_back('S'); put16((uint16_t)r.num_errors); }
        else { out.push_back('I'); put32((int32_t)r.num_errors); }
    }
    if (long_cigar) {
        out.push_back('C'); out.push_back('G'); out.push_back('B'); out.push_back('I');
        put32((int32_t)r.cigar_length);
        for (uint32_t c = 0; c < r.cigar_length; ++c) put32((int32_t)cig[c]);
    }
    int32_t const bs = (int32_t)(out.size() - start) - 4;
    memcpy(out.data() + start, &bs, 4);
    return true;
}
}  // namespace

extern "C" int flx_sam_set_threads(flx_sam_writer* w, uint32_t n_threads) {
    if (!w) { set_error("null writer"); return FLX_ERR_INVALID; }
    w->threads = std::max(1u, std::min(n_threads, 64u));
    return FLX_OK;
}

extern "C" int flx_sam_write(flx_sam_writer* w, const char* const* read_ids, const uint8_t* read_pool, const uint64_t* read_offsets,
                             const char* const* quals, const flx_record* records, uint64_t n_records, const uint32_t* cigar_words) {
    if (!w || (n_records && (!records || !read_ids || !read_pool || !read_offsets))) { set_error("flx_sam_write: null argument"); return FLX_ERR_INVALID; }
    if (w->failed) { set_error("write error on the alignment output"); return FLX_ERR_IO; }
    if (w->bam) {
        // BAM: a worker formats a fixed number of records at a time into a small buffer and deflates every 64 KB of it into BGZF blocks as
        // it goes (the last block of a part is short: BGZF blocks need not be full), so the uncompressed records - 280 KB per read at default
        // flags, 18 GB for 65 536 reads - never leave the cache; the parts' blocks are written in order. Parts are cut by record count, not
        // by thread count: the file's bytes do not depend on --threads. (Round 3 formatted a whole batch, copied it into one pending
        // buffer and compressed that: three passes over gigabytes of freshly faulted-in memory, 12 of the CLI's 13 s at default flags.)
        if (!w->pending.empty() && !bgzf_flush(w, true)) { w->failed = true; set_error("write error on the alignment output"); return FLX_ERR_IO; }
        constexpr uint64_t PART_RECORDS = 256;
        size_t const n_parts = (size_t)((n_records + PART_RECORDS - 1) / PART_RECORDS);
        std::vector<std::vector<uint8_t>> parts(n_parts);
        std::vector<std::string> errs(n_parts);
        io_parallel(n_parts, w->threads, [&](size_t p) {
            uint64_t const r0 = (uint64_t)p * PART_RECORDS, r1 = std::min<uint64_t>(n_records, r0 + PART_RECORDS);
            thread_local std::vector<uint8_t> raw;
            raw.clear();
            std::vector<uint8_t>& outv = parts[p];
            size_t done = 0;                                   // bytes of raw already compressed
            size_t base = 0;                                   // bytes of the part's stream that have left raw: stream position = base + index
            // what the deflate encoder is told: a record whose CIGAR array is the one of the record before it (same offset and length in the
            // run's pool) repeats it at the distance of the two arrays; stream positions
            struct StreamHint { size_t pos, len, dist; };
            thread_local std::vector<StreamHint> hints;
            thread_local std::vector<LzHint> block_hints;
            hints.clear();
            size_t hint_first = 0;                             // hints before this one lie wholly in front of `done`
            auto deflate_full_blocks = [&](bool all) {
                while (raw.size() - done >= BGZF_BLOCK || (all && raw.size() > done)) {
                    size_t const len = std::min(BGZF_BLOCK, raw.size() - done);
                    size_t const b0 = base + done, b1 = b0 + len;
                    block_hints.clear();
                    while (hint_first < hints.size() && hints[hint_first].pos + hints[hint_first].len <= b0) ++hint_first;
                    for (size_t h = hint_first; h < hints.size() && hints[h].pos < b1; ++h) {
                        size_t const lo = std::max(hints[h].pos, b0), to = std::min(hints[h].pos + hints[h].len, b1);
                        if (hints[h].dist == 0 || hints[h].dist > 32768) { block_hints.push_back(LzHint{(uint32_t)(lo - b0), (uint32_t)(to - lo), 0u}); continue; }
                        // the part of the repeat whose source lies in this block too; what is in front of it has nothing to repeat here
                        size_t const from = std::min(std::max(lo, b0 + hints[h].dist), to);
                        if (lo < from) block_hints.push_back(LzHint{(uint32_t)(lo - b0), (uint32_t)(from - lo), 0u});
                        if (from < to) block_hints.push_back(LzHint{(uint32_t)(from - b0), (uint32_t)(to - from), (uint32_t)hints[h].dist});
                    }
                    size_t const at = outv.size();
                    outv.resize(at + BGZF_MAX_OUT);
                    size_t const c = bgzf_compress_block(raw.data() + done, len, outv.data() + at, block_hints.data(), block_hints.size());
                    if (c == 0) { errs[p] = "BGZF compression failed"; outv.resize(at); return; }
                    outv.resize(at + c);
                    done += len;
                }
                if (done == raw.size()) { base += raw.size(); raw.clear(); done = 0; }
                else if (done >= (4u << 20)) { base += done; raw.erase(raw.begin(), raw.begin() + (long)done); done = 0; }
            };
            size_t prev_cigar_at = SIZE_MAX;                   // stream position of the previous record's CIGAR array
            for (uint64_t i = r0; i < r1 && errs[p].empty(); ++i) {
                size_t cigar_at = SIZE_MAX;
                uint64_t const t_format = prof_ns();
                if (!format_record(w, records[i], read_ids, read_pool, read_offsets, quals, cigar_words, raw, errs[p], &cigar_at)) break;
                if (writer_profile()) g_ns_format += prof_ns() - t_format;
                if (cigar_at != SIZE_MAX) {
                    cigar_at += base;
                    bool const same = prev_cigar_at != SIZE_MAX && i > r0 && records[i].cigar_offset == records[i - 1].cigar_offset && records[i].cigar_length == records[i - 1].cigar_length;
                    hints.push_back(StreamHint{cigar_at, 4 * (size_t)records[i].cigar_length, same ? cigar_at - prev_cigar_at : 0});
                }
                prev_cigar_at = cigar_at;
                deflate_full_blocks(false);
            }
            if (errs[p].empty()) deflate_full_blocks(true);
        });
        for (auto const& e : errs) if (!e.empty()) { set_error(e); return FLX_ERR_INVALID; }
        for (auto const& part : parts)
            if (!part.empty() && fwrite(part.data(), 1, part.size(), w->f) != part.size()) { w->failed = true; break; }
        if (w->failed) { set_error("write error on the alignment output"); return FLX_ERR_IO; }
        return FLX_OK;
    }
    // SAM: records are formatted in parallel (contiguous ranges, one buffer each), then written in order
    size_t const n_parts = std::max<size_t>(1, std::min<size_t>(w->threads * 4, n_records / 64));
    std::vector<std::vector<uint8_t>> parts(n_parts);
    std::vector<std::string> errs(n_parts);
    io_parallel(n_parts, w->threads, [&](size_t p) {
        uint64_t const r0 = n_records * p / n_parts, r1 = n_records * (p + 1) / n_parts;
        {
            // the part's size, roughly (about 8 bytes per CIGAR operation and 2 per base of a record with SEQ / QUAL), so that the buffer is
            // allocated once instead of doubling its way up through copies
            size_t guess = 0;
            for (uint64_t i = r0; i < r1; ++i) {
                flx_record const& r = records[i];
                bool const with_seq = (r.flag & 4u) || !(r.flag & 256u);
                size_t const slen = with_seq ? (size_t)(read_offsets[r.read_index + 1] - read_offsets[r.read_index]) : 0;
                guess += 96 + 4 * (size_t)r.cigar_length + slen + slen / 2;
            }
            parts[p].reserve(2 * guess);
        }
        for (uint64_t i = r0; i < r1 && errs[p].empty(); ++i)
            if (!format_record(w, records[i], read_ids, read_pool, read_offsets, quals, cigar_words, parts[p], errs[p])) break;
    });
    for (auto const& e : errs) if (!e.empty()) { set_error(e); return FLX_ERR_INVALID; }
    for (auto const& part : parts) {
        if (part.empty()) continue;
        if (fwrite(part.data(), 1, part.size(), w->f) != part.size()) w->failed = true;
        if (w->failed) break;
    }
    if (w->failed) { set_error("write error on the alignment output"); return FLX_ERR_IO; }
    return FLX_OK;
}

extern "C" int flx_sam_close(flx_sam_writer* w) {
    if (!w) return FLX_OK;
    if (writer_profile() && w->bam)
        fprintf(stderr, "[flx writer profile] summed over the I/O threads: formatting %.2f s, deflate %.2f s, checksums %.2f s\n", g_ns_format.load() / 1e9, g_ns_deflate.load() / 1e9, g_ns_crc.load() / 1e9);
    bool ok = !w->failed;
    if (w->bam) {
        ok = bgzf_flush(w, true) && ok;
        uint8_t eof_block[BGZF_MAX_OUT];
        size_t const n = bgzf_compress_block(nullptr, 0, eof_block);   // BGZF EOF marker block
        ok = n != 0 && fwrite(eof_block, 1, n, w->f) == n && ok;
    }
    ok = (fclose(w->f) == 0) && ok;
    delete w;
    if (!ok) { set_error("write error while closing the alignment output"); return FLX_ERR_IO; }
    return FLX_OK;
}
