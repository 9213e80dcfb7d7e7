#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include <chrono>
// The deflate encoders of the BAM writer (flx_io.cpp: lz_deflate, the default: hash-chainless string matching + one dynamic-Huffman block;
// literal_deflate: literals only) against zlib's inflate: every block must come back
// byte for byte - empty and tiny blocks, one symbol, all 256 symbols, skewed distributions whose Huffman trees are deeper than the 15
// bits deflate allows, incompressible data (stored block), full-size blocks. Test infrastructure; the encoder is pulled in by including
// the source (it lives in an anonymous namespace).
#include "../floxer_amd/csrc/flx_io.cpp"
static bool roundtrip_with(const std::vector<uint8_t>& in, bool lz) {
    std::vector<uint8_t> out(in.size() + 2048);
    size_t const c = lz ? lz_deflate(in.data(), in.size(), out.data()) : literal_deflate(in.data(), in.size(), out.data());
    if (c > in.size() + 600) { printf("FAIL n=%zu: %zu bytes out\n", in.size(), c); return false; }
    std::vector<uint8_t> back(in.size() + 16);
    z_stream zs; memset(&zs, 0, sizeof(zs));
    inflateInit2(&zs, -15);
    zs.next_in = out.data(); zs.avail_in = (uInt)c; zs.next_out = back.data(); zs.avail_out = (uInt)back.size();
    int rc = inflate(&zs, Z_FINISH);
    bool ok = rc == Z_STREAM_END && zs.total_out == in.size() && memcmp(back.data(), in.data(), in.size()) == 0 && zs.avail_in == 0;
    inflateEnd(&zs);
    if (!ok) printf("FAIL (%s) n=%zu rc=%d out=%lu c=%zu\n", lz ? "lz" : "literal", in.size(), rc, zs.total_out, c);
    return ok;
}
// lz_deflate with hints: the stream must come back whatever the hints say (true repeats, stretches declared free of repeats, wrong ones)
static bool roundtrip_hinted(const std::vector<uint8_t>& in, const std::vector<LzHint>& hints, size_t* clen = nullptr) {
    std::vector<uint8_t> out(in.size() + 2048);
    size_t const c = lz_deflate(in.data(), in.size(), out.data(), hints.data(), hints.size());
    if (clen) *clen = c;
    std::vector<uint8_t> back(in.size() + 16);
    z_stream zs; memset(&zs, 0, sizeof(zs));
    inflateInit2(&zs, -15);
    zs.next_in = out.data(); zs.avail_in = (uInt)c; zs.next_out = back.data(); zs.avail_out = (uInt)back.size();
    int rc = inflate(&zs, Z_FINISH);
    bool ok = rc == Z_STREAM_END && zs.total_out == in.size() && memcmp(back.data(), in.data(), in.size()) == 0 && zs.avail_in == 0;
    inflateEnd(&zs);
    if (!ok) printf("FAIL (hinted, %zu hints) n=%zu rc=%d out=%lu c=%zu\n", hints.size(), in.size(), rc, zs.total_out, c);
    return ok;
}
static bool roundtrip(const std::vector<uint8_t>& in) { return roundtrip_with(in, false) & roundtrip_with(in, true); }
// BAM-like payload: records of a fixed head + a CIGAR array of `words` words; `repeat` consecutive records share the array
static std::vector<uint8_t> bam_like(std::mt19937& rng, size_t n, unsigned words, unsigned repeat) {
    std::vector<uint8_t> v;
    std::vector<uint32_t> cig;
    unsigned k = 0;
    while (v.size() < n) {
        if (k++ % repeat == 0) { cig.clear(); for (unsigned w = 0; w < words; ++w) cig.push_back(w % 2 ? ((1u << 4) | (rng() % 3 == 0 ? 8 : rng() % 2 + 1)) : (((rng() % 23) + 1) << 4 | 7)); }
        uint8_t head[36 + 21]; for (auto& b : head) b = (uint8_t)rng();
        // (record sizes: a multiple of eight - a repeat sits at exactly the table's stride -, odd, and anything)
        v.insert(v.end(), head, head + (words == 1600 ? (repeat == 40 ? 40 : 57) : 36 + words % 5));
        const uint8_t* p = (const uint8_t*)cig.data(); v.insert(v.end(), p, p + cig.size() * 4);
    }
    v.resize(n);
    return v;
}
int main() {
    std::mt19937 rng(1);
    bool ok = true;
    for (size_t n : {0ul, 1ul, 2ul, 3ul, 7ul, 255ul, 256ul, 1000ul, 65279ul, 65280ul}) {
        for (int kind = 0; kind < 6; ++kind) {
            std::vector<uint8_t> v(n);
            for (size_t i = 0; i < n; ++i) {
                switch (kind) {
                    case 0: v[i] = 0; break;
                    case 1: v[i] = (uint8_t)rng(); break;
                    case 2: v[i] = (uint8_t)(i % 3 == 0 ? rng() % 7 : 0); break;
                    case 3: v[i] = (uint8_t)(i & 255); break;
                    case 4: { unsigned r = rng() % 65536; unsigned s = 0; while (r & 1) { r >>= 1; ++s; } v[i] = (uint8_t)s; } break;   // geometric: deep tree
                    case 5: { double u = (rng() % 1000000) / 1e6; v[i] = (uint8_t)(u * u * u * u * u * u * 255); } break;
                }
            }
            ok &= roundtrip(v);
        }
    }
    // fibonacci-like frequencies force lengths > 15
    { std::vector<uint8_t> v; unsigned a = 1, b = 1; for (unsigned s = 0; s < 24 && v.size() < 65000; ++s) { for (unsigned k = 0; k < a && v.size() < 65280; ++k) v.push_back((uint8_t)s); unsigned t = a + b; a = b; b = t; } ok &= roundtrip(v); }
    // repeats: records that share their CIGAR array (default flags), at distances below and above the 32-KB window, runs longer than 258,
    // overlapping matches (runs of one byte / one word), a repeat that ends at the block's last byte
    for (unsigned words : {3u, 70u, 1600u, 9000u}) for (unsigned rep : {1u, 2u, 40u}) for (size_t n : {500ul, 65280ul}) ok &= roundtrip(bam_like(rng, n, words, rep));
    { std::vector<uint8_t> v(65280, 7); ok &= roundtrip(v); for (size_t i = 0; i < v.size(); ++i) v[i] = (uint8_t)("ACGT"[i % 4]); ok &= roundtrip(v); }
    { std::vector<uint8_t> v(40000); for (auto& b : v) b = (uint8_t)rng(); std::vector<uint8_t> w = v; w.insert(w.end(), v.begin(), v.begin() + 25280); ok &= roundtrip(w); }
    for (int r = 0; r < 200; ++r) {                                       // random mixtures of literal stretches and copies
        std::vector<uint8_t> v;
        size_t const n = 64 + rng() % 65217;
        while (v.size() < n) {
            if (v.size() > 8 && rng() % 3) { size_t const d = 1 + rng() % std::min<size_t>(v.size(), 40000), l = 3 + rng() % 700; for (size_t k = 0; k < l && v.size() < n; ++k) v.push_back(v[v.size() - d]); }
            else { size_t const l = 1 + rng() % 300; for (size_t k = 0; k < l && v.size() < n; ++k) v.push_back((uint8_t)(rng() % (1 + rng() % 255))); }
        }
        ok &= roundtrip(v);
    }
    // hints (flx_sam_write tells the encoder which CIGAR arrays repeat the one before and which are new): records of 40 + 6400 bytes that share
    // their array; all hints true, no hints, literal hints over true repeats (larger, still valid), wrong distances and ranges past the end
    {
        std::vector<uint8_t> b = bam_like(rng, 65280, 1600, 40);
        size_t const rec = 40 + 6400;
        std::vector<LzHint> good, lits, wrong;
        for (size_t r = 0; r * rec + rec <= b.size(); ++r) {
            uint32_t const at = (uint32_t)(r * rec + 40);
            good.push_back(LzHint{at, 6400u, r ? (uint32_t)rec : 0u});
            lits.push_back(LzHint{at, 6400u, 0u});
            wrong.push_back(LzHint{at + 3, 6400u, r ? (uint32_t)(rec - 4 * (r % 3)) : 0u});
        }
        wrong.push_back(LzHint{65000u, 5000u, 100u});          // runs past the end of the block
        size_t c_plain = 0, c_good = 0, c_lits = 0, c_wrong = 0;
        ok &= roundtrip_hinted(b, {}, &c_plain) & roundtrip_hinted(b, good, &c_good) & roundtrip_hinted(b, lits, &c_lits) & roundtrip_hinted(b, wrong, &c_wrong);
        if (c_good > c_plain + 96 || c_lits < 4 * c_good) { printf("FAIL hinted sizes: plain %zu, true hints %zu, literal hints %zu, wrong hints %zu\n", c_plain, c_good, c_lits, c_wrong); ok = false; }
        for (int r = 0; r < 100; ++r) {                        // random hints over random mixtures
            std::vector<uint8_t> v = bam_like(rng, 2000 + rng() % 63000, 1 + rng() % 2000, 1 + rng() % 5);
            std::vector<LzHint> h;
            for (uint32_t at = rng() % 500; at + 64 < v.size();) { uint32_t const l = 1 + rng() % 9000; h.push_back(LzHint{at, l, rng() % 3 ? (uint32_t)(rng() % 40000) : 0u}); at += l + rng() % 3000; }
            ok &= roundtrip_hinted(v, h);
        }
        auto t0h = std::chrono::steady_clock::now();
        for (int r = 0; r < 2000; ++r) lz_deflate(b.data(), b.size(), std::vector<uint8_t>(70000).data(), good.data(), good.size());
        double const sh = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0h).count();
        printf("records sharing a CIGAR: 40  lz+hints %6.0f MB/s  ratio %6.2f (no hints: %.2f)\n", 2000 * 65280 / 1e6 / sh, 65280.0 / c_good, 65280.0 / c_plain);
    }
    // the block checksum by carry-less multiplication against zlib's, all lengths around the 16- and 64-byte steps, odd starts
    {
        std::vector<uint8_t> v(70000); for (auto& x : v) x = (uint8_t)rng();
        bool crc_ok = true;
        for (size_t len = 0; len < 700; ++len) for (size_t off : {0ul, 1ul, 5ul}) crc_ok &= block_crc32(v.data() + off, len) == (uint32_t)crc32(0, v.data() + off, (uInt)len);
        for (size_t len : {4096ul, 65279ul, 65280ul, 65536ul, 69990ul}) crc_ok &= block_crc32(v.data() + 3, len) == (uint32_t)crc32(0, v.data() + 3, (uInt)len);
        if (!crc_ok) { printf("FAIL block_crc32\n"); ok = false; }
    }
    // speed and size on BAM-like data: one record per read (-I) and forty records sharing a CIGAR (default flags), against zlib level 1
    for (unsigned rep : {1u, 40u, 41u}) {
        std::vector<uint8_t> b = bam_like(rng, 65280, 1600, rep);
        std::vector<uint8_t> o(70000);
        for (int enc = 0; enc < 3; ++enc) {
            auto t0 = std::chrono::steady_clock::now(); size_t c = 0;
            int const reps = enc == 2 ? 300 : 2000;
            for (int r = 0; r < reps; ++r) {
                if (enc == 0) c = lz_deflate(b.data(), b.size(), o.data());
                else if (enc == 1) c = literal_deflate(b.data(), b.size(), o.data());
                else { uLongf dl = (uLongf)o.size(); compress2(o.data(), &dl, b.data(), (uLong)b.size(), 1); c = dl; }
            }
            double const s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("records sharing a CIGAR: %2u  %-8s %6.0f MB/s  ratio %6.2f\n", rep, enc == 0 ? "lz" : enc == 1 ? "literal" : "zlib -1", reps * 65280 / 1e6 / s, 65280.0 / c);
        }
    }
    // speed on CIGAR-like data
    std::vector<uint8_t> buf(65280);
    for (size_t i = 0; i + 4 <= buf.size(); i += 4) { uint32_t w = (i / 4) % 2 ? ((1u << 4) | (rng() % 3 == 0 ? 8 : rng() % 2 + 1)) : (((rng() % 23) + 1) << 4 | 7); memcpy(&buf[i], &w, 4); }
    std::vector<uint8_t> out(70000);
    auto t0 = std::chrono::steady_clock::now(); size_t c = 0;
    for (int r = 0; r < 2000; ++r) c = literal_deflate(buf.data(), buf.size(), out.data());
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%s; literal_deflate: %.0f MB/s, ratio %.2f\n", ok ? "bgzf_check ok" : "FAILURES", 2000 * 65280 / 1e6 / s, 65280.0 / c);
    return ok ? 0 : 1;
}
