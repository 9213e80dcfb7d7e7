#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include <chrono>
// The literal-only deflate encoder of the BAM writer (flx_io.cpp: literal_deflate) against zlib's inflate: every block must come back
// byte for byte - empty and tiny blocks, one symbol, all 256 symbols, skewed distributions whose Huffman trees are deeper than the 15
// bits deflate allows, incompressible data (stored block), full-size blocks. Test infrastructure; the encoder is pulled in by including
// the source (it lives in an anonymous namespace).
#include "../floxer_amd/csrc/flx_io.cpp"
static bool roundtrip(const std::vector<uint8_t>& in) {
    std::vector<uint8_t> out(in.size() + 1024);
    size_t const c = literal_deflate(in.data(), in.size(), out.data());
    std::vector<uint8_t> back(in.size() + 16);
    z_stream zs; memset(&zs, 0, sizeof(zs));
    inflateInit2(&zs, -15);
    zs.next_in = out.data(); zs.avail_in = (uInt)c; zs.next_out = back.data(); zs.avail_out = (uInt)back.size();
    int rc = inflate(&zs, Z_FINISH);
    bool ok = rc == Z_STREAM_END && zs.total_out == in.size() && memcmp(back.data(), in.data(), in.size()) == 0 && zs.avail_in == 0;
    inflateEnd(&zs);
    if (!ok) printf("FAIL n=%zu rc=%d out=%lu c=%zu\n", in.size(), rc, zs.total_out, c);
    return ok;
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
