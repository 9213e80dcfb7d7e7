// AddressSanitizer / UBSan driver for the HIP-free host sources of the product (GPU ASan is not available on the pool, SURVEY.md
// section 5): PEX trees, expanded search schemes, host index construction (SA-IS) + save / load / corrupt files, the simulator, the
// statistics object, the SAM / BAM writer. Built and run by tests/test_host_cpu.py::test_sanitizers (make -C tests/sanitize).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../floxer_amd/csrc/flx_internal.hpp"
#include "../../floxer_amd/csrc/flx_stats.hpp"

using namespace flx;

// the one device entry point the host sources reference; never reached here (hip_device < 0 everywhere)
int DeviceApi::index_arrays(int, const u8*, u64, u32*, u8*, u8*, OccBlock*, OccBlock*) { return 1; }

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "sanitize_host: check failed: %s (line %d)\n", #x, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    std::string const tmp = argc > 1 ? argv[1] : "/tmp";
    // ---- PEX trees and schemes over a range of shapes
    for (u64 len : {12ull, 30ull, 97ull, 1000ull, 10007ull})
        for (u64 k : {0ull, 1ull, 3ull, 14ull, 80ull})
            for (int bottom_up = 0; bottom_up < 2; ++bottom_up) {
                if (k >= len) continue;
                PexTree const t = build_pex_tree(len, k, 2, bottom_up != 0);
                CHECK(!t.leaves.empty());
                for (auto const& l : t.leaves) { CHECK(l.to < len && l.from <= l.to); (void)expanded_scheme(std::min<u32>(l.num_errors, 3u), l.to - l.from + 1); }
            }
    // ---- simulator, index, file round trip, corrupt files
    std::vector<u8> genome(60000);
    CHECK(flx_sim_genome(genome.size(), 7, genome.data()) == FLX_OK);
    for (size_t i = 1000; i < 1100; ++i) genome[i] = 5;
    u64 lens[3] = {40000, 19999, 1};
    std::vector<u64> offs(51);
    std::vector<u8> pool(50 * 1100);
    std::vector<u32> chrom(50);
    std::vector<u64> pos(50);
    std::vector<u8> rev(50);
    u64 sim_lens[2] = {40000, 19999};
    CHECK(flx_sim_reads(genome.data(), sim_lens, 2, 50, 1000, 0.08, 0.5, 3, pool.data(), pool.size(), offs.data(), chrom.data(), pos.data(), rev.data()) == FLX_OK);
    CHECK(offs[50] <= pool.size());
    flx_index* idx = nullptr;
    CHECK(flx_index_build(genome.data(), lens, 3, &idx) == FLX_OK);
    CHECK(flx_index_num_references(idx) == 3);
    std::vector<u8> meta(4096);
    u64 mlen = meta.size();
    CHECK(flx_index_meta_export(idx, meta.data(), &mlen) == FLX_OK);
    flx_index* light = nullptr;
    CHECK(flx_index_meta_import(meta.data(), mlen, &light) == FLX_OK);
    flx_index_free(light);
    std::string const path = tmp + "/sanitize.idx";
    CHECK(flx_index_save(idx, path.c_str()) == FLX_OK);
    flx_index* loaded = nullptr;
    CHECK(flx_index_load(path.c_str(), &loaded) == FLX_OK);
    CHECK(flx_index_matches_reference(loaded, genome.data(), lens, 3) == FLX_OK);
    flx_index_free(loaded);
    {   // truncated and bit-flipped copies must be refused, never crash
        FILE* f = fopen(path.c_str(), "rb");
        std::vector<u8> bytes;
        u8 buf[65536];
        for (size_t n; (n = fread(buf, 1, sizeof(buf), f)) > 0;) bytes.insert(bytes.end(), buf, buf + n);
        fclose(f);
        for (size_t cut : {(size_t)5, (size_t)100, bytes.size() / 3, bytes.size() - 7}) {
            std::string const p2 = tmp + "/sanitize_cut.idx";
            f = fopen(p2.c_str(), "wb"); fwrite(bytes.data(), 1, cut, f); fclose(f);
            flx_index* bad = nullptr;
            CHECK(flx_index_load(p2.c_str(), &bad) != FLX_OK);
        }
        for (size_t at : {(size_t)8, (size_t)16, (size_t)72, (size_t)80}) {       // n, C[0], a length word
            std::vector<u8> b2 = bytes;
            b2[at + 6] ^= 0x40;
            std::string const p2 = tmp + "/sanitize_flip.idx";
            f = fopen(p2.c_str(), "wb"); fwrite(b2.data(), 1, b2.size(), f); fclose(f);
            flx_index* bad = nullptr;
            int const rc = flx_index_load(p2.c_str(), &bad);
            if (rc == FLX_OK) flx_index_free(bad);
        }
    }
    // ---- statistics
    flx_stats* st = nullptr;
    CHECK(flx_stats_create("simulated", &st) == FLX_OK);
    u64 n = 0;
    (void)flx_stats_format(st, 1, nullptr, &n);
    std::vector<char> text(n);
    CHECK(flx_stats_format(st, 1, text.data(), &n) == FLX_OK);
    CHECK(flx_stats_format(st, 0, text.data(), &n) == FLX_ERR_CAPACITY || n <= text.size());
    flx_stats_free(st);
    // ---- writer
    for (const char* ext : {".sam", ".bam"}) {
        std::string const out = tmp + "/sanitize" + ext;
        const char* ref_ids[2] = {"chromosome_0", "chromosome_1"};
        flx_sam_writer* w = nullptr;
        CHECK(flx_sam_open(out.c_str(), ref_ids, sim_lens, 2, &w) == FLX_OK);
        CHECK(flx_sam_set_threads(w, 3) == FLX_OK);
        std::vector<std::string> names, quals;
        std::vector<const char*> idp, qp;
        for (int r = 0; r < 50; ++r) { names.push_back("read_" + std::to_string(r)); quals.push_back(std::string(offs[r + 1] - offs[r], 'I')); }
        for (int r = 0; r < 50; ++r) { idp.push_back(names[r].c_str()); qp.push_back(quals[r].c_str()); }
        std::vector<flx_record> recs;
        std::vector<u32> cig;
        for (int r = 0; r < 50; ++r) {
            u32 const L = (u32)(offs[r + 1] - offs[r]);
            recs.push_back(flx_record{(u64)r, rev[r] ? 16u : 0u, (int32_t)chrom[r], (int32_t)pos[r], 80, cig.size(), 3, 0});
            cig.push_back((L / 2) << 4 | 7); cig.push_back(1u << 4 | 8); cig.push_back((L - L / 2 - 1) << 4 | 7);
            recs.push_back(flx_record{(u64)r, 256u, (int32_t)chrom[r], (int32_t)pos[r] + 5, 99, cig.size() - 3, 3, 0});
            if (r % 10 == 0) recs.push_back(flx_record{(u64)r, 4u, -1, 0, 0, 0, 0, 0});
        }
        CHECK(flx_sam_write(w, idp.data(), pool.data(), offs.data(), qp.data(), recs.data(), recs.size(), cig.data()) == FLX_OK);
        CHECK(flx_sam_close(w) == FLX_OK);
    }
    flx_index_free(idx);
    printf("sanitize_host ok\n");
    return 0;
}
