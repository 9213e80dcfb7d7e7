// simulated_dataset — the reference's evaluation aid (src/main/simulated_dataset.cpp) on top of this build's generator:
//   simulated_dataset create --genomes genome.fasta --reads reads.fastq [-c chromosome-length] [-n num-chromosomes]
//                            [-l read-length] [-m num-reads] [-e error-rate] [-s random-seed]        (:225-331)
//   simulated_dataset verify --alignments alignments.sam|.bam [-p allowed-pos-diff]                  (:383-472)
// `create` writes a uniform genome and reads with floor(error_rate * read_length) mutated positions whose names carry their
// origin (id_{i}_chromosome_{c}_position_{p}_max_errors_{e}, :207-213); `verify` reads an aligner's output and prints, per query,
// whether an alignment within the expected number of errors sits at the simulated position — the accuracy table for synthetic
// reads. The random numbers are this build's (flx_sim_genome / flx_sim_reads), not std::mt19937's (not portable, see there).
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/floxer_amd.h"

namespace {

struct Args {
    std::map<std::string, std::string> kv;
    std::string get(std::string const& long_id, std::string const& short_id, std::string const& def = "") const {
        auto it = kv.find(long_id);
        if (it != kv.end()) return it->second;
        it = kv.find(short_id);
        return it != kv.end() ? it->second : def;
    }
    bool has(std::string const& long_id, std::string const& short_id) const { return kv.count(long_id) || kv.count(short_id); }
};
bool parse_args(int argc, char** argv, int first, Args& a) {
    for (int i = first; i < argc; ++i) {
        std::string k = argv[i];
        if (k.rfind("--", 0) == 0) k = k.substr(2);
        else if (k.rfind("-", 0) == 0) k = k.substr(1);
        else { fprintf(stderr, "[CLI PARSER ERROR]\nunexpected argument %s\n", argv[i]); return false; }
        size_t const eq = k.find('=');
        if (eq != std::string::npos) { a.kv[k.substr(0, eq)] = k.substr(eq + 1); continue; }
        if (i + 1 >= argc) { fprintf(stderr, "[CLI PARSER ERROR]\nmissing value for option %s\n", argv[i]); return false; }
        a.kv[k] = argv[++i];
    }
    return true;
}
bool has_ext(std::string const& p, std::initializer_list<const char*> exts) {
    size_t const dot = p.rfind('.');
    if (dot == std::string::npos) return false;
    std::string const e = p.substr(dot + 1);
    for (auto x : exts) if (e == x) return true;
    return false;
}

int create(int argc, char** argv) {
    Args a;
    if (!parse_args(argc, argv, 2, a)) return -1;
    std::string const genome_path = a.get("genomes", "g"), read_path = a.get("reads", "r");
    if (genome_path.empty() || !has_ext(genome_path, {"fa", "fasta", "fna", "ffn", "fas", "faa", "mpfa", "frn"})) { fprintf(stderr, "[CLI PARSER ERROR]\nOption -g/--genomes is required (fa|fasta|fna|ffn|fas|faa|mpfa|frn).\n"); return -1; }
    if (read_path.empty() || !has_ext(read_path, {"fq", "fastq"})) { fprintf(stderr, "[CLI PARSER ERROR]\nOption -r/--reads is required (fq|fastq).\n"); return -1; }
    uint64_t const chromosome_length = strtoull(a.get("chromosome-length", "c", "50000000").c_str(), nullptr, 10);    // defaults of :233-239
    uint64_t const num_chromosomes = strtoull(a.get("num-chromosomes", "n", "10").c_str(), nullptr, 10);
    uint64_t const base_read_length = strtoull(a.get("read-length", "l", "20000").c_str(), nullptr, 10);
    uint64_t const num_reads = strtoull(a.get("num-reads", "m", "8000").c_str(), nullptr, 10);
    double const error_rate = atof(a.get("error-rate", "e", "0.07").c_str());
    uint64_t const seed = strtoull(a.get("random-seed", "s", "7267281").c_str(), nullptr, 10);
    double const revcomp = atof(a.get("revcomp-fraction", "R", "0").c_str());         // extension: the reference only emits forward reads
    if (num_chromosomes < 1 || num_chromosomes > 50 || chromosome_length > 1000000000ull || base_read_length > 1000000 || num_reads < 1 ||
        error_rate < 0.00001 || error_rate > 0.99999) { fprintf(stderr, "[CLI PARSER ERROR]\nan option value is out of range (simulated_dataset.cpp:256-300)\n"); return -1; }
    if (chromosome_length <= base_read_length) { fprintf(stderr, "[error] Chromomsome length %llu must be larger than read length %llu\n", (unsigned long long)chromosome_length, (unsigned long long)base_read_length); return -1; }

    std::vector<uint8_t> genome(chromosome_length * num_chromosomes);
    if (flx_sim_genome(genome.size(), seed, genome.data()) != FLX_OK) { fprintf(stderr, "[error] %s\n", flx_last_error()); return -1; }
    static const char letters[] = "$ACGTN";
    {
        FILE* f = fopen(genome_path.c_str(), "w");
        if (!f) { fprintf(stderr, "[error] cannot write %s\n", genome_path.c_str()); return -1; }
        std::string line;
        for (uint64_t c = 0; c < num_chromosomes; ++c) {
            fprintf(f, ">chromosome_%llu\n", (unsigned long long)c);
            const uint8_t* s = genome.data() + c * chromosome_length;
            for (uint64_t i = 0; i < chromosome_length; i += 80) {
                uint64_t const n = std::min<uint64_t>(80, chromosome_length - i);
                line.assign(n, 'N');
                for (uint64_t j = 0; j < n; ++j) line[j] = letters[s[i + j]];
                line += '\n';
                fwrite(line.data(), 1, line.size(), f);
            }
        }
        fclose(f);
    }
    std::vector<uint64_t> lens(num_chromosomes, chromosome_length), offsets(num_reads + 1), pos(num_reads);
    std::vector<uint32_t> chrom(num_reads);
    std::vector<uint8_t> rev(num_reads);
    uint64_t const num_errors = (uint64_t)(error_rate * (double)base_read_length);
    std::vector<uint8_t> pool(num_reads * (base_read_length + num_errors) + 1);
    if (flx_sim_reads(genome.data(), lens.data(), (uint32_t)num_chromosomes, num_reads, (uint32_t)base_read_length, error_rate, revcomp, seed + 1,
                      pool.data(), pool.size(), offsets.data(), chrom.data(), pos.data(), rev.data()) != FLX_OK) { fprintf(stderr, "[error] %s\n", flx_last_error()); return -1; }
    FILE* f = fopen(read_path.c_str(), "w");
    if (!f) { fprintf(stderr, "[error] cannot write %s\n", read_path.c_str()); return -1; }
    std::string rec;
    for (uint64_t r = 0; r < num_reads; ++r) {
        uint64_t const n = offsets[r + 1] - offsets[r];
        rec = "@id_" + std::to_string(r) + "_chromosome_" + std::to_string(chrom[r]) + "_position_" + std::to_string(pos[r]) + "_max_errors_" + std::to_string(num_errors) + "\n";
        size_t const at = rec.size();
        rec.resize(at + n);
        for (uint64_t j = 0; j < n; ++j) rec[at + j] = letters[pool[offsets[r] + j]];
        rec += "\n+\n";
        rec.append(n, 'I');
        rec += '\n';
        fwrite(rec.data(), 1, rec.size(), f);
    }
    fclose(f);
    return 0;
}

// ---------------------------------------------------------------- verify
struct Alignment { uint64_t chromosome_id, position, num_errors; };
struct Origin { uint64_t chromosome_id, position, max_num_errors; };

bool parse_query_id(std::string const& id, Origin& o) {                               // :337-362
    std::vector<std::string> parts;
    size_t at = 0;
    while (at <= id.size()) {
        size_t const u = id.find('_', at);
        parts.push_back(id.substr(at, u == std::string::npos ? std::string::npos : u - at));
        if (u == std::string::npos) break;
        at = u + 1;
    }
    if (parts.size() < 9 || parts[0] != "id" || parts[2] != "chromosome" || parts[4] != "position" || parts[6] != "max" || parts[7] != "errors") return false;
    o = Origin{strtoull(parts[3].c_str(), nullptr, 10), strtoull(parts[5].c_str(), nullptr, 10), strtoull(parts[8].c_str(), nullptr, 10)};
    return true;
}
uint64_t parse_chromosome_id(std::string const& name) {                                // :364-371: the number after the first underscore
    size_t const u = name.find('_');
    return u == std::string::npos ? 0 : strtoull(name.c_str() + u + 1, nullptr, 10);
}

struct Collected {
    std::vector<std::string> order;                                                    // query ids in order of first appearance
    std::unordered_map<std::string, std::vector<Alignment>> by_query;
    void add(std::string const& id, Alignment a) {
        auto it = by_query.find(id);
        if (it == by_query.end()) { order.push_back(id); it = by_query.emplace(id, std::vector<Alignment>{}).first; }
        it->second.push_back(a);
    }
};

bool read_sam(std::string const& path, Collected& c, std::string& err) {
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    std::vector<char> buf(1 << 22);
    std::string line;
    auto handle = [&](std::string const& l) {
        if (l.empty() || l[0] == '@') return;
        std::vector<std::string> cols;
        size_t at = 0;
        while (true) { size_t const t = l.find('\t', at); cols.push_back(l.substr(at, t == std::string::npos ? std::string::npos : t - at)); if (t == std::string::npos) break; at = t + 1; }
        if (cols.size() < 11) return;
        unsigned const flag = (unsigned)strtoul(cols[1].c_str(), nullptr, 10);
        if (flag & 4u) return;
        uint64_t nm = 0;
        for (size_t i = 11; i < cols.size(); ++i) if (cols[i].rfind("NM:i:", 0) == 0) nm = strtoull(cols[i].c_str() + 5, nullptr, 10);
        c.add(cols[0], Alignment{parse_chromosome_id(cols[2]), strtoull(cols[3].c_str(), nullptr, 10) - 1, nm});     // 0-based like seqan3's reader
    };
    while (gzgets(f, buf.data(), (int)buf.size())) {
        line += buf.data();
        if (!line.empty() && line.back() == '\n') { line.pop_back(); handle(line); line.clear(); }
    }
    if (!line.empty()) handle(line);
    gzclose(f);
    return true;
}

bool read_bam(std::string const& path, Collected& c, std::string& err) {
    gzFile f = gzopen(path.c_str(), "rb");                                             // BGZF is a series of gzip members
    if (!f) { err = "cannot open " + path; return false; }
    auto rd = [&](void* p, size_t n) { return gzread(f, p, (unsigned)n) == (int)n; };
    char magic[4];
    int32_t l_text = 0, n_ref = 0;
    if (!rd(magic, 4) || memcmp(magic, "BAM\1", 4) != 0 || !rd(&l_text, 4)) { err = "not a BAM file"; gzclose(f); return false; }
    std::vector<char> text((size_t)l_text);
    if (l_text && !rd(text.data(), (size_t)l_text)) { err = "truncated BAM header"; gzclose(f); return false; }
    if (!rd(&n_ref, 4)) { err = "truncated BAM header"; gzclose(f); return false; }
    std::vector<std::string> ref_names;
    for (int32_t i = 0; i < n_ref; ++i) {
        int32_t l_name = 0, l_ref = 0;
        if (!rd(&l_name, 4)) { err = "truncated BAM header"; gzclose(f); return false; }
        std::string name((size_t)l_name, '\0');
        if (!rd(&name[0], (size_t)l_name) || !rd(&l_ref, 4)) { err = "truncated BAM header"; gzclose(f); return false; }
        name.resize(strlen(name.c_str()));
        ref_names.push_back(name);
    }
    std::vector<uint8_t> rec;
    while (true) {
        int32_t block_size = 0;
        int const got = gzread(f, &block_size, 4);
        if (got == 0) break;
        if (got != 4 || block_size < 32) { err = "truncated BAM record"; gzclose(f); return false; }
        rec.resize((size_t)block_size);
        if (!rd(rec.data(), rec.size())) { err = "truncated BAM record"; gzclose(f); return false; }
        int32_t ref_id, pos, l_seq;
        uint16_t n_cigar, flag;
        memcpy(&ref_id, rec.data(), 4); memcpy(&pos, rec.data() + 4, 4);
        uint8_t const l_read_name = rec[8];
        memcpy(&n_cigar, rec.data() + 12, 2); memcpy(&flag, rec.data() + 14, 2); memcpy(&l_seq, rec.data() + 16, 4);
        if (flag & 4u || ref_id < 0 || ref_id >= (int32_t)ref_names.size()) continue;
        std::string const name((const char*)rec.data() + 32);
        size_t at = 32 + (size_t)l_read_name + 4 * (size_t)n_cigar + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
        uint64_t nm = 0;
        while (at + 3 <= rec.size()) {                                                 // tags: NM as C / S / I (flx_io.cpp), others skipped
            char const t0 = (char)rec[at], t1 = (char)rec[at + 1], ty = (char)rec[at + 2];
            at += 3;
            size_t sz = 0;
            if (ty == 'c' || ty == 'C' || ty == 'A') sz = 1; else if (ty == 's' || ty == 'S') sz = 2; else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
            else if (ty == 'Z' || ty == 'H') { sz = strlen((const char*)rec.data() + at) + 1; }
            else if (ty == 'B') { char const sub = (char)rec[at]; uint32_t cnt; memcpy(&cnt, rec.data() + at + 1, 4); size_t const es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4; sz = 5 + es * cnt; }
            else break;
            if (t0 == 'N' && t1 == 'M') { uint32_t v = 0; memcpy(&v, rec.data() + at, std::min<size_t>(sz, 4)); nm = v; }
            at += sz;
        }
        c.add(name, Alignment{parse_chromosome_id(ref_names[(size_t)ref_id]), (uint64_t)pos, nm});
    }
    gzclose(f);
    return true;
}

int verify(int argc, char** argv) {
    Args a;
    if (!parse_args(argc, argv, 2, a)) return -1;
    std::string const path = a.get("alignments", "a");
    if (path.empty()) { fprintf(stderr, "[CLI PARSER ERROR]\nOption -a/--alignments is required but not set.\n"); return -1; }
    uint64_t const allowed_pos_diff = strtoull(a.get("allowed-pos-diff", "p", "0").c_str(), nullptr, 10);
    Collected c;
    std::string err;
    bool const ok = has_ext(path, {"bam"}) ? read_bam(path, c, err) : read_sam(path, c, err);
    if (!ok) { fprintf(stderr, "[error] %s\n", err.c_str()); return -1; }
    // :415-466, literally: both differences start at the largest uint32 and "not found" is tested against the largest size_t, so a
    // query without any alignment on its chromosome is reported as FoundSuboptimal with both differences 4294967295
    printf("queries = [\n");
    uint64_t n_optimal = 0, n_suboptimal = 0, n_not_found = 0;
    for (auto const& id : c.order) {
        Origin origin{};
        if (!parse_query_id(id, origin)) { fprintf(stderr, "[warning] query id %s does not carry an origin\n", id.c_str()); continue; }
        uint64_t pos_diff = std::numeric_limits<uint32_t>::max(), pos_diff_higher = std::numeric_limits<uint32_t>::max();
        for (auto const& al : c.by_query[id]) {
            if (origin.chromosome_id != al.chromosome_id) continue;
            uint64_t const d = std::max(origin.position, al.position) - std::min(origin.position, al.position);
            if (al.num_errors > origin.max_num_errors) pos_diff_higher = std::min(d, pos_diff_higher);
            else pos_diff = std::min(d, pos_diff);
            if (pos_diff == 0) break;
        }
        printf("    { id = \"%s\", status = { ", id.c_str());
        if (pos_diff <= allowed_pos_diff) { printf("FoundOptimal = {}"); ++n_optimal; }
        else if (pos_diff == std::numeric_limits<size_t>::max() && pos_diff_higher == std::numeric_limits<size_t>::max()) { printf("NotFound = {}"); ++n_not_found; }
        else { printf("FoundSuboptimal = { pos_diff_expected_num_errors = %llu, pos_diff_higher_num_errors = %llu }", (unsigned long long)pos_diff, (unsigned long long)pos_diff_higher); ++n_suboptimal; }
        printf(" } },\n");
    }
    printf("]\n");
    // summary on stderr (stdout is the reference's table, nothing else)
    fprintf(stderr, "[info] %zu mapped queries: FoundOptimal %llu, FoundSuboptimal %llu, NotFound %llu (allowed position difference %llu)\n", c.order.size(),
            (unsigned long long)n_optimal, (unsigned long long)n_suboptimal, (unsigned long long)n_not_found, (unsigned long long)allowed_pos_diff);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc >= 2 && std::string(argv[1]) == "create") return create(argc, argv);
    if (argc >= 2 && std::string(argv[1]) == "verify") return verify(argc, argv);
    fprintf(stderr, "usage:\n  simulated_dataset create --genomes genome.fasta --reads reads.fastq [-c N] [-n N] [-l N] [-m N] [-e R] [-s SEED]\n"
                    "  simulated_dataset verify --alignments alignments.sam [-p allowed-pos-diff]\n");
    return -1;
}
