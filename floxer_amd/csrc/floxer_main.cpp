// floxer-compatible command line over libfloxer_amd (drop-in for the process boundary):
//   ./floxer --reference ref.fa --queries reads.fq --error-probability 0.08 --output out.bam   (README.md:35)
// Options, short ids, defaults and validators follow include/floxer_cli.hpp:41-70 and src/lib/floxer_cli.cpp:173-435;
// main() follows src/main/floxer.cpp:35-195. Diagnostics go to stderr only, stdout stays empty
// (floxer_whole_program_via_cli_test.cpp:125-126); exit code 0 on success, 255 (-1) on any error.
#include <zlib.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <future>
#include <memory>
#include <string>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <cerrno>
#include <thread>
#include <vector>

#include "../../include/floxer_amd.h"

namespace {

bool g_debug = false;
FILE* g_logfile = nullptr;

void log_line(const char* level, const char* fmt, ...) {
    char buf[4096];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    bool const is_debug = strcmp(level, "debug") == 0;
    if (!is_debug || g_debug) fprintf(stderr, "[floxer] [%s] %s\n", level, buf);
    if (g_logfile) { fprintf(g_logfile, "[%s] %s\n", level, buf); fflush(g_logfile); }
}

struct Options {
    std::string reference, queries, output, index, logfile;
    bool console_debug_logs = false;
    bool has_query_errors = false, has_error_probability = false;
    uint64_t query_errors = 0;
    double error_probability = NAN;
    uint64_t seed_errors = 2, max_anchors_hard = 500, max_anchors_soft = 50;
    std::string anchor_group_order = "count_first", anchor_choice_strategy = "round_robin";
    uint64_t seed_sampling_step_size = 1;
    bool dont_erase_useless_anchors = false, bottom_up_pex_tree = false, interval_optimization = false;
    double extra_verification_ratio = 0.05;
    bool direct_full_verification = false;
    uint64_t num_anchors_per_task = 3000;
    bool without_cigar = false;
    uint64_t threads = 1, timeout = 0;
    bool has_timeout = false;
    std::string stats, stats_input_hint;
    std::string devices;
};

struct OptDef { char short_id; const char* long_id; bool flag; };
const OptDef OPTS[] = {
    {'r', "reference", false}, {'q', "queries", false}, {'o', "output", false}, {'i', "index", false}, {'l', "logfile", false},
    {'c', "console-debug-logs", true}, {'e', "query-errors", false}, {'p', "error-probability", false}, {'s', "seed-errors", false},
    {'M', "max-anchors-hard", false}, {'m', "max-anchors-soft", false}, {'g', "anchor-group-order", false},
    {'y', "anchor-choice-strategy", false}, {'C', "seed-sampling-step-size", false}, {'E', "dont-erase-useless-anchors", true},
    {'b', "bottom-up-pex-tree", true}, {'I', "interval-optimization", true}, {'v', "extra-verification-ratio", false},
    {'d', "direct-full-verification", true}, {'u', "num-anchors-per-task", false}, {'w', "without-cigar", true}, {'t', "threads", false},
    {'x', "timeout", false}, {'S', "stats", false}, {'H', "stats-input-hint", false},
    // not in the reference: the HIP devices to run on ("0", "0-7", "0,2,4", "all"; default: FLX_DEVICES, else device 0)
    {'G', "devices", false},
};

struct CliError { std::string msg; };

bool ends_with(std::string const& s, std::string const& suf) { return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0; }
bool has_ext(std::string const& path, std::vector<std::string> const& exts, bool allow_gz) {
    for (auto const& e : exts) {
        if (ends_with(path, "." + e)) return true;
        if (allow_gz && ends_with(path, "." + e + ".gz")) return true;
    }
    return false;
}
uint64_t parse_u64(std::string const& name, std::string const& v) {
    char* end = nullptr;
    if (v.empty() || v[0] == '-') throw CliError{"Value parse failed for --" + name + ": Argument " + v + " could not be parsed as type unsigned integer."};
    unsigned long long const x = strtoull(v.c_str(), &end, 10);
    if (*end) throw CliError{"Value parse failed for --" + name + ": Argument " + v + " could not be parsed as type unsigned integer."};
    return x;
}
double parse_double(std::string const& name, std::string const& v) {
    char* end = nullptr;
    double const x = strtod(v.c_str(), &end);
    if (v.empty() || *end) throw CliError{"Value parse failed for --" + name + ": Argument " + v + " could not be parsed as type double."};
    return x;
}
void range_check(std::string const& name, double v, double lo, double hi) {
    if (v < lo || v > hi) throw CliError{"Validation failed for option --" + name + ": Value " + std::to_string(v) + " is not in range [" + std::to_string(lo) + "," + std::to_string(hi) + "]."};
}
// (by permission, not by opening: opening and closing a FIFO would break the pipe under its writer before the reader gets to it)
bool file_readable(std::string const& p) { return access(p.c_str(), R_OK) == 0; }

Options parse_cli(int argc, char** argv) {
    Options o;
    bool seen_ref = false, seen_q = false, seen_out = false;
    for (int a = 1; a < argc; ++a) {
        std::string arg = argv[a];
        const OptDef* def = nullptr;
        std::string inline_value;
        bool has_inline = false;
        if (arg.size() > 2 && arg[0] == '-' && arg[1] == '-') {
            std::string name = arg.substr(2);
            size_t const eq = name.find('=');
            if (eq != std::string::npos) { inline_value = name.substr(eq + 1); name = name.substr(0, eq); has_inline = true; }
            for (auto const& d : OPTS) if (name == d.long_id) def = &d;
        } else if (arg.size() == 2 && arg[0] == '-') {
            for (auto const& d : OPTS) if (arg[1] == d.short_id) def = &d;
        }
        if (arg == "-h" || arg == "--help") {
            fprintf(stderr, "floxer (MI355X-native path) - usage: ./floxer --reference hg38.fasta --queries reads.fastq --error-probability 0.07 --output mapped_reads.bam\n");
            for (auto const& d : OPTS) fprintf(stderr, "  -%c, --%s%s\n", d.short_id, d.long_id, d.flag ? "" : " <value>");
            exit(0);
        }
        if (arg == "--version") { fprintf(stderr, "%s\n", flx_version()); exit(0); }
        if (!def) throw CliError{"Unknown option " + arg + ". In case this is meant to be a non-option/argument/parameter, please specify the start of non-options with '--'."};
        std::string value;
        if (!def->flag) {
            if (has_inline) value = inline_value;
            else { if (a + 1 >= argc) throw CliError{std::string("Missing value for option --") + def->long_id}; value = argv[++a]; }
        }
        std::string const n = def->long_id;
        if (n == "reference") { o.reference = value; seen_ref = true; }
        else if (n == "queries") { o.queries = value; seen_q = true; }
        else if (n == "output") { o.output = value; seen_out = true; }
        else if (n == "index") o.index = value;
        else if (n == "logfile") o.logfile = value;
        else if (n == "console-debug-logs") o.console_debug_logs = true;
        else if (n == "query-errors") { o.query_errors = parse_u64(n, value); range_check(n, (double)o.query_errors, 0, 4096); o.has_query_errors = true; }
        else if (n == "error-probability") { o.error_probability = parse_double(n, value); range_check(n, o.error_probability, 0.00001, 0.99999); o.has_error_probability = true; }
        else if (n == "seed-errors") { o.seed_errors = parse_u64(n, value); range_check(n, (double)o.seed_errors, 0, 3); }
        else if (n == "max-anchors-hard") o.max_anchors_hard = parse_u64(n, value);
        else if (n == "max-anchors-soft") o.max_anchors_soft = parse_u64(n, value);
        else if (n == "anchor-group-order") {
            if (value != "count_first" && value != "errors_first" && value != "none") throw CliError{"Validation failed for option --" + n + ": Value " + value + " is not one of [count_first,errors_first,none]."};
            o.anchor_group_order = value;
        } else if (n == "anchor-choice-strategy") {
            if (value != "round_robin" && value != "full_groups" && value != "first_reported") throw CliError{"Validation failed for option --" + n + ": Value " + value + " is not one of [round_robin,full_groups,first_reported]."};
            o.anchor_choice_strategy = value;
        } else if (n == "seed-sampling-step-size") o.seed_sampling_step_size = parse_u64(n, value);
        else if (n == "dont-erase-useless-anchors") o.dont_erase_useless_anchors = true;
        else if (n == "bottom-up-pex-tree") o.bottom_up_pex_tree = true;
        else if (n == "interval-optimization") o.interval_optimization = true;
        else if (n == "extra-verification-ratio") o.extra_verification_ratio = parse_double(n, value);
        else if (n == "direct-full-verification") o.direct_full_verification = true;
        else if (n == "num-anchors-per-task") { o.num_anchors_per_task = parse_u64(n, value); if (o.num_anchors_per_task < 1) throw CliError{"Validation failed for option --" + n + ": must be at least 1."}; }
        else if (n == "without-cigar") o.without_cigar = true;
        else if (n == "threads") { o.threads = parse_u64(n, value); range_check(n, (double)o.threads, 1, 4096); }
        else if (n == "timeout") { o.timeout = parse_u64(n, value); o.has_timeout = true; }
        else if (n == "stats") o.stats = value;
        else if (n == "devices") o.devices = value;
        else if (n == "stats-input-hint") {
            if (value != "real_nanopore" && value != "simulated") throw CliError{"Validation failed for option --" + n + ": Value " + value + " is not one of [real_nanopore,simulated]."};
            o.stats_input_hint = value;
        }
    }
    if (!seen_ref) throw CliError{"Option -r/--reference is required but not set."};
    if (!seen_q) throw CliError{"Option -q/--queries is required but not set."};
    if (!seen_out) throw CliError{"Option -o/--output is required but not set."};
    if (!has_ext(o.reference, {"fa", "fasta", "fna", "ffn", "fas", "faa", "mpfa", "frn"}, true)) throw CliError{"Validation failed for option -r/--reference: Expected one of the following valid extensions: [fa,fasta,fna,ffn,fas,faa,mpfa,frn](.gz)."};
    if (!file_readable(o.reference)) throw CliError{"Validation failed for option -r/--reference: The file " + o.reference + " does not exist!"};
    if (!has_ext(o.queries, {"fq", "fastq"}, true)) throw CliError{"Validation failed for option -q/--queries: Expected one of the following valid extensions: [fq,fastq](.gz)."};
    if (!file_readable(o.queries)) throw CliError{"Validation failed for option -q/--queries: The file " + o.queries + " does not exist!"};
    if (!has_ext(o.output, {"bam", "sam"}, false)) throw CliError{"Validation failed for option -o/--output: Expected one of the following valid extensions: [bam,sam]."};
    // cross validation, floxer_cli.cpp:173-204
    if (!o.has_query_errors && !o.has_error_probability) throw CliError{"Either a fixed number of errors in the query or an error probability must be given."};
    if (!o.has_error_probability && o.query_errors < o.seed_errors)
        throw CliError{"The number of errors per query (" + std::to_string(o.query_errors) + ") must be greater or equal than the number of errors in the PEX tree leaves (" + std::to_string(o.seed_errors) + ")."};
    if (o.max_anchors_hard < o.max_anchors_soft)
        throw CliError{"The hard maximum number of anchors (" + std::to_string(o.max_anchors_hard) + ") should not be smaller than the soft maximum number of anchors (" + std::to_string(o.max_anchors_soft) + ")."};
    if (o.seed_sampling_step_size == 0) throw CliError{"Validation failed for option --seed-sampling-step-size: must be at least 1."};
    return o;
}

// ---------------------------------------------------------------- FASTA / FASTQ (plain or gz), input.cpp:36-148
struct LineReader {
    gzFile f;
    std::vector<char> buf;
    explicit LineReader(const char* path) : f(gzopen(path, "rb")), buf(1 << 16) { if (f) gzbuffer(f, 1 << 20); }
    ~LineReader() { if (f) gzclose(f); }
    bool getline(std::string& out) {
        out.clear();
        while (true) {
            if (!gzgets(f, buf.data(), (int)buf.size())) return !out.empty();
            size_t const n = strlen(buf.data());
            out.append(buf.data(), n);
            if (n && out.back() == '\n') { out.pop_back(); if (!out.empty() && out.back() == '\r') out.pop_back(); return true; }
        }
    }
};

std::string record_id(std::string const& tag) { return tag.substr(0, tag.find(' ')); }     // input.cpp:161-163

struct Reference { std::vector<std::string> ids; std::vector<uint64_t> lens; std::vector<uint8_t> pool; };

bool read_references(std::string const& path, Reference& ref, std::string& err) {
    LineReader in(path.c_str());
    if (!in.f) { err = "cannot open " + path; return false; }
    std::string line, id, seq;
    bool have = false;
    auto flush = [&]() {
        if (!have) return;
        if (seq.empty()) { log_line("warning", "The record %s in the reference file has an empty sequence and will be skipped.", id.c_str()); return; }
        ref.ids.push_back(id);
        ref.lens.push_back(seq.size());
        size_t const off = ref.pool.size();
        ref.pool.resize(off + seq.size());
        flx_chars_to_rank_sequence(seq.data(), seq.size(), ref.pool.data() + off);
        log_line("debug", "read reference, id: %s, length %zu", id.c_str(), seq.size());
    };
    while (in.getline(line)) {
        if (!line.empty() && line[0] == '>') { flush(); id = record_id(line.substr(1)); seq.clear(); have = true; }
        else if (have) seq += line;
    }
    flush();
    if (ref.ids.empty()) { err = "The reference file is empty, which is not allowed."; return false; }
    return true;
}

// ---------------------------------------------------------------- FASTQ, block-parallel (input.cpp:83-148)
// The file is read (and, for .gz, inflated: one gzip stream is serial) in blocks by the calling thread; everything after that is
// parallel over the records of a batch: line boundaries, ids, rank encoding. A batch keeps its raw text: ids and qualities are
// pointers into it (terminated in place), so nothing is copied per record.
unsigned io_threads(uint64_t cli_threads) {
    if (const char* env = getenv("FLX_IO_THREADS")) { unsigned const v = (unsigned)strtoul(env, nullptr, 10); if (v) return std::min(v, 64u); }
    unsigned const hw = std::max(1u, std::thread::hardware_concurrency());
    return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::max<uint64_t>(cli_threads, std::min(hw, 8u)), 64));
}
template <class F>
void parallel_for(size_t n, unsigned threads, F&& body) {      // body(first, last) on disjoint ranges
    unsigned const t = (unsigned)std::min<size_t>(threads, std::max<size_t>(1, n / 256));
    if (t <= 1) { body((size_t)0, n); return; }
    std::vector<std::thread> pool;
    for (unsigned i = 0; i < t; ++i) pool.emplace_back([&, i] { body(n * i / t, n * (i + 1) / t); });
    for (auto& th : pool) th.join();
}

// std::vector<char> value-initialises what resize() adds: 16 MB of zeroes in front of every 16-MB read. This allocator leaves it alone.
template <class T>
struct DefaultInit : std::allocator<T> {
    template <class U> struct rebind { using other = DefaultInit<U>; };
    template <class U> void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void*>(p)) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};

struct ReadBatch {
    std::vector<char, DefaultInit<char>> raw;   // the batch's FASTQ text; ids / quals point into it
    std::vector<size_t> nl;                     // line ends of the text (FastqReader::next_text), four per record
    size_t n_rec = 0;                           // records the text holds
    std::vector<const char*> ids, quals;
    std::vector<uint8_t> pool;
    std::vector<uint64_t> offsets{0};
    size_t size() const { return ids.size(); }
};

struct FastqReader {
    gzFile f = nullptr;                         // gzip input
    int fd = -1;                                // plain input: read() straight into the batch's text (zlib's transparent mode copies every byte twice more)
    std::vector<char, DefaultInit<char>> carry; // text behind the last complete record of the previous batch
    bool eof = false;
    unsigned threads;
    size_t file_pos = 0, last_batch_bytes = 0;  // plain input: where the next read starts; the text the last batch took
    double s_read = 0, s_scan = 0, s_records = 0; // seconds spent reading, finding line ends, and on the records (FLX_CLI_PROFILE)
    static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    FastqReader(const char* path, unsigned threads_) : threads(threads_) {
        int const probe = open(path, O_RDONLY);
        if (probe < 0) return;
        // the pread path needs a file that can be read at an offset; a FIFO, /dev/stdin or a socket (plain or gzip: the reference reads a
        // pipe named *.fastq through an ifstream just as well) goes through zlib, whose transparent mode passes plain text through
        struct stat st;
        if (fstat(probe, &st) != 0 || !S_ISREG(st.st_mode)) {
            f = gzdopen(probe, "rb");
            if (f) gzbuffer(f, 1 << 20); else close(probe);
            return;
        }
        unsigned char magic[2] = {0, 0};
        ssize_t const got = pread(probe, magic, 2, 0);
        if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
            close(probe);
            f = gzopen(path, "rb");
            if (f) gzbuffer(f, 1 << 20);
        } else fd = probe;
    }
    ~FastqReader() { if (f) gzclose(f); if (fd >= 0) close(fd); }
    bool is_open() const { return f != nullptr || fd >= 0; }

    // the next batch of up to max_reads records; false at the end of the file or on a malformed record (err set)
    bool next(ReadBatch& b, size_t max_reads, std::string& err) {
        if (!next_text(b, max_reads, err)) return false;
        double const t_records = now_s();
        bool const ok = parse_records(b, threads, err);
        s_records += now_s() - t_records;
        return ok;
    }
    // Stage one, the reader's own thread: the text of the next batch (whole records) and its line ends. Stage two (parse_records) needs
    // nothing but the batch: the CLI runs it in the batch's task, next to the other batches' alignment, so that the one thread that has to
    // read the file in order does nothing else (it was the bound of the CLI with -I: 2.9 of 3.4 s).
    bool next_text(ReadBatch& b, size_t max_reads, std::string& err) {
        // (a recycled batch keeps its buffers: 330 MB of text and 165 MB of ranks per 16384 reads of 10 kb are not faulted in again)
        b.ids.clear(); b.quals.clear(); b.pool.clear(); b.offsets.assign(1, 0);
        auto& raw = b.raw;
        raw.clear();
        raw.insert(raw.end(), carry.begin(), carry.end());
        carry.clear();
        // ---- read until the text holds max_reads records (4 lines each) or the file ends
        size_t lines = 0, scanned = 0;
        std::vector<size_t>& nl = b.nl;         // positions of the line ends
        nl.clear();
        b.n_rec = 0;
        auto scan = [&]() {                     // line ends of raw[scanned, size): pieces of at least 8 MB, one thread each
            double const t_scan = now_s();
            size_t const lo = scanned, hi = raw.size();
            unsigned const t = (unsigned)std::min<size_t>(std::max(1u, threads), std::max<size_t>(1, (hi - lo) >> 23));
            std::vector<std::vector<size_t>> found(t);
            auto piece = [&](unsigned i) {
                size_t at = lo + (hi - lo) * i / t;
                size_t const end = lo + (hi - lo) * (i + 1) / t;
                while (at < end) {
                    const char* p = (const char*)memchr(raw.data() + at, '\n', end - at);
                    if (!p) break;
                    found[i].push_back((size_t)(p - raw.data()));
                    at = (size_t)(p - raw.data()) + 1;
                }
            };
            std::vector<std::thread> pool;
            for (unsigned i = 1; i < t; ++i) pool.emplace_back(piece, i);
            piece(0);
            for (auto& th : pool) th.join();
            for (auto const& v : found) { nl.insert(nl.end(), v.begin(), v.end()); lines += v.size(); }
            scanned = hi;
            s_scan += now_s() - t_scan;
        };
        scan();
        while (!eof && lines < 4 * max_reads) {
            // plain input: as much as the last batch took (and a little more) in one go, the pieces read by several threads (pread);
            // gzip input: 16 MB at a time through zlib
            size_t const at = raw.size(), want = f ? (size_t)16 << 20 : std::max<size_t>((size_t)16 << 20, last_batch_bytes > at ? last_batch_bytes - at + (1u << 20) : (size_t)16 << 20);
            double const t_read = now_s();
            raw.resize(at + want);
            long got;
            if (f) got = gzread(f, raw.data() + at, (unsigned)want);
            else {
                unsigned const t = (unsigned)std::min<size_t>(std::max(1u, threads), std::max<size_t>(1, want >> 24));      // >= 16 MB per thread
                std::vector<long> part(t, 0);
                std::vector<std::thread> pool;
                auto piece = [&](unsigned i) {
                    size_t const lo = want * i / t, hi = want * (i + 1) / t;
                    size_t done = 0;
                    while (lo + done < hi) {
                        ssize_t const r = pread(fd, raw.data() + at + lo + done, hi - lo - done, (off_t)(file_pos + lo + done));
                        if (r < 0) { if (errno == EINTR) continue; part[i] = -1; return; }
                        if (r == 0) break;
                        done += (size_t)r;
                    }
                    part[i] = (long)done;
                };
                for (unsigned i = 1; i < t; ++i) pool.emplace_back(piece, i);
                piece(0);
                for (auto& th : pool) th.join();
                got = 0;
                for (unsigned i = 0; i < t && got >= 0; ++i) {
                    if (part[i] < 0) got = -1;
                    else { got += part[i]; if ((size_t)part[i] < want * (i + 1) / t - want * i / t) break; }      // a short piece: the file ends inside it
                }
                if (got > 0) file_pos += (size_t)got;
            }
            if (got < 0) { err = "read error on the query file"; return false; }
            raw.resize(at + (size_t)got);
            s_read += now_s() - t_read;
            if (f && (size_t)got < want) {
                // a short read is the end of the data only when zlib agrees (a truncated .gz ends with Z_BUF_ERROR / Z_DATA_ERROR)
                int zerr = Z_OK;
                (void)gzerror(f, &zerr);
                if (zerr != Z_OK && zerr != Z_STREAM_END) { err = "the query file is truncated or corrupt (gzip stream error)"; return false; }
                eof = true;
            }
            if (!f && got == 0) eof = true;
            scan();
        }
        if (eof && !raw.empty() && raw.back() != '\n') { nl.push_back(raw.size()); raw.push_back('\n'); ++lines; }   // last line without a line end
        // blank lines at the very end of the file are not records
        while (eof && lines % 4 != 0 && lines > 0 && nl.size() >= 1 && (lines == 1 ? nl[0] == 0 : nl[lines - 1] == nl[lines - 2] + 1)) { nl.pop_back(); --lines; }
        size_t n_rec = std::min(lines / 4, max_reads);
        if (eof && lines % 4 != 0 && lines / 4 < max_reads) { err = "truncated FASTQ record at the end of the query file"; return false; }
        size_t const used = n_rec ? nl[4 * n_rec - 1] + 1 : 0;
        carry.assign(raw.begin() + (long)used, raw.end());
        raw.resize(used);
        last_batch_bytes = used;
        if (n_rec == 0) return false;
        b.n_rec = n_rec;
        return true;
    }
    // Stage two: ids, sequences (rank-encoded into the batch's pool) and qualities of the batch's records; false on a malformed record
    static bool parse_records(ReadBatch& b, unsigned threads, std::string& err) {
        auto& raw = b.raw;
        std::vector<size_t> const& nl = b.nl;
        size_t const n_rec = b.n_rec;
        // ---- records: id (up to the first blank, input.cpp:161-163), sequence, quality; terminated in place
        struct Rec { size_t id, seq, seq_len, qual; bool keep; };
        std::vector<Rec> recs(n_rec);
        std::atomic<bool> bad{false}, bad_qual{false};
        parallel_for(n_rec, threads, [&](size_t r0, size_t r1) {
            for (size_t r = r0; r < r1; ++r) {
                size_t const l0 = r ? nl[4 * r - 1] + 1 : 0, e0 = nl[4 * r], l1 = e0 + 1, e1 = nl[4 * r + 1], l3 = nl[4 * r + 2] + 1, e3 = nl[4 * r + 3];
                if (raw[l0] != '@' || raw[nl[4 * r + 1] + 1] != '+') { bad = true; continue; }
                size_t id_end = e0;
                if (id_end > l0 && raw[id_end - 1] == '\r') --id_end;
                for (size_t i = l0 + 1; i < id_end; ++i) if (raw[i] == ' ') { id_end = i; break; }
                raw[id_end] = 0;
                size_t s_end = e1, q_end = e3;
                if (s_end > l1 && raw[s_end - 1] == '\r') --s_end;
                if (q_end > l3 && raw[q_end - 1] == '\r') --q_end;
                raw[q_end] = 0;
                if (q_end - l3 != s_end - l1) { bad_qual = true; continue; }             // input.cpp:137 asserts qual.size() == seq.size()
                recs[r] = Rec{l0 + 1, l1, s_end - l1, l3, true};
            }
        });
        if (bad) { err = "malformed FASTQ record in the query file"; return false; }
        if (bad_qual) { err = "a FASTQ record's sequence and quality differ in length"; return false; }
        for (auto& rc : recs) {                 // the reference's filters on the way in (input.cpp:95-110)
            if (rc.seq_len == 0) { log_line("warning", "The record %s in the query file has an empty sequence and will be skipped.", raw.data() + rc.id); rc.keep = false; }
            else if (rc.seq_len > 100000) { log_line("warning", "skipping too large query: %s", raw.data() + rc.id); rc.keep = false; }
        }
        recs.erase(std::remove_if(recs.begin(), recs.end(), [](Rec const& x) { return !x.keep; }), recs.end());
        b.ids.resize(recs.size());
        b.quals.resize(recs.size());
        b.offsets.assign(recs.size() + 1, 0);
        for (size_t r = 0; r < recs.size(); ++r) b.offsets[r + 1] = b.offsets[r] + recs[r].seq_len;
        b.pool.resize(b.offsets.back() + 1);
        parallel_for(recs.size(), threads, [&](size_t r0, size_t r1) {
            for (size_t r = r0; r < r1; ++r) {
                b.ids[r] = raw.data() + recs[r].id;
                b.quals[r] = raw.data() + recs[r].qual;
                flx_chars_to_rank_sequence(raw.data() + recs[r].seq, recs[r].seq_len, b.pool.data() + b.offsets[r]);
            }
        });
        return true;
    }
};

std::vector<int> parse_devices(std::string spec, int n_available, std::string& err) {
    std::vector<int> out;
    if (spec.empty()) if (const char* env = getenv("FLX_DEVICES")) spec = env;
    if (spec.empty()) { if (const char* env = getenv("FLX_DEVICE")) spec = env; }
    if (spec.empty()) return {0};
    if (spec == "all") { for (int d = 0; d < n_available; ++d) out.push_back(d); return out; }
    size_t at = 0;
    while (at <= spec.size()) {
        size_t const comma = spec.find(',', at);
        std::string const part = spec.substr(at, comma == std::string::npos ? std::string::npos : comma - at);
        size_t const dash = part.find('-');
        char* end = nullptr;
        long const lo = strtol(part.c_str(), &end, 10);
        long hi = lo;
        if (dash != std::string::npos) hi = strtol(part.c_str() + dash + 1, &end, 10);
        if (part.empty() || *end || lo < 0 || hi < lo) { err = "cannot parse the device list " + spec; return {}; }
        for (long d = lo; d <= hi; ++d) out.push_back((int)d);
        if (comma == std::string::npos) break;
        at = comma + 1;
    }
    for (int d : out) if (d >= n_available) { err = "device " + std::to_string(d) + " of the device list does not exist (" + std::to_string(n_available) + " HIP devices)"; return {}; }
    return out;
}

}  // namespace

int main(int argc, char** argv) {
    Options o;
    try { o = parse_cli(argc, argv); }
    catch (CliError const& e) { fprintf(stderr, "[CLI PARSER ERROR]\n%s\n", e.msg.c_str()); return -1; }
    g_debug = o.console_debug_logs;
    if (!o.logfile.empty()) g_logfile = fopen(o.logfile.c_str(), "a");
    log_line("info", "successfully parsed CLI input ... starting");
    if (getenv("FLX_CLI_PARSE_ONLY")) {          // diagnostic: the FASTQ reader alone (no GPU): batches, records, seconds per stage
        FastqReader qin(o.queries.c_str(), io_threads(o.threads));
        if (!qin.is_open()) { log_line("error", "cannot open %s", o.queries.c_str()); return -1; }
        ReadBatch batch;
        std::string perr;
        uint64_t n = 0, batches = 0;
        double const t0 = FastqReader::now_s();
        while (qin.next(batch, 16384, perr)) { n += batch.ids.size(); ++batches; }
        double const secs = FastqReader::now_s() - t0;
        if (!perr.empty()) { log_line("error", "%s", perr.c_str()); return -1; }
        fprintf(stderr, "[flx cli parse only] %llu records in %llu batches, %.2f s (%.0f reads/s): reading %.2f s, line ends %.2f s, records %.2f s\n",
                (unsigned long long)n, (unsigned long long)batches, secs, n / secs, qin.s_read, qin.s_scan, qin.s_records);
        return 0;
    }

    Reference ref;
    std::string err;
    log_line("info", "reading reference sequences from %s", o.reference.c_str());
    if (!read_references(o.reference, ref, err)) { log_line("error", "An error occured while trying to read the reference from the file %s.\n%s", o.reference.c_str(), err.c_str()); return -1; }

    flx_index* index = nullptr;
    struct stat st;
    if (!o.index.empty() && stat(o.index.c_str(), &st) == 0) {
        log_line("info", "loading index from %s", o.index.c_str());
        if (flx_index_load(o.index.c_str(), &index) != FLX_OK) { log_line("error", "An error occured while trying to load the index from the file %s.\n%s", o.index.c_str(), flx_last_error()); return -1; }
        if (flx_index_matches_reference(index, ref.pool.data(), ref.lens.data(), (uint32_t)ref.ids.size()) != FLX_OK) {
            log_line("error", "The index in the file %s was not built from the reference in %s.\n%s", o.index.c_str(), o.reference.c_str(), flx_last_error());
            return -1;
        }
    } else {
        log_line("info", "building index with %llu thread%s", (unsigned long long)o.threads, o.threads == 1 ? "" : "s");
        auto const t0 = std::chrono::steady_clock::now();
        // suffix arrays on the GPU (FLX_INDEX_ON_HOST=1: the host's SA-IS; the index is the same either way)
        int const brc = getenv("FLX_INDEX_ON_HOST") ? flx_index_build(ref.pool.data(), ref.lens.data(), (uint32_t)ref.ids.size(), &index)
                                                    : flx_index_build_on_device(0, ref.pool.data(), ref.lens.data(), (uint32_t)ref.ids.size(), &index);
        if (brc != FLX_OK) { log_line("error", "index construction failed: %s", flx_last_error()); return -1; }
        log_line("info", "building index took %.3f seconds", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        if (!o.index.empty() && flx_index_save(index, o.index.c_str()) != FLX_OK)
            log_line("warning", "An error occured while trying to write the index to the file %s.\nContinuing without saving the index.\n%s", o.index.c_str(), flx_last_error());
    }

    // One context per device of the device list, all uploading the one index built / loaded above; batches of reads are dealt to
    // them in turn and written in input order (floxer.cpp:141-171: the reference's unit of parallelism is the read, too).
    int n_hip = flx_device_count();
    if (n_hip <= 0) { log_line("error", "no HIP device available; floxer_amd has no CPU fallback"); return -1; }
    std::vector<int> const devices = parse_devices(o.devices, n_hip, err);
    if (devices.empty()) { log_line("error", "%s", err.c_str()); return -1; }
    flx_stats* stats = nullptr;
    if (!o.stats.empty() && flx_stats_create(o.stats_input_hint.empty() ? nullptr : o.stats_input_hint.c_str(), &stats) != FLX_OK) { log_line("error", "%s", flx_last_error()); return -1; }
    std::vector<flx_ctx*> ctxs;
    for (int d : devices) {
        flx_ctx* c = nullptr;
        if (flx_ctx_create(d, index, &c) != FLX_OK) { log_line("error", "cannot set up the GPU context on device %d: %s", d, flx_last_error()); return -1; }
        if (stats) flx_ctx_set_stats(c, stats);
        ctxs.push_back(c);
    }
    log_line("info", "running on %zu HIP device context%s", ctxs.size(), ctxs.size() == 1 ? "" : "s");

    std::vector<const char*> ref_id_ptrs;
    for (auto const& s : ref.ids) ref_id_ptrs.push_back(s.c_str());
    flx_sam_writer* out = nullptr;
    if (flx_sam_open(o.output.c_str(), ref_id_ptrs.data(), ref.lens.data(), (uint32_t)ref.ids.size(), &out) != FLX_OK) { log_line("error", "%s", flx_last_error()); return -1; }

    flx_params p;
    flx_params_default(&p);
    p.query_error_probability = o.has_error_probability ? o.error_probability : -1.0;      // the probability wins (input.cpp:27-33)
    p.query_num_errors = o.query_errors;
    p.pex_seed_num_errors = o.seed_errors;
    p.search.max_num_anchors_hard = o.max_anchors_hard;
    p.search.max_num_anchors_soft = o.max_anchors_soft;
    p.search.anchor_group_order = o.anchor_group_order == "errors_first" ? FLX_ORDER_ERRORS_FIRST : o.anchor_group_order == "none" ? FLX_ORDER_NONE : FLX_ORDER_COUNT_FIRST;
    p.search.anchor_choice_strategy = o.anchor_choice_strategy == "full_groups" ? FLX_CHOICE_FULL_GROUPS : o.anchor_choice_strategy == "first_reported" ? FLX_CHOICE_FIRST_REPORTED : FLX_CHOICE_ROUND_ROBIN;
    p.search.erase_useless_anchors = !o.dont_erase_useless_anchors;
    p.seed_sampling_step_size = o.seed_sampling_step_size;
    p.bottom_up_pex_tree_building = o.bottom_up_pex_tree;
    p.use_interval_optimization = o.interval_optimization;
    p.extra_verification_ratio = o.extra_verification_ratio;
    p.direct_full_verification = o.direct_full_verification;
    p.without_cigar = o.without_cigar;
    p.num_anchors_per_verification_task = o.num_anchors_per_task;

    struct stat qst;
    stat(o.queries.c_str(), &qst);
    log_line("info", "aligning queries from a %lld bytes large file against %zu references on the GPU and writing output file to %s", (long long)qst.st_size, ref.ids.size(), o.output.c_str());
    auto const t_align = std::chrono::steady_clock::now();
    unsigned const n_io = io_threads(o.threads);
    FastqReader qin(o.queries.c_str(), n_io);
    if (!qin.is_open()) { log_line("error", "cannot open %s", o.queries.c_str()); return -1; }
    flx_sam_set_threads(out, n_io);
    size_t batch_reads = 16384;      // 1024 reads per lane and chunk (see flx_align_reads_resident)
    if (const char* env = getenv("FLX_BATCH_READS")) { size_t const v = strtoull(env, nullptr, 10); if (v) batch_reads = v; }
    // Batches are independent: up to three are in a context at a time (their chunks share its lanes), the next one is parsed
    // while they run, and results are written in input order.
    struct Finished { std::unique_ptr<ReadBatch> batch; std::vector<flx_record> recs; std::vector<uint32_t> cig; std::vector<uint8_t> skipped; int rc = FLX_OK; std::string err; bool reader_error = false; };
    // FLX_CLI_PROFILE=1: seconds this run spent parsing (this thread), aligning (sum over the batches' tasks) and writing (the writer
    // thread) on stderr at the end: which of the three stages bounds the end-to-end rate
    std::atomic<uint64_t> us_parse{0}, us_align{0}, us_copy{0}, us_write{0}, us_records{0};
    auto const now_us = [] { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto align_batch = [&](std::unique_ptr<ReadBatch> b, flx_ctx* ctx) {
        Finished f;
        flx_run* run = nullptr;
        {
            // the second stage of the reader, in this batch's own task
            uint64_t const tr = now_us();
            std::string perr;
            bool const ok = FastqReader::parse_records(*b, n_io, perr);
            us_records += now_us() - tr;
            if (!ok) { f.rc = FLX_ERR_INVALID; f.err = perr; f.reader_error = true; f.batch = std::move(b); return f; }
        }
        uint64_t const t0 = now_us();
        f.rc = flx_align_reads(ctx, &p, b->pool.data(), b->offsets.data(), b->ids.size(), &run);
        us_align += now_us() - t0;
        if (f.rc != FLX_OK) { f.err = flx_last_error(); f.batch = std::move(b); return f; }
        f.recs.resize(flx_run_num_records(run));
        f.cig.resize(flx_run_num_cigar_words(run) + 1);
        f.skipped.resize(b->ids.size());
        uint64_t const t1 = now_us();
        flx_run_copy(run, f.recs.data(), f.cig.data(), f.skipped.data());
        flx_run_free(run);
        us_copy += now_us() - t1;
        f.batch = std::move(b);
        return f;
    };
    // Three stages run side by side: this thread parses the next batch, the batches in flight are aligned (one task each), a writer
    // thread takes them in input order and formats / compresses / writes them (itself on the writer's I/O threads).
    std::deque<std::future<Finished>> in_flight;           // guarded by q_mu
    std::vector<std::unique_ptr<ReadBatch>> spare_batches; // written batches, recycled by the reader (guarded by q_mu)
    std::mutex q_mu;
    std::condition_variable q_cv;
    bool no_more = false;
    uint64_t total_reads = 0, total_records = 0, n_batches = 0;
    std::atomic<bool> failed{false};
    bool eof = false, timed_out = false;
    auto write_one = [&](Finished& f) {
        if (failed.load()) return;
        if (f.rc != FLX_OK) {
            if (f.reader_error) log_line("error", "An error occured while trying to read the queries from the file %s.\n%s", o.queries.c_str(), f.err.c_str());
            else log_line("error", "An error occurred while aligning a batch of queries.\nShutting down. The output file is likely incomplete. Error message:\n%s", f.err.c_str());
            failed.store(true);
            return;
        }
        ReadBatch const& batch = *f.batch;
        for (size_t i = 0; i < f.skipped.size(); ++i)
            if (f.skipped[i]) log_line("warning", "skipping query: %s due to bad configuration regarding the number of errors.", batch.ids[i]);
        uint64_t const t0 = now_us();
        if (flx_sam_write(out, batch.ids.data(), batch.pool.data(), batch.offsets.data(), batch.quals.data(), f.recs.data(), f.recs.size(), f.cig.data()) != FLX_OK) { log_line("error", "%s", flx_last_error()); failed.store(true); }
        us_write += now_us() - t0;
        total_reads += batch.ids.size();
        total_records += f.recs.size();
        log_line("debug", "finished a batch: %llu queries, %llu records so far", (unsigned long long)total_reads, (unsigned long long)total_records);
    };
    std::thread writer([&] {
        while (true) {
            std::future<Finished> next;
            {
                std::unique_lock<std::mutex> g(q_mu);
                q_cv.wait(g, [&] { return !in_flight.empty() || no_more; });
                if (in_flight.empty()) return;
                next = std::move(in_flight.front());
                in_flight.pop_front();
            }
            q_cv.notify_all();
            Finished f = next.get();
            write_one(f);
            if (f.batch) { std::lock_guard<std::mutex> g(q_mu); spare_batches.push_back(std::move(f.batch)); }      // its buffers serve a later batch
        }
    });
    size_t const max_in_flight = 3 * ctxs.size() + 1;
    while (!eof && !failed.load()) {
        if (o.has_timeout && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_align).count() > (double)o.timeout) {
            log_line("warning", "Timeout happened. Shutting down now. The output file might be incomplete.");
            timed_out = true;
            break;
        }
        std::unique_ptr<ReadBatch> batch;
        {
            std::lock_guard<std::mutex> g(q_mu);
            if (!spare_batches.empty()) { batch = std::move(spare_batches.back()); spare_batches.pop_back(); }
        }
        if (!batch) batch = std::make_unique<ReadBatch>();
        uint64_t const t_parse = now_us();
        bool const got = qin.next_text(*batch, batch_reads, err);
        us_parse += now_us() - t_parse;
        if (!got) {
            eof = true;
            if (!err.empty()) { log_line("error", "An error occured while trying to read the queries from the file %s.\n%s", o.queries.c_str(), err.c_str()); failed.store(true); }
            break;
        }
        flx_ctx* const target = ctxs[n_batches++ % ctxs.size()];
        {
            std::unique_lock<std::mutex> g(q_mu);
            q_cv.wait(g, [&] { return in_flight.size() < max_in_flight; });
            in_flight.push_back(std::async(std::launch::async, align_batch, std::move(batch), target));
        }
        q_cv.notify_all();
    }
    { std::lock_guard<std::mutex> g(q_mu); no_more = true; }
    q_cv.notify_all();
    writer.join();
    if (flx_sam_close(out) != FLX_OK) { log_line("error", "%s", flx_last_error()); failed.store(true); }
    // (the alignment phase ends with the output file: floxer.cpp:154-179 stops its watch there; giving 40 GB of HBM back is not part of it)
    double const secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_align).count();
    for (flx_ctx* c : ctxs) flx_ctx_destroy(c);
    flx_index_free(index);
    if (failed.load() || timed_out) return -1;
    log_line("info", "finished aligning successfully in %.3f seconds (%llu queries, %llu records)", secs, (unsigned long long)total_reads, (unsigned long long)total_records);
    if (getenv("FLX_CLI_PROFILE")) {
        fprintf(stderr, "[flx cli profile] reader thread: reading %.2f s, line ends %.2f s; records (ids, rank encoding: in the batches' own tasks) %.2f s\n", qin.s_read, qin.s_scan, us_records.load() / 1e6);
        fprintf(stderr, "[flx cli profile] wall %.2f s: parsing %.2f s (reader thread), aligning %.2f s summed over %llu batches (up to %zu in flight), copying results %.2f s, writing %.2f s (writer thread, %u I/O threads)\n",
                secs, us_parse.load() / 1e6, us_align.load() / 1e6, (unsigned long long)n_batches, max_in_flight, us_copy.load() / 1e6, us_write.load() / 1e6, n_io);
    }
    if (stats) {                                                                       // floxer.cpp:182-192
        uint64_t len = 0;
        flx_stats_format(stats, o.stats != "terminal", nullptr, &len);
        std::string text(len, '\0');
        if (flx_stats_format(stats, o.stats != "terminal", &text[0], &len) == FLX_OK) {
            text.resize(strlen(text.c_str()));
            if (o.stats == "terminal") {
                for (size_t at = 0; at < text.size();) {
                    size_t const e = text.find("\n\n", at);
                    log_line("info", "%s", text.substr(at, e == std::string::npos ? std::string::npos : e - at).c_str());
                    if (e == std::string::npos) break;
                    at = e + 2;
                }
            } else if (FILE* f = fopen(o.stats.c_str(), "w")) { fwrite(text.data(), 1, text.size(), f); fclose(f); }
            else log_line("warning", "cannot write the statistics to %s", o.stats.c_str());
        }
        flx_stats_free(stats);
    }
    if (g_logfile) fclose(g_logfile);
    return 0;
}
