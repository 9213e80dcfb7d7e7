// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of floxer's seed-and-verify hot path (reference @ /root/reference, v0.2.0).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link, load or call
// anything in this directory, and only as the checker / reported baseline. The product
// (floxer_amd/) never includes, links or calls this code.
//
// PARITY STATUS
//   * pinned by the reference's own tests (see tests/test_oracle_pins.py, tests/golden/):
//       math (math_test.cpp), rank encoding (input_test.cpp), PEX trees (pex_test.cpp),
//       erase_useless_anchors (search_test.cpp:138-184), interval semantics (intervals_test.cpp),
//       alignment score/begin/CIGAR (alignment_test.cpp), hierarchical verification + span
//       arithmetic (verification_test.cpp), end-to-end records (floxer_whole_program_via_cli_test.cpp).
//   * "parity unpinned" (no reference test asserts it, third-party source absent from
//     /root/reference, restated from the published algorithms):
//       - fmindex-collection @ b0e311f: search_ng21 enumeration order / duplicate hits / info-flag
//         pruning, search_schemes::expand part-length rule, optimum(0,3) scheme;
//       - seqan3 @ bfa237e: trace priority between "up" and "left" (up>diag and left>diag are
//         evidenced; up vs left is a best recollection).
//     Every such spot is tagged [3P-UNVERIFIED] below.
//
// The reference cannot be built here (CPM downloads 9 dependencies at configure time, no network),
// so there is no oracle/_ref; see DESIGN.md.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace orc {

// ---------------------------------------------------------------- math (include/math.hpp:10-27)
int32_t saturate_value_to_int32_max(uint64_t value);
uint64_t ceil_div(uint64_t a, uint64_t b);
uint64_t floating_point_error_aware_ceil(double value);

// ---------------------------------------------------------------- input (src/lib/input.cpp)
uint8_t char_to_rank(char c);                       // input.cpp:165-176 + ivs::d_dna5
char rank_to_char(uint8_t r);
std::vector<uint8_t> chars_to_rank_sequence(const char* s, size_t n);
std::vector<uint8_t> reverse_complement_rank(const std::vector<uint8_t>& seq);
std::string extract_record_id(const std::string& tag);     // input.cpp:161-163

// ---------------------------------------------------------------- PEX tree (src/lib/pex.cpp)
struct pex_node {
    static constexpr uint64_t null_id = ~uint64_t(0);
    uint64_t parent_id;
    uint64_t from;      // inclusive
    uint64_t to;        // inclusive
    uint64_t num_errors;
    uint64_t length() const { return to - from + 1; }
    bool is_root() const { return parent_id == null_id; }
};

struct pex_tree {
    std::vector<pex_node> inner_nodes;
    std::vector<pex_node> leaves;
    uint64_t no_error_seed_length = 0;
    uint64_t leaf_max_num_errors = 0;

    pex_tree(uint64_t total_query_length, uint64_t query_num_errors, uint64_t leaf_max_num_errors, bool bottom_up);
    const pex_node& root() const { return inner_nodes.empty() ? leaves.at(0) : inner_nodes.at(0); }

private:
    void add_nodes_recursive(uint64_t from1, uint64_t to1, uint64_t num_errors, uint64_t parent_id);
    void add_nodes_bottom_up(uint64_t total_len, uint64_t query_num_errors);
};

// ---------------------------------------------------------------- FM index (fmindex-collection BiFMIndex)
struct fm_index {
    uint64_t n = 0;                         // padded text length
    uint32_t sampling = 4;
    std::vector<uint8_t> text;              // concatenated ranks + zero padding
    std::vector<uint64_t> seq_start;        // start of each sequence in text
    std::vector<uint64_t> seq_len;
    std::vector<uint32_t> sa;               // full suffix array of text, n < 2^32 (only sampled rows are *used* by locate)
    std::vector<uint8_t> bwt, bwt_rev;
    uint64_t C[7];                          // C[c] = #symbols < c
    // occ checkpoints every 64 positions
    std::vector<uint32_t> occ_cp, occ_rev_cp;   // [(n/64+1)][6]

    uint64_t occ(bool rev, uint8_t c, uint64_t i) const;     // #c in bwt[0,i)
    void all_occ(bool rev, uint64_t i, uint64_t out[6]) const;
    // LF-walk locate, BiFMIndex::locate: returns (seq id, position)
    void locate(uint64_t row, uint64_t& seq_id, uint64_t& pos, uint64_t* lf_steps = nullptr) const;
};

fm_index build_index(const std::vector<std::vector<uint8_t>>& refs, uint32_t sampling);
// The same index with its suffix array and the two BWTs taken as data instead of being sorted here (the suffix array of a text is
// unique, so this is the index build_index() would make; used where sorting 3 G suffixes on the CPU is out of reach: the
// benchmark's cpu_baseline leg and the scale tests). The arrays are checked for consistency on a sample of rows.
fm_index import_index(const std::vector<std::vector<uint8_t>>& refs, uint32_t sampling, const uint32_t* sa, const uint8_t* bwt,
                      const uint8_t* bwt_rev);

struct cursor { uint64_t lb, lb_rev, len; };

// ---------------------------------------------------------------- search schemes (search_schemes::generator::optimum, expand)
struct search_def { std::vector<uint32_t> pi, l, u; };
std::vector<search_def> optimum_scheme(uint32_t k);                          // [3P-UNVERIFIED for k==3]
std::vector<search_def> expand_scheme(const std::vector<search_def>& s, uint64_t len);   // empty if not expandable

struct anchor_group { cursor cur; uint64_t num_errors; };

struct search_counters {
    uint64_t n_extend_all = 0;     // all-symbol cursor extensions (2 rank positions each)
    uint64_t n_extend_one = 0;     // single-symbol extensions (2 rank positions each)
    uint64_t n_locates = 0;
    uint64_t n_lf_steps = 0;
};

// search_ng21::search_n restated; groups in delegate order
void search_n(const fm_index& idx, const uint8_t* query, uint64_t len, uint32_t k, uint64_t n,
              std::vector<anchor_group>& out, search_counters* ctr = nullptr);

// ---------------------------------------------------------------- search.cpp
struct anchor_t {
    uint64_t pex_leaf_index, reference_id, reference_position, num_errors;
};
static constexpr uint64_t erase_marker = ~uint64_t(0);

enum group_order { ORDER_ERRORS_FIRST = 0, ORDER_COUNT_FIRST = 1, ORDER_NONE = 2 };
enum choice_strategy { CHOICE_ROUND_ROBIN = 0, CHOICE_FULL_GROUPS = 1, CHOICE_FIRST_REPORTED = 2 };

struct search_config {
    uint64_t max_num_anchors_hard = 500;
    uint64_t max_num_anchors_soft = 50;
    int anchor_group_order = ORDER_COUNT_FIRST;
    int anchor_choice_strategy = CHOICE_ROUND_ROBIN;
    bool erase_useless_anchors = true;
};

struct seed_t { const uint8_t* seq; uint64_t len; uint64_t num_errors; uint64_t query_position; uint64_t pex_leaf_index; };

struct anchors_of_seed {
    uint64_t num_kept_useful_anchors = 0, num_kept_raw_anchors = 0, num_excluded_raw_anchors_by_soft_cap = 0;
    std::vector<std::vector<anchor_t>> anchors_by_reference;   // empty if fully excluded
};

uint64_t erase_useless_anchors(std::vector<std::vector<anchor_t>>& anchors_by_reference);   // search.cpp:352-389
std::vector<anchors_of_seed> search_seeds(const fm_index& idx, const std::vector<seed_t>& seeds,
                                          const search_config& cfg, search_counters* ctr = nullptr);   // search.cpp:143-324
std::vector<seed_t> generate_seeds(const pex_tree& tree, const uint8_t* query, uint64_t step);        // pex.cpp:258-277

// ---------------------------------------------------------------- alignment (alignment.cpp + seqan3 edit distance)
enum align_mode { MODE_EXISTS = 0, MODE_WITHOUT_CIGAR = 1, MODE_WITH_CIGAR = 2 };
struct align_result {
    bool exists = false;
    uint64_t num_errors = 0;
    uint64_t begin = 0;                 // position in the given reference window
    std::vector<uint32_t> cigar;        // BAM encoding len<<4|op, ops: I=1 D=2 '='=7 X=8
};
struct align_counters { uint64_t word_steps = 0; uint64_t cells = 0; };
// algo 0: plain O(nm) DP matrix (definition of the semantics); algo 1: Myers/Hyyro bit-vector with stored trace planes
align_result align(const uint8_t* ref, uint64_t n, const uint8_t* query, uint64_t m, uint64_t k, int mode, int algo,
                   align_counters* ctr = nullptr);

// ---------------------------------------------------------------- intervals (intervals.cpp)
struct half_open_interval { uint64_t start, end; };
int relationship_with(half_open_interval a, half_open_interval b);      // enum order of intervals.hpp:15-23
half_open_interval trim_from_both_sides(half_open_interval a, uint64_t amount);
struct verified_intervals {
    bool active = true;
    std::vector<half_open_interval> ivs;
    void insert(half_open_interval iv);
    bool contains(half_open_interval iv) const;
};

// ---------------------------------------------------------------- verification (verification.cpp)
struct span_config { uint64_t offset, length, extra; };
span_config compute_reference_span_start_and_length(uint64_t anchor_pos, const pex_node& node, uint64_t leaf_from,
                                                    uint64_t full_reference_length, double extra_ratio);

struct query_alignment { uint64_t start_in_reference, num_errors; bool reverse; std::vector<uint32_t> cigar; };

struct params {
    double error_probability = -1.0;       // <0: use query_num_errors
    uint64_t query_num_errors = 0;
    uint64_t seed_errors = 2;
    search_config search;
    uint64_t seed_sampling_step = 1;
    bool bottom_up = false;
    bool interval_optimization = false;
    double extra_verification_ratio = 0.05;
    bool direct_full = false;
    uint64_t anchors_per_task = 3000;
    bool without_cigar = false;
    int align_algo = 1;
};

struct verify_counters { uint64_t inner_jobs = 0, root_jobs = 0, inner_word_steps = 0, root_word_steps = 0, ref_query_bytes = 0; };

// The values the reference feeds its statistics (statistics.hpp:24-172; statistics.cpp:223-240 gives the order of the histograms), as
// raw lists: [0] query lengths, [1] seed lengths, [2] errors per seed, [3] seeds per query, [4] fully excluded seeds per query, [5] kept
// anchors per query, [6] / [7] raw anchors excluded by the soft cap / by erase-useless per query, [8] kept anchors per kept seed,
// [9] / [10] excluded by the soft cap / by erase-useless per kept seed, [11] / [12] reference span sizes aligned of inner nodes / of roots,
// [13] root spans avoided (verification.cpp:130), [14] alignments per query, [15] alignments' edit distance. The two wall-clock
// histograms (milliseconds per query) are not restated. Binning (statistics.cpp:80-94) is left to the checker.
constexpr int N_STAT_LISTS = 16;
struct run_statistics {
    uint64_t completely_excluded_queries = 0;
    std::vector<uint64_t> values[N_STAT_LISTS];
    void merge(run_statistics const& o) {
        completely_excluded_queries += o.completely_excluded_queries;
        for (int i = 0; i < N_STAT_LISTS; ++i) values[i].insert(values[i].end(), o.values[i].begin(), o.values[i].end());
    }
};

// one anchor, query_verifier::verify
void verify_anchor(const pex_tree& tree, const anchor_t& anchor, const uint8_t* query, bool reverse,
                   const uint8_t* reference, uint64_t reference_len, const params& p, verified_intervals& ivs,
                   std::vector<query_alignment>& out, verify_counters* ctr = nullptr, run_statistics* st = nullptr);

// ---------------------------------------------------------------- whole path for a set of reads (parallelization.cpp, output.cpp)
struct record {
    uint64_t read_index;
    uint32_t flag;              // 0, 16, 256, 272, 4
    int64_t ref_id;             // -1 unmapped
    int32_t pos;                // 0-based, saturated
    uint32_t nm;
    uint64_t cigar_off, cigar_len;
};
struct run_output {
    std::vector<record> records;
    std::vector<uint32_t> cigars;
    std::vector<uint8_t> skipped;         // per read: 1 if filtered (input.cpp:95-129) -> no record
    search_counters sc;
    verify_counters vc;
    run_statistics st;
};
run_output align_reads(const fm_index& idx, const std::vector<std::vector<uint8_t>>& refs,
                       const std::vector<std::vector<uint8_t>>& reads, const params& p, unsigned threads);

std::string cigar_to_string(const uint32_t* c, uint64_t n);

}  // namespace orc
