/*
 * floxer_amd — C ABI of the MI355X-native seed-and-verify path (libfloxer_amd.so).
 *
 * floxer has no plugin/FFI interface; its hot path sits behind four C++ seams. Each entry point
 * below replaces one of those seams and cites it. Conventions: caller owns every buffer; plain pointers + sizes; no
 * exceptions cross the boundary — every function returns FLX_OK (0) or a negative flx_status and flx_last_error()
 * describes the failure (the reference throws C++ exceptions that its task wrappers turn into a stop flag,
 * parallelization.cpp:149-157). A flx_ctx owns one HIP device, its HBM-resident index and a set of lanes (a HIP stream with
 * its workspaces each); calls on different contexts are independent. The compute calls (flx_search_seeds, flx_search_groups,
 * flx_align_batch, flx_align_reads*, flx_reads_upload) may be issued from several host threads on one context: each takes a
 * free lane and waits when there is none, so batches overlap on the GPU. Configuration calls (flx_ctx_set_stream,
 * flx_ctx_enable_kernel_timing, flx_ctx_reset_kernel_stats, flx_ctx_destroy) must not overlap anything else.
 *
 * Sequences are rank sequences as the reference stores them (input.cpp:165-176): $=0 A=1 C=2 G=3 T=4 N/other=5.
 * CIGARs are BAM words (len<<4|op) with ops I=1 D=2 '='=7 X=8 (extended CIGAR, alignment.cpp:178).
 */
#ifndef FLOXER_AMD_H
#define FLOXER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum flx_status {
    FLX_OK = 0,
    FLX_ERR_INVALID = -1,      /* bad argument / shape */
    FLX_ERR_NO_DEVICE = -2,    /* no HIP device / HIP runtime failure: the product never falls back to a CPU path */
    FLX_ERR_CAPACITY = -3,     /* caller buffer too small; required size is reported through the size out-parameter */
    FLX_ERR_UNSUPPORTED = -4,
    FLX_ERR_INTERNAL = -5,
    FLX_ERR_IO = -6
} flx_status;

const char* flx_last_error(void);
const char* flx_version(void);

/* ------------------------------------------------------------------------------------------------ host-side arithmetic
 * math.hpp:10-27, input.cpp:26-34 — must be bit-identical incl. the double arithmetic, so it is host code. */
uint64_t flx_ceil_div(uint64_t a, uint64_t b);
uint64_t flx_floating_point_error_aware_ceil(double value);
int32_t flx_saturate_value_to_int32_max(uint64_t value);
/* input.cpp:161-176 + ivs::reverse_complement_rank (input.cpp:132) */
void flx_chars_to_rank_sequence(const char* chars, uint64_t n, uint8_t* out_ranks);
void flx_reverse_complement_rank(const uint8_t* ranks, uint64_t n, uint8_t* out_ranks);

/* ------------------------------------------------------------------------------------------------ PEX tree
 * replaces pex::pex_tree::pex_tree (pex.hpp:57-126, pex.cpp:84-256). Nodes: inner nodes first (root = inner[0], or
 * leaves[0] when the tree is a single node), then leaves. parent_id indexes the inner nodes; FLX_NULL_ID for the root. */
#define FLX_NULL_ID 0xFFFFFFFFu
typedef struct flx_pex_node {
    uint32_t parent_id;
    uint32_t from;        /* inclusive */
    uint32_t to;          /* inclusive */
    uint32_t num_errors;
} flx_pex_node;
int flx_pex_tree_build(uint64_t query_length, uint64_t query_num_errors, uint64_t leaf_max_num_errors, int bottom_up,
                       flx_pex_node* nodes, uint64_t capacity, uint64_t* n_inner, uint64_t* n_leaves);

/* ------------------------------------------------------------------------------------------------ index lifetime
 * replaces fmindex(refs, sampling_rate=4, threads) / load_index / save_index (floxer.cpp:62-107, input.cpp:150-157,
 * output.cpp:25-40). Built on the host; own versioned file format (not cereal-compatible). */
typedef struct flx_index flx_index;
int flx_index_build(const uint8_t* ref_ranks_concat, const uint64_t* ref_lens, uint32_t n_refs, flx_index** out);
/* The same index with its two suffix arrays built on a HIP device (prefix doubling with radix sorts instead of the host's SA-IS:
 * seconds instead of a minute for a chromosome-sized reference); 36 bytes of HBM per reference symbol while it runs. */
int flx_index_build_on_device(int hip_device, const uint8_t* ref_ranks_concat, const uint64_t* ref_lens, uint32_t n_refs, flx_index** out);
int flx_index_save(const flx_index* index, const char* path);
int flx_index_load(const char* path, flx_index** out);
void flx_index_free(flx_index* index);
uint64_t flx_index_text_length(const flx_index* index);     /* concatenated text incl. sentinel padding */
uint32_t flx_index_num_references(const flx_index* index);
uint64_t flx_index_device_bytes(const flx_index* index);    /* HBM footprint of the index image once uploaded */
/* what a context adds to the image when the device has room for it: the inverse suffix array (4 B per text symbol) and the presence
   filter of the seeding kernels (4^K / 8 bytes, K = ceil(log4 n) + 2); a context made without either gives the same results, slower */
uint64_t flx_index_derived_device_bytes(const flx_index* index);
/* FLX_OK iff the index was built from exactly these reference sequences (guards --index against a stale file, floxer.cpp:63-79) */
int flx_index_matches_reference(const flx_index* index, const uint8_t* ref_ranks_concat, const uint64_t* ref_lens, uint32_t n_refs);
/* test hooks: suffix array / BWT as built (text_length entries) */
int flx_index_copy_sa(const flx_index* index, uint64_t* out);
int flx_index_copy_sa_u32(const flx_index* index, uint32_t* out);   /* the same as stored (text < 2^32 symbols) */
int flx_index_copy_bwt(const flx_index* index, int reversed, uint8_t* out);

/* ------------------------------------------------------------------------------------------------ device context */
typedef struct flx_ctx flx_ctx;
int flx_device_count(void);                                  /* HIP devices visible to the process (0: none / no runtime) */
int flx_ctx_create(int hip_device, const flx_index* index, flx_ctx** out);   /* uploads index + reference text to HBM */
void flx_ctx_destroy(flx_ctx* ctx);
/* Index replicas across the GPUs of a job (the FM index is replicated per GPU). The HBM image of an index is five
 * device buffers (occurrence table of the text, of the reversed text, suffix array, text with its guard bytes, k-mer table). A rank
 * that built or loaded the index uploads the image into buffers it owns (flx_index_image_upload), sends them to the other ranks
 * (RCCL broadcast over xGMI: floxer_amd/distributed.py) together with the small host part (flx_index_meta_export), and every rank
 * makes its context on its copy (flx_ctx_create_on_image; the buffers must outlive the context): no rank but the first holds the
 * index in host memory, builds it or reads it from a file. */
typedef struct flx_index_image { uint64_t bytes[5]; } flx_index_image;
int flx_index_image_layout(const flx_index* index, flx_index_image* out);
int flx_index_image_upload(const flx_index* index, int hip_device, void* const device_buffers[5]);
int flx_index_meta_export(const flx_index* index, uint8_t* buf, uint64_t* len /* in: capacity, out: needed */);
int flx_index_meta_import(const uint8_t* buf, uint64_t len, flx_index** out);   /* an index without arrays: for flx_ctx_create_on_image */
/* sizes: the bytes the caller's five buffers hold; they must be the index's layout (a stale image, e.g. of another build's block size,
 * is refused instead of read out of bounds) */
int flx_ctx_create_on_image(int hip_device, const flx_index* index, void* const device_buffers[5], const flx_index_image* sizes, flx_ctx** out);
/* use a caller-owned HIP stream (hipStream_t passed as void*) for all launches; NULL restores the context's own stream */
int flx_ctx_set_stream(flx_ctx* ctx, void* hip_stream);

/* ------------------------------------------------------------------------------------------------ seam 1: seeding
 * replaces search_result searcher::search_seeds(std::vector<seed> const&) const (search.hpp:104-112, search.cpp:143-324) */
typedef struct flx_seed {            /* search::seed, search.hpp:17-22 */
    uint64_t seq_offset;             /* into the sequence pool */
    uint32_t length;
    uint32_t num_errors;             /* 0..3 */
    uint32_t pex_leaf_index;
    uint32_t reserved;
} flx_seed;

enum { FLX_ORDER_ERRORS_FIRST = 0, FLX_ORDER_COUNT_FIRST = 1, FLX_ORDER_NONE = 2 };             /* search.hpp:44-46 */
enum { FLX_CHOICE_ROUND_ROBIN = 0, FLX_CHOICE_FULL_GROUPS = 1, FLX_CHOICE_FIRST_REPORTED = 2 };  /* search.hpp:50-52 */

typedef struct flx_search_config {   /* search::search_config, search.hpp:56-62; defaults floxer_cli.hpp:52-56 */
    uint64_t max_num_anchors_hard;
    uint64_t max_num_anchors_soft;
    int32_t anchor_group_order;
    int32_t anchor_choice_strategy;
    int32_t erase_useless_anchors;
    int32_t reserved;
} flx_search_config;

typedef struct flx_anchor {          /* search::anchor_t, search.hpp:27-38 */
    uint32_t seed_index;             /* index into the seeds array of the call */
    uint32_t pex_leaf_index;
    uint32_t reference_id;
    uint32_t num_errors;
    uint64_t reference_position;
} flx_anchor;

typedef struct flx_seed_stats {      /* search_result::anchors_of_seed, search.hpp:80-87 */
    uint32_t num_kept_useful_anchors;
    uint32_t num_kept_raw_anchors;
    uint32_t num_excluded_raw_anchors_by_soft_cap;
    uint32_t fully_excluded;
} flx_seed_stats;

/* anchors are written in search_result::anchor_iterator order (seed, reference, position; search.cpp:78-100).
 * n_anchors: in = capacity, out = number produced (FLX_ERR_CAPACITY if larger than capacity). */
int flx_search_seeds(flx_ctx* ctx, const uint8_t* seq_pool, uint64_t seq_pool_len, const flx_seed* seeds, uint64_t n_seeds,
                     const flx_search_config* cfg, flx_anchor* out_anchors, uint64_t* n_anchors, flx_seed_stats* out_stats);

/* raw search_ng21::search_n emission for the seeds (test hook for kernel K1): rows {seed_index, lb, len, errors} */
typedef struct flx_hit_group { uint32_t seed_index, lb, len, num_errors; } flx_hit_group;
int flx_search_groups(flx_ctx* ctx, const uint8_t* seq_pool, uint64_t seq_pool_len, const flx_seed* seeds, uint64_t n_seeds,
                      uint64_t max_hits_per_seed, flx_hit_group* out, uint64_t* n_out);

/* ------------------------------------------------------------------------------------------------ seam 2: alignment
 * replaces alignment_result align(span<const u8> reference, span<const u8> query, alignment_config const&)
 * (alignment.hpp:57-77, alignment.cpp:83-181), batched. */
enum { FLX_MODE_EXISTS = 0, FLX_MODE_WITHOUT_CIGAR = 1, FLX_MODE_WITH_CIGAR = 2 };                /* alignment.hpp:53-55 */
typedef struct flx_align_job {
    uint64_t ref_offset;     /* into ref_pool, or into the context's reference text when ref_pool == NULL */
    uint64_t query_offset;   /* into query_pool */
    uint32_t ref_length;
    uint32_t query_length;
    uint32_t num_allowed_errors;
    uint32_t mode;
} flx_align_job;
typedef struct flx_align_result {
    uint32_t exists;         /* alignment_outcome::alignment_exists */
    uint32_t num_errors;
    uint64_t begin;          /* start in the given reference window (caller adds reference_span_offset) */
    uint64_t cigar_offset;   /* into cigar_pool (words) */
    uint32_t cigar_length;
    uint32_t reserved;
} flx_align_result;
int flx_align_batch(flx_ctx* ctx, const uint8_t* ref_pool, uint64_t ref_pool_len, const uint8_t* query_pool,
                    uint64_t query_pool_len, const flx_align_job* jobs, uint64_t n_jobs, flx_align_result* out,
                    uint32_t* cigar_pool, uint64_t* cigar_pool_words /* in: capacity, out: used */);

/* ------------------------------------------------------------------------------------------------ seam 3: whole path
 * replaces parallelization::spawn_search_task + spawn_verification_task + query_verifier::verify +
 * alignment_output::write_alignments_for_query (parallelization.cpp:45-293, verification.hpp:22-48, output.cpp:49-108)
 * for a batch of reads, with --threads 1 record order. */
typedef struct flx_params {          /* cli::command_line_input, floxer_cli.hpp:41-70 */
    double query_error_probability;  /* < 0: use query_num_errors */
    uint64_t query_num_errors;
    uint64_t pex_seed_num_errors;    /* default 2 */
    flx_search_config search;
    uint64_t seed_sampling_step_size;/* default 1 */
    int32_t bottom_up_pex_tree_building;
    int32_t use_interval_optimization;
    double extra_verification_ratio; /* default 0.05 */
    int32_t direct_full_verification;
    int32_t without_cigar;
    uint64_t num_anchors_per_verification_task;   /* default 3000 */
} flx_params;
void flx_params_default(flx_params* p);

typedef struct flx_record {          /* one SAM/BAM record, output.cpp:49-108 */
    uint64_t read_index;
    uint32_t flag;                   /* 0 / 16 / 256 / 272 / 4 */
    int32_t reference_id;            /* -1 when unmapped */
    int32_t position;                /* 0-based, saturated to int32 (output.cpp:85) */
    uint32_t num_errors;             /* NM */
    uint64_t cigar_offset;
    uint32_t cigar_length;
    uint32_t reserved;
} flx_record;

typedef struct flx_run flx_run;      /* result of one batch */
/* Reads are given as one rank pool; read i = pool[offsets[i], offsets[i+1]). The read filters of input.cpp:95-129 apply
 * (filtered reads produce no record and are flagged in the skipped array). */
int flx_align_reads(flx_ctx* ctx, const flx_params* params, const uint8_t* read_pool, const uint64_t* read_offsets,
                    uint64_t n_reads, flx_run** out);
/* The same with the reads already resident in HBM (the measured configuration of bench.py): upload once, align many times. */
typedef struct flx_reads flx_reads;
int flx_reads_upload(flx_ctx* ctx, const uint8_t* read_pool, const uint64_t* read_offsets, uint64_t n_reads, flx_reads** out);
void flx_reads_free(flx_reads* reads);
int flx_align_reads_resident(flx_ctx* ctx, const flx_params* params, const flx_reads* reads, flx_run** out);
uint64_t flx_run_num_records(const flx_run* run);
uint64_t flx_run_num_cigar_words(const flx_run* run);
int flx_run_copy(const flx_run* run, flx_record* records, uint32_t* cigar_words, uint8_t* skipped);
void flx_run_free(flx_run* run);

/* ------------------------------------------------------------------------------------------------ measurement
 * Per-kernel accounting (bench.py): when enabled every launch is bracketed with HIP events on the launch stream. */
typedef struct flx_kernel_stat {
    char name[32];
    uint64_t launches;
    double device_ms;            /* sum of hipEventElapsedTime over the launches */
    uint64_t algorithmic_bytes;  /* bytes the algorithm must move, summed over launches */
    uint64_t work_units;         /* word-steps / rank queries / locates, per kernel */
} flx_kernel_stat;
int flx_ctx_enable_kernel_timing(flx_ctx* ctx, int enable);
int flx_ctx_reset_kernel_stats(flx_ctx* ctx);
int flx_ctx_get_kernel_stats(flx_ctx* ctx, flx_kernel_stat* out, uint32_t* n /* in: capacity, out: count */);

/* Counters of the path since the context was made / they were reset (all batches of all host threads): how many seeds the device
 * handled by itself, how much of the requested DP work was run after de-duplication. Cheap, always on. */
typedef struct flx_path_counters {
    uint64_t seeds, seeds_with_anchors, seeds_excluded_by_hard_cap, seeds_selected_on_host, anchors, cursor_extensions;
    uint64_t inner_tests_requested, root_alignments_requested, root_alignments_found, records, reads;
    uint64_t search_reruns;      /* search launches repeated because a chunk's hits or queued subtrees outgrew their buffers */
    uint64_t reserved[4];
} flx_path_counters;
int flx_ctx_get_path_counters(flx_ctx* ctx, flx_path_counters* out);
int flx_ctx_reset_path_counters(flx_ctx* ctx);

/* ------------------------------------------------------------------------------------------------ statistics (--stats)
 * replaces statistics::search_and_alignment_statistics (include/statistics.hpp:24-172, src/lib/statistics.cpp): the reference's
 * count and eighteen histograms, same names, thresholds and renderings. input_hint: NULL / "real_nanopore" / "simulated"
 * (statistics.cpp:209-221). A context with a statistics object attached adds every read of every batch it aligns (the flx_stats
 * outlives that; not a configuration call: attach before the first batch). flx_stats_format: toml != 0 the TOML file of
 * format_statistics_as_toml, else the terminal entries of format_statistics_for_stdout separated by blank lines; len: in =
 * capacity, out = bytes needed incl. the terminating 0 (FLX_ERR_CAPACITY when too small). */
typedef struct flx_stats flx_stats;
int flx_stats_create(const char* input_hint, flx_stats** out);
void flx_stats_free(flx_stats* stats);
int flx_stats_merge(flx_stats* into, const flx_stats* other);
uint64_t flx_stats_num_queries(const flx_stats* stats);
int flx_stats_format(const flx_stats* stats, int toml, char* buf, uint64_t* len);
int flx_ctx_set_stats(flx_ctx* ctx, flx_stats* stats);

/* ------------------------------------------------------------------------------------------------ file boundary
 * FASTA/FASTQ in (input.cpp:36-148), SAM/BAM out (output.cpp:49-108, 197-212) — used by the floxer-compatible CLI. */
typedef struct flx_sam_writer flx_sam_writer;
int flx_sam_open(const char* path /* .sam or .bam */, const char* const* ref_ids, const uint64_t* ref_lens, uint32_t n_refs,
                 flx_sam_writer** out);
int flx_sam_write(flx_sam_writer* w, const char* const* read_ids, const uint8_t* read_pool, const uint64_t* read_offsets,
                  const char* const* quals, const flx_record* records, uint64_t n_records, const uint32_t* cigar_words);
int flx_sam_close(flx_sam_writer* w);
/* record formatting and BGZF block compression of flx_sam_write on n_threads host threads (default 1; output bytes do not depend on it) */
int flx_sam_set_threads(flx_sam_writer* w, uint32_t n_threads);

/* ------------------------------------------------------------------------------------------------ synthetic inputs
 * The reference's simulator (src/main/simulated_dataset.cpp:30-49, 81-223), multi-threaded and with a portable generator:
 * uniform genome over ACGT; reads = substrings of base_len with exactly floor(error_rate * base_len) distinct positions mutated
 * (mismatch / insertion / deletion uniformly), reverse-complemented with probability revcomp_fraction. Read r depends on
 * (seed, r) only. out_offsets has n_reads + 1 entries; out_chrom / out_pos / out_reverse (may be NULL) receive the truth. */
int flx_sim_genome(uint64_t length, uint64_t seed, uint8_t* out_ranks);
/* the same length of repeat-rich sequence (interspersed repeat families, tandem repeats, low complexity, runs of N, segmental
 * duplications: about half of the bases unique, as in a human genome): the workload floxer's caps -M / -m (search.cpp:190-272) exist for */
int flx_sim_genome_repeats(uint64_t length, uint64_t seed, uint8_t* out_ranks);
int flx_sim_reads(const uint8_t* genome_concat, const uint64_t* chrom_lens, uint32_t n_chrom, uint64_t n_reads, uint32_t base_len,
                  double error_rate, double revcomp_fraction, uint64_t seed, uint8_t* out_pool, uint64_t pool_capacity,
                  uint64_t* out_offsets, uint32_t* out_chrom, uint64_t* out_pos, uint8_t* out_reverse);

#ifdef __cplusplus
}
#endif
#endif /* FLOXER_AMD_H */
