// floxer_amd internal declarations shared by the host sources and the HIP translation unit.
#pragma once

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/floxer_amd.h"

namespace flx {

using u8 = uint8_t;
using u16 = uint16_t;
using u32 = uint32_t;
using u64 = uint64_t;
using i64 = int64_t;

void set_error(const std::string& msg);

// ------------------------------------------------------------------------------------------------ host block pool
// The per-batch lists of the host pipeline (hits, anchors, requests, CIGAR slabs) are tens of MB each and are built and dropped
// once per chunk. Blocks of 256 KB and more come from a process-wide pool of size classes and go back to it when a list is
// destroyed, so after the first batches nothing is mapped, faulted in or unmapped any more (glibc maps every block above 32 MB
// afresh, and concurrent page faults of 16 lanes serialise on the process's mmap lock). FLX_HOST_POOL_MB caps what the pool keeps
// (default 16384).
void* host_pool_get(size_t bytes);
void host_pool_put(void* p, size_t bytes);
// Called for every block the pool takes from / returns to the C library (pin = 1 / 0). The device side of the library sets it to
// page-lock the blocks (hipHostRegister): the lists then go to and come from the GPU by DMA instead of through the runtime's
// staging copies. Null in the HIP-free builds.
// (process-wide, installed once: the first context decides; FLX_NO_PIN is read then)
extern std::atomic<void (*)(void* p, size_t bytes, int pin)> host_pool_pin_hook;
template <class T>
struct PoolAlloc {
    using value_type = T;
    PoolAlloc() = default;
    template <class U> PoolAlloc(PoolAlloc<U> const&) {}
    T* allocate(size_t n) { return static_cast<T*>(host_pool_get(n * sizeof(T))); }
    void deallocate(T* p, size_t n) { host_pool_put(p, n * sizeof(T)); }
    template <class U> bool operator==(PoolAlloc<U> const&) const { return true; }
    template <class U> bool operator!=(PoolAlloc<U> const&) const { return false; }
};
template <class T> using hvec = std::vector<T, PoolAlloc<T>>;

// ------------------------------------------------------------------------------------------------ HBM data layout
// Occurrence table block: 32 BWT positions in 32 bytes, 2 bytes of HBM per text symbol and direction (hg38: 6.2 GB per direction
// out of 288 GB). One lane serves one seed and reads a block with two 16-byte loads (a third load per block was what bound the
// kernel: the L1's miss handling, not HBM; scripts/micro/gather_cost.hip), four blocks share a 128-byte line:
//   w[0..4]  cnt[c] = number of symbol c in bwt[0, 32*b) for c = 0..4, absolute (text < 2^32 symbols); the count of symbol 5 (N)
//            is the position minus the other five
//   w[5..7]  bit-planes of the 32 positions: p0, p1, p2 (plane k = bit k of the symbol; symbols 0..5, positions past the end of
//            the text filled with 7)
struct alignas(32) OccBlock {
    u32 w[8];
};
constexpr u32 OCC_BLOCK_POS = 32;
constexpr u32 TEXT_PAD = 256;     // bytes of padding in front of and behind the device copy of a reference text (the text walk copies up to 176 symbols around a position)

struct HostIndex {
    u64 n = 0;                            // padded text length
    std::vector<u8> text;                 // concatenated references, each followed by 4-(len%4) zero sentinels
    std::vector<u64> seq_start, seq_len;
    std::vector<u32> sa;                  // full suffix array (u32: n < 2^32)
    std::vector<OccBlock> occ[2];         // 0: BWT of text (extendLeft), 1: BWT of reversed text (extendRight)
    u64 C[7] = {0, 0, 0, 0, 0, 0, 0};
    std::vector<u8> bwt[2];               // kept for tests (flx_index_copy_bwt); not uploaded
    // bidirectional cursors {lb, lb_rev, len} of every KMER_Q-mer over A,C,G,T (first symbol most significant): the exact first
    // part of a search starts from a table entry instead of KMER_Q rank pairs
    std::vector<u32> kmer_table;
};
constexpr u32 KMER_Q = 8;
inline bool index_has_arrays(HostIndex const& H) { return H.text.size() == H.n && H.sa.size() == H.n && !H.occ[0].empty() && !H.occ[1].empty() && !H.kmer_table.empty(); }

// Device-side view handed to kernels (plain pointers into HBM)
struct DevIndex {
    const OccBlock* occ[2];
    const u32* sa;
    const u8* text;        // points at text[0]; TEXT_PAD readable bytes on both sides
    const u32* kmer;       // KMER_Q-mer cursor table, 3 words per entry
    // derived from text and sa on the device when a context is made (flx_search.hip): not part of the index file or image
    const u32* isa;        // inverse suffix array: row of the suffix that starts at a text position
    const u64* filter;     // presence bits of the text's filter_k-mers (flx_fm_core.hpp); null: no filter
    const u64* filter_m;   // the same bits under the mirrored code (rightmost symbol lowest): the children of a leftward extension share a word; may be null
    u32 filter_k, filter_tmin;   // strings of filter_tmin .. filter_k symbols can be asked for
    u32 C[7];
    u32 n;
};

// ------------------------------------------------------------------------------------------------ K1: FM search
// Expanded search-scheme entry for one query character in search order (search_schemes::expand):
//   bits 0..19 query position, 20..22 lower bound, 23..25 upper bound, 26 extension direction (1 = right),
//   27 set while the entry belongs to the leading run of exact (lower = upper = 0), rightward, consecutive characters
constexpr u32 SCH_POS_MASK = 0xFFFFF;
inline u32 sch_pack(u32 pos, u32 l, u32 u, bool right, bool exact_prefix) {
    return pos | (l << 20) | (u << 23) | ((right ? 1u : 0u) << 26) | ((exact_prefix ? 1u : 0u) << 27);
}

struct DevSeed {
    u64 seq_off;        // into the device sequence pool
    u64 stack_off;      // first frame of this seed's DFS stack
    u32 length;
    u32 scheme_off;     // first entry of this (length, errors) expanded scheme; searches are consecutive, `length` entries each
    u32 frames_searches;// frames reserved for the DFS stack (bits 0..23) | number of searches (bits 24..)
    u32 id;             // index of the seed in the caller's list (the launch order is by expected cost, see search_seeds_device)
    u32 flags;          // SEED_HAS_DELIM | SEED_NOT_ACGT (flx_fm_core.hpp): what the seed's read may hold
    u32 pad;
};

struct DevFrame {       // 64 bytes: one branching node of the DFS, written when the node is made (four 16-byte stores of one lane)
    // v[0..4]   abs of the child cursors of symbols 1..5 on the extended side (C[c] + occ)
    // v[5..10]  number of rows of the children of symbols 0..5 (their bounds on the other side are prefix sums of these)
    // v[11..14] { lb, lb_rev, abs of the child of symbol 0, state } of the node itself (its number of rows is the sum of its
    //           children's); state = x:20 | e:3 | linfo:2 | rinfo:2 | next_sym:3 | right:1
    // v[15]     mask of the children not taken yet (rewritten when a frame is put on top of this one)
    u32 v[16];
};
static_assert(sizeof(DevFrame) == 64, "frame is four 16-byte slots");

// errors: bits 0..7 the hit's error count, bits 8.. its ordinal among the hits of its seed in the order the kernel found them.
// key: position of the hit in search_n's emission order among the hits of its seed (see fm_search_kernel; 0 from the ordered kernel,
// whose ordinals are the emission order already)
struct DevHit { u32 seed, lb, len, errors; u64 key; };
// ---- seeds of a chunk written on the device (flx_search.hip: seed_build_kernel): a seed is a function of (read, orientation, sampled leaf of
// the read's PEX tree), so the host only describes the reads, their trees' leaves and where each (read, seed class) starts in launch order
struct DevSeedLeaf { u32 from, length, cls, rank; };                  // cls: class (errors, length) index within its tree; rank among the tree's leaves of that class
struct DevSeedClass { u32 pos_base, count, scheme_off, frames_searches; };   // per (read, class of its tree): launch position of the read's forward seeds of the class (reverse: + count)
struct DevSeedRead { u64 pool_fwd, pool_rev; u32 leaf_first, n_leaves, seed_base, class_first, flags, pad; };
struct SeedGen {                                                      // host side of one chunk
    std::vector<DevSeedRead, PoolAlloc<DevSeedRead>> reads;
    std::vector<DevSeedLeaf, PoolAlloc<DevSeedLeaf>> leaves;
    std::vector<DevSeedClass, PoolAlloc<DevSeedClass>> classes;
    std::vector<u64, PoolAlloc<u64>> scheme_table;
    u64 n_seeds = 0;
    u32 max_errors = 0, max_length = 0;
};
struct DevSelStat { u8 useful, raw, flag, excluded; u32 excluded_soft; };   // per seed, from the device-side selection
struct DevOutAnchor { u32 seed_index, leaf, ref_id, errors; u64 pos; };   // = HostAnchor (leaf is filled by the host)

// ------------------------------------------------------------------------------------------------ K3/K4: alignment
struct DevAlignJob {
    u64 ref_off;        // into the device text the launch uses
    u64 q_off;          // into the device query pool (also addresses the Peq planes)
    u64 trace_off;      // first 16-byte slot of this job's trace arena (TRACE launches)
    u32 n, m, k;
    u32 out_index;      // where the result goes
    u64 lastrow_off;    // first entry of this job's last-row values (launches with a last-row buffer)
};
struct DevAlignOut { u32 score; u32 end_col; };    // score 0xFFFFFFFF: no alignment within k

// one window inside a job's column range: its best end column is the rightmost minimum of the job's last row over [first, first+n)
struct DevRowWindow { u64 first; u32 n, k; u32 out_index, pad; };

struct DevTraceJob {
    u64 ref_off, q_off, trace_off, cigar_off;      // cigar_off: first word of this job's CIGAR slab
    u32 n, m, lanes, words_per_lane;
    u32 end_col, cigar_cap, out_index;
    u32 k;                                         // allowed errors (band of the checkpointed trace)
};
struct DevTraceOut { u32 begin; u32 cigar_start; u32 cigar_len; u32 pad; };   // cigar_start relative to the slab

// ------------------------------------------------------------------------------------------------ verification rounds on the device
// The inner PEX levels as device-resident state: every anchor of a chunk with the node it is about to test (flx_rounds.hip).
struct DevVrAnchor {            // 48 bytes
    i64 diag_rel;               // anchor position minus the leaf's first query row (relative to its reference sequence, may be < 0)
    u64 seq_start, seq_len;     // the reference sequence in the padded text
    u64 q_base;                 // pool offset of the read in the anchor's orientation
    u32 tree_base;              // first node of the read's tree in the node table
    u32 query;                  // ordinal of (read, orientation) in the chunk
};
struct DevVrNode { u32 parent, from, rows, errors; };          // parent = index within the tree, 0xFFFFFFFF for the root
enum : u8 { VR_CLIMBING = 0, VR_DEAD = 1, VR_AT_ROOT = 2, VR_SOLO = 3 };      // VR_SOLO: climbing, its next test alone

// ---- the rounds as three launches each (flx_rounds.hip)
// scalars (u32 words): jobs of the round (two counters, taken in turn), anchors left to climb and their smallest node after the round,
// two 64-bit running sums for the accounting (word-steps and sequence bytes of the jobs so far), requests so far (anchors in rounds),
// blocks of vr2_apply that have finished
enum : u32 { VR2_N_JOBS = 0 /* and 1: by the round's parity */, VR2_N_CLIMBING = 2, VR2_SMALLEST = 3, VR2_WORD_STEPS = 4, VR2_BYTES = 6, VR2_N_REQ = 8, VR2_DONE = 9,
             VR2_QUEUE = 10 /* job queue head of the lane-per-job existence kernel */, VR2_QUEUE_ERR = 11 /* windows its buffers did not hold */,
             VR2_PENDING = 12 /* and 13, by parity: anchors the next round will ask for (rounds queued without the host in between) */,
             VR2_SCALARS = 16 };
struct Vr2Buffers {
    const DevVrAnchor* anchors; const DevVrNode* nodes;
    const u32* q_first;                          // per query (read x orientation): its first anchor; n_queries + 1 entries
    u32* node; u8* status;                       // per anchor, mutable
    u32* a_slot;                                 // per anchor in the round: first job slot of its cluster << 2 | union job << 1 | own / intersection job; else ~0
    DevAlignJob* jobs; DevAlignOut* outs;        // the round's job list (at most two per anchor in the round) and its results
    u32* scalars;
};

// launch geometry for one alignment job shape. queue: hand-over slots per job in LDS (a power of two; 0: the ring never waits, see ring_delay)
struct AlignShape { u32 words_per_lane; u32 lanes_per_job; u32 banded; u32 queue = 0; };

// Banded TRACE launches (K4) do not store the trace itself. A lane computes its word group 16 columns (one block) per step, group g
// takes block b at block-step T = b + g; per (block-step, ring lane, word) the launch keeps the 16 pairs of horizontal-delta bits
// that enter the word from above (one u32) and the word's vertical delta vectors {vp, vn} before the block (one 16-byte slot). Any
// word's trace bits over any block are recomputed from those by the traceback kernel (1.25 B instead of 16 B per column and word).
struct TraceLayout { u64 steps, carry_words, carry_slots, ckpt_slots; };   // steps = block-steps; slots = 16-byte units; carry region first
#if defined(__HIPCC__)
#define FLX_HD __host__ __device__
#else
#define FLX_HD
#endif

// The ring schedule of the banded block kernels (ed_block_body). A job's word groups (64 W rows each) go round its R lanes: lane p takes
// the groups p, p + R, ...; group g computes the blocks (16 columns) b_lo(g) .. b_hi(g) of its band, block b at block-step
// T = b + ring_offset(g), one step behind the group above it, whose bottom row it needs. With R so large that group g + R starts only after
// group g has ended (64 W (R - 1) + R + 1 > diagonals: what round 3 asked of a shape) a lane idles from the end of one group to the start of
// its next: 40 % of the block-steps of a 10-kb root alignment on 16 lanes. With fewer lanes a lane is still at work when its next group
// could start: every revolution of the ring then begins `delay` block-steps later than the one above it would allow, offset(g) =
// g + (g / R) delay, the lanes work back to back, and what lane R - 1 hands down to lane 0 waits those steps in a queue in LDS.
constexpr u32 RING_QUEUE_MAX = 128;          // most hand-over slots per job: delay + 1 of them are in use
FLX_HD inline void ring_group_blocks(int n, int m, int k, int W, int Lg, int pad, int g, int& b_lo, int& b_hi) {
    int const band_hi = n - m + k;
    int const r0 = 64 * W * g - pad > 0 ? 64 * W * g - pad : 0, r1 = 64 * W * (g + 1) - pad;
    b_lo = (r0 - k > 0 ? r0 - k : 0) >> 4;
    int const hi = r1 - 1 + band_hi < n - 1 ? r1 - 1 + band_hi : n - 1;
    b_hi = hi >> 4;
    if (g + 1 < Lg) { int const next_lo = (r1 - k > 0 ? r1 - k : 0) >> 4; if (next_lo > b_hi) b_hi = next_lo; }      // (kept going for the block in which the group below starts)
}
FLX_HD inline u32 ring_delay(u32 n, u32 m, u32 k, u32 W, u32 R) {
    if (n == 0 || m == 0 || (u64)n + k < m) return 0;
    int const nw = (int)((m + 63u) / 64u), Lg = (nw + (int)W - 1) / (int)W, pad = Lg * 64 * (int)W - (int)m;
    int delay = 0;
    for (int g = 0; g + (int)R < Lg; ++g) {
        int lo0, hi0, lo1, hi1;
        ring_group_blocks((int)n, (int)m, (int)k, (int)W, Lg, pad, g, lo0, hi0);
        ring_group_blocks((int)n, (int)m, (int)k, (int)W, Lg, pad, g + (int)R, lo1, hi1);
        int const d = hi0 - lo1 + 1 - (int)R;             // the lane is free at block-step hi0 + offset(g) + 1; its next group wants lo1 + offset(g) + R + delay
        if (d > delay) delay = d;
    }
    return (u32)delay;
}
FLX_HD inline u64 ring_offset(u32 g, u32 R, u32 delay) { return (u64)g + (u64)(g / R) * delay; }
// block-steps of a job: the last group's last block is the window's last
FLX_HD inline u64 ring_steps(u32 n, u32 m, u32 k, u32 W, u32 R) {
    u64 const nw = (m + 63u) / 64u, groups = (nw + W - 1) / W;
    return (n ? (u64)(n - 1) / 16 : 0) + ring_offset((u32)groups - 1u, R, ring_delay(n, m, k, W, R)) + 1;
}
FLX_HD inline TraceLayout ckpt_trace_layout(u32 n, u32 m, u32 k, u32 W, u32 R) {
    TraceLayout l;
    l.steps = ring_steps(n, m, k, W, R);
    l.carry_words = l.steps * R * W;
    l.carry_slots = (l.carry_words + 3) / 4;
    l.ckpt_slots = l.steps * R * W;
    return l;
}
// parallel = false: the shape that occupies the fewest wave slots (launches with many jobs); true: the shape with the shortest
// per-step chain and the most waves (launches whose jobs would not fill the GPU otherwise)
AlignShape choose_align_shape(u32 n, u32 m, u32 k, bool parallel = false);
u64 align_trace_slots(u32 n, u32 m, u32 k, AlignShape sh);     // 16-byte trace slots a TRACE launch of this shape needs for one job
u32 align_supported_max_query();
u32 fm_search_max_keyed_length();

// ------------------------------------------------------------------------------------------------ device launchers (flx_device.hip)
struct KernelTimer;   // opaque, owned by the context

struct DeviceApi {
    // all return 0 or a hipError_t (non-zero)
    // suffix array of text[0, n) on the device (a suffix that is a prefix of another sorts first); out: n entries on the host
    static int suffix_array(int hip_device, const u8* text, u64 n, u32* out);
    // suffix array, BWT of the text and of the reversed text and both occurrence tables (n / 64 + 1 blocks each), built on the
    // device; all outputs on the host
    static int index_arrays(int hip_device, const u8* text, u64 n, u32* out_sa, u8* out_bwt0, u8* out_bwt1, OccBlock* out_occ0, OccBlock* out_occ1);
    static int build_peq(void* stream, const u8* d_seq, u64 len, u64* d_peq);
    // The DFS in the reference's order, frames on per-seed stacks in HBM (DevSeed::stack_off / frames into d_stack): for first_reported
    // (the first n rows in emission order) and the raw-emission hook. d_seed_cnt (may be null): number of hits of every seed; each hit
    // then carries its ordinal within its seed in errors >> 8.
    static int search(void* stream, const DevIndex& idx, const u8* d_seq, const u64* d_scheme, const DevSeed* d_seeds,
                      u32 n_seeds, u32 max_hits_per_seed, DevFrame* d_stack, DevHit* d_hits, u32 hit_cap,
                      u32* d_counters, u32* d_seed_cnt = nullptr);
    // flx_search.hip: the walk with its stack in LDS plus the presence filter (d_qpack: 2-bit form of d_seq, null: no filter) and the
    // text walk of one-row subtrees (d_items: room for item_cap queued subtrees, null: none are queued). d_counters: 32 zeroed words.
    // d_seed_rows (n_seeds zeroed words, or null): rows reported per seed over all lanes; with it the lanes of a wave share the subtrees
    // of heavy seeds once the seed queue is dry, and busy waves hand subtrees to waves that have run out of work through d_mailboxes
    // (see fm_search_filter_kernel)
    static int search_filtered(void* stream, const DevIndex& idx, const u8* d_seq, const u32* d_qpack, const u64* d_scheme, const DevSeed* d_seeds,
                               u32 n_seeds, u32 max_hits_per_seed, u32 frame_levels, DevHit* d_hits, u32 hit_cap, DevHit* d_items, u32 item_cap,
                               u32* d_counters, u32* d_seed_cnt, u32* d_seed_rows, void* d_mailboxes, u32 mailbox_waves, u32 concurrent_launches,
                               bool long_seeds = false);
    // long_seeds: most seeds of the launch have more than 64 symbols (the text walk then takes its larger LDS windows: up to 160 symbols)
    // the mailboxes through which the waves of one launch of the filter walk hand subtrees to each other (room for `waves` waves; may be null)
    static size_t mailbox_bytes(u32 waves);
    // the tables a context derives from text and suffix array: bytes of isa + filter for a text of n symbols; derive_index fills
    // d_isa (n words) and d_filter (null: no filter) and sets idx.isa / filter / filter_k / filter_tmin
    static size_t derived_bytes(u64 n, u32* filter_k_out);
    static int derive_index(void* stream, DevIndex& idx, u32* d_isa, u64* d_filter, u64* d_filter_mirrored = nullptr);
    // reserves the hardware queue's scratch for the pipeline's kernels (see scratch_warm_kernel)
    static int warm_scratch(void* stream);
    // DevSeeds of a chunk in launch order from the chunk's description (all pointers on the device)
    static int build_seeds(void* stream, const DevSeedRead* reads, u32 n_reads, const DevSeedLeaf* leaves, const DevSeedClass* classes, DevSeed* out);
    // 2-bit form of a sequence pool (pack_words_for(len) words, flx_fm_core.hpp)
    static int pack_pool(void* stream, const u8* d_seq, u64 len, u32* d_qpack);
    // anchor selection on the device (see seed_select_kernel). d_seed_cnt, d_hit_offset, d_n_out, d_out_offset: n_seeds + 1 entries
    // (the caller zeroes the last entry of d_seed_cnt and d_n_out); d_stat: one DevSelStat per seed;
    // d_grouped: as many entries as d_hits; d_out: one entry per selected anchor (at most the number of rows of the handled seeds);
    // d_lists: 3 * n_seeds + 3 entries (the lists of light, heavy and many-group seeds and their lengths)
    static size_t select_scan_bytes(u32 n_seeds);
    static int select(void* stream, const DevHit* d_hits, const u32* d_counters, u32 hit_cap, u32* d_seed_cnt, u32* d_hit_offset,
                      DevHit* d_grouped, u32 n_seeds, const DevIndex& idx, const u64* d_seq_start, u32 n_ref, u32 hard_cap, u32 soft_cap,
                      bool erase, void* d_stat, u32* d_n_out, u32* d_out_offset, DevOutAnchor* d_out, u32 out_cap, u32* d_rows,
                      u32* d_row_offset, DevOutAnchor* d_sparse, u32 sparse_cap, void* d_scan_tmp, size_t scan_bytes, u32* d_lists);
    static int locate(void* stream, const DevIndex& idx, const u32* d_rows, u32 n, u32* d_out);
    // d_lastrow (banded TRACE launches only, may be null): D[m][c] of every computed column c, 0xFFFF elsewhere (pre-filled by the caller)
    static int align(void* stream, const u8* d_text, const u64* d_peq, const DevAlignJob* d_jobs, u32 n_jobs, AlignShape shape,
                     bool trace, u64* d_trace, DevAlignOut* d_out, u16* d_lastrow = nullptr);
    // flx_rounds.hip: a round = vr2_request (job list and count on the device), align_exists_counted on it, vr2_apply
    // check_pending: the round is one of a queued series; it does nothing when the round before it counted no anchor for it (next_limit of
    // that round's vr2_apply = this round's limit; parity as for the job counters)
    static int vr2_request(void* stream, Vr2Buffers const& B, u32 n_queries, u32 limit, u32 acct_words, u32 width_cap, u32 parity, bool check_pending = false);
    static int vr2_apply(void* stream, Vr2Buffers const& B, u32 n_anchors, u32* host_scalars, u32 next_limit = 0xFFFFFFFFu, u32 parity = 0);
    // existence tests, one lane per job with Ukkonen's cutoff (flx_lanes.hip): at most `waves` waves take the jobs from a queue (*d_queue = 0
    // at the start, d_queue[1] counts jobs whose windows the rows did not hold); d_n_jobs (optional): the number of jobs is on the device
    // (at most max_jobs); cap_blocks >= (64 + n - m + 2k) / 16 + 3 of every job, exists_lane_lds_bytes(cap_blocks) <= 160 KB;
    // d_stats (optional): 8 x u64 counters; lanes_per_job (1, 2, 4, 8, 16): a team of lanes per job, each on every P-th word group, one block
    // behind the lane of the group above (jobs with long chains and too few of them to fill the chip one lane each)
    static size_t exists_lane_lds_bytes(u32 cap_blocks);
    static int align_exists_lanes(void* stream, const u8* d_text, const u64* d_peq, const DevAlignJob* d_jobs, u32 max_jobs, const u32* d_n_jobs,
                                  u32* d_queue, u32 waves, u32 cap_blocks, DevAlignOut* d_out, unsigned long long* d_stats, u32 lanes_per_job = 1);
    static int align_exists_counted(void* stream, const u8* d_text, const u64* d_peq, const DevAlignJob* d_jobs, u32 max_jobs, const u32* d_n_jobs,
                                    AlignShape shape, u32 max_waves, DevAlignOut* d_out, u32* d_err = nullptr);
    static AlignShape shape_holding(u32 nw, i64 width, bool parallel);
    static u64 shape_width_cap(u32 nw, AlignShape sh);       // the widest band (n - m + 2k) the shape holds for jobs of nw words
    static int lastrow_min(void* stream, const u16* d_lastrow, const DevRowWindow* d_windows, u32 n_windows, DevAlignOut* d_out);
    static int traceback(void* stream, const u8* d_text, const u8* d_query, const u64* d_peq, const u64* d_trace,
                         const DevTraceJob* d_jobs, u32 n_jobs, bool checkpointed, u32* d_cigar, DevTraceOut* d_out);
};

// ------------------------------------------------------------------------------------------------ host logic
u64 ceil_div(u64 a, u64 b);
u64 fp_aware_ceil(double v);
int32_t saturate_i32(u64 v);
u8 char_to_rank(char c);
char rank_to_char(u8 r);
void reverse_complement(const u8* in, u64 n, u8* out);

struct PexTree {
    std::vector<flx_pex_node> inner, leaves;
    const flx_pex_node& root() const { return inner.empty() ? leaves[0] : inner[0]; }
};
PexTree build_pex_tree(u64 len, u64 k, u64 s, bool bottom_up);

struct SearchDef { std::vector<u32> pi, l, u; };
const std::vector<SearchDef>& optimum_scheme(u32 k);
// expanded + packed entries (low word sch_pack, high word see flx_fm_core.hpp) for all searches of optimum(0,k) at this length,
// `len` entries per search; empty if the scheme cannot be expanded (len < number of parts)
std::vector<u64> expanded_scheme(u32 k, u32 len);

// hip_device >= 0: the two suffix arrays are built on that device (prefix doubling), else on the host (SA-IS)
HostIndex* build_host_index(const u8* concat, const u64* lens, u32 n_refs, int hip_device = -1);
int save_host_index(const HostIndex& idx, const char* path);
HostIndex* load_host_index(const char* path);

}  // namespace flx

struct flx_index { flx::HostIndex* host = nullptr; };
