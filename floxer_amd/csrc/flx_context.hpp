// Device context: HBM-resident index, grow-only workspaces, stream, per-kernel accounting.
#pragma once

#include <hip/hip_runtime.h>

#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "flx_internal.hpp"

namespace flx {

#define FLX_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t const e__ = (expr);                                                                  \
        if (e__ != hipSuccess) {                                                                        \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e__));                              \
            return FLX_ERR_NO_DEVICE;                                                                   \
        }                                                                                               \
    } while (0)

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t cap = 0;
    DeviceBuffer() = default;
    DeviceBuffer(DeviceBuffer const&) = delete;
    DeviceBuffer& operator=(DeviceBuffer const&) = delete;
    ~DeviceBuffer() { release(); }   // (error paths drop half-made owners: nothing stays allocated)
    int ensure(size_t bytes, bool exact = false);      // grow-only; contents are NOT preserved; exact: no growth slack
    void release();
    void take(DeviceBuffer& other) { release(); ptr = other.ptr; cap = other.cap; other.ptr = nullptr; other.cap = 0; }      // this one owns other's memory now
    template <class T> T* as() const { return reinterpret_cast<T*>(ptr); }
};

struct PendingTiming { std::string name; hipEvent_t start, stop; u64 bytes, units; };

}  // namespace flx


struct flx_ctx;
struct flx_stats;

namespace flx {
// One execution lane: a HIP stream with its own grow-only workspaces. A context runs the slices of a read batch on several
// lanes concurrently (one host thread each) so that the host-side phases of one slice overlap the kernels of the others.
struct Lane {
    flx_ctx* ctx = nullptr;
    int id = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    DeviceBuffer seq, seq_rev, peq, peq_rev, scheme, seeds, stack, hits, counters, rows, rows_out, qpack, items;
    DeviceBuffer jobs, job_out, trace, tjobs, tjob_out, cigar, user_text, user_text_rev, lastrow, row_windows, row_out,
        seed_cnt, hit_off, grouped, sel_stat, sel_n, sel_off, sel_out, sel_tmp, sel_rows, sel_row_off, sel_sparse, sel_lists, vr, lane_rows, seed_gen, mailboxes;
    size_t trace_budget_bytes = 0;
    std::vector<PendingTiming> pending;
    std::vector<hipEvent_t> event_pool;
    hipEvent_t sync_event = nullptr;
    u32* vr_host_scalars = nullptr;  // page-locked, mapped: a round's scalars, written by the device (flx_rounds.hip)
    bool has_run = false;            // a chunk has run here (its workspaces have their working sizes)
    double hits_per_seed = 0, items_per_seed = 0, sel_rows_per_seed = 0;      // of the last search here: the next one's buffers are sized for that and a margin
    int wait_idle();                 // the stream has drained (the thread sleeps on a blocking event unless FLX_SPIN_SYNC is set)
    int sync();                      // wait_idle + fold pending timings into the context's statistics
    std::vector<DeviceBuffer*> workspaces();
    int size_like(Lane& other);      // grow this lane's workspaces to the other lane's capacities                      // stream synchronize + fold pending timings into the context's statistics
    hipEvent_t get_event();
    void release_all();
};
}  // namespace flx

struct flx_ctx {
    ~flx_ctx();                      // releases streams, workspaces and the index image it owns (also on a half-made context)
    int device = 0;
    const flx::HostIndex* hidx = nullptr;
    flx::DevIndex didx{};
    flx::DeviceBuffer occ0, occ1, sa, text, text_rev, kmer, seq_start;
    flx::DeviceBuffer isa, filter, filter_m;   // derived from text and suffix array when the context is made (flx_search.hip)
    bool text_rev_ready = false;
    std::mutex mu;                   // guards text_rev upload and the statistics
    std::vector<std::unique_ptr<flx::Lane>> lanes;
    hipStream_t upload_stream = nullptr;   // flx_reads_upload copies here, so that it can run while the lanes are busy
    // Device buffers of read batches that have been freed (flx_reads_free), by role (pool, 2-bit form, Peq planes, reversed pool, its Peq
    // planes): the next batch takes them instead of allocating. hipFree waits for the whole device, so a caller that hands batches over in
    // host memory (flx_align_reads: three buffers made and freed per batch) used to stall every lane three times per batch.
    std::mutex spare_mu;
    std::vector<std::unique_ptr<flx::DeviceBuffer>> spare_read_buffers[5];
    bool external_stream = false;    // a caller-owned stream is installed on lane 0: run on that lane only
    // accounting
    bool timing = false;
    std::map<std::string, flx_kernel_stat> stats;
    std::vector<std::string> stat_order;
    flx_path_counters path{};        // guarded by mu
    flx_stats* read_stats = nullptr; // flx_ctx_set_stats: every batch adds its reads (flx_stats.cpp); not owned

    // lanes are handed out one holder at a time, so calls on one context may overlap (each waits for a free lane)
    std::mutex lane_mu;
    std::condition_variable lane_cv;
    std::vector<int> free_lanes;     // indices into `lanes`
    flx::Lane* acquire_lane(int wanted = -1);   // blocks; wanted >= 0: that lane. Prefers the lane released last.
    void release_lane(flx::Lane* lane);
    void warm_one_cold_lane(flx::Lane* like);   // gives one lane that never ran the workspace sizes of `like` (still held)

    // K1 launches in flight at a time (see search_seeds_device): a counting semaphore
    int k1_tokens = 0;               // guarded by lane_mu; 0 = unlimited
    int k1_running = 0;
    void k1_acquire();
    void k1_release();

    flx::Lane* lane0() { return lanes[0].get(); }
    int sync_all();
    void account(const char* name, flx::u64 bytes, flx::u64 units, hipEvent_t start, hipEvent_t stop);   // caller holds mu
};

namespace flx {

struct LaneLease {
    flx_ctx* ctx;
    Lane* lane;
    explicit LaneLease(flx_ctx* c, int wanted = -1) : ctx(c), lane(c->acquire_lane(wanted)) {}
    ~LaneLease() { ctx->release_lane(lane); }
    LaneLease(LaneLease const&) = delete;
    LaneLease& operator=(LaneLease const&) = delete;
};

// brackets a launch with events when timing is enabled
template <class F>
int timed_launch(Lane* lane, const char* name, u64 bytes, u64 units, F&& launch) {
    if (!lane->ctx->timing) {
        int const rc = launch();
        if (rc != 0) { set_error(std::string(name) + ": launch failed: " + hipGetErrorString((hipError_t)rc)); return FLX_ERR_NO_DEVICE; }
        return FLX_OK;
    }
    hipEvent_t const a = lane->get_event(), b = lane->get_event();
    FLX_HIP(hipEventRecord(a, lane->stream));
    int const rc = launch();
    FLX_HIP(hipEventRecord(b, lane->stream));
    if (rc != 0) { set_error(std::string(name) + ": launch failed: " + hipGetErrorString((hipError_t)rc)); return FLX_ERR_NO_DEVICE; }
    lane->pending.push_back(PendingTiming{name, a, b, bytes, units});
    return FLX_OK;
}

// ---- pipeline pieces (flx_pipeline.cpp)
struct HostAnchor { u32 seed_index, leaf, ref_id, errors; u64 pos; };
struct SeedStats { u32 useful, raw, excluded_soft, fully_excluded; };

// d_seq_pool_or_null: the pool is resident (then d_qpack_or_null may be its 2-bit form); seed_flags (per seed, SEED_* of
// flx_fm_core.hpp) may be null when the host pool is given (they are read off it)
int search_seeds_device(Lane* lane, const u8* d_seq_pool_or_null, const u8* h_seq_pool, u64 pool_len, const flx_seed* seeds,
                        u64 n_seeds, const flx_search_config& cfg, hvec<HostAnchor>& anchors, hvec<SeedStats>& stats,
                        hvec<DevHit>* raw_hits, u64 raw_max_hits, const u32* d_qpack_or_null = nullptr, const u8* seed_flags = nullptr,
                        const SeedGen* gen = nullptr);
// gen: `seeds` is null and the chunk's seeds are written on the device from this description (their ids = the order the caller would have
// listed them in: read by read, forward then reverse complement, leaf by leaf); the anchors' leaf is left to the caller; returns
// SEARCH_NEEDS_HOST_SEEDS (nothing done that counts) for the forms that read the list (ordered walk, host-side grouping): call again with it.
constexpr int SEARCH_NEEDS_HOST_SEEDS = 1;

}  // namespace flx
