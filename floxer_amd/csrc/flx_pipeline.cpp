// Host orchestration of the device path: seeding (K1/K2 + anchor selection, search.cpp:143-324), alignment batches
// (K0/K3/K4/K5, alignment.cpp:83-181) and the level-synchronous PEX verification driver (verification.cpp:8-245) with
// --threads 1 record order (parallelization.cpp:14-43, 230-276; output.cpp:49-108).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <numeric>
#include <queue>
#include <set>
#include <thread>
#include <unordered_map>

#include "flx_context.hpp"
#include "flx_fm_core.hpp"
#include "flx_stats.hpp"

namespace flx {

// ================================================================================================ buffers / context
int DeviceBuffer::ensure(size_t bytes, bool exact) {
    if (bytes <= cap && ptr) return FLX_OK;
    static int const debug = getenv("FLX_ALLOC_DEBUG") ? 1 : 0;
    if (debug) fprintf(stderr, "[flx alloc] %.3f device buffer grows %zu -> %zu bytes\n", std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(), cap, bytes);
    release();
    size_t const want = exact ? std::max<size_t>(bytes, 4096) : std::max<size_t>(bytes + bytes / 2, 4096);       // 50 % slack: batches of a run differ by a few per cent
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    size_t got = want;
    if (e != hipSuccess) { (void)hipGetLastError(); got = bytes; e = hipMalloc(&p, bytes); }   // retry without slack
    if (e != hipSuccess) { set_error(std::string("hipMalloc of ") + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e)); return FLX_ERR_NO_DEVICE; }
    ptr = p;
    cap = got;
    if (debug) fprintf(stderr, "[flx alloc] buffer %p .. %p (%zu bytes, asked %zu)\n", p, (void*)((char*)p + got), got, bytes);
    return FLX_OK;
}
void DeviceBuffer::release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
}

}  // namespace flx

using namespace flx;

namespace {
// FLX_HOST_PROFILE=1 prints wall-clock milliseconds of the host phases of flx_align_reads_resident to stderr
struct PhaseTimer {
    struct Row { const char* name; double wall, cpu; };
    bool on;
    std::chrono::steady_clock::time_point t;
    double cpu_t = 0;
    hvec<Row> rows;
    const char* what;
    static double thread_cpu_ms() {
        timespec ts{};
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
        return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
    }
    explicit PhaseTimer(const char* what_ = "slice") : on(getenv("FLX_HOST_PROFILE") != nullptr), t(std::chrono::steady_clock::now()), what(what_) {
        if (on) cpu_t = thread_cpu_ms();
    }
    void mark(const char* name) {
        if (!on) return;
        auto const now = std::chrono::steady_clock::now();
        double const cpu_now = thread_cpu_ms();
        rows.push_back({name, std::chrono::duration<double, std::milli>(now - t).count(), cpu_now - cpu_t});
        t = now;
        cpu_t = cpu_now;
    }
    ~PhaseTimer() {                                  // name=wall/cpu of the calling thread, milliseconds
        if (!on) return;
        double total = 0, cpu = 0;
        for (auto& r : rows) { total += r.wall; cpu += r.cpu; }
        fprintf(stderr, "[flx host profile] %s total %.2f/%.2f ms:", what, total, cpu);
        for (auto& r : rows) fprintf(stderr, " %s=%.2f/%.2f", r.name, r.wall, r.cpu);
        fprintf(stderr, "\n");
    }
};
}  // namespace

hipEvent_t Lane::get_event() {
    if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
hvec<DeviceBuffer*> lane_workspaces(Lane& l) {
    return {&l.seq, &l.seq_rev, &l.peq, &l.peq_rev, &l.scheme, &l.seeds, &l.stack, &l.hits, &l.counters, &l.rows, &l.rows_out, &l.jobs,
            &l.job_out, &l.trace, &l.tjobs, &l.tjob_out, &l.cigar, &l.user_text, &l.user_text_rev, &l.lastrow, &l.row_windows, &l.row_out,
            &l.seed_cnt, &l.hit_off, &l.grouped, &l.sel_stat, &l.sel_n, &l.sel_off, &l.sel_out, &l.sel_tmp, &l.sel_rows, &l.sel_row_off, &l.sel_sparse, &l.sel_lists, &l.vr, &l.lane_rows,
            &l.qpack, &l.items, &l.seed_gen, &l.mailboxes};
}
// FLX_ALLOC_DEBUG: the address ranges of a lane's workspaces (a GPU memory fault reports an address)
static void dump_lane_buffers(Lane& l, const char* when) {
    static const char* const names[] = {"seq", "seq_rev", "peq", "peq_rev", "scheme", "seeds", "stack", "hits", "counters", "rows", "rows_out", "jobs", "job_out",
        "trace", "tjobs", "tjob_out", "cigar", "user_text", "user_text_rev", "lastrow", "row_windows", "row_out", "seed_cnt", "hit_off", "grouped", "sel_stat",
        "sel_n", "sel_off", "sel_out", "sel_tmp", "sel_rows", "sel_row_off", "sel_sparse", "sel_lists", "vr", "lane_rows", "qpack", "items", "seed_gen", "mailboxes"};
    auto const ws = lane_workspaces(l);
    for (size_t i = 0; i < ws.size(); ++i)
        if (ws[i]->ptr) fprintf(stderr, "[flx alloc] lane %d %s %s %p .. %p\n", l.id, when, names[i], ws[i]->ptr, (void*)((char*)ws[i]->ptr + ws[i]->cap));
}
std::vector<DeviceBuffer*> Lane::workspaces() { auto v = lane_workspaces(*this); return std::vector<DeviceBuffer*>(v.begin(), v.end()); }
int Lane::size_like(Lane& other) {
    auto mine = workspaces(), theirs = other.workspaces();
    for (size_t i = 0; i < mine.size(); ++i)
        if (theirs[i]->cap > mine[i]->cap) {
            // the other lane's capacity already holds the growth slack: take exactly that
            void* p = nullptr;
            if (hipMalloc(&p, theirs[i]->cap) != hipSuccess) { (void)hipGetLastError(); return FLX_OK; }     // best effort
            mine[i]->release();
            mine[i]->ptr = p;
            mine[i]->cap = theirs[i]->cap;
        }
    return FLX_OK;
}
void Lane::release_all() {
    for (DeviceBuffer* b : workspaces()) b->release();
    for (auto& p : pending) { (void)hipEventDestroy(p.start); (void)hipEventDestroy(p.stop); }
    for (auto e : event_pool) (void)hipEventDestroy(e);
    pending.clear();
    event_pool.clear();
    if (own_stream) (void)hipStreamDestroy(own_stream);
    own_stream = stream = nullptr;
    if (vr_host_scalars) { (void)hipHostFree(vr_host_scalars); vr_host_scalars = nullptr; }
}
void flx_ctx::account(const char* name, u64 bytes, u64 units, hipEvent_t start, hipEvent_t stop) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, start, stop);
    auto it = stats.find(name);
    if (it == stats.end()) {
        flx_kernel_stat st{};
        strncpy(st.name, name, sizeof(st.name) - 1);
        it = stats.emplace(name, st).first;
        stat_order.push_back(name);
    }
    it->second.launches += 1;
    it->second.device_ms += ms;
    it->second.algorithmic_bytes += bytes;
    it->second.work_units += units;
}
int Lane::wait_idle() {
    // hipStreamSynchronize and hipEventSynchronize keep the calling core busy for as long as the GPU works (also with
    // hipEventBlockingSync on this runtime); polling an event with short sleeps leaves the core to the other lanes' host work.
    static int const spin = getenv("FLX_SPIN_SYNC") ? 1 : 0;
    if (spin) { FLX_HIP(hipStreamSynchronize(stream)); return FLX_OK; }
    if (!sync_event) FLX_HIP(hipEventCreateWithFlags(&sync_event, hipEventDisableTiming));
    FLX_HIP(hipEventRecord(sync_event, stream));
    // (the sleeps grow with the time already waited: a long kernel is not polled thousands of times, a short one is not overslept
    // by more than a fifth of its duration)
    static unsigned const max_sleep = getenv("FLX_POLL_MAX_US") ? (unsigned)atoi(getenv("FLX_POLL_MAX_US")) : 1000u;
    for (unsigned sleep_us = 20;;) {
        hipError_t const e = hipEventQuery(sync_event);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) { set_error(std::string("hipEventQuery: ") + hipGetErrorString(e)); return FLX_ERR_NO_DEVICE; }
        std::this_thread::sleep_for(std::chrono::microseconds(sleep_us));
        sleep_us = std::min(max_sleep, sleep_us + sleep_us / 4 + 1);
    }
    return FLX_OK;
}
int Lane::sync() {
    if (int const rc = wait_idle()) return rc;
    if (!pending.empty()) {
        std::lock_guard<std::mutex> g(ctx->mu);
        for (auto& p : pending) {
            ctx->account(p.name.c_str(), p.bytes, p.units, p.start, p.stop);
            event_pool.push_back(p.start);
            event_pool.push_back(p.stop);
        }
        pending.clear();
    }
    return FLX_OK;
}
flx::Lane* flx_ctx::acquire_lane(int wanted) {
    std::unique_lock<std::mutex> g(lane_mu);
    while (true) {
        for (size_t i = free_lanes.size(); i-- > 0;)            // the lane released last first: its workspaces are warm
            if (wanted < 0 || free_lanes[i] == wanted) {
                int const id = free_lanes[i];
                free_lanes.erase(free_lanes.begin() + (long)i);
                return lanes[(size_t)id].get();
            }
        lane_cv.wait(g);
    }
}
void flx_ctx::warm_one_cold_lane(flx::Lane* like) {
    // A lane allocates its workspaces (the trace arena alone is GBs) the first time a chunk runs on it. The thread that has
    // just finished a chunk pays that for one lane that has not run yet, so that lanes first used later in a run, when more
    // batches are in flight, start warm.
    flx::Lane* cold = nullptr;
    {
        std::lock_guard<std::mutex> g(lane_mu);
        for (size_t i = 0; i < free_lanes.size(); ++i)
            if (!lanes[(size_t)free_lanes[i]]->has_run) {
                cold = lanes[(size_t)free_lanes[i]].get();
                free_lanes.erase(free_lanes.begin() + (long)i);
                break;
            }
    }
    if (!cold) return;
    (void)cold->size_like(*like);
    cold->has_run = true;
    release_lane(cold);
}
void flx_ctx::release_lane(flx::Lane* lane) {
    { std::lock_guard<std::mutex> g(lane_mu); free_lanes.push_back(lane->id); }
    lane_cv.notify_all();
}
void flx_ctx::k1_acquire() {
    if (k1_tokens <= 0) return;
    std::unique_lock<std::mutex> g(lane_mu);
    lane_cv.wait(g, [&] { return k1_running < k1_tokens; });
    ++k1_running;
}
void flx_ctx::k1_release() {
    if (k1_tokens <= 0) return;
    { std::lock_guard<std::mutex> g(lane_mu); --k1_running; }
    lane_cv.notify_all();
}
int flx_ctx::sync_all() {
    for (auto& l : lanes) { int rc = l->sync(); if (rc) return rc; }
    return FLX_OK;
}

namespace flx {

static int h2d(Lane* ctx, DeviceBuffer& buf, const void* src, size_t bytes, size_t extra_zero_tail = 0) {
    int rc = buf.ensure(bytes + extra_zero_tail + 16);
    if (rc) return rc;
    if (bytes) FLX_HIP(hipMemcpyAsync(buf.ptr, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (extra_zero_tail) FLX_HIP(hipMemsetAsync((char*)buf.ptr + bytes, 0, extra_zero_tail, ctx->stream));
    return FLX_OK;
}
static int d2h(Lane* ctx, void* dst, const void* src, size_t bytes) {
    // A copy into pageable memory makes the calling thread wait, spinning, for everything queued before it. Waiting for the
    // stream on a blocking event first lets the thread sleep while the kernels run, so its core serves another lane.
    if (bytes) {
        int const rc = ctx->wait_idle();
        if (rc) return rc;
        FLX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    return FLX_OK;
}
// upload a byte sequence with TEXT_PAD zero bytes in front and behind; returns pointer to element 0
static int upload_padded(Lane* ctx, DeviceBuffer& buf, const u8* src, u64 len, const u8** d_first) {
    int rc = buf.ensure(len + 2 * TEXT_PAD + 16);
    if (rc) return rc;
    FLX_HIP(hipMemsetAsync(buf.ptr, 0, TEXT_PAD, ctx->stream));
    if (len) FLX_HIP(hipMemcpyAsync((char*)buf.ptr + TEXT_PAD, src, len, hipMemcpyHostToDevice, ctx->stream));
    FLX_HIP(hipMemsetAsync((char*)buf.ptr + TEXT_PAD + len, 0, TEXT_PAD + 16, ctx->stream));
    *d_first = (const u8*)buf.ptr + TEXT_PAD;
    return FLX_OK;
}

// ================================================================================================ seeding
namespace {

struct Group { u32 lb, len, errors; };

bool anchor_better(u64 pos_a, u64 err_a, u64 pos_b, u64 err_b) {                         // search.cpp:38-44
    u64 const d = pos_a < pos_b ? pos_b - pos_a : pos_a - pos_b;
    return err_a <= err_b && d <= err_b - err_a;
}

constexpr u64 ERASED = ~0ull;
struct RefAnchor { u64 pos; u64 errors; };

// search.cpp:352-389 for one (seed, reference) bucket
void erase_useless(hvec<RefAnchor>& v) {
    if (v.empty()) return;
    std::sort(v.begin(), v.end(), [](RefAnchor const& a, RefAnchor const& b) { return a.pos < b.pos; });
    for (size_t cur = 0; cur + 1 < v.size();) {
        size_t other = cur + 1;
        while (other < v.size() && anchor_better(v[cur].pos, v[cur].errors, v[other].pos, v[other].errors)) {
            v[other].errors = ERASED;
            ++other;
        }
        if (other < v.size() && anchor_better(v[other].pos, v[other].errors, v[cur].pos, v[cur].errors)) v[cur].errors = ERASED;
        cur = other;
    }
    v.erase(std::remove_if(v.begin(), v.end(), [](RefAnchor const& a) { return a.errors == ERASED; }), v.end());
}

}  // namespace

int search_seeds_device(Lane* ctx, const u8* d_seq_pool_or_null, const u8* h_seq_pool, u64 pool_len, const flx_seed* seeds,
                        u64 n_seeds, const flx_search_config& cfg, hvec<HostAnchor>& anchors, hvec<SeedStats>& stats,
                        hvec<DevHit>* raw_hits, u64 raw_max_hits, const u32* d_qpack_or_null, const u8* seed_flags, const SeedGen* gen) {
    anchors.clear();
    if (gen) n_seeds = gen->n_seeds;
    stats.assign(n_seeds, SeedStats{0, 0, 0, 0});
    if (n_seeds == 0) return FLX_OK;
    if (n_seeds >= (1ull << 31)) { set_error("too many seeds in one call"); return FLX_ERR_INVALID; }
    HostIndex const& H = *ctx->ctx->hidx;
    PhaseTimer sprof("search");

    // ---- expanded schemes (search_scheme_cache, search.cpp:328-350), DFS stack reservations and the launch order
    // Launch order = expected cost, heaviest class first (more errors, then shorter): the work of a seed grows steeply with its
    // errors (k = 2 leaves of a 5-kb read cost 4x the k = 1 leaves), and what a wave still holds when the seed queue runs dry
    // is the tail of the kernel. Within a class the caller's order is kept. Hits carry the seed's id, not its launch position.
    // Two passes over the caller's seeds (a chunk of 10-kb reads has a million of them): classes and their sizes, then every DevSeed
    // written once, at its launch position.
    struct SeedClass { u32 scheme_off, frames_searches, count, next; };
    std::map<u32, SeedClass> classes;                                   // key: (3 - errors) << 24 | length -> heaviest first
    hvec<u64> scheme_table;
    u64 frames = 0;
    u32 max_errors = 0, max_length = 0;
    auto class_key = [](flx_seed const& s) { return ((3u - s.num_errors) << 24) | s.length; };
    if (gen) { scheme_table = gen->scheme_table; max_errors = gen->max_errors; max_length = gen->max_length; }
    else {
        u32 last = 0xFFFFFFFFu;
        SeedClass* slot = nullptr;                                      // consecutive seeds are mostly of one class
        for (u64 i = 0; i < n_seeds; ++i) {
            flx_seed const& s = seeds[i];
            if (s.num_errors > 3) { set_error("seed errors must be in [0,3] (floxer_cli.cpp:299)"); return FLX_ERR_INVALID; }
            if (s.length == 0 || s.length > SCH_POS_MASK || s.seq_offset + s.length > pool_len) { set_error("seed outside the sequence pool"); return FLX_ERR_INVALID; }
            u32 const key = class_key(s);
            if (key != last) {
                auto it = classes.find(key);
                if (it == classes.end()) {
                    auto const e = expanded_scheme(s.num_errors, s.length);
                    u32 const nsearch = e.empty() ? 0 : (u32)(e.size() / s.length);
                    it = classes.emplace(key, SeedClass{(u32)scheme_table.size(), (s.length + s.num_errors + 3) | (nsearch << 24), 0, 0}).first;
                    scheme_table.insert(scheme_table.end(), e.begin(), e.end());
                    max_errors = std::max(max_errors, s.num_errors);
                    max_length = std::max(max_length, s.length);
                }
                slot = &it->second;
                last = key;
            }
            ++slot->count;
        }
        u32 pos = 0;
        for (auto& kv : classes) { kv.second.next = pos; pos += kv.second.count; }
    }
    hvec<DevSeed> dseeds(gen ? 0 : n_seeds);
    // what a seed's symbols may be (SEED_* of flx_fm_core.hpp): given by the caller per seed, or read off the host pool
    auto flags_of = [&](u64 i) -> u32 {
        if (seed_flags) return seed_flags[i];
        if (!h_seq_pool) return SEED_HAS_DELIM | SEED_NOT_ACGT;
        u32 f = 0;
        const u8* p = h_seq_pool + seeds[i].seq_offset;
        for (u32 j = 0; j < seeds[i].length; ++j) { if (p[j] == 0) f |= SEED_HAS_DELIM; if (p[j] - 1u > 3u) f |= SEED_NOT_ACGT; }
        return f;
    };
    if (!gen) {
        u32 last = 0xFFFFFFFFu;
        SeedClass* slot = nullptr;
        for (u64 i = 0; i < n_seeds; ++i) {
            flx_seed const& s = seeds[i];
            u32 const key = class_key(s);
            if (key != last) { slot = &classes.find(key)->second; last = key; }
            DevSeed& d = dseeds[slot->next++];
            d.seq_off = s.seq_offset;
            d.length = s.length;
            d.scheme_off = slot->scheme_off;
            d.frames_searches = slot->frames_searches;
            d.stack_off = frames;                                       // (reserved in the caller's order: only the ordered walk uses it)
            d.id = (u32)i;
            d.flags = flags_of(i);
            d.pad = 0;
            frames += slot->frames_searches & 0xFFFFFFu;
        }
    }
    if (scheme_table.empty()) scheme_table.push_back(0);
    // most seeds longer than 64 symbols (20-kb reads at 2 %: leaves of 98 .. 147)? The text walk then takes its larger LDS windows.
    bool long_seeds = false;
    {
        u64 n_long = 0, n_all = 0;
        if (gen) { for (auto const& lf : gen->leaves) { n_long += lf.length > 64u; ++n_all; } }
        else for (u64 i = 0; i < n_seeds; ++i) { n_long += seeds[i].length > 64u; ++n_all; }
        long_seeds = 2 * n_long > n_all;
    }

    sprof.mark("prep");
    int rc;
    const u8* d_seq = d_seq_pool_or_null;
    if (!d_seq) {
        if ((rc = h2d(ctx, ctx->seq, h_seq_pool, pool_len, 64))) return rc;
        d_seq = ctx->seq.as<u8>();
    }
    if ((rc = h2d(ctx, ctx->scheme, scheme_table.data(), scheme_table.size() * 8))) return rc;
    if (!gen) { if ((rc = h2d(ctx, ctx->seeds, dseeds.data(), dseeds.size() * sizeof(DevSeed)))) return rc; }
    else {
        // the chunk's description (a few hundred KB) up, the DevSeeds written where the search reads them
        size_t const b_reads = gen->reads.size() * sizeof(DevSeedRead), b_leaves = gen->leaves.size() * sizeof(DevSeedLeaf), b_classes = gen->classes.size() * sizeof(DevSeedClass);
        size_t const o_leaves = (b_reads + 255) / 256 * 256, o_classes = o_leaves + (b_leaves + 255) / 256 * 256;
        if ((rc = ctx->seed_gen.ensure(o_classes + b_classes + 256))) return rc;
        if ((rc = ctx->seeds.ensure(n_seeds * sizeof(DevSeed)))) return rc;
        char* const g = (char*)ctx->seed_gen.ptr;
        FLX_HIP(hipMemcpyAsync(g, gen->reads.data(), b_reads, hipMemcpyHostToDevice, ctx->stream));
        FLX_HIP(hipMemcpyAsync(g + o_leaves, gen->leaves.data(), b_leaves, hipMemcpyHostToDevice, ctx->stream));
        FLX_HIP(hipMemcpyAsync(g + o_classes, gen->classes.data(), b_classes, hipMemcpyHostToDevice, ctx->stream));
        int const e = DeviceApi::build_seeds(ctx->stream, (const DevSeedRead*)g, (u32)gen->reads.size(), (const DevSeedLeaf*)(g + o_leaves), (const DevSeedClass*)(g + o_classes),
                                             ctx->seeds.as<DevSeed>());
        if (e) { set_error(std::string("seed_build: ") + hipGetErrorString((hipError_t)e)); return FLX_ERR_NO_DEVICE; }
    }
    // The DFS in the reference's order (frames on a per-seed stack in HBM) where the order of discovery matters: the raw-emission
    // hook and first_reported, which want the first n rows; everywhere else the walk with its stack in LDS, whose hits carry keys
    // that restore the emission order.
    bool const ordered = (raw_hits && !getenv("FLX_FM_KEYED_RAW")) || cfg.anchor_choice_strategy == FLX_CHOICE_FIRST_REPORTED || max_length > fm_search_max_keyed_length() ||
                         max_errors > 3 || getenv("FLX_FM_ORDERED");      // (FLX_FM_ORDERED=1: the ordered walk for everything, for comparisons)
    if (ordered && (rc = ctx->stack.ensure(frames * sizeof(DevFrame)))) return rc;
    if ((rc = ctx->counters.ensure(128))) return rc;
    // the walk of flx_search.hip (presence filter, one-row subtrees against the text) unless the order of discovery matters
    bool const filtered = !ordered;
    const u32* d_qpack = d_qpack_or_null;
    if (filtered && !d_qpack && ctx->ctx->didx.filter) {
        if ((rc = ctx->qpack.ensure(pack_words_for(pool_len) * 4 + 64))) return rc;
        int const e = DeviceApi::pack_pool(ctx->stream, d_seq, pool_len, ctx->qpack.as<u32>());
        if (e) { set_error(std::string("pack_pool: ") + hipGetErrorString((hipError_t)e)); return FLX_ERR_NO_DEVICE; }
        d_qpack = ctx->qpack.as<u32>();
    }
    // (10-kb reads at 8 % on a random text: 6.3 one-row subtrees and 0.4 hits per seed; on a repeat-rich text several times that: what the
    // last search on this lane needed, and a quarter more, is the starting size; a search that outgrows its buffers runs again)
    u64 item_cap = filtered && ctx->ctx->didx.isa ? std::max<u64>(n_seeds * 8, (u64)(ctx->items_per_seed * 1.25 * (double)n_seeds)) + 4096 * 64 : 0;

    u32 const max_hits = raw_hits ? (u32)std::min<u64>(raw_max_hits, 0xFFFFFFF0u)
                                  : (cfg.anchor_choice_strategy == FLX_CHOICE_FIRST_REPORTED
                                         ? (u32)cfg.max_num_anchors_soft
                                         : (u32)std::max(cfg.max_num_anchors_hard, cfg.max_num_anchors_hard + 1));
    u64 const hit_slack = 4096 * 64;            // unused ends of the per-wave slot ranges (FM_MAX_WAVES x FM_HIT_GRAB)
    u64 hit_cap = std::max<u64>(n_seeds * 6, (u64)(ctx->hits_per_seed * 1.25 * (double)n_seeds)) + hit_slack;
    // Anchor selection on the device (K1b) for the default group order and anchor choice; seeds it does not handle come back
    // flagged and go through the host code below.
    bool const device_select = !raw_hits && cfg.anchor_group_order == FLX_ORDER_COUNT_FIRST && cfg.anchor_choice_strategy == FLX_CHOICE_ROUND_ROBIN &&
                               cfg.max_num_anchors_soft >= 1 && !getenv("FLX_HOST_SELECT");
    if (gen && (!device_select || ordered)) return SEARCH_NEEDS_HOST_SEEDS;      // (those paths read the seed list)
    size_t const scan_bytes = device_select ? DeviceApi::select_scan_bytes((u32)n_seeds) : 0;
    hvec<DevSelStat> sel_stat;                // per seed
    u32 sel_total = 0, sel_rows_total = 0;
    if (device_select) {
        if ((rc = ctx->seed_cnt.ensure((n_seeds + 1) * 4))) return rc;
        if ((rc = ctx->hit_off.ensure((n_seeds + 1) * 4))) return rc;
        if ((rc = ctx->sel_stat.ensure(n_seeds * sizeof(DevSelStat) + 16))) return rc;
        if ((rc = ctx->sel_n.ensure((n_seeds + 1) * 4))) return rc;
        if ((rc = ctx->sel_off.ensure((n_seeds + 1) * 4))) return rc;
        if ((rc = ctx->sel_rows.ensure((n_seeds + 1) * 4))) return rc;
        if ((rc = ctx->sel_row_off.ensure((n_seeds + 1) * 4))) return rc;
        if ((rc = ctx->sel_tmp.ensure(scan_bytes + 64))) return rc;
        if ((rc = ctx->sel_lists.ensure((3 * n_seeds + 3) * 4))) return rc;
        sel_stat.resize(n_seeds);
    }
    // the mailboxes through which the waves of a search launch hand subtrees to each other (at most 4096 waves per launch, 6 KB each)
    u32 const mailbox_waves = device_select ? 4096u : 0u;
    if (mailbox_waves && (rc = ctx->mailboxes.ensure(DeviceApi::mailbox_bytes(mailbox_waves)))) return rc;
    u32 counters[32];
    u64 sel_cap = (u64)(ctx->sel_rows_per_seed * 1.25 * (double)n_seeds);      // entries of the selected-anchor list (at least hit_cap, below)
    struct K1Token { flx_ctx* c; explicit K1Token(flx_ctx* c_) : c(c_) { c->k1_acquire(); } ~K1Token() { c->k1_release(); } };
    for (int attempt = 0;; ++attempt) {
        K1Token const token(ctx->ctx);           // (held until this attempt's kernels have finished)
        static int const alloc_debug = getenv("FLX_ALLOC_DEBUG") ? 1 : 0;
        sel_cap = std::max(sel_cap, hit_cap);
        if ((rc = ctx->hits.ensure(hit_cap * sizeof(DevHit)))) return rc;
        if (item_cap && (rc = ctx->items.ensure(item_cap * sizeof(DevHit)))) return rc;
        FLX_HIP(hipMemsetAsync(ctx->counters.ptr, 0, 128, ctx->stream));
        if (device_select) {
            // one selected anchor per hit row at most; rows <= hits * SEL_MAX would be the hard bound, the seeds the device
            // handles have at most soft-cap rows each and nearly all hits have one row: hit_cap entries, checked after the run
            if ((rc = ctx->grouped.ensure(hit_cap * sizeof(DevHit)))) return rc;
            if ((rc = ctx->sel_out.ensure(sel_cap * sizeof(DevOutAnchor)))) return rc;
            if ((rc = ctx->sel_sparse.ensure(sel_cap * sizeof(DevOutAnchor)))) return rc;
            // (the search counts a seed's rows here while it runs; seed_rows_kernel then writes every entry but the last)
            FLX_HIP(hipMemsetAsync(ctx->sel_rows.ptr, 0, (n_seeds + 1) * 4, ctx->stream));
            FLX_HIP(hipMemsetAsync(ctx->seed_cnt.ptr, 0, (n_seeds + 1) * 4, ctx->stream));
            FLX_HIP(hipMemsetAsync((char*)ctx->sel_n.ptr + n_seeds * 4, 0, 4, ctx->stream));
        }
        if (alloc_debug) { dump_lane_buffers(*ctx, "search"); fprintf(stderr, "[flx alloc] lane %d search: seeds %llu hit_cap %llu item_cap %llu pool %p qpack %p\n", ctx->id, (unsigned long long)n_seeds, (unsigned long long)hit_cap, (unsigned long long)item_cap, (const void*)d_seq, (const void*)d_qpack); }
        rc = timed_launch(ctx, "fm_search", 0, n_seeds, [&] {
            u32 const concurrent = ctx->ctx->external_stream ? 1u : (u32)ctx->ctx->lanes.size();
            if (filtered)
                return DeviceApi::search_filtered(ctx->stream, ctx->ctx->didx, d_seq, d_qpack, ctx->scheme.as<u64>(), ctx->seeds.as<DevSeed>(), (u32)n_seeds,
                                                  max_hits, max_errors, ctx->hits.as<DevHit>(), (u32)std::min<u64>(hit_cap, 0xFFFFFFFFu),
                                                  item_cap ? ctx->items.as<DevHit>() : nullptr, (u32)std::min<u64>(item_cap, 0xFFFFFFFFu),
                                                  ctx->counters.as<u32>(), device_select ? ctx->seed_cnt.as<u32>() : nullptr,
                                                  device_select ? ctx->sel_rows.as<u32>() : nullptr, device_select ? ctx->mailboxes.ptr : nullptr, mailbox_waves, concurrent, long_seeds);
            return DeviceApi::search(ctx->stream, ctx->ctx->didx, d_seq, ctx->scheme.as<u64>(), ctx->seeds.as<DevSeed>(), (u32)n_seeds,
                                     max_hits, ctx->stack.as<DevFrame>(), ctx->hits.as<DevHit>(), (u32)std::min<u64>(hit_cap, 0xFFFFFFFFu),
                                     ctx->counters.as<u32>(), device_select ? ctx->seed_cnt.as<u32>() : nullptr);
        });
        if (rc) return rc;
        if (device_select) {
            rc = timed_launch(ctx, "fm_select", n_seeds * 5, n_seeds, [&] {
                return DeviceApi::select(ctx->stream, ctx->hits.as<DevHit>(), ctx->counters.as<u32>(), (u32)std::min<u64>(hit_cap, 0xFFFFFFFFu),
                                         ctx->seed_cnt.as<u32>(), ctx->hit_off.as<u32>(), ctx->grouped.as<DevHit>(), (u32)n_seeds, ctx->ctx->didx,
                                         ctx->ctx->seq_start.as<u64>(), (u32)H.seq_start.size(), (u32)std::min<u64>(cfg.max_num_anchors_hard, 0xFFFFFFFFu),
                                         (u32)std::min<u64>(cfg.max_num_anchors_soft, 0xFFFFFFFFu), cfg.erase_useless_anchors != 0, ctx->sel_stat.ptr,
                                         ctx->sel_n.as<u32>(), ctx->sel_off.as<u32>(), ctx->sel_out.as<DevOutAnchor>(), (u32)std::min<u64>(sel_cap, 0xFFFFFFFFu),
                                         ctx->sel_rows.as<u32>(), ctx->sel_row_off.as<u32>(), ctx->sel_sparse.as<DevOutAnchor>(),
                                         (u32)std::min<u64>(sel_cap, 0xFFFFFFFFu), ctx->sel_tmp.ptr, scan_bytes, ctx->sel_lists.as<u32>());
            });
            if (rc) return rc;
        }
        if ((rc = d2h(ctx, counters, ctx->counters.ptr, 128))) return rc;
        if (device_select) {
            if ((rc = d2h(ctx, &sel_total, (char*)ctx->sel_off.ptr + n_seeds * 4, 4))) return rc;
            if ((rc = d2h(ctx, &sel_rows_total, (char*)ctx->sel_row_off.ptr + n_seeds * 4, 4))) return rc;
            if ((rc = d2h(ctx, sel_stat.data(), ctx->sel_stat.ptr, n_seeds * sizeof(DevSelStat)))) return rc;
        }
        if ((rc = ctx->sync())) return rc;
        if (getenv("FLX_SEARCH_DEBUG")) fprintf(stderr, "[fm_search] seeds %llu ext %u (of single-row intervals %u) wave-iterations %u (max per wave %u) busy pair-iterations %u, after the queue ran dry %u (max %u), subtrees handed over %u, from wave to wave %u, walks abandoned over the cap %u\n", (unsigned long long)n_seeds, counters[2], counters[3], counters[4], counters[5], counters[6], counters[8], counters[9], counters[14], counters[15], counters[20]);
        if (getenv("FLX_SEARCH_DEBUG") && filtered) fprintf(stderr, "[fm_search filtered] subtrees queued %u (slots %u of %llu), filter words asked %u, children dropped %u, searches ended by the prefix lookup %u; text walk: lane-steps %u, wave-iterations %u in %u waves (longest %u)\n", counters[3], counters[16], (unsigned long long)item_cap, counters[10], counters[11], counters[12], counters[18], counters[19], counters[22], counters[21]);
        if (counters[1]) { set_error(counters[1] & 2u ? "fm_search: a subtree handed between waves was not taken" : "fm_search: DFS stack reservation exceeded"); return FLX_ERR_INTERNAL; }
        bool const items_fit = !item_cap || counters[16] <= item_cap;
        if (items_fit && counters[0] <= hit_cap && (!device_select || sel_rows_total <= sel_cap)) break;      // (selected anchors <= rows)
        if (attempt >= 3) { set_error("fm_search: hit buffer could not be sized"); return FLX_ERR_INTERNAL; }
        { std::lock_guard<std::mutex> g(ctx->ctx->mu); ++ctx->ctx->path.search_reruns; }
        // (a wave reserves 64 slots at a time and leaves the rest of a range unused when a ballot's records do not fit into it: the slots
        // reserved are at most twice the records written plus one range per wave of both kernels, however the waves were scheduled)
        u64 const wave_ranges = (u64)(4096 + 8192 + 64) * 64;
        if (!items_fit) item_cap = std::max<u64>((u64)counters[16], 2 * (u64)counters[3]) + wave_ranges;      // queued subtrees were dropped: run again with room for all
        else if (counters[0] > hit_cap) hit_cap = std::max<u64>((u64)counters[0], 2 * (u64)counters[13]) + wave_ranges;      // the number of hits is known now; run again
        else sel_cap = (u64)sel_rows_total + 1024;
    }
    // The kernel's accounting: the bytes THIS walk has to touch, from its own device counters (work units = rank pairs). Random accesses
    // count at the 64-B size the memory system fetches them in; records that stream count at their size:
    //   rank pair                2 x 64 B   (a 32-B block at either end of the interval; both ends in one block still count twice)
    //   filter lookup            64 B       (one 64-bit word of the presence table)
    //   queued one-row subtree   24 B written + 24 B read (the record) + 64 B (SA[row]) + 64 B (the text next to it) + 64 B (its seed's record)
    //   hit                      24 B written; a hit of the text walk reads ISA[position] (64 B): charged for every hit
    //   seed                     40 B (its record) + 64 B (its symbols) + 64 B (their 2-bit form, filter walk only)
    // The ordered walk (no filter, no text walk) prices its rank pairs and its 64-B frames written and read back.
    // SURVEY.md 8(d)'s figure - 128 B per cursor extension of the REFERENCE's walk - is computed by bench.py from the oracle's count and
    // reported beside this one; it is not a fraction of the HBM peak for a walk that answers with fewer rank queries.
    if (ctx->ctx->timing) {
        u64 bytes = (u64)counters[2] * 128;
        if (filtered) bytes += (u64)counters[10] * 64 + (u64)counters[3] * (24 + 24 + 64 + 64 + 64) + (u64)counters[13] * (24 + 64) + n_seeds * (u64)(40 + 64 + (d_qpack ? 64 : 0));
        else bytes += (u64)counters[2] * 128 + n_seeds * (u64)(40 + 64);
        std::lock_guard<std::mutex> g(ctx->ctx->mu);
        auto it = ctx->ctx->stats.find("fm_search");
        if (it != ctx->ctx->stats.end()) { it->second.algorithmic_bytes += bytes; it->second.work_units += counters[2]; }
    }
    ctx->hits_per_seed = (double)counters[0] / (double)n_seeds;
    ctx->items_per_seed = (double)counters[16] / (double)n_seeds;
    ctx->sel_rows_per_seed = (double)sel_rows_total / (double)n_seeds;
    sprof.mark("kernel");
    // path counters of this call (folded into the context's at every way out of the selection below)
    u64 const n_extensions = counters[2];
    auto count_path = [&](u64 on_host) {
        u64 with = 0, excl = 0;
        for (auto const& st : stats) { with += st.useful != 0; excl += st.fully_excluded != 0; }
        std::lock_guard<std::mutex> g(ctx->ctx->mu);
        flx_path_counters& pc = ctx->ctx->path;
        pc.seeds += n_seeds; pc.seeds_with_anchors += with; pc.seeds_excluded_by_hard_cap += excl; pc.seeds_selected_on_host += on_host;
        pc.anchors += anchors.size(); pc.cursor_extensions += n_extensions;
    };
    // ---- what the device selected; host_seed[si] != 0: this seed still goes through the host code
    hvec<HostAnchor> dev_anchors;
    hvec<u8> host_seed;
    if (device_select) {
        static_assert(sizeof(HostAnchor) == sizeof(DevOutAnchor), "the compact list is read as HostAnchor");
        dev_anchors.resize(sel_total);
        if (sel_total) {
            if ((rc = d2h(ctx, dev_anchors.data(), ctx->sel_out.ptr, (size_t)sel_total * sizeof(HostAnchor)))) return rc;
            if ((rc = ctx->sync())) return rc;
        }
        if (seeds) for (auto& a : dev_anchors) a.leaf = seeds[a.seed_index].pex_leaf_index;
        host_seed.assign(n_seeds, 0);
        bool any = false;
        for (u64 si = 0; si < n_seeds; ++si) {
            DevSelStat const st = sel_stat[si];
            if (st.flag) { host_seed[si] = 1; any = true; }
            else stats[si] = SeedStats{st.useful, st.raw, st.excluded_soft, st.excluded};
        }
        sprof.mark("device-select");
        if (getenv("FLX_SEARCH_DEBUG")) {
            u64 flagged = 0, with_anchors = 0, excl = 0;
            for (u64 si = 0; si < n_seeds; ++si) { DevSelStat const st = sel_stat[si]; flagged += st.flag; with_anchors += st.useful != 0; excl += st.excluded; }
            fprintf(stderr, "[fm_select] seeds %llu: with anchors %llu, excluded %llu, left to the host %llu; anchors %u\n", (unsigned long long)n_seeds,
                    (unsigned long long)with_anchors, (unsigned long long)excl, (unsigned long long)flagged, sel_total);
        }
        if (!any) { anchors.swap(dev_anchors); count_path(0); return FLX_OK; }
    }
    // ---- the hits per seed in emission order: `by_seed`, seed si owns [first[si], first[si+1]). With device-side selection the
    //      device has grouped them already (only the seeds left to the host are looked at below); else the host groups them.
    hvec<u32> first(n_seeds + 1, 0);
    hvec<DevHit> by_seed;
    if (device_select) {
        if ((rc = d2h(ctx, first.data(), ctx->hit_off.ptr, (n_seeds + 1) * 4))) return rc;
        if ((rc = ctx->sync())) return rc;
        by_seed.resize(first[n_seeds]);
        if ((rc = d2h(ctx, by_seed.data(), ctx->grouped.ptr, (size_t)first[n_seeds] * sizeof(DevHit)))) return rc;
        if ((rc = ctx->sync())) return rc;
        // the segments the host is going to look at, into emission order (the device sorts its own seeds' hits where it reads them)
        if (!ordered)
            for (u64 si = 0; si < n_seeds; ++si)
                if (host_seed[si] && first[si + 1] - first[si] > 1)
                    std::stable_sort(by_seed.begin() + first[si], by_seed.begin() + first[si + 1], [](DevHit const& a, DevHit const& b) { return a.key < b.key; });
        sprof.mark("d2h-hits");
    } else {
        u32 const n_slots = counters[0];      // reserved slots; unused ones carry seed 0xFFFFFFFF
        hvec<DevHit> hits(n_slots);
        if ((rc = d2h(ctx, hits.data(), ctx->hits.ptr, (size_t)n_slots * sizeof(DevHit)))) return rc;
        if ((rc = ctx->sync())) return rc;
        sprof.mark("d2h-hits");
        // a seed stays on one wave, whose slot ranges and slots within a range are handed out in increasing order
        for (auto const& h : hits) if (h.seed != 0xFFFFFFFFu) first[h.seed + 1]++;
        for (u64 i = 0; i < n_seeds; ++i) first[i + 1] += first[i];
        by_seed.resize(first[n_seeds]);
        hvec<u32> cursor(first.begin(), first.end() - 1);
        for (auto const& h : hits) if (h.seed != 0xFFFFFFFFu) by_seed[cursor[h.seed]++] = h;
        // into the reference's emission order (the keys of the walk with its stack in LDS; the ordered walk's hits are in it already)
        if (!ordered)
            for (u64 si = 0; si < n_seeds; ++si)
                if (first[si + 1] - first[si] > 1)
                    std::stable_sort(by_seed.begin() + first[si], by_seed.begin() + first[si + 1], [](DevHit const& a, DevHit const& b) { return a.key < b.key; });
    }
    if (raw_hits) { *raw_hits = std::move(by_seed); return FLX_OK; }
    hvec<u32> todo;                           // the seeds the host selects for, ascending
    if (host_seed.empty()) { todo.resize(n_seeds); std::iota(todo.begin(), todo.end(), 0u); }
    else for (u64 si = 0; si < n_seeds; ++si) if (host_seed[si]) todo.push_back((u32)si);

    sprof.mark("group");
    // ---- hard cap, group order, anchor choice (search.cpp:190-302)
    struct RowReq { u32 seed, errors, row; };
    hvec<RowReq> reqs;
    hvec<u64> total_raw(n_seeds, 0);
    hvec<u8> excluded(n_seeds, 0);
    hvec<Group> groups;
    hvec<u32> alive;
    for (u32 const si : todo) {
        if (first[si] == first[si + 1]) continue;               // no hit at all: nothing to select
        if (first[si] + 1 == first[si + 1] && by_seed[first[si]].len == 1 && cfg.max_num_anchors_hard >= 1 && cfg.max_num_anchors_soft >= 1) {
            // one group of one row (most seeds of a read that has a single locus): every order and strategy keeps exactly it
            total_raw[si] = 1;
            reqs.push_back(RowReq{(u32)si, by_seed[first[si]].errors, by_seed[first[si]].lb});
            continue;
        }
        groups.clear();
        u64 total = 0;
        for (u32 h = first[si]; h < first[si + 1]; ++h) { groups.push_back(Group{by_seed[h].lb, by_seed[h].len, by_seed[h].errors}); total += by_seed[h].len; }
        total_raw[si] = total;
        if (total > cfg.max_num_anchors_hard && cfg.anchor_choice_strategy != FLX_CHOICE_FIRST_REPORTED) { excluded[si] = 1; continue; }
        switch (cfg.anchor_group_order) {
            case FLX_ORDER_COUNT_FIRST:
                std::sort(groups.begin(), groups.end(), [](Group const& a, Group const& b) {
                    if (a.len != b.len) return a.len < b.len;
                    return a.errors < b.errors;
                });
                break;
            case FLX_ORDER_ERRORS_FIRST:     // literally as written in search.cpp:215-222
                std::sort(groups.begin(), groups.end(), [](Group const& a, Group const& b) {
                    if (a.errors != b.errors) return a.len < b.len;
                    return a.errors < b.errors;
                });
                break;
            default: break;
        }
        u64 kept = 0;
        if (cfg.anchor_choice_strategy == FLX_CHOICE_ROUND_ROBIN) {
            // search.cpp:239-272: cycle through the groups that still have rows, taking row lb + round from each; a group leaves
            // the cycle after its last row. (The reference keeps the remaining indices in a std::set; a compacting vector visits
            // them in the same ascending order.)
            alive.resize(groups.size());
            for (size_t g = 0; g < groups.size(); ++g) alive[g] = (u32)g;
            u64 round = 0;
            while (kept != cfg.max_num_anchors_soft && !alive.empty()) {
                size_t w = 0;
                for (size_t a = 0; a < alive.size(); ++a) {
                    if (kept == cfg.max_num_anchors_soft) { alive[w++] = alive[a]; continue; }
                    Group const& g = groups[alive[a]];
                    reqs.push_back(RowReq{(u32)si, g.errors, (u32)(g.lb + round)});
                    ++kept;
                    if (g.len != round + 1) alive[w++] = alive[a];
                }
                alive.resize(w);
                ++round;
            }
        } else {
            size_t gi = 0;
            while (kept != cfg.max_num_anchors_soft && gi < groups.size()) {
                Group const& g = groups[gi];
                for (u32 r = 0; r < g.len; ++r) {
                    reqs.push_back(RowReq{(u32)si, g.errors, g.lb + r});
                    if (++kept == cfg.max_num_anchors_soft) break;
                }
                ++gi;
            }
        }
    }

    sprof.mark("select");
    // ---- locate (search.cpp:253, 284) as one SA gather
    hvec<u32> rows(reqs.size()), textpos(reqs.size());
    for (size_t i = 0; i < reqs.size(); ++i) rows[i] = reqs[i].row;
    if (!reqs.empty()) {
        if ((rc = h2d(ctx, ctx->rows, rows.data(), rows.size() * 4))) return rc;
        if ((rc = ctx->rows_out.ensure(rows.size() * 4))) return rc;
        rc = timed_launch(ctx, "fm_locate", rows.size() * 8, rows.size(), [&] {
            return DeviceApi::locate(ctx->stream, ctx->ctx->didx, ctx->rows.as<u32>(), (u32)rows.size(), ctx->rows_out.as<u32>());
        });
        if (rc) return rc;
        if ((rc = d2h(ctx, textpos.data(), ctx->rows_out.ptr, rows.size() * 4))) return rc;
        if ((rc = ctx->sync())) return rc;
    }

    sprof.mark("locate");
    // ---- per seed: bucket per reference, erase useless anchors, flatten (search.cpp:78-100, 304-318)
    size_t const nref = H.seq_len.size();
    hvec<hvec<RefAnchor>> by_ref(nref);
    hvec<u32> touched;                    // references that received an anchor of the current seed
    size_t ri = 0;
    for (u32 const si : todo) {
        if (excluded[si]) { stats[si] = SeedStats{0, 0, 0, 1}; continue; }
        if (ri >= reqs.size() || reqs[ri].seed != si) continue;      // nothing kept: stats stay zero
        if (ri + 1 == reqs.size() || reqs[ri + 1].seed != si) {
            // a single anchor: its bucket holds nothing that could make it useless
            u64 const p = textpos[ri];
            if (p >= H.n) { set_error("fm_locate returned a position outside the text"); return FLX_ERR_INTERNAL; }
            size_t const s = nref == 1 ? 0 : std::upper_bound(H.seq_start.begin(), H.seq_start.end(), p) - H.seq_start.begin() - 1;
            stats[si] = SeedStats{1, 1, (u32)(total_raw[si] - 1), 0};
            anchors.push_back(HostAnchor{(u32)si, (seeds ? seeds[si].pex_leaf_index : 0u), (u32)s, reqs[ri].errors, p - H.seq_start[s]});
            ++ri;
            continue;
        }
        touched.clear();
        u32 raw = 0;
        while (ri < reqs.size() && reqs[ri].seed == si) {
            u64 const p = textpos[ri];
            if (p >= H.n) { set_error("fm_locate returned a position outside the text"); return FLX_ERR_INTERNAL; }
            size_t const s = nref == 1 ? 0 : std::upper_bound(H.seq_start.begin(), H.seq_start.end(), p) - H.seq_start.begin() - 1;
            if (by_ref[s].empty()) touched.push_back((u32)s);
            by_ref[s].push_back(RefAnchor{p - H.seq_start[s], reqs[ri].errors});
            ++raw;
            ++ri;
        }
        std::sort(touched.begin(), touched.end());                   // anchors are reported by reference id (search.cpp:78-100)
        u32 useful = raw;
        if (cfg.erase_useless_anchors) {
            useful = 0;
            for (u32 r : touched) { erase_useless(by_ref[r]); useful += (u32)by_ref[r].size(); }
        }
        stats[si] = SeedStats{useful, raw, (u32)(total_raw[si] - raw), 0};
        for (u32 r : touched) {
            for (auto const& a : by_ref[r]) anchors.push_back(HostAnchor{(u32)si, (seeds ? seeds[si].pex_leaf_index : 0u), r, (u32)a.errors, a.pos});
            by_ref[r].clear();
        }
    }
    sprof.mark("erase+flatten");
    if (!dev_anchors.empty()) {               // both lists are in seed order
        hvec<HostAnchor> merged(anchors.size() + dev_anchors.size());
        std::merge(anchors.begin(), anchors.end(), dev_anchors.begin(), dev_anchors.end(), merged.begin(),
                   [](HostAnchor const& a, HostAnchor const& b) { return a.seed_index < b.seed_index; });
        anchors.swap(merged);
    }
    count_path(todo.size());
    return FLX_OK;
}

// ================================================================================================ alignment batches
namespace {

struct AlignRequest { u64 ref_off, q_off; u32 n, m, k; };

// word-steps the launch really performs for one job: the whole matrix, or only the band -k <= col-row <= n-m+k
u64 job_word_steps(u32 n, u32 m, u32 k, AlignShape sh) {
    u64 const nw = (m + 63) / 64;
    if (!sh.banded) return (u64)n * nw;
    i64 const W = sh.words_per_lane, band_hi = (i64)n - (i64)m + (i64)k;
    u64 total = 0;
    for (i64 g = 0; g * W < (i64)nw; ++g) {
        i64 const r0 = 64 * W * g, r1 = std::min<i64>(m, r0 + 64 * W);
        i64 const lo = std::max<i64>(0, r0 - (i64)k), hi = std::min<i64>((i64)n - 1, r1 - 1 + band_hi);
        if (hi >= lo) total += (u64)(hi - lo + 1) * (u64)std::min<i64>(W, (i64)nw - g * W);
    }
    return total;
}

struct ShapeKey {
    u32 w, g, banded;
    bool operator<(ShapeKey const& o) const { return w != o.w ? w < o.w : g != o.g ? g < o.g : banded < o.banded; }
};

// Anchors of one locus produce many identical (window, node) jobs (sibling leaves share their parent's window, anchors with the
// same indel drift share the root window). Identical inputs give identical outputs, so each distinct job runs once.
struct ReqKey {
    u64 ref_off, q_off; u32 n, m, k;
    bool operator==(ReqKey const& o) const { return ref_off == o.ref_off && q_off == o.q_off && n == o.n && m == o.m && k == o.k; }
};
struct ReqKeyHash {
    size_t operator()(ReqKey const& r) const {
        u64 h = r.ref_off * 0x9E3779B97F4A7C15ull ^ (r.q_off + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
        h ^= ((u64)r.n << 40) ^ ((u64)r.m << 20) ^ r.k;
        h ^= h >> 29;
        return (size_t)(h * 0xBF58476D1CE4E5B9ull);
    }
};
void dedup_requests(hvec<AlignRequest> const& reqs, hvec<AlignRequest>& uniq, hvec<u32>& uniq_of) {
    // open-addressing table of indices into `uniq` (power-of-two size, linear probing)
    size_t cap = 16;
    while (cap < reqs.size() * 2 + 1) cap <<= 1;
    hvec<u32> table(cap, 0xFFFFFFFFu);
    ReqKeyHash const hasher;
    uniq.clear();
    uniq.reserve(reqs.size());
    uniq_of.resize(reqs.size());
    for (size_t i = 0; i < reqs.size(); ++i) {
        AlignRequest const& r = reqs[i];
        ReqKey const key{r.ref_off, r.q_off, r.n, r.m, r.k};
        size_t h = hasher(key) & (cap - 1);
        while (true) {
            u32 const e = table[h];
            if (e == 0xFFFFFFFFu) { table[h] = (u32)uniq.size(); uniq_of[i] = (u32)uniq.size(); uniq.push_back(r); break; }
            AlignRequest const& u = uniq[e];
            if (u.ref_off == r.ref_off && u.q_off == r.q_off && u.n == r.n && u.m == r.m && u.k == r.k) { uniq_of[i] = e; break; }
            h = (h + 1) & (cap - 1);
        }
    }
}

// Shapes for the jobs of one call. Many jobs: each gets the shape that costs the fewest wave slots. Few jobs (they would leave
// most SIMDs without a wave): all get one common shape with the fewest words per lane, i.e. more, shorter-running waves and a
// single launch.
// a round tests the nodes of [smallest, smallest * span / 100] rows (FLX_ROUND_SPAN overrides the percentage)
u64 round_span_percent() {
    static u64 const v = getenv("FLX_ROUND_SPAN") ? std::max<u64>(100, strtoull(getenv("FLX_ROUND_SPAN"), nullptr, 10)) : 150;
    return v;
}
u64 align_few_waves() {          // FLX_ALIGN_FEW_WAVES overrides the threshold (tests force either form)
    const char* env = getenv("FLX_ALIGN_FEW_WAVES");
    return env ? strtoull(env, nullptr, 10) : 512;
}
int choose_shapes(hvec<AlignRequest> const& reqs, hvec<AlignShape>& shapes) {
    shapes.resize(reqs.size());
    u64 lanes = 0;
    for (size_t i = 0; i < reqs.size(); ++i) {
        shapes[i] = choose_align_shape(reqs[i].n, reqs[i].m, reqs[i].k);
        if (shapes[i].words_per_lane == 0) { set_error("query longer than the supported maximum"); return FLX_ERR_UNSUPPORTED; }
        lanes += shapes[i].lanes_per_job;
    }
    auto fits = [](AlignRequest const& r, AlignShape const& sh) {
        u32 const nw = (r.m + 63) / 64, W = sh.words_per_lane, R = sh.lanes_per_job;
        i64 const width = (i64)r.n - (i64)r.m + 2 * (i64)r.k;
        if ((nw + W - 1) / W <= R || (sh.banded && (i64)64 * W * (R - 1) + R + 1 > width)) return true;
        return sh.banded && sh.queue != 0 && ring_delay(r.n, r.m, r.k, W, R) + 1u <= RING_QUEUE_MAX;      // (a ring that waits: DeviceApi::align gives it the largest queue)
    };
    if (!reqs.empty() && lanes / 64 >= align_few_waves()) {
        // A launch lasts at least as long as its longest job, and the jobs of a batch differ by a few columns (unions of a locus' windows): the
        // shape is chosen per class of query words, for the class's widest band - a ring's delay is the job's own (ring_delay), so the narrower
        // jobs of the class lose nothing on it. (Per job, 10-kb root alignments over a repeat-rich reference fell into a dozen launches of two
        // shapes and took 171 ms per 16384 reads instead of 46.)
        std::map<u32, size_t> widest;                        // query words -> request with the widest band
        auto width_of = [](AlignRequest const& r) { return (i64)r.n - (i64)r.m + 2 * (i64)r.k; };
        for (size_t i = 0; i < reqs.size(); ++i) {
            u32 const nw = (reqs[i].m + 63) / 64;
            auto it = widest.find(nw);
            if (it == widest.end() || width_of(reqs[i]) > width_of(reqs[it->second])) widest[nw] = i;
        }
        for (size_t i = 0; i < reqs.size(); ++i) {
            AlignShape const cand = shapes[widest[(reqs[i].m + 63) / 64]];
            if (fits(reqs[i], cand)) shapes[i] = cand;
        }
        // a handful of jobs with a shape of their own join the most common shape that can hold them instead of getting a launch
        std::map<ShapeKey, std::pair<u32, u32>> count;       // jobs, queue
        for (auto const& sh : shapes) { auto& c = count[ShapeKey{sh.words_per_lane, sh.lanes_per_job, sh.banded}]; c.first++; c.second = std::max(c.second, sh.queue); }
        if (count.size() > 1) {
            for (size_t i = 0; i < reqs.size(); ++i) {
                ShapeKey const mine{shapes[i].words_per_lane, shapes[i].lanes_per_job, shapes[i].banded};
                if (count[mine].first >= 64) continue;
                u32 best_n = 0;
                AlignShape best = shapes[i];
                for (auto const& kv : count) {
                    AlignShape const cand{kv.first.w, kv.first.g, kv.first.banded, kv.second.second};
                    if (kv.second.first >= 64 && kv.second.first > best_n && fits(reqs[i], cand)) { best_n = kv.second.first; best = cand; }
                }
                shapes[i] = best;
            }
        }
        return FLX_OK;
    }
    if (reqs.empty()) return FLX_OK;
    AlignShape common{0, 0, shapes[0].banded};
    for (size_t i = 0; i < reqs.size(); ++i) {
        AlignShape const p = choose_align_shape(reqs[i].n, reqs[i].m, reqs[i].k, true);
        if (p.words_per_lane > common.words_per_lane) common.words_per_lane = p.words_per_lane;
    }
    // lanes each job needs at the common words per lane
    for (size_t i = 0; i < reqs.size(); ++i) {
        u32 const nw = (reqs[i].m + 63) / 64, W = common.words_per_lane;
        i64 const width = (i64)reqs[i].n - (i64)reqs[i].m + 2 * (i64)reqs[i].k;
        u32 r = 1;
        while (r < 64 && !((nw + W - 1) / W <= r || (common.banded && (i64)64 * W * (r - 1) + r + 1 > width))) r *= 2;
        if (r > common.lanes_per_job) common.lanes_per_job = r;
    }
    for (auto& sh : shapes) sh = common;
    return FLX_OK;
}

// FLX_EXISTS_LANES=1: the existence tests through the lane-per-job kernel with Ukkonen's cutoff (flx_lanes.hip) instead of the ring form of
// flx_device.hip. It computes a ninth of the blocks, and the pipeline runs as fast with either (it is not short of issue slots,
// profiles/r03_experiments.txt 5-6); alone the ring form is the faster one (40 against 60 ms per 16384 reads: a job's word groups run
// side by side on its ring, one after the other on its lane), so the ring form is the default.
static bool exists_lane_form() { return getenv("FLX_EXISTS_LANES") != nullptr; }
// its waves and the blocks its per-lane rows hold for windows of at most `width` diagonals (n - m + 2k); false: the rows would not fit the LDS
// lanes a job gets in the lane form (ed_exists_team_kernel): one for small nodes (many jobs, short chains), more for the large ones of
// which there are few with thousands of blocks each; rows = the smallest node of the launch. FLX_EXISTS_TEAM fixes it.
static u32 exists_team_size(u64 rows) {
    if (const char* e = getenv("FLX_EXISTS_TEAM")) { int const fixed = atoi(e); if (fixed > 0) return (u32)fixed; }      // (read per call: tests switch it)
    u64 const groups = (rows + 63) / 64;
    return groups >= 48 ? 16u : groups >= 20 ? 8u : groups >= 8 ? 4u : groups >= 4 ? 2u : 1u;
}
static bool exists_lane_setup(u64 max_jobs, i64 width, u32& waves, u32& cap_blocks, u32 team = 1) {
    cap_blocks = (u32)((64 + std::max<i64>(width, 0)) / 16 + 3) | 1u;                   // (odd: the lanes' rows start in different banks)
    static u32 const max_waves = [] { const char* e = getenv("FLX_EXISTS_LANE_WAVES"); return (u32)(e ? std::max(1, atoi(e)) : 4096); }();
    waves = (u32)std::max<u64>(std::min<u64>(max_waves, (max_jobs * team + 63) / 64), 1);
    return DeviceApi::exists_lane_lds_bytes(cap_blocks) <= 150 * 1024;
}

// score + end column for every (distinct) request (no trace)
int run_score_jobs_unique(Lane* ctx, const u8* d_text, const u64* d_peq, hvec<AlignRequest> const& reqs,
                          hvec<DevAlignOut>& outs, const char* kernel_name) {
    outs.assign(reqs.size(), DevAlignOut{0xFFFFFFFFu, 0});
    if (reqs.empty()) return FLX_OK;
    PhaseTimer jprof("score-jobs");
    i64 lane_width = 0;
    for (auto const& r : reqs) lane_width = std::max<i64>(lane_width, (i64)r.n - (i64)r.m + 2 * (i64)r.k);
    u32 waves = 0, cap_blocks = 0;
    u64 rows_min = ~0ull;
    for (auto const& r : reqs) rows_min = std::min<u64>(rows_min, r.m);
    u32 const team = exists_team_size(rows_min);
    if (exists_lane_form() && choose_align_shape(reqs[0].n, reqs[0].m, reqs[0].k).banded && exists_lane_setup(reqs.size(), lane_width, waves, cap_blocks, team)) {
        // one launch for every shape: a lane per job
        hvec<DevAlignJob> jobs(reqs.size());
        u64 steps = 0, bytes = 0;
        for (u32 i = 0; i < reqs.size(); ++i) {
            AlignRequest const& r = reqs[i];
            if (r.m > align_supported_max_query()) { set_error("query longer than the supported maximum"); return FLX_ERR_UNSUPPORTED; }
            jobs[i] = DevAlignJob{r.ref_off, r.q_off, 0, r.n, r.m, r.k, i, 0};
            steps += job_word_steps(r.n, r.m, r.k, AlignShape{1, 1, 1});
            bytes += (u64)r.n + r.m;
        }
        int rc;
        if ((rc = h2d(ctx, ctx->jobs, jobs.data(), jobs.size() * sizeof(DevAlignJob)))) return rc;
        if ((rc = ctx->job_out.ensure(reqs.size() * sizeof(DevAlignOut)))) return rc;
        if ((rc = ctx->counters.ensure(128))) return rc;
        FLX_HIP(hipMemsetAsync(ctx->counters.ptr, 0, 128, ctx->stream));
        rc = timed_launch(ctx, kernel_name, bytes, steps, [&] {
            return DeviceApi::align_exists_lanes(ctx->stream, d_text, d_peq, ctx->jobs.as<DevAlignJob>(), (u32)jobs.size(), nullptr, ctx->counters.as<u32>(),
                                                 waves, cap_blocks, ctx->job_out.as<DevAlignOut>(),
                                                 getenv("FLX_ALIGN_DEBUG") ? (unsigned long long*)((char*)ctx->counters.ptr + 64) : nullptr, team);
        });
        if (rc) return rc;
        u32 cnt[32];
        if ((rc = d2h(ctx, outs.data(), ctx->job_out.ptr, reqs.size() * sizeof(DevAlignOut)))) return rc;
        if ((rc = d2h(ctx, cnt, ctx->counters.ptr, 128))) return rc;
        jprof.mark("launch");
        rc = ctx->sync();
        jprof.mark("wait");
        if (rc) return rc;
        if (cnt[1]) { set_error("existence tests: a window did not fit the row buffers"); return FLX_ERR_INTERNAL; }
        if (getenv("FLX_ALIGN_DEBUG")) {
            unsigned long long st[8];
            memcpy(st, (char*)cnt + 64, sizeof(st));
            fprintf(stderr, "[%s lanes] jobs %zu waves %u cap %u: blocks %llu, wave iterations %llu, lane iterations %llu, groups %llu\n", kernel_name, jobs.size(), waves, cap_blocks, st[0], st[1], st[2], st[3]);
        }
        return FLX_OK;
    }
    std::map<ShapeKey, hvec<u32>> by_shape;
    {
        hvec<AlignShape> shapes;
        if (int const rc = choose_shapes(reqs, shapes)) return rc;
        jprof.mark("shapes");
        for (u32 i = 0; i < reqs.size(); ++i) by_shape[ShapeKey{shapes[i].words_per_lane, shapes[i].lanes_per_job, shapes[i].banded}].push_back(i);
    }
    jprof.mark("by-shape");
    hvec<DevAlignJob> jobs;
    jobs.reserve(reqs.size());
    struct Launch { ShapeKey key; u32 first, count; u64 word_steps, bytes; };
    hvec<Launch> launches;
    for (auto& kv : by_shape) {
        auto& ids = kv.second;
        Launch l{kv.first, (u32)jobs.size(), (u32)ids.size(), 0, 0};
        for (u32 id : ids) {
            AlignRequest const& r = reqs[id];
            jobs.push_back(DevAlignJob{r.ref_off, r.q_off, 0, r.n, r.m, r.k, id, 0});
            l.word_steps += job_word_steps(r.n, r.m, r.k, AlignShape{kv.first.w, kv.first.g, kv.first.banded});
            l.bytes += (u64)r.n + r.m;
        }
        launches.push_back(l);
    }
    jprof.mark("job-list");
    int rc;
    if ((rc = h2d(ctx, ctx->jobs, jobs.data(), jobs.size() * sizeof(DevAlignJob)))) return rc;
    if ((rc = ctx->job_out.ensure(reqs.size() * sizeof(DevAlignOut)))) return rc;
    for (auto const& l : launches) {
        if (getenv("FLX_ALIGN_DEBUG")) fprintf(stderr, "[%s] W %u R %u banded %u jobs %u word-steps %llu n0 %u m0 %u k0 %u\n", kernel_name, l.key.w, l.key.g, l.key.banded, l.count, (unsigned long long)l.word_steps, jobs[l.first].n, jobs[l.first].m, jobs[l.first].k);
        rc = timed_launch(ctx, kernel_name, l.bytes, l.word_steps, [&] {
            return DeviceApi::align(ctx->stream, d_text, d_peq, ctx->jobs.as<DevAlignJob>() + l.first, l.count,
                                    AlignShape{l.key.w, l.key.g, l.key.banded}, false, nullptr, ctx->job_out.as<DevAlignOut>());
        });
        if (rc) return rc;
    }
    if ((rc = d2h(ctx, outs.data(), ctx->job_out.ptr, reqs.size() * sizeof(DevAlignOut)))) return rc;
    jprof.mark("launch");
    rc = ctx->sync();
    jprof.mark("wait");
    return rc;
}

int run_score_jobs(Lane* ctx, const u8* d_text, const u64* d_peq, hvec<AlignRequest> const& reqs,
                   hvec<DevAlignOut>& outs, const char* kernel_name) {
    hvec<AlignRequest> uniq;
    hvec<u32> uniq_of;
    dedup_requests(reqs, uniq, uniq_of);
    hvec<DevAlignOut> uouts;
    int rc = run_score_jobs_unique(ctx, d_text, d_peq, uniq, uouts, kernel_name);
    if (rc) return rc;
    outs.resize(reqs.size());
    for (size_t i = 0; i < reqs.size(); ++i) outs[i] = uouts[uniq_of[i]];
    return FLX_OK;
}

struct TraceResult { bool exists = false; u32 nm = 0; u32 begin = 0; u64 cigar_off = 0; u32 cigar_len = 0; };

int run_trace_jobs_unique(Lane* ctx, const u8* d_text, const u8* d_query, const u64* d_peq, hvec<AlignRequest> const& reqs,
                          hvec<TraceResult>& results, hvec<u32>& cigar_pool);
int choose_shapes(hvec<AlignRequest> const& reqs, hvec<AlignShape>& shapes);

// score, begin position and CIGAR for every request (alignment.cpp:147-180); CIGAR words land in cigar_pool (shared by duplicates)
int run_trace_jobs(Lane* ctx, const u8* d_text, const u8* d_query, const u64* d_peq, hvec<AlignRequest> const& reqs,
                   hvec<TraceResult>& results, hvec<u32>& cigar_pool) {
    hvec<AlignRequest> uniq;
    hvec<u32> uniq_of;
    dedup_requests(reqs, uniq, uniq_of);
    hvec<TraceResult> ures;
    int rc = run_trace_jobs_unique(ctx, d_text, d_query, d_peq, uniq, ures, cigar_pool);
    if (rc) return rc;
    results.resize(reqs.size());
    for (size_t i = 0; i < reqs.size(); ++i) results[i] = ures[uniq_of[i]];
    return FLX_OK;
}

// Existence tests of one locus. Anchors of the same read at the same locus test the same node in windows shifted by their indel
// drift. Existence is monotone in the window: an alignment inside the intersection I of such windows lies inside every one of them,
// and if their union U holds none then neither does any of them. So a cluster first tests I (one job instead of one per member);
// only if that fails it tests U, and only if U holds an alignment that I does not are the members tested one by one.
// outs[i].score is 0xFFFFFFFF for "no alignment within k" and some score <= k of a contained alignment otherwise (callers of
// this function only look at that distinction); outs[i].end_col is not meaningful.
thread_local double g_exists_ms[4] = {0, 0, 0, 0};            // dedup, cluster, GPU round trip, scatter (FLX_HOST_PROFILE)
int run_exists_jobs(Lane* ctx, const u8* d_text, const u64* d_peq, hvec<AlignRequest> const& reqs, hvec<DevAlignOut>& outs) {
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](int slot) { auto const t1 = std::chrono::steady_clock::now(); g_exists_ms[slot] += std::chrono::duration<double, std::milli>(t1 - t0).count(); t0 = t1; };
    // The requests come in anchor order, and the anchors of a read and orientation are in leaf order: all requests for one node
    // (the leaves below it are a contiguous range) form one run of equal (query rows, errors). Sorting a run by reference
    // position puts equal windows and the windows of one locus next to each other: no hash table, no sort of the whole round.
    // (Requests for one node that are not adjacent would only be tested more than once.)
    static int const disabled = (getenv("FLX_NO_UNION") || getenv("FLX_NO_EXISTS_CLUSTERS")) ? 1 : 0;
    hvec<AlignRequest> uniq;
    hvec<u32> uniq_of(reqs.size());
    hvec<u32> order;                                          // position in `uniq` (identity: kept for the code below)
    struct Cluster { u32 first, count; u64 lo_start, hi_start, lo_end, hi_end; };     // members = uniq[first .. first+count)
    hvec<Cluster> clusters;
    uniq.reserve(reqs.size());
    hvec<u32> run;
    for (size_t i0 = 0; i0 < reqs.size();) {
        size_t i1 = i0 + 1;
        while (i1 < reqs.size() && reqs[i1].q_off == reqs[i0].q_off && reqs[i1].m == reqs[i0].m && reqs[i1].k == reqs[i0].k) ++i1;
        run.resize(i1 - i0);
        for (size_t j = 0; j < run.size(); ++j) run[j] = (u32)(i0 + j);
        if (run.size() > 1)
            std::sort(run.begin(), run.end(), [&](u32 x, u32 y) { return reqs[x].ref_off != reqs[y].ref_off ? reqs[x].ref_off < reqs[y].ref_off : reqs[x].n < reqs[y].n; });
        bool first_of_run = true;
        for (u32 idx : run) {
            AlignRequest const& r = reqs[idx];
            if (!first_of_run && uniq.back().ref_off == r.ref_off && uniq.back().n == r.n) { uniq_of[idx] = (u32)uniq.size() - 1; continue; }
            uniq_of[idx] = (u32)uniq.size();
            if (!first_of_run && !disabled) {
                Cluster& c = clusters.back();
                if (r.ref_off <= uniq[c.first].ref_off + std::max<u64>(8, r.m / 8)) {
                    c.count++;
                    c.hi_start = std::max(c.hi_start, r.ref_off);
                    c.lo_end = std::min(c.lo_end, r.ref_off + r.n);
                    c.hi_end = std::max(c.hi_end, r.ref_off + r.n);
                    uniq.push_back(r);
                    continue;
                }
            }
            clusters.push_back(Cluster{(u32)uniq.size(), 1, r.ref_off, r.ref_off, r.ref_off + r.n, r.ref_off + r.n});
            uniq.push_back(r);
            first_of_run = false;
        }
        i0 = i1;
    }
    order.resize(uniq.size());
    std::iota(order.begin(), order.end(), 0u);
    lap(0);
    hvec<DevAlignOut> uouts(uniq.size(), DevAlignOut{0xFFFFFFFFu, 0});
    // ---- one launch: single windows on their own, clusters on their intersection and (speculatively: a separate round trip
    //      to the GPU costs a chunk more than the extra jobs) on their union
    hvec<AlignRequest> jobs;
    hvec<u32> job_cluster;                                   // cluster index, bit 31 set for the union job
    for (u32 ci = 0; ci < clusters.size(); ++ci) {
        Cluster const& c = clusters[ci];
        AlignRequest r = uniq[order[c.first]];
        if (c.count == 1) { jobs.push_back(r); job_cluster.push_back(ci); continue; }
        if (c.lo_end > c.hi_start) {                         // the common columns (none: straight to the union and the members)
            AlignRequest i = r;
            i.ref_off = c.hi_start;
            i.n = (u32)(c.lo_end - c.hi_start);
            jobs.push_back(i);
            job_cluster.push_back(ci);
        }
        r.ref_off = c.lo_start;
        r.n = (u32)(c.hi_end - c.lo_start);
        jobs.push_back(r);
        job_cluster.push_back(ci | 0x80000000u);
    }
    hvec<DevAlignOut> jouts;
    lap(1);
    int rc = run_score_jobs_unique(ctx, d_text, d_peq, jobs, jouts, "ed_align_exists");
    if (rc) return rc;
    lap(2);
    hvec<u8> state(clusters.size(), 0);                      // 0 undecided, 1 all pass, 2 all fail
    hvec<u32> pass_score(clusters.size(), 0);
    for (size_t j = 0; j < jobs.size(); ++j) {
        u32 const ci = job_cluster[j] & 0x7FFFFFFFu;
        bool const is_union = job_cluster[j] >> 31;
        bool const found = jouts[j].score != 0xFFFFFFFFu;
        if (!is_union) {
            if (found) { state[ci] = 1; pass_score[ci] = jouts[j].score; }
            else if (clusters[ci].count == 1) state[ci] = 2;
        } else if (!found) state[ci] = 2;                   // (an intersection cannot hold an alignment the union does not)
    }
    // ---- phase C: members of the clusters that are still undecided, one by one
    jobs.clear();
    hvec<u32> job_member;
    for (u32 ci = 0; ci < clusters.size(); ++ci) {
        Cluster const& c = clusters[ci];
        if (state[ci] != 0) continue;
        for (u32 j = 0; j < c.count; ++j) { jobs.push_back(uniq[order[c.first + j]]); job_member.push_back(order[c.first + j]); }
    }
    if (!jobs.empty()) {
        if ((rc = run_score_jobs_unique(ctx, d_text, d_peq, jobs, jouts, "ed_align_exists"))) return rc;
        for (size_t j = 0; j < jobs.size(); ++j) uouts[job_member[j]] = jouts[j];
    }
    for (u32 ci = 0; ci < clusters.size(); ++ci) {
        if (state[ci] == 0) continue;
        Cluster const& c = clusters[ci];
        for (u32 j = 0; j < c.count; ++j) uouts[order[c.first + j]] = DevAlignOut{state[ci] == 1 ? pass_score[ci] : 0xFFFFFFFFu, 0};
    }
    if (getenv("FLX_ALIGN_DEBUG")) {
        size_t multi = 0, decided_a = 0;
        for (u32 ci = 0; ci < clusters.size(); ++ci) if (clusters[ci].count > 1) { ++multi; if (state[ci] == 1) ++decided_a; }
        fprintf(stderr, "[exists clusters] requests %zu distinct %zu clusters %zu (of several windows %zu, passed on the intersection %zu) one by one %zu\n",
                reqs.size(), uniq.size(), clusters.size(), multi, decided_a, jobs.size());
    }
    outs.resize(reqs.size());
    for (size_t i = 0; i < reqs.size(); ++i) outs[i] = uouts[uniq_of[i]];
    lap(3);
    return FLX_OK;
}

// Root alignments of one locus. Anchors of the same read at the same locus ask for windows that differ by a few columns (their
// indel drift), ten per read with floxer's defaults, and nearly always get the same alignment. One DP over the union U of such
// windows serves them all, exactly:
//   * a window w is a column range of U, and D_U <= D_w cell by cell (U only adds start columns), with equality on every cell of
//     a D_U-optimal path that starts inside w;
//   * let j be the rightmost column of w with the minimal D_U[m][.] = v over w. If the path traced back from (m, j) in D_U starts at
//     a column of w, then D_w = D_U along it, so min D_w = v, j is also the rightmost minimum of D_w (right of j D_w >= D_U > v), and
//     the trace decisions along the path agree (a move D_U rejects is rejected by D_w as well, a move D_U takes leads to a cell of
//     the path): score, end, begin and CIGAR of w are those read off U;
//   * v > k: no alignment in w either; the path starts left of w (rare): w is aligned on its own as before.
// The band of U contains the band of every member, and a banded value is exact whenever it is <= k.
constexpr u64 UNION_MAX_SHIFT = 256;      // members start within this many columns of the first member of their union

int run_trace_jobs_union(Lane* ctx, const u8* d_text, const u8* d_query, const u64* d_peq, hvec<AlignRequest> const& reqs,
                         hvec<TraceResult>& results, hvec<u32>& cigar_pool) {
    hvec<AlignRequest> uniq;
    hvec<u32> uniq_of;
    dedup_requests(reqs, uniq, uniq_of);
    hvec<TraceResult> ures(uniq.size());
    bool usable = !uniq.empty() && choose_align_shape(uniq[0].n, uniq[0].m, uniq[0].k).banded != 0 && !getenv("FLX_NO_UNION");
    for (auto const& r : uniq) usable = usable && r.k < 0xFFFFu;
    // ---- unions: same query rows, starts within UNION_MAX_SHIFT of the first member
    struct Union { AlignRequest req; u32 first_member, n_members; };
    hvec<u32> order(uniq.size());
    hvec<Union> unions;
    hvec<u32> members;                       // indices into uniq, grouped by union
    if (usable) {
        std::iota(order.begin(), order.end(), 0u);
        std::sort(order.begin(), order.end(), [&](u32 a, u32 b) {
            AlignRequest const &x = uniq[a], &y = uniq[b];
            if (x.q_off != y.q_off) return x.q_off < y.q_off;
            if (x.m != y.m) return x.m < y.m;
            if (x.k != y.k) return x.k < y.k;
            return x.ref_off < y.ref_off;
        });
        for (u32 id : order) {
            AlignRequest const& r = uniq[id];
            if (!unions.empty()) {
                Union& u = unions.back();
                if (u.req.q_off == r.q_off && u.req.m == r.m && u.req.k == r.k && r.ref_off <= u.req.ref_off + UNION_MAX_SHIFT) {
                    u64 const end = std::max<u64>(u.req.ref_off + u.req.n, r.ref_off + r.n);
                    u.req.n = (u32)(end - u.req.ref_off);
                    u.n_members++;
                    members.push_back(id);
                    continue;
                }
            }
            unions.push_back(Union{r, (u32)members.size(), 1});
            members.push_back(id);
        }
    }
    if (!usable || unions.size() == uniq.size()) {          // nothing to share: the plain path
        int const rc = run_trace_jobs_unique(ctx, d_text, d_query, d_peq, uniq, ures, cigar_pool);
        if (rc) return rc;
        results.resize(reqs.size());
        for (size_t i = 0; i < reqs.size(); ++i) results[i] = ures[uniq_of[i]];
        return FLX_OK;
    }

    hvec<AlignRequest> ureqs(unions.size());
    for (size_t i = 0; i < unions.size(); ++i) ureqs[i] = unions[i].req;
    hvec<AlignShape> shapes;
    if (int const src = choose_shapes(ureqs, shapes)) return src;
    hvec<u64> slots(ureqs.size());
    u64 const budget_slots = std::max<u64>(ctx->trace_budget_bytes / 16, 1);
    for (size_t i = 0; i < ureqs.size(); ++i) {
        slots[i] = align_trace_slots(ureqs[i].n, ureqs[i].m, ureqs[i].k, shapes[i]);
        if (slots[i] > budget_slots) { set_error("one alignment needs more trace memory than the configured budget (FLX_TRACE_ARENA_MB)"); return FLX_ERR_CAPACITY; }
    }
    hvec<AlignRequest> fallback;
    hvec<u32> fallback_of;                   // uniq index of each fallback request
    int rc;
    size_t next = 0;
    while (next < ureqs.size()) {
        size_t const begin = next;
        u64 used = 0;
        while (next < ureqs.size() && used + slots[next] <= budget_slots) { used += slots[next]; ++next; }
        size_t const count = next - begin;
        if ((rc = ctx->trace.ensure(std::max<size_t>(used * 16 + 64, ctx->trace.ptr ? 0 : std::min<size_t>(ctx->trace_budget_bytes, (size_t)budget_slots * 16) / 3 * 2)))) return rc;

        // ---- K4 over the unions of this arena chunk, with their last rows
        std::map<ShapeKey, hvec<u32>> by_shape;
        for (size_t i = begin; i < next; ++i) by_shape[ShapeKey{shapes[i].words_per_lane, shapes[i].lanes_per_job, shapes[i].banded}].push_back((u32)i);
        hvec<DevAlignJob> jobs;
        hvec<u64> trace_off(count), row_off(count);
        struct Launch { ShapeKey key; u32 first, count; u64 word_steps, bytes; };
        hvec<Launch> launches;
        u64 off = 0, rows = 0;
        for (auto& kv : by_shape) {
            auto& ids = kv.second;
            std::stable_sort(ids.begin(), ids.end(), [&](u32 a, u32 b) { return ureqs[a].n > ureqs[b].n; });
            Launch l{kv.first, (u32)jobs.size(), (u32)ids.size(), 0, 0};
            for (u32 id : ids) {
                AlignRequest const& r = ureqs[id];
                trace_off[id - begin] = off;
                row_off[id - begin] = rows;
                jobs.push_back(DevAlignJob{r.ref_off, r.q_off, off, r.n, r.m, r.k, (u32)(id - begin), rows});
                off += slots[id];
                rows += ((u64)r.n + 15) / 16 * 16;   // K4 stores a block's 16 last-row values as two 16-byte words
                l.word_steps += job_word_steps(r.n, r.m, r.k, shapes[id]);
                TraceLayout const tl = ckpt_trace_layout(r.n, r.m, r.k, shapes[id].words_per_lane, shapes[id].lanes_per_job);
                l.bytes += (u64)r.n + r.m + (tl.carry_slots + tl.ckpt_slots) * 16 + 2ull * r.n;
            }
            launches.push_back(l);
        }
        if ((rc = h2d(ctx, ctx->jobs, jobs.data(), jobs.size() * sizeof(DevAlignJob)))) return rc;
        if ((rc = ctx->job_out.ensure(count * sizeof(DevAlignOut)))) return rc;
        if ((rc = ctx->lastrow.ensure(rows * 2 + 64))) return rc;
        FLX_HIP(hipMemsetAsync(ctx->lastrow.ptr, 0xFF, rows * 2, ctx->stream));
        for (auto const& l : launches) {
            if (getenv("FLX_ALIGN_DEBUG")) fprintf(stderr, "[ed_align_trace] unions W %u R %u jobs %u word-steps %llu n0 %u m0 %u k0 %u\n", l.key.w, l.key.g, l.count, (unsigned long long)l.word_steps, jobs[l.first].n, jobs[l.first].m, jobs[l.first].k);
            rc = timed_launch(ctx, "ed_align_trace", l.bytes, l.word_steps, [&] {
                return DeviceApi::align(ctx->stream, d_text, d_peq, ctx->jobs.as<DevAlignJob>() + l.first, l.count,
                                        AlignShape{l.key.w, l.key.g, l.key.banded}, true, ctx->trace.as<u64>(), ctx->job_out.as<DevAlignOut>(),
                                        ctx->lastrow.as<u16>());
            });
            if (rc) return rc;
        }
        // ---- every member's rightmost minimum over its own columns
        hvec<DevRowWindow> wins;
        hvec<u32> win_member;                // uniq index per window
        hvec<u32> win_union;                 // union index (absolute) per window
        for (size_t ui = begin; ui < next; ++ui)
            for (u32 j = 0; j < unions[ui].n_members; ++j) {
                u32 const id = members[unions[ui].first_member + j];
                AlignRequest const& r = uniq[id];
                wins.push_back(DevRowWindow{row_off[ui - begin] + (r.ref_off - ureqs[ui].ref_off), r.n, r.k, (u32)wins.size(), 0});
                win_member.push_back(id);
                win_union.push_back((u32)ui);
            }
        if ((rc = h2d(ctx, ctx->row_windows, wins.data(), wins.size() * sizeof(DevRowWindow)))) return rc;
        if ((rc = ctx->row_out.ensure(wins.size() * sizeof(DevAlignOut)))) return rc;
        rc = timed_launch(ctx, "ed_lastrow_min", rows * 2, wins.size(), [&] {
            return DeviceApi::lastrow_min(ctx->stream, ctx->lastrow.as<u16>(), ctx->row_windows.as<DevRowWindow>(), (u32)wins.size(), ctx->row_out.as<DevAlignOut>());
        });
        if (rc) return rc;
        hvec<DevAlignOut> wouts(wins.size());
        if ((rc = d2h(ctx, wouts.data(), ctx->row_out.ptr, wins.size() * sizeof(DevAlignOut)))) return rc;
        if ((rc = ctx->sync())) return rc;

        // ---- one traceback per distinct (union, end column)
        hvec<DevTraceJob> tjobs;
        hvec<u32> win_tjob(wins.size(), 0xFFFFFFFFu);
        u64 cigar_words = 0, path_steps = 0;
        {
            size_t w0 = 0;
            while (w0 < wins.size()) {                     // windows of one union are consecutive
                size_t w1 = w0;
                while (w1 < wins.size() && win_union[w1] == win_union[w0]) ++w1;
                u32 const ui = win_union[w0];
                AlignRequest const& ur = ureqs[ui];
                AlignShape const sh = shapes[ui];
                for (size_t w = w0; w < w1; ++w) {
                    if (wouts[w].score == 0xFFFFFFFFu) continue;
                    u32 const end_in_union = (u32)(uniq[win_member[w]].ref_off - ur.ref_off) + wouts[w].end_col;
                    for (size_t v = w0; v < w; ++v)
                        if (win_tjob[v] != 0xFFFFFFFFu && tjobs[win_tjob[v]].end_col == end_in_union) { win_tjob[w] = win_tjob[v]; break; }
                    if (win_tjob[w] != 0xFFFFFFFFu) continue;
                    u32 const cap = 2 * wouts[w].score + 2;
                    win_tjob[w] = (u32)tjobs.size();
                    tjobs.push_back(DevTraceJob{ur.ref_off, ur.q_off, trace_off[ui - begin], cigar_words, ur.n, ur.m, sh.lanes_per_job, sh.words_per_lane,
                                                end_in_union, cap, (u32)tjobs.size(), ur.k});
                    cigar_words += cap;
                    path_steps += (u64)ur.m + wouts[w].score;
                }
                w0 = w1;
            }
        }
        hvec<DevTraceOut> touts(tjobs.size());
        size_t const pool_base = cigar_pool.size();
        if (!tjobs.empty()) {
            if ((rc = h2d(ctx, ctx->tjobs, tjobs.data(), tjobs.size() * sizeof(DevTraceJob)))) return rc;
            if ((rc = ctx->tjob_out.ensure(tjobs.size() * sizeof(DevTraceOut)))) return rc;
            if ((rc = ctx->cigar.ensure(cigar_words * 4 + 16))) return rc;
            rc = timed_launch(ctx, "ed_traceback", path_steps * 18, path_steps, [&] {
                return DeviceApi::traceback(ctx->stream, d_text, d_query, d_peq, ctx->trace.as<u64>(), ctx->tjobs.as<DevTraceJob>(),
                                            (u32)tjobs.size(), true, ctx->cigar.as<u32>(), ctx->tjob_out.as<DevTraceOut>());
            });
            if (rc) return rc;
            cigar_pool.resize(pool_base + cigar_words);
            if ((rc = d2h(ctx, touts.data(), ctx->tjob_out.ptr, touts.size() * sizeof(DevTraceOut)))) return rc;
            if ((rc = d2h(ctx, cigar_pool.data() + pool_base, ctx->cigar.ptr, cigar_words * 4))) return rc;
            if ((rc = ctx->sync())) return rc;
        }
        // ---- members take the union's alignment when its path starts inside their window
        for (size_t w = 0; w < wins.size(); ++w) {
            u32 const id = win_member[w];
            if (wouts[w].score == 0xFFFFFFFFu) continue;                       // no alignment within k in this window
            DevTraceOut const& t = touts[win_tjob[w]];
            if (t.cigar_len == 0xFFFFFFFFu) { set_error("ed_traceback: CIGAR slab overflow"); return FLX_ERR_INTERNAL; }
            u64 const shift = uniq[id].ref_off - ureqs[win_union[w]].ref_off;
            static int const force_own = getenv("FLX_UNION_ALIGN_OWN") ? 1 : 0;        // test hook: as if every path left its window
            if (t.begin < shift || force_own) { fallback_of.push_back(id); fallback.push_back(uniq[id]); continue; }
            TraceResult& res = ures[id];
            res.exists = true;
            res.nm = wouts[w].score;
            res.begin = (u32)(t.begin - shift);
            res.cigar_off = pool_base + tjobs[win_tjob[w]].cigar_off + t.cigar_start;
            res.cigar_len = t.cigar_len;
        }
    }
    if (!fallback.empty()) {
        hvec<TraceResult> fres;
        if ((rc = run_trace_jobs_unique(ctx, d_text, d_query, d_peq, fallback, fres, cigar_pool))) return rc;
        for (size_t i = 0; i < fallback.size(); ++i) ures[fallback_of[i]] = fres[i];
    }
    if (getenv("FLX_ALIGN_DEBUG")) fprintf(stderr, "[root unions] requests %zu distinct %zu unions %zu aligned on their own %zu\n", reqs.size(), uniq.size(), unions.size(), fallback.size());
    results.resize(reqs.size());
    for (size_t i = 0; i < reqs.size(); ++i) results[i] = ures[uniq_of[i]];
    return FLX_OK;
}

int run_trace_jobs_unique(Lane* ctx, const u8* d_text, const u8* d_query, const u64* d_peq, hvec<AlignRequest> const& reqs,
                          hvec<TraceResult>& results, hvec<u32>& cigar_pool) {
    results.assign(reqs.size(), TraceResult{});
    if (reqs.empty()) return FLX_OK;
    PhaseTimer tprof("trace-jobs");
    hvec<AlignShape> shapes;
    if (int const src = choose_shapes(reqs, shapes)) return src;
    hvec<u64> slots(reqs.size());
    u64 const budget_slots = std::max<u64>(ctx->trace_budget_bytes / 16, 1);
    for (size_t i = 0; i < reqs.size(); ++i) {
        slots[i] = align_trace_slots(reqs[i].n, reqs[i].m, reqs[i].k, shapes[i]);
        if (slots[i] > budget_slots) { set_error("one alignment needs more trace memory than the configured budget (FLX_TRACE_ARENA_MB)"); return FLX_ERR_CAPACITY; }
    }
    int rc;
    size_t next = 0;
    while (next < reqs.size()) {
        // ---- chunk of jobs whose trace planes fit the arena
        size_t begin = next;
        u64 used = 0;
        while (next < reqs.size() && used + slots[next] <= budget_slots) { used += slots[next]; ++next; }
        size_t const count = next - begin;
        // the arena is taken whole on first use (its size is the configured budget): no reallocation between batches
        if ((rc = ctx->trace.ensure(std::max<size_t>(used * 16 + 64, ctx->trace.ptr ? 0 : std::min<size_t>(ctx->trace_budget_bytes, (size_t)budget_slots * 16) / 3 * 2)))) return rc;

        std::map<ShapeKey, hvec<u32>> by_shape;
        for (size_t i = begin; i < next; ++i) by_shape[ShapeKey{shapes[i].words_per_lane, shapes[i].lanes_per_job, shapes[i].banded}].push_back((u32)i);
        hvec<DevAlignJob> jobs;
        hvec<u64> trace_off(count);
        struct Launch { ShapeKey key; u32 first, count; u64 word_steps, bytes; };
        hvec<Launch> launches;
        u64 off = 0;
        for (auto& kv : by_shape) {
            auto& ids = kv.second;
            std::stable_sort(ids.begin(), ids.end(), [&](u32 a, u32 b) { return reqs[a].n > reqs[b].n; });
            Launch l{kv.first, (u32)jobs.size(), (u32)ids.size(), 0, 0};
            for (u32 id : ids) {
                AlignRequest const& r = reqs[id];
                trace_off[id - begin] = off;
                jobs.push_back(DevAlignJob{r.ref_off, r.q_off, off, r.n, r.m, r.k, (u32)(id - begin), 0});
                off += slots[id];
                u64 const ws = job_word_steps(r.n, r.m, r.k, shapes[id]);
                l.word_steps += ws;
                // reference + query symbols read; trace written: full form 16 B per word-step, checkpointed form its carry and
                // checkpoint regions
                if (shapes[id].banded) {
                    TraceLayout const tl = ckpt_trace_layout(r.n, r.m, r.k, shapes[id].words_per_lane, shapes[id].lanes_per_job);
                    l.bytes += (u64)r.n + r.m + (tl.carry_slots + tl.ckpt_slots) * 16;
                } else l.bytes += (u64)r.n + r.m + ws * 16;
            }
            launches.push_back(l);
        }
        tprof.mark("prep");
        if ((rc = h2d(ctx, ctx->jobs, jobs.data(), jobs.size() * sizeof(DevAlignJob)))) return rc;
        if ((rc = ctx->job_out.ensure(count * sizeof(DevAlignOut)))) return rc;
        for (auto const& l : launches) {
            if (getenv("FLX_ALIGN_DEBUG")) fprintf(stderr, "[ed_align_trace] W %u R %u banded %u jobs %u word-steps %llu n0 %u m0 %u k0 %u\n", l.key.w, l.key.g, l.key.banded, l.count, (unsigned long long)l.word_steps, jobs[l.first].n, jobs[l.first].m, jobs[l.first].k);
            rc = timed_launch(ctx, "ed_align_trace", l.bytes, l.word_steps, [&] {
                return DeviceApi::align(ctx->stream, d_text, d_peq, ctx->jobs.as<DevAlignJob>() + l.first, l.count,
                                        AlignShape{l.key.w, l.key.g, l.key.banded}, true, ctx->trace.as<u64>(), ctx->job_out.as<DevAlignOut>());
            });
            if (rc) return rc;
        }
        hvec<DevAlignOut> outs(count);
        if ((rc = d2h(ctx, outs.data(), ctx->job_out.ptr, count * sizeof(DevAlignOut)))) return rc;
        if ((rc = ctx->sync())) return rc;
        tprof.mark("K4");

        // ---- traceback for the jobs that have an alignment within k
        hvec<DevTraceJob> tjobs;
        hvec<u32> tjob_req;
        u64 cigar_words = 0, path_steps = 0;
        for (size_t c = 0; c < count; ++c) {
            if (outs[c].score == 0xFFFFFFFFu) continue;
            size_t const id = begin + c;
            AlignRequest const& r = reqs[id];
            AlignShape const sh = shapes[id];
            u32 const nw = (r.m + 63) / 64;
            u32 const L = sh.banded ? sh.lanes_per_job : (nw + sh.words_per_lane - 1) / sh.words_per_lane;
            u32 const cap = 2 * outs[c].score + 2;      // runs <= 2*NM + 1
            tjobs.push_back(DevTraceJob{r.ref_off, r.q_off, trace_off[c], cigar_words, r.n, r.m, L, sh.words_per_lane, outs[c].end_col,
                                        cap, (u32)tjob_req.size(), r.k});
            tjob_req.push_back((u32)id);
            cigar_words += cap;
            path_steps += (u64)r.m + outs[c].score;
        }
        if (!tjobs.empty()) {
            if ((rc = h2d(ctx, ctx->tjobs, tjobs.data(), tjobs.size() * sizeof(DevTraceJob)))) return rc;
            if ((rc = ctx->tjob_out.ensure(tjobs.size() * sizeof(DevTraceOut)))) return rc;
            if ((rc = ctx->cigar.ensure(cigar_words * 4 + 16))) return rc;
            rc = timed_launch(ctx, "ed_traceback", path_steps * 18, path_steps, [&] {
                return DeviceApi::traceback(ctx->stream, d_text, d_query, d_peq, ctx->trace.as<u64>(), ctx->tjobs.as<DevTraceJob>(),
                                            (u32)tjobs.size(), shapes[begin].banded != 0, ctx->cigar.as<u32>(), ctx->tjob_out.as<DevTraceOut>());
            });
            if (rc) return rc;
            tprof.mark("tb-prep");
            hvec<DevTraceOut> touts(tjobs.size());
            size_t const pool_base = cigar_pool.size();
            cigar_pool.resize(pool_base + cigar_words);          // slabs are kept as they are (gaps included): no host repacking
            tprof.mark("pool-resize");
            if ((rc = d2h(ctx, touts.data(), ctx->tjob_out.ptr, touts.size() * sizeof(DevTraceOut)))) return rc;
            if ((rc = d2h(ctx, cigar_pool.data() + pool_base, ctx->cigar.ptr, cigar_words * 4))) return rc;
            if ((rc = ctx->sync())) return rc;
            tprof.mark("K5+d2h");
            for (size_t j = 0; j < tjobs.size(); ++j) {
                if (touts[j].cigar_len == 0xFFFFFFFFu) { set_error("ed_traceback: CIGAR slab overflow"); return FLX_ERR_INTERNAL; }
                TraceResult& res = results[tjob_req[j]];
                res.exists = true;
                res.nm = outs[tjob_req[j] - begin].score;
                res.begin = touts[j].begin;
                res.cigar_off = pool_base + tjobs[j].cigar_off + touts[j].cigar_start;
                res.cigar_len = touts[j].cigar_len;
            }
        }
    }
    return FLX_OK;
}

int build_peq(Lane* ctx, const u8* d_seq, u64 len, DeviceBuffer& peq) {
    u64 const n_words = len / 64 + 2;
    int rc = peq.ensure(n_words * 6 * 8 + 64);
    if (rc) return rc;
    return timed_launch(ctx, "peq_build", len + n_words * 48, n_words, [&] { return DeviceApi::build_peq(ctx->stream, d_seq, len, peq.as<u64>()); });
}

int ensure_reversed_text(Lane* lane) {
    flx_ctx* ctx = lane->ctx;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (ctx->text_rev_ready) return FLX_OK;
    HostIndex const& H = *ctx->hidx;
    hvec<u8> rev(H.n);
    if (H.text.size() == H.n) std::reverse_copy(H.text.begin(), H.text.end(), rev.begin());
    else {                                       // a context on a received image: the text is in HBM only
        FLX_HIP(hipMemcpy(rev.data(), ctx->didx.text, H.n, hipMemcpyDeviceToHost));
        std::reverse(rev.begin(), rev.end());
    }
    const u8* first = nullptr;
    int rc = upload_padded(lane, ctx->text_rev, rev.data(), rev.size(), &first);
    if (rc) return rc;
    FLX_HIP(hipStreamSynchronize(lane->stream));
    ctx->text_rev_ready = true;
    return FLX_OK;
}

}  // namespace

}  // namespace flx

// ================================================================================================ C ABI: seams 1 and 2
extern "C" int flx_search_seeds(flx_ctx* ctx, const uint8_t* seq_pool, uint64_t seq_pool_len, const flx_seed* seeds, uint64_t n_seeds,
                                const flx_search_config* cfg, flx_anchor* out_anchors, uint64_t* n_anchors, flx_seed_stats* out_stats) {
    if (!ctx || !cfg || !n_anchors || (n_seeds && (!seeds || !seq_pool))) { set_error("flx_search_seeds: null argument"); return FLX_ERR_INVALID; }
    if (cfg->max_num_anchors_hard < cfg->max_num_anchors_soft) { set_error("max-anchors-hard must not be smaller than max-anchors-soft (floxer_cli.cpp:194)"); return FLX_ERR_INVALID; }
    FLX_HIP(hipSetDevice(ctx->device));
    hvec<HostAnchor> anchors;
    hvec<SeedStats> stats;
    LaneLease lease(ctx, ctx->external_stream ? 0 : -1);
    int rc = search_seeds_device(lease.lane, nullptr, seq_pool, seq_pool_len, seeds, n_seeds, *cfg, anchors, stats, nullptr, 0);
    if (rc) return rc;
    uint64_t const cap = *n_anchors;
    *n_anchors = anchors.size();
    if (out_stats) for (uint64_t i = 0; i < n_seeds; ++i) out_stats[i] = flx_seed_stats{stats[i].useful, stats[i].raw, stats[i].excluded_soft, stats[i].fully_excluded};
    if (anchors.size() > cap) { set_error("anchor buffer too small"); return FLX_ERR_CAPACITY; }
    for (size_t i = 0; i < anchors.size(); ++i)
        out_anchors[i] = flx_anchor{anchors[i].seed_index, anchors[i].leaf, anchors[i].ref_id, anchors[i].errors, anchors[i].pos};
    return FLX_OK;
}

extern "C" int flx_search_groups(flx_ctx* ctx, const uint8_t* seq_pool, uint64_t seq_pool_len, const flx_seed* seeds, uint64_t n_seeds,
                                 uint64_t max_hits_per_seed, flx_hit_group* out, uint64_t* n_out) {
    if (!ctx || !n_out || (n_seeds && (!seeds || !seq_pool))) { set_error("flx_search_groups: null argument"); return FLX_ERR_INVALID; }
    FLX_HIP(hipSetDevice(ctx->device));
    hvec<HostAnchor> anchors;
    hvec<SeedStats> stats;
    hvec<DevHit> hits;
    flx_search_config cfg{};
    LaneLease lease(ctx, ctx->external_stream ? 0 : -1);
    int rc = search_seeds_device(lease.lane, nullptr, seq_pool, seq_pool_len, seeds, n_seeds, cfg, anchors, stats, &hits, max_hits_per_seed);
    if (rc) return rc;
    uint64_t const cap = *n_out;
    *n_out = hits.size();
    if (hits.size() > cap) { set_error("hit buffer too small"); return FLX_ERR_CAPACITY; }
    for (size_t i = 0; i < hits.size(); ++i) out[i] = flx_hit_group{hits[i].seed, hits[i].lb, hits[i].len, hits[i].errors};
    return FLX_OK;
}

extern "C" int flx_align_batch(flx_ctx* ctx, const uint8_t* ref_pool, uint64_t ref_pool_len, const uint8_t* query_pool,
                               uint64_t query_pool_len, const flx_align_job* jobs, uint64_t n_jobs, flx_align_result* out,
                               uint32_t* cigar_pool, uint64_t* cigar_pool_words) {
    if (!ctx || (n_jobs && (!jobs || !out || !query_pool))) { set_error("flx_align_batch: null argument"); return FLX_ERR_INVALID; }
    FLX_HIP(hipSetDevice(ctx->device));
    if (n_jobs >= (1ull << 31)) { set_error("too many jobs in one call"); return FLX_ERR_INVALID; }
    u64 const text_len = ref_pool ? ref_pool_len : ctx->hidx->n;
    bool any_rev = false, any_trace = false;
    for (uint64_t i = 0; i < n_jobs; ++i) {
        flx_align_job const& j = jobs[i];
        if (j.query_length == 0 || j.query_offset + j.query_length > query_pool_len || j.ref_offset + j.ref_length > text_len || j.mode > 2) {
            set_error("flx_align_batch: job outside its pools"); return FLX_ERR_INVALID;
        }
        if (j.query_length > align_supported_max_query()) { set_error("query longer than the supported maximum"); return FLX_ERR_UNSUPPORTED; }
        any_rev |= j.mode == FLX_MODE_WITHOUT_CIGAR;
        any_trace |= j.mode == FLX_MODE_WITH_CIGAR;
    }
    int rc;
    LaneLease lease(ctx, ctx->external_stream ? 0 : -1);
    Lane* L = lease.lane;
    const u8* d_text = ctx->didx.text;
    const u8* d_text_rev = nullptr;
    hvec<u8> tmp;
    if (ref_pool) {
        if ((rc = upload_padded(L, L->user_text, ref_pool, ref_pool_len, &d_text))) return rc;
        if (any_rev) {
            tmp.assign(ref_pool, ref_pool + ref_pool_len);
            std::reverse(tmp.begin(), tmp.end());
            if ((rc = upload_padded(L, L->user_text_rev, tmp.data(), tmp.size(), &d_text_rev))) return rc;
            if ((rc = L->sync())) return rc;
        }
    } else if (any_rev) {
        if ((rc = ensure_reversed_text(L))) return rc;
        d_text_rev = ctx->text_rev.as<u8>() + TEXT_PAD;
    }
    if ((rc = h2d(L, L->seq, query_pool, query_pool_len, 192))) return rc;
    if ((rc = build_peq(L, L->seq.as<u8>(), query_pool_len, L->peq))) return rc;
    hvec<u8> qrev;
    if (any_rev) {
        qrev.assign(query_pool, query_pool + query_pool_len);
        std::reverse(qrev.begin(), qrev.end());
        if ((rc = h2d(L, L->seq_rev, qrev.data(), qrev.size(), 64))) return rc;
        if ((rc = build_peq(L, L->seq_rev.as<u8>(), query_pool_len, L->peq_rev))) return rc;
    }
    hvec<AlignRequest> score_reqs, rev_reqs, trace_reqs;
    hvec<u32> score_ids, rev_ids, trace_ids;
    for (uint64_t i = 0; i < n_jobs; ++i) {
        flx_align_job const& j = jobs[i];
        if (j.mode == FLX_MODE_EXISTS) { score_reqs.push_back({j.ref_offset, j.query_offset, j.ref_length, j.query_length, j.num_allowed_errors}); score_ids.push_back((u32)i); }
        else if (j.mode == FLX_MODE_WITHOUT_CIGAR) {
            rev_reqs.push_back({text_len - j.ref_offset - j.ref_length, query_pool_len - j.query_offset - j.query_length, j.ref_length, j.query_length, j.num_allowed_errors});
            rev_ids.push_back((u32)i);
        } else { trace_reqs.push_back({j.ref_offset, j.query_offset, j.ref_length, j.query_length, j.num_allowed_errors}); trace_ids.push_back((u32)i); }
    }
    for (uint64_t i = 0; i < n_jobs; ++i) out[i] = flx_align_result{0, 0, 0, 0, 0, 0};
    hvec<DevAlignOut> outs;
    if ((rc = run_score_jobs(L, d_text, L->peq.as<u64>(), score_reqs, outs, "ed_align_exists"))) return rc;
    for (size_t i = 0; i < outs.size(); ++i)
        if (outs[i].score != 0xFFFFFFFFu) { out[score_ids[i]].exists = 1; out[score_ids[i]].num_errors = outs[i].score; }
    if ((rc = run_score_jobs(L, d_text_rev, L->peq_rev.as<u64>(), rev_reqs, outs, "ed_align_exists"))) return rc;
    for (size_t i = 0; i < outs.size(); ++i)
        if (outs[i].score != 0xFFFFFFFFu) {
            flx_align_result& r = out[rev_ids[i]];
            r.exists = 1; r.num_errors = outs[i].score; r.begin = rev_reqs[i].n - outs[i].end_col;      // alignment.cpp:135
        }
    hvec<TraceResult> tres;
    hvec<u32> cig;
    if ((rc = run_trace_jobs(L, d_text, L->seq.as<u8>(), L->peq.as<u64>(), trace_reqs, tres, cig))) return rc;
    uint64_t const cap = cigar_pool_words ? *cigar_pool_words : 0;
    if (cigar_pool_words) *cigar_pool_words = cig.size();
    if (any_trace && (!cigar_pool || cig.size() > cap)) { set_error("cigar pool too small"); return FLX_ERR_CAPACITY; }
    if (!cig.empty()) memcpy(cigar_pool, cig.data(), cig.size() * 4);
    for (size_t i = 0; i < tres.size(); ++i)
        if (tres[i].exists) {
            flx_align_result& r = out[trace_ids[i]];
            r.exists = 1; r.num_errors = tres[i].nm; r.begin = tres[i].begin; r.cigar_offset = tres[i].cigar_off; r.cigar_length = tres[i].cigar_len;
        }
    return FLX_OK;
}

// ================================================================================================ seam 3: whole path
namespace {

struct half_open { u64 start, end; };
half_open trim_both(half_open a, u64 amount) {                                                  // intervals.cpp:48-58
    u64 const new_end = std::max(a.start + 1, amount > a.end ? 0 : a.end - amount);
    u64 const new_start = std::min(new_end - 1, a.start + amount);
    return {new_start, new_end};
}
struct VerifiedIntervals {                                                                     // intervals.cpp:84-127
    hvec<half_open> ivs;
    bool contains(half_open t) const {
        for (auto const& e : ivs) if (e.start <= t.start && e.end >= t.end) return true;      // equal or contains
        return false;
    }
    void insert(half_open t) { if (!contains(t)) ivs.push_back(t); }
};

struct Span { u64 offset, length, extra; };
Span compute_span(u64 anchor_pos, flx_pex_node const& node, u64 leaf_from, u64 reflen, double ratio) {   // verification.cpp:157-184
    u64 const base = (u64)(node.to - node.from + 1) + 2ull * node.num_errors + 1;
    u64 const extra = ratio == 0.0 ? 0 : fp_aware_ceil(base * ratio);        // (inner nodes: no extension, fp_aware_ceil(0) = 0)
    i64 const start_signed = (i64)anchor_pos - (i64)(leaf_from - node.from) - (i64)node.num_errors - (i64)extra;
    u64 const start = start_signed >= 0 ? (u64)start_signed : 0;
    u64 const length = std::min(base + 2 * extra, reflen - start);
    return {start, length, extra};
}

struct pr_task { int priority; int id; bool operator<(pr_task const& o) const { return priority < o.priority; } };
// order in which one worker runs the verification packages of a read (BS::thread_pool's priority queue), parallelization.cpp:131-148
hvec<int> package_order(int n) {
    std::priority_queue<pr_task> q;
    for (int i = 0; i < n; ++i) q.push(pr_task{16383, i});
    q.push(pr_task{-16384, -1});
    hvec<int> order;
    while (!q.empty()) { pr_task t = q.top(); q.pop(); if (t.id < 0) break; order.push_back(t.id); }
    return order;
}

struct ReadState {
    u64 read_index;
    u32 len, k;
    u64 pool_off[2];            // forward, reverse complement
    const PexTree* tree_ptr = nullptr;      // reads of one length share one tree (it depends on (length, errors) only)
    PexTree const& tree_ref() const { return *tree_ptr; }
    hvec<u32> anchor_ids[2];
};

struct AnchorState {
    u32 read;                   // index into kept reads
    u8 orientation;
    u32 leaf, ref_id;
    u64 pos;
    u32 node;                   // inner node under test
    u32 node_rows = 0;          // its number of query rows (kept here: the rounds scan it)
    bool alive = true, at_root = false, wants_root = false;
};

}  // namespace

struct flx_run {
    hvec<flx_record> records;     // cigar_offset relative to this object's `cigars`
    hvec<u32> cigars;
    hvec<u8> skipped;
    hvec<flx_run> parts;          // a batch result is the in-order list of its slices (no concatenation on the host)
};

extern "C" void flx_params_default(flx_params* p) {
    memset(p, 0, sizeof(*p));
    p->query_error_probability = -1.0;
    p->pex_seed_num_errors = 2;
    p->search.max_num_anchors_hard = 500;
    p->search.max_num_anchors_soft = 50;
    p->search.anchor_group_order = FLX_ORDER_COUNT_FIRST;
    p->search.anchor_choice_strategy = FLX_CHOICE_ROUND_ROBIN;
    p->search.erase_useless_anchors = 1;
    p->seed_sampling_step_size = 1;
    p->extra_verification_ratio = 0.05;
    p->num_anchors_per_verification_task = 3000;
}

struct flx_reads {
    flx_ctx* ctx = nullptr;
    uint64_t n_reads = 0;
    hvec<u64> lens;            // per read
    hvec<u64> pool_off;        // per read: offset of the forward sequence; reverse complement follows at +len
    hvec<u8> pool;             // host copy (forward + reverse complement per read)
    hvec<u8> flags;            // per read: SEED_HAS_DELIM | SEED_NOT_ACGT (flx_fm_core.hpp) when it holds such symbols
    flx::DeviceBuffer d_pool;         // HBM-resident copy
    mutable flx::DeviceBuffer d_pack; // its 2-bit form (K1's presence filter), built with the Peq planes
    // Peq planes of the whole pool (K0), built by the first flx_align_reads_resident call on these reads and shared by all
    // lanes and later calls (they depend on the pool only)
    mutable std::mutex peq_mu;
    mutable bool peq_built = false;
    mutable flx::DeviceBuffer d_peq;
    mutable hipEvent_t peq_event = nullptr;      // recorded behind K0; every lane's stream waits for it before its first DP launch
    // --without-cigar aligns the reversed sequences (alignment.cpp:115-145): the reversed pool and its Peq planes, made by the first
    // chunk that needs them and shared like d_peq
    mutable bool rev_built = false;
    mutable flx::DeviceBuffer d_pool_rev, d_peq_rev;
    mutable hipEvent_t rev_event = nullptr;
};

namespace {
// a freed batch's buffer of this role, if the context keeps one (the largest): buf owns it afterwards
void take_spare_read_buffer(flx_ctx* ctx, int role, DeviceBuffer& buf) {
    if (buf.ptr) return;
    std::lock_guard<std::mutex> g(ctx->spare_mu);
    auto& v = ctx->spare_read_buffers[role];
    if (v.empty()) return;
    size_t best = 0;
    for (size_t i = 1; i < v.size(); ++i) if (v[i]->cap > v[best]->cap) best = i;
    buf.take(*v[best]);
    v.erase(v.begin() + (long)best);
}
void keep_spare_read_buffer(flx_ctx* ctx, int role, DeviceBuffer& buf) {
    if (!buf.ptr) return;
    {
        std::lock_guard<std::mutex> g(ctx->spare_mu);
        auto& v = ctx->spare_read_buffers[role];
        if (v.size() < 6) {                               // (as many batches as a caller keeps in flight, and a few)
            v.emplace_back(new DeviceBuffer());
            v.back()->take(buf);
            return;
        }
    }
    buf.release();
}
}  // namespace

extern "C" int flx_reads_upload(flx_ctx* ctx, const uint8_t* read_pool, const uint64_t* read_offsets, uint64_t n_reads, flx_reads** out) {
    if (!ctx || !out || (n_reads && (!read_pool || !read_offsets))) { set_error("flx_reads_upload: null argument"); return FLX_ERR_INVALID; }
    FLX_HIP(hipSetDevice(ctx->device));
    auto rd = std::make_unique<flx_reads>();
    rd->ctx = ctx;
    rd->n_reads = n_reads;
    rd->lens.resize(n_reads);
    rd->pool_off.resize(n_reads);
    rd->flags.assign(n_reads, 0);
    u64 total = 0;
    for (u64 i = 0; i < n_reads; ++i) {
        if (read_offsets[i + 1] < read_offsets[i]) { set_error("read offsets must be non-decreasing"); return FLX_ERR_INVALID; }
        rd->lens[i] = read_offsets[i + 1] - read_offsets[i];
        total += 2 * rd->lens[i];
    }
    rd->pool.resize(total);
    {
        u64 off = 0;
        for (u64 i = 0; i < n_reads; ++i) { rd->pool_off[i] = off; off += 2 * rd->lens[i]; }
    }
    // forward copy, reverse complement and symbol classes of every read, on several threads for large batches (a 16384-read batch is
    // 330 MB of pool: 80 ms on one thread, inside the clock of a caller that hands reads over in host memory)
    std::atomic<bool> bad_rank{false};
    auto fill = [&](u64 r0, u64 r1) {
        for (u64 i = r0; i < r1; ++i) {
            u64 const len = rd->lens[i], off = rd->pool_off[i];
            const u8* src = read_pool + read_offsets[i];
            u8 seen = 0;                                   // bit r: rank r occurs
            for (u64 b = 0; b < len; ++b) seen |= (u8)(1u << (src[b] < 6 ? src[b] : 7));
            if (seen & 0x80) { bad_rank = true; return; }
            rd->flags[i] = (u8)(((seen & 1) ? SEED_HAS_DELIM : 0) | ((seen & 0x21) ? SEED_NOT_ACGT : 0));
            memcpy(rd->pool.data() + off, src, len);
            reverse_complement(src, len, rd->pool.data() + off + len);
        }
    };
    unsigned const n_threads = total >= (8u << 20) ? std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency() / 2)) : 1u;
    if (n_threads <= 1) fill(0, n_reads);
    else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_threads; ++t) pool.emplace_back(fill, n_reads * t / n_threads, n_reads * (t + 1) / n_threads);
        for (auto& th : pool) th.join();
    }
    if (bad_rank) { set_error("read rank > 5"); return FLX_ERR_INVALID; }
    take_spare_read_buffer(ctx, 0, rd->d_pool);
    int rc = rd->d_pool.ensure(total + 256);
    if (rc) return rc;
    hipStream_t const s0 = ctx->external_stream ? ctx->lane0()->stream : ctx->upload_stream;
    if (total) FLX_HIP(hipMemcpyAsync(rd->d_pool.ptr, rd->pool.data(), total, hipMemcpyHostToDevice, s0));
    FLX_HIP(hipMemsetAsync((char*)rd->d_pool.ptr + total, 0, 192, s0));
    FLX_HIP(hipStreamSynchronize(s0));
    *out = rd.release();
    return FLX_OK;
}

extern "C" void flx_reads_free(flx_reads* reads) {
    if (!reads) return;
    if (reads->ctx) (void)hipSetDevice(reads->ctx->device);
    if (reads->ctx) {
        keep_spare_read_buffer(reads->ctx, 0, reads->d_pool);
        keep_spare_read_buffer(reads->ctx, 1, reads->d_pack);
        keep_spare_read_buffer(reads->ctx, 2, reads->d_peq);
        keep_spare_read_buffer(reads->ctx, 3, reads->d_pool_rev);
        keep_spare_read_buffer(reads->ctx, 4, reads->d_peq_rev);
    }
    reads->d_pool.release();
    reads->d_pack.release();
    reads->d_peq.release();
    reads->d_pool_rev.release();
    reads->d_peq_rev.release();
    if (reads->peq_event) (void)hipEventDestroy(reads->peq_event);
    if (reads->rev_event) (void)hipEventDestroy(reads->rev_event);
    delete reads;
}

extern "C" int flx_align_reads(flx_ctx* ctx, const flx_params* P, const uint8_t* read_pool, const uint64_t* read_offsets,
                               uint64_t n_reads, flx_run** out) {
    flx_reads* rd = nullptr;
    int rc = flx_reads_upload(ctx, read_pool, read_offsets, n_reads, &rd);
    if (rc) return rc;
    rc = flx_align_reads_resident(ctx, P, rd, out);
    flx_reads_free(rd);
    return rc;
}

namespace {

// one contiguous slice of the batch on one lane; produces the slice's records (read_index relative to the whole batch)
int align_slice(Lane* lane, const flx_params* P, const flx_reads* RD, u64 first_read, u64 end_read, flx_run* run) {
    flx_ctx* ctx = lane->ctx;
    FLX_HIP(hipSetDevice(ctx->device));
    HostIndex const& H = *ctx->hidx;
    hvec<u8> const& pool = RD->pool;
    PhaseTimer prof;

    // ---- reads -> PEX trees, seeds on the forward and reverse-complement sequence (parallelization.cpp:77-98)
    hvec<ReadState> reads;
    hvec<flx_seed> seeds;
    hvec<u8> seed_flags;
    std::map<std::pair<u64, u64>, std::unique_ptr<PexTree>> tree_cache;      // (length, errors) -> tree
    {
        // (the lists below grow to a seed per ~40 read bases: sized once instead of doubling their way up)
        u64 bases = 0;
        for (u64 i = first_read; i < end_read; ++i) bases += RD->lens[i];
        u64 const guess = 2 * (bases / 32 + (end_read - first_read)) / std::max<u64>(1, P->seed_sampling_step_size) + 64;
        (void)guess;
        reads.reserve(end_read - first_read);
    }
    for (u64 i = first_read; i < end_read; ++i) {
        u64 const len = RD->lens[i];
        if (len == 0 || len > 100000) { run->skipped[i] = 1; continue; }                       // input.cpp:95-110
        u64 const k = P->query_error_probability >= 0 ? fp_aware_ceil(len * P->query_error_probability) : P->query_num_errors;
        if (len <= k || k < P->pex_seed_num_errors) { run->skipped[i] = 1; continue; }         // input.cpp:115-129
        if (len > align_supported_max_query()) { set_error("read longer than the supported maximum"); return FLX_ERR_UNSUPPORTED; }
        ReadState rs;
        rs.read_index = i;
        rs.len = (u32)len;
        rs.k = (u32)k;
        {
            auto it = tree_cache.find(std::make_pair(len, k));
            if (it == tree_cache.end())
                it = tree_cache.emplace(std::make_pair(len, k), std::make_unique<PexTree>(build_pex_tree(len, k, P->pex_seed_num_errors, P->bottom_up_pex_tree_building != 0))).first;
            rs.tree_ptr = it->second.get();
        }
        rs.pool_off[0] = RD->pool_off[i];
        rs.pool_off[1] = RD->pool_off[i] + len;
        reads.push_back(std::move(rs));
    }
    // ---- the seeds: every step-th leaf of a read's tree, forward then reverse complement (pex.cpp:258-277). Seed s of the chunk =
    //      (read, orientation, leaf) by the reads' seed ranges: seed_first[r] .. seed_first[r + 1], n_sampled(r) per orientation.
    u64 const step = std::max<u64>(1, P->seed_sampling_step_size);
    auto n_sampled = [&](ReadState const& r) { return (u32)((r.tree_ref().leaves.size() + step - 1) / step); };
    hvec<u32> seed_first(reads.size() + 1, 0);
    for (size_t r = 0; r < reads.size(); ++r) seed_first[r + 1] = seed_first[r] + 2u * n_sampled(reads[r]);
    u64 const n_seeds_total = seed_first[reads.size()];
    auto build_host_seeds = [&]() {                                    // the list form (the host's selection, statistics, FLX_HOST_SEEDS=1)
        seeds.clear(); seed_flags.clear();
        seeds.reserve(n_seeds_total); seed_flags.reserve(n_seeds_total);
        for (size_t r = 0; r < reads.size(); ++r)
            for (int o = 0; o < 2; ++o)
                for (u64 l = 0; l < reads[r].tree_ref().leaves.size(); l += step) {
                    flx_pex_node const& leaf = reads[r].tree_ref().leaves[l];
                    seeds.push_back(flx_seed{reads[r].pool_off[o] + leaf.from, leaf.to - leaf.from + 1, leaf.num_errors, (u32)l, 0});
                    seed_flags.push_back(RD->flags[reads[r].read_index]);
                }
    };
    // the same as a description the device writes the seeds from: per tree its sampled leaves with their class (errors, length) and rank
    // within the class, per read where its seeds of each class start in launch order (heaviest class first: more errors, then shorter)
    SeedGen gen;
    bool const use_gen = !getenv("FLX_HOST_SEEDS") && n_seeds_total > 0 && n_seeds_total < (1ull << 31);
    if (use_gen) {
        struct TreePlan { u32 leaf_first; hvec<u32> class_key, class_count; };
        std::map<const PexTree*, TreePlan> plans;
        struct GlobalClass { u64 pos = 0; u32 scheme_off = 0, nsearch = 0; };
        std::map<u32, GlobalClass> global;                             // class key -> seeds of the class in the chunk, then its next launch position; its scheme
        for (auto const& rs : reads) {
            auto it = plans.find(rs.tree_ptr);
            if (it == plans.end()) {
                TreePlan tp;
                tp.leaf_first = (u32)gen.leaves.size();
                for (u64 l = 0; l < rs.tree_ref().leaves.size(); l += step) {
                    flx_pex_node const& leaf = rs.tree_ref().leaves[l];
                    u32 const length = leaf.to - leaf.from + 1, key = ((3u - std::min<u32>(leaf.num_errors, 3u)) << 24) | length;
                    size_t c = 0;
                    while (c < tp.class_key.size() && tp.class_key[c] != key) ++c;
                    if (c == tp.class_key.size()) { tp.class_key.push_back(key); tp.class_count.push_back(0); }
                    gen.leaves.push_back(DevSeedLeaf{leaf.from, length, (u32)c, tp.class_count[c]++});
                    gen.max_errors = std::max(gen.max_errors, leaf.num_errors);
                    gen.max_length = std::max(gen.max_length, length);
                }
                it = plans.emplace(rs.tree_ptr, std::move(tp)).first;
            }
            for (size_t c = 0; c < it->second.class_key.size(); ++c) global[it->second.class_key[c]].pos += 2ull * it->second.class_count[c];
        }
        if (gen.max_errors > 3) { set_error("seed errors must be in [0,3] (floxer_cli.cpp:299)"); return FLX_ERR_INVALID; }
        u64 pos = 0;
        for (auto& kv : global) {                                      // ascending key = heaviest class first
            u32 const errors = 3u - (kv.first >> 24), length = kv.first & 0xFFFFFFu;
            auto const e = expanded_scheme(errors, length);
            kv.second.scheme_off = (u32)gen.scheme_table.size();
            kv.second.nsearch = e.empty() ? 0 : (u32)(e.size() / length);
            gen.scheme_table.insert(gen.scheme_table.end(), e.begin(), e.end());
            u64 const n = kv.second.pos;
            kv.second.pos = pos;
            pos += n;
        }
        gen.reads.reserve(reads.size());
        for (size_t r = 0; r < reads.size(); ++r) {
            ReadState const& rs = reads[r];
            TreePlan const& tp = plans.find(rs.tree_ptr)->second;
            gen.reads.push_back(DevSeedRead{rs.pool_off[0], rs.pool_off[1], tp.leaf_first, n_sampled(rs), seed_first[r], (u32)gen.classes.size(), RD->flags[rs.read_index], 0});
            for (size_t c = 0; c < tp.class_key.size(); ++c) {
                auto& g = global[tp.class_key[c]];
                u32 const errors = 3u - (tp.class_key[c] >> 24), length = tp.class_key[c] & 0xFFFFFFu;
                gen.classes.push_back(DevSeedClass{(u32)g.pos, tp.class_count[c], g.scheme_off, (length + errors + 3) | (g.nsearch << 24)});
                g.pos += 2ull * tp.class_count[c];
            }
        }
        gen.n_seeds = n_seeds_total;
    } else build_host_seeds();

    int rc;
    const u8* d_pool = RD->d_pool.as<u8>();
    prof.mark("pex+seeds");
    // statistics in the reference's form (flx_stats.cpp), when the context has a statistics object attached
    std::unique_ptr<Stats> st_local;
    if (ctx->read_stats) st_local = std::make_unique<Stats>(stats_simulated(ctx->read_stats));
    auto const t_slice = std::chrono::steady_clock::now();


    // ---- seeding
    hvec<HostAnchor> anchors;
    hvec<SeedStats> sstats;
    // (K1 starts behind K0 and the pool's 2-bit form: the event is recorded when the first call on these reads has queued both)
    FLX_HIP(hipStreamWaitEvent(lane->stream, RD->peq_event, 0));
    rc = use_gen ? search_seeds_device(lane, d_pool, pool.data(), pool.size(), nullptr, 0, P->search, anchors, sstats, nullptr, 0,
                                       RD->d_pack.ptr ? RD->d_pack.as<u32>() : nullptr, nullptr, &gen)
                 : SEARCH_NEEDS_HOST_SEEDS;
    if (rc == SEARCH_NEEDS_HOST_SEEDS) {
        if (use_gen) build_host_seeds();
        rc = search_seeds_device(lane, d_pool, pool.data(), pool.size(), seeds.data(), seeds.size(), P->search, anchors, sstats, nullptr, 0,
                                 RD->d_pack.ptr ? RD->d_pack.as<u32>() : nullptr, seed_flags.data());
    }
    if (rc) return rc;

    prof.mark("search");
    double const search_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_slice).count();
    if (st_local) {
        // per query: its length, its seeds (forward then reverse complement: one contiguous run of the seed list) and their
        // selection counters (statistics.cpp:283-295, 367-419)
        hvec<SeedStatRow> rows;
        for (size_t r = 0; r < reads.size(); ++r) {
            st_local->at(Stats::QUERY_LENGTHS).add(reads[r].len);
            rows.clear();
            u32 const nl = n_sampled(reads[r]);
            for (u32 si = seed_first[r]; si < seed_first[r + 1]; ++si) {
                flx_pex_node const& leaf = reads[r].tree_ref().leaves[(u64)((si - seed_first[r]) % nl) * step];
                st_local->at(Stats::ERRORS_PER_SEED).add(leaf.num_errors);
                st_local->at(Stats::SEED_LENGTHS).add(leaf.to - leaf.from + 1);
                rows.push_back(SeedStatRow{sstats[si].useful, sstats[si].raw, sstats[si].excluded_soft});
            }
            st_local->at(Stats::SEEDS_PER_QUERY).add(rows.size());
            st_local->add_search_result(rows.data(), rows.size());
            st_local->at(Stats::MS_SEARCH).add((u64)(search_ms / (double)std::max<size_t>(1, reads.size())));
        }
    }
    hvec<AnchorState> A(anchors.size());
    u32 rd_i = 0;
    for (size_t a = 0; a < anchors.size(); ++a) {
        // the anchor's seed -> (read, orientation, leaf); the anchors come seed by seed, so the read mostly stays or moves on by one
        u32 const si = anchors[a].seed_index;
        if (si < seed_first[rd_i] || si >= seed_first[rd_i + 1]) {
            if (si >= seed_first[rd_i + 1] && rd_i + 2 < seed_first.size() && si < seed_first[rd_i + 2]) ++rd_i;
            else rd_i = (u32)(std::upper_bound(seed_first.begin(), seed_first.end(), si) - seed_first.begin() - 1);
        }
        u32 const nl = n_sampled(reads[rd_i]), local = si - seed_first[rd_i];
        struct { u32 read; u8 orientation; } const so{rd_i, (u8)(local >= nl ? 1 : 0)};
        A[a].read = so.read;
        A[a].orientation = so.orientation;
        A[a].leaf = (u32)((u64)(local - (so.orientation ? nl : 0u)) * step);
        A[a].ref_id = anchors[a].ref_id;
        A[a].pos = anchors[a].pos;
        reads[so.read].anchor_ids[so.orientation].push_back((u32)a);
    }

    // ---- verification order of each read: packages (forward then reverse complement, <= N anchors each) in the order one
    //      worker would run them (parallelization.cpp:14-43, 230)
    hvec<hvec<u32>> exec_order(reads.size());
    for (size_t r = 0; r < reads.size(); ++r) {
        hvec<std::pair<u32, u32>> pkgs;          // (first, count) into a concatenated list
        hvec<u32> concat;
        for (int o = 0; o < 2; ++o) {
            auto const& ids = reads[r].anchor_ids[o];
            for (size_t i = 0; i < ids.size(); i += P->num_anchors_per_verification_task) {
                u32 const cnt = (u32)std::min<size_t>(P->num_anchors_per_verification_task, ids.size() - i);
                pkgs.emplace_back((u32)concat.size(), cnt);
                concat.insert(concat.end(), ids.begin() + i, ids.begin() + i + cnt);
            }
        }
        for (int pid : package_order((int)pkgs.size()))
            for (u32 j = 0; j < pkgs[pid].second; ++j) exec_order[r].push_back(concat[pkgs[pid].first + j]);
    }

    prof.mark("anchors+order");
    // ---- Peq planes of the whole pool
    const u64* const d_peq = RD->d_peq.as<u64>();              // built once per resident read set (flx_align_reads_resident)
    FLX_HIP(hipStreamWaitEvent(lane->stream, RD->peq_event, 0));
    const u8* d_text = ctx->didx.text;

    auto window_request = [&](AnchorState const& a, flx_pex_node const& node, double ratio, Span* span_out) {
        ReadState const& rs = reads[a.read];
        flx_pex_node const& leaf = rs.tree_ref().leaves[a.leaf];
        Span const sp = compute_span(a.pos, node, leaf.from, H.seq_len[a.ref_id], ratio);
        if (span_out) *span_out = sp;
        return AlignRequest{H.seq_start[a.ref_id] + sp.offset, rs.pool_off[a.orientation] + node.from, (u32)sp.length,
                            node.to - node.from + 1, node.num_errors};
    };

    // ---- hierarchical verification, level-synchronous (verification.cpp:44-117): inner nodes only test existence and do
    //      not depend on the interval cache, so all anchors climb together; an anchor stops at its first failing node.
    for (auto& a : A) {
        ReadState const& rs = reads[a.read];
        flx_pex_node const& leaf = rs.tree_ref().leaves[a.leaf];
        if (P->direct_full_verification || leaf.parent_id == FLX_NULL_ID) { a.at_root = true; continue; }   // verification.cpp:23-42, 52-72
        a.node = leaf.parent_id;
        if (rs.tree_ref().inner[a.node].parent_id == FLX_NULL_ID) a.at_root = true;
    }
    // Anchors do not wait for each other and their tests do not depend on any order, so a round tests the anchors whose
    // current node is in the smallest size class still pending (PEX trees are unbalanced: the same node is reached after a
    // different number of steps from different leaves). All tests of a node size then share one launch, and identical
    // (window, node) tests requested by anchors that started at different depths are found by the de-duplication.
    u64 n_inner_requested = 0;
    hvec<u32> climbing, selected, waiting, survivors;    // anchors that still have an inner node to test (in anchor order)
    double g_build_ms = 0;
    for (u32 ai = 0; ai < A.size(); ++ai) if (A[ai].alive && !A[ai].at_root) climbing.push_back(ai);
    // The rounds run with the anchors' state resident on the device (requests, de-duplication, clusters and the moves up the trees
    // are kernels; the host launches K3 on each round's job list and decides per cluster). FLX_HOST_ROUNDS=1, or a statistics
    // object on the context (it wants every request's window), selects the host form below; both give the same records.
    static int const host_rounds = getenv("FLX_HOST_ROUNDS") ? 1 : 0;
    if (!host_rounds && !st_local && !climbing.empty()) {
        u32 const n = (u32)A.size();
        PhaseTimer vprof("rounds");
        // ---- node table of the chunk's trees, anchors, the anchors of every query (read x orientation: contiguous, the anchors are in
        //      seed order)
        std::map<const PexTree*, u32> tree_base;
        hvec<DevVrNode> nodes;
        for (auto const& kv : tree_cache) {
            PexTree const& t = *kv.second;
            tree_base[&t] = (u32)nodes.size();
            for (auto const& nd : t.inner) nodes.push_back(DevVrNode{nd.parent_id, nd.from, nd.to - nd.from + 1, nd.num_errors});
        }
        if (nodes.empty()) nodes.push_back(DevVrNode{0xFFFFFFFFu, 0, 1, 0});
        u32 const n_queries = (u32)(2 * reads.size());
        hvec<u32> q_first(n_queries + 1, 0);
        hvec<DevVrAnchor> da(n);
        hvec<u32> h_node(n);
        hvec<u8> h_status(n);
        u32 n_climbing = 0, smallest = 0xFFFFFFFFu;
        for (u32 i = 0; i < n; ++i) {
            AnchorState const& a = A[i];
            ReadState const& rs = reads[a.read];
            flx_pex_node const& leaf = rs.tree_ref().leaves[a.leaf];
            u32 const tb = tree_base[rs.tree_ptr];
            u32 const query = 2u * a.read + a.orientation;
            q_first[query + 1]++;
            da[i] = DevVrAnchor{(i64)a.pos - (i64)leaf.from, H.seq_start[a.ref_id], H.seq_len[a.ref_id], rs.pool_off[a.orientation], tb, query};
            bool const climbs = a.alive && !a.at_root;
            h_node[i] = climbs ? a.node : 0u;
            h_status[i] = climbs ? VR_CLIMBING : a.at_root ? VR_AT_ROOT : VR_DEAD;
            if (climbs) { ++n_climbing; smallest = std::min(smallest, nodes[tb + a.node].rows); }
            if (i > 0 && 2u * A[i - 1].read + A[i - 1].orientation > query) { set_error("verification rounds: anchors out of query order"); return FLX_ERR_INTERNAL; }
        }
        for (u32 qi = 0; qi < n_queries; ++qi) q_first[qi + 1] += q_first[qi];
        // ---- one device buffer cut into the arrays of Vr2Buffers
        size_t off = 0;
        auto take = [&](size_t bytes) { size_t const at = off; off += (bytes + 255) & ~(size_t)255; return at; };
        size_t const o_anchors = take((size_t)n * sizeof(DevVrAnchor)), o_nodes = take(nodes.size() * sizeof(DevVrNode)), o_qfirst = take(((size_t)n_queries + 1) * 4),
                     o_node = take((size_t)n * 4), o_status = take(n), o_slot = take((size_t)n * 4), o_jobs = take((size_t)n * 2 * sizeof(DevAlignJob)),
                     o_outs = take((size_t)n * 2 * sizeof(DevAlignOut)), o_scalars = take(VR2_SCALARS * 4);
        vprof.mark("anchor-table");
        if ((rc = lane->vr.ensure(off))) return rc;
        char* const base = (char*)lane->vr.ptr;
        Vr2Buffers B{};
        B.anchors = (const DevVrAnchor*)(base + o_anchors); B.nodes = (const DevVrNode*)(base + o_nodes); B.q_first = (const u32*)(base + o_qfirst);
        B.node = (u32*)(base + o_node); B.status = (u8*)(base + o_status); B.a_slot = (u32*)(base + o_slot);
        B.jobs = (DevAlignJob*)(base + o_jobs); B.outs = (DevAlignOut*)(base + o_outs); B.scalars = (u32*)(base + o_scalars);
        FLX_HIP(hipMemcpyAsync(base + o_anchors, da.data(), (size_t)n * sizeof(DevVrAnchor), hipMemcpyHostToDevice, lane->stream));
        FLX_HIP(hipMemcpyAsync(base + o_nodes, nodes.data(), nodes.size() * sizeof(DevVrNode), hipMemcpyHostToDevice, lane->stream));
        FLX_HIP(hipMemcpyAsync(base + o_qfirst, q_first.data(), ((size_t)n_queries + 1) * 4, hipMemcpyHostToDevice, lane->stream));
        FLX_HIP(hipMemcpyAsync(base + o_node, h_node.data(), (size_t)n * 4, hipMemcpyHostToDevice, lane->stream));
        FLX_HIP(hipMemcpyAsync(base + o_status, h_status.data(), n, hipMemcpyHostToDevice, lane->stream));
        FLX_HIP(hipMemsetAsync(base + o_slot, 0xFF, (size_t)n * 4, lane->stream));
        FLX_HIP(hipMemsetAsync(B.scalars, 0, VR2_SCALARS * 4, lane->stream));
        if (!lane->vr_host_scalars) FLX_HIP(hipHostMalloc((void**)&lane->vr_host_scalars, VR2_SCALARS * 4, hipHostMallocMapped));
        vprof.mark("upload");
        u64 const few_waves = align_few_waves();
        u64 prev_jobs = n_climbing / 2, acc_steps = 0, acc_bytes = 0, acc_req = 0;
        unsigned long long* lane_stats = nullptr;               // FLX_ALIGN_DEBUG: the lane-per-job kernel's counters, per round
        if (getenv("FLX_ALIGN_DEBUG")) {
            if ((rc = lane->counters.ensure(128))) return rc;
            lane_stats = (unsigned long long*)((char*)lane->counters.ptr + 64);
        }
        // ---- The rounds whose size classes are known in advance (lane-per-job existence kernel: no launch shape to choose from the last
        //      round's job count) are queued back to back, two per class (the members of a union window without an alignment are tested
        //      alone in the round after it): a round reads what the round before it left on the device, one that finds nothing to ask for
        //      returns at once, and the host waits once, behind the last. What is still climbing then goes through the loop below.
        u32 round = 0;
        if (exists_lane_form() && !lane_stats && getenv("FLX_ROUNDS_QUEUED") && n_climbing > 0) {      // (measured: 130-136 k reads/s queued, 135-142 k with a wait per round)
            struct RoundClass { u64 limit; u32 nw_max; i64 width_max; };
            hvec<RoundClass> plan;
            {
                hvec<std::pair<u32, u32>> sizes;                       // (rows, errors) of every inner node below a root
                for (auto const& nd : nodes) if (nd.parent != 0xFFFFFFFFu) sizes.emplace_back(nd.rows, nd.errors);
                std::sort(sizes.begin(), sizes.end());
                size_t i = 0;
                while (i < sizes.size() && sizes[i].first < smallest) ++i;
                while (i < sizes.size()) {
                    RoundClass c{(u64)sizes[i].first * round_span_percent() / 100, 0, 0};
                    for (; i < sizes.size() && sizes[i].first <= c.limit; ++i) {
                        c.nw_max = std::max(c.nw_max, (sizes[i].first + 63u) / 64u);
                        c.width_max = std::max<i64>(c.width_max, 4 * (i64)sizes[i].second + 1);
                    }
                    plan.push_back(c);
                    plan.push_back(c);
                }
            }
            bool queued = !plan.empty();
            u32 const max_jobs = (u32)std::min<u64>(2ull * n_climbing, 2ull * n);
            for (size_t r = 0; queued && r < plan.size(); ++r) {
                RoundClass const& c = plan[r];
                AlignShape const shape = DeviceApi::shape_holding(c.nw_max, c.width_max, false);
                u32 lane_waves = 0, lane_cap = 0;
                u64 const cap = shape.words_per_lane == 0 ? 0 : DeviceApi::shape_width_cap(c.nw_max, shape);
                u64 const width_cap = std::min<u64>(cap, std::max<u64>(8 * (u64)c.width_max, 1024));
                if (shape.words_per_lane == 0 || !shape.banded || !exists_lane_setup(max_jobs, (i64)std::max<u64>(width_cap, (u64)c.width_max), lane_waves, lane_cap)) {
                    queued = false;                                    // (a class the lane form does not take: the loop below from here on)
                    break;
                }
                if (r == 0) FLX_HIP(hipMemsetAsync(B.scalars + VR2_PENDING, 1, 4, lane->stream));      // (non-zero: the first queued round always runs)
                int const e1 = DeviceApi::vr2_request(lane->stream, B, n_queries, (u32)std::min<u64>(c.limit, 0xFFFFFFFFu), shape.words_per_lane,
                                                      (u32)std::min<u64>(width_cap, 0xFFFFFFFFull), round, true);
                if (e1) { set_error(std::string("verification round: ") + hipGetErrorString((hipError_t)e1)); return FLX_ERR_NO_DEVICE; }
                rc = timed_launch(lane, "ed_align_exists", 0, 0, [&] {
                    return DeviceApi::align_exists_lanes(lane->stream, d_text, d_peq, B.jobs, max_jobs, B.scalars + VR2_N_JOBS + (round & 1u), B.scalars + VR2_QUEUE,
                                                         lane_waves, lane_cap, B.outs, nullptr);
                });
                if (rc) return rc;
                u64 const next_limit = r + 1 < plan.size() ? plan[r + 1].limit : 0xFFFFFFFFull;
                int const e2 = DeviceApi::vr2_apply(lane->stream, B, n, lane->vr_host_scalars, (u32)std::min<u64>(next_limit, 0xFFFFFFFFu), round);
                if (e2) { set_error(std::string("verification round: ") + hipGetErrorString((hipError_t)e2)); return FLX_ERR_NO_DEVICE; }
                ++round;
            }
            if (round > 0) {
                if ((rc = lane->sync())) return rc;
                u32 sc[VR2_SCALARS];
                memcpy(sc, lane->vr_host_scalars, sizeof(sc));
                if (sc[VR2_QUEUE_ERR]) { set_error("existence tests: a window did not fit the row buffers"); return FLX_ERR_INTERNAL; }
                u64 ws, by;
                memcpy(&ws, &sc[VR2_WORD_STEPS], 8);
                memcpy(&by, &sc[VR2_BYTES], 8);
                if (ctx->timing) {
                    std::lock_guard<std::mutex> g(ctx->mu);
                    auto it = ctx->stats.find("ed_align_exists");
                    if (it != ctx->stats.end()) { it->second.algorithmic_bytes += by; it->second.work_units += ws; }
                }
                n_inner_requested += sc[VR2_N_REQ];
                acc_steps = ws; acc_bytes = by; acc_req = sc[VR2_N_REQ];
                prev_jobs = sc[VR2_N_JOBS + ((round - 1u) & 1u)];
                n_climbing = sc[VR2_N_CLIMBING];
                smallest = sc[VR2_SMALLEST];
                vprof.mark("round");
            }
        }
        for (; n_climbing > 0; ++round) {
            u64 const limit = (u64)smallest * round_span_percent() / 100;
            // One launch shape for the round: the cheapest that holds the window of every node in the round's size class, or the one
            // with the fewest words per lane when the round has few jobs (they would leave most SIMDs without a wave; the last round's
            // job count is the estimate: either shape holds every job)
            u32 nw_max = 0;
            i64 width_max = 0;
            for (auto const& nd : nodes)
                if (nd.rows >= smallest && nd.rows <= limit) {
                    nw_max = std::max(nw_max, (nd.rows + 63u) / 64u);
                    width_max = std::max<i64>(width_max, 4 * (i64)nd.errors + 1);            // a window of its own: n - m + 2k = (2e + 1) + 2e
                }
            AlignShape const shape_t = DeviceApi::shape_holding(nw_max, width_max, false), shape_p = DeviceApi::shape_holding(nw_max, width_max, true);
            if (shape_t.words_per_lane == 0 || shape_p.words_per_lane == 0) { set_error("query longer than the supported maximum"); return FLX_ERR_UNSUPPORTED; }
            AlignShape const shape = prev_jobs * shape_t.lanes_per_job / 64 >= few_waves ? shape_t : shape_p;
            // what the shape holds beyond that goes to the clusters' union windows (a shape holds a job when every word group has a lane of
            // its own, when the ring's lanes are free again before their next group starts: 64 W (R - 1) + R + 1 > diagonals, or when the
            // steps a revolution of the ring has to wait fit the launch's hand-over slots: flx_internal.hpp, ring_delay)
            u64 const cap = DeviceApi::shape_width_cap(nw_max, shape);
            // (the lane-per-job kernel holds any window its row buffers hold: unions up to the ring shape's cap, or eight windows' width)
            u32 lane_waves = 0, lane_cap = 0;
            u32 const max_jobs = (u32)std::min<u64>(2ull * n_climbing, 2ull * n);
            u64 const lane_width_cap = std::min<u64>(cap, std::max<u64>(8 * (u64)width_max, 1024));
            u32 const team = exists_team_size(smallest);
            bool const lane_form = exists_lane_form() && shape.banded && exists_lane_setup(max_jobs, (i64)std::max<u64>(lane_width_cap, (u64)width_max), lane_waves, lane_cap, team);
            u64 const width_cap = lane_form ? lane_width_cap : cap;
            int const e1 = DeviceApi::vr2_request(lane->stream, B, n_queries, (u32)std::min<u64>(limit, 0xFFFFFFFFu), shape.words_per_lane,
                                                  (u32)std::min<u64>(width_cap, 0xFFFFFFFFull), round);
            if (e1) { set_error(std::string("verification round: ") + hipGetErrorString((hipError_t)e1)); return FLX_ERR_NO_DEVICE; }
            rc = timed_launch(lane, "ed_align_exists", 0, 0, [&] {
                if (lane_form)
                    return DeviceApi::align_exists_lanes(lane->stream, d_text, d_peq, B.jobs, max_jobs, B.scalars + VR2_N_JOBS + (round & 1u), B.scalars + VR2_QUEUE,
                                                         lane_waves, lane_cap, B.outs, lane_stats, team);
                // (a fixed grid of at most this many waves takes the round's job groups in turn; FLX_EXISTS_MAX_WAVES: how much of the chip one
                // round's launch may hold while the other lanes' kernels want room)
                static u32 const exists_waves = [] { const char* e = getenv("FLX_EXISTS_MAX_WAVES"); return (u32)(e ? std::max(64, atoi(e)) : 8192); }();
                return DeviceApi::align_exists_counted(lane->stream, d_text, d_peq, B.jobs, max_jobs, B.scalars + VR2_N_JOBS + (round & 1u), shape, exists_waves, B.outs, B.scalars + VR2_QUEUE_ERR);
            });
            if (rc) return rc;
            int const e2 = DeviceApi::vr2_apply(lane->stream, B, n, lane->vr_host_scalars);
            if (e2) { set_error(std::string("verification round: ") + hipGetErrorString((hipError_t)e2)); return FLX_ERR_NO_DEVICE; }
            if ((rc = lane->sync())) return rc;
            u32 sc[VR2_SCALARS];
            memcpy(sc, lane->vr_host_scalars, sizeof(sc));            // (left there by the last block of vr2_apply)
            if (sc[VR2_QUEUE_ERR]) { set_error("existence tests: a window did not fit the row buffers"); return FLX_ERR_INTERNAL; }
            u64 ws, by;
            memcpy(&ws, &sc[VR2_WORD_STEPS], 8);
            memcpy(&by, &sc[VR2_BYTES], 8);
            if (ctx->timing) {        // the round's word-steps and sequence bytes were counted on the device: fold them into the kernel's accounting
                std::lock_guard<std::mutex> g(ctx->mu);
                auto it = ctx->stats.find("ed_align_exists");
                if (it != ctx->stats.end()) { it->second.algorithmic_bytes += by - acc_bytes; it->second.work_units += ws - acc_steps; }
            }
            u64 const round_req = sc[VR2_N_REQ] - acc_req;
            u64 const round_ws = ws - acc_steps;
            acc_steps = ws; acc_bytes = by; acc_req = sc[VR2_N_REQ];
            n_inner_requested += round_req;
            if (round_req == 0 && sc[VR2_N_CLIMBING] >= n_climbing) { set_error("verification rounds do not advance"); return FLX_ERR_INTERNAL; }
            prev_jobs = sc[VR2_N_JOBS + (round & 1u)];
            if (lane_stats) {
                unsigned long long st[8];
                FLX_HIP(hipMemcpy(st, lane_stats, sizeof(st), hipMemcpyDeviceToHost));
                FLX_HIP(hipMemset(lane_stats, 0, sizeof(st)));
                fprintf(stderr, "[exists round %u] rows %u..%llu jobs %llu waves %u cap %u: blocks %llu, wave iterations %llu (x64 = %llu), lane iterations %llu, groups %llu; word-steps %llu\n", round, smallest,
                        (unsigned long long)limit, (unsigned long long)prev_jobs, lane_waves, lane_cap, st[0], st[1], st[1] * 64, st[2], st[3], (unsigned long long)round_ws);
            }
            n_climbing = sc[VR2_N_CLIMBING];
            smallest = sc[VR2_SMALLEST];
            vprof.mark("round");
        }
        if ((rc = d2h(lane, h_status.data(), B.status, n))) return rc;
        if ((rc = d2h(lane, h_node.data(), B.node, (size_t)n * 4))) return rc;
        if ((rc = lane->sync())) return rc;
        for (u32 i = 0; i < n; ++i) {
            AnchorState& a = A[i];
            if (h_status[i] == VR_DEAD && a.alive && !a.at_root) a.alive = false;
            else if (h_status[i] == VR_AT_ROOT && !a.at_root) { a.at_root = true; a.node = h_node[i]; }
        }
        vprof.mark("read-back");
    } else {
    hvec<AlignRequest> reqs;
        hvec<DevAlignOut> outs;
        auto rows_of = [&](AnchorState const& a) { flx_pex_node const& nd = reads[a.read].tree_ref().inner[a.node]; return nd.to - nd.from + 1; };
        // `climbing` carries each anchor's node size next to its index (the rounds scan it): {anchor, rows}
        struct Climber { u32 anchor, rows; };
        hvec<Climber> climbers, sel, wait, surv;
        u32 smallest = 0xFFFFFFFFu;
        climbers.reserve(climbing.size());
        for (u32 ai : climbing) { u32 const r = rows_of(A[ai]); climbers.push_back(Climber{ai, r}); smallest = std::min(smallest, r); }
        while (!climbers.empty()) {
            u64 const limit = (u64)smallest * round_span_percent() / 100;
            auto const tb0 = std::chrono::steady_clock::now();
            sel.clear();
            wait.clear();
            surv.clear();
            reqs.clear();
            u32 next_smallest = 0xFFFFFFFFu;
            for (Climber const& c : climbers) {                  // both parts stay in anchor order
                if (c.rows <= limit) {
                    sel.push_back(c);
                    reqs.push_back(window_request(A[c.anchor], reads[A[c.anchor].read].tree_ref().inner[A[c.anchor].node], 0.0, nullptr));
                } else { wait.push_back(c); next_smallest = std::min(next_smallest, c.rows); }
            }
            g_build_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count();
            n_inner_requested += reqs.size();
            // (statistics: the inner tests are counted per anchor in the interval pass below - under -I the reference never starts on an
            // anchor whose root window is already verified, verification.cpp:45)
            if ((rc = run_exists_jobs(lane, d_text, d_peq, reqs, outs))) return rc;
            for (size_t i = 0; i < outs.size(); ++i) {
                AnchorState& a = A[sel[i].anchor];
                if (outs[i].score == 0xFFFFFFFFu) { a.alive = false; continue; }
                a.node = reads[a.read].tree_ref().inner[a.node].parent_id;
                if (reads[a.read].tree_ref().inner[a.node].parent_id == FLX_NULL_ID) a.at_root = true;
                else { u32 const r = rows_of(a); surv.push_back(Climber{sel[i].anchor, r}); next_smallest = std::min(next_smallest, r); }
            }
            climbers.resize(wait.size() + surv.size());
            std::merge(wait.begin(), wait.end(), surv.begin(), surv.end(), climbers.begin(), [](Climber const& x, Climber const& y) { return x.anchor < y.anchor; });
            smallest = next_smallest;
        }

    }
    prof.mark("inner-levels");
    if (prof.on) {
        fprintf(stderr, "[flx host profile] exists rounds: dedup=%.2f cluster=%.2f gpu-round-trip=%.2f scatter=%.2f build-requests=%.2f ms\n",
                g_exists_ms[0], g_exists_ms[1], g_exists_ms[2], g_exists_ms[3], g_build_ms);
        g_exists_ms[0] = g_exists_ms[1] = g_exists_ms[2] = g_exists_ms[3] = 0;
    }
    // ---- interval pass in verification order (verification.cpp:45, 106-109, 119-136): decides which anchors align the root
    hvec<AlignRequest> root_reqs;
    hvec<u32> root_anchor;
    hvec<Span> root_spans;
    // statistics: the inner nodes an anchor tested (verification.cpp:241) - its leaf's parent, upwards, to the node it failed at or to
    // the node below the root
    auto add_inner_spans = [&](AnchorState const& a) {
        auto const& tree = reads[a.read].tree_ref();
        flx_pex_node const& leaf = tree.leaves[a.leaf];
        if (P->direct_full_verification || leaf.parent_id == FLX_NULL_ID) return;
        for (u32 nd = leaf.parent_id; tree.inner[nd].parent_id != FLX_NULL_ID; nd = tree.inner[nd].parent_id) {
            st_local->at(Stats::SPAN_INNER).add(window_request(a, tree.inner[nd], 0.0, nullptr).n);
            if (!a.alive && nd == a.node) break;
        }
    };
    for (size_t r = 0; r < reads.size(); ++r) {
        hvec<VerifiedIntervals> cache[2];
        if (P->use_interval_optimization) { cache[0].resize(H.seq_len.size()); cache[1].resize(H.seq_len.size()); }
        for (u32 ai : exec_order[r]) {
            AnchorState& a = A[ai];
            ReadState const& rs = reads[a.read];
            Span sp;
            AlignRequest const req = window_request(a, rs.tree_ref().root(), P->extra_verification_ratio, &sp);
            if (P->use_interval_optimization) {
                auto& ivs = cache[a.orientation][a.ref_id];
                if (ivs.contains(trim_both({sp.offset, sp.offset + sp.length}, sp.extra))) {           // root_was_already_verified
                    if (st_local) st_local->at(Stats::SPAN_ROOT_AVOIDED).add(sp.length);                // verification.cpp:130
                    continue;
                }
                if (st_local) add_inner_spans(a);
                if (!(a.alive && a.at_root)) continue;
                ivs.insert({sp.offset, sp.offset + sp.length});
            } else {
                if (st_local) add_inner_spans(a);
                if (!(a.alive && a.at_root)) continue;
            }
            a.wants_root = true;
            if (st_local) st_local->at(Stats::SPAN_ROOT).add(sp.length);                               // verification.cpp:239
            root_reqs.push_back(req);
            root_anchor.push_back(ai);
            root_spans.push_back(sp);
        }
    }

    prof.mark("interval-pass");
    // ---- root alignments (alignment.cpp:115-180)
    struct RootAlignment { bool exists = false; u64 start = 0; u32 nm = 0; u64 cigar_off = 0; u32 cigar_len = 0; };
    hvec<RootAlignment> root_res(root_reqs.size());
    hvec<u32> cig;
    if (P->without_cigar) {
        if ((rc = ensure_reversed_text(lane))) return rc;
        {
            std::lock_guard<std::mutex> g(RD->peq_mu);
            if (!RD->rev_built) {
                hvec<u8> qrev(pool.rbegin(), pool.rend());
                take_spare_read_buffer(ctx, 3, RD->d_pool_rev);
                take_spare_read_buffer(ctx, 4, RD->d_peq_rev);
                if ((rc = RD->d_pool_rev.ensure(qrev.size() + 256))) return rc;
                FLX_HIP(hipMemcpyAsync(RD->d_pool_rev.ptr, qrev.data(), qrev.size(), hipMemcpyHostToDevice, lane->stream));
                FLX_HIP(hipMemsetAsync((char*)RD->d_pool_rev.ptr + qrev.size(), 0, 192, lane->stream));
                if ((rc = build_peq(lane, RD->d_pool_rev.as<u8>(), qrev.size(), RD->d_peq_rev))) return rc;
                if (!RD->rev_event) FLX_HIP(hipEventCreateWithFlags(&RD->rev_event, hipEventDisableTiming));
                FLX_HIP(hipEventRecord(RD->rev_event, lane->stream));
                FLX_HIP(hipStreamSynchronize(lane->stream));        // qrev leaves scope
                RD->rev_built = true;
            }
        }
        FLX_HIP(hipStreamWaitEvent(lane->stream, RD->rev_event, 0));
        hvec<AlignRequest> rev(root_reqs.size());
        for (size_t i = 0; i < rev.size(); ++i)
            rev[i] = AlignRequest{H.n - root_reqs[i].ref_off - root_reqs[i].n, pool.size() - root_reqs[i].q_off - root_reqs[i].m,
                                  root_reqs[i].n, root_reqs[i].m, root_reqs[i].k};
        hvec<DevAlignOut> outs;
        if ((rc = run_score_jobs(lane, ctx->text_rev.as<u8>() + TEXT_PAD, RD->d_peq_rev.as<u64>(), rev, outs, "ed_align_exists"))) return rc;
        for (size_t i = 0; i < outs.size(); ++i)
            if (outs[i].score != 0xFFFFFFFFu) { root_res[i].exists = true; root_res[i].nm = outs[i].score; root_res[i].start = root_spans[i].offset + (root_reqs[i].n - outs[i].end_col); }
    } else {
        hvec<TraceResult> tres;
        if ((rc = run_trace_jobs_union(lane, d_text, d_pool, d_peq, root_reqs, tres, cig))) return rc;
        for (size_t i = 0; i < tres.size(); ++i)
            if (tres[i].exists) root_res[i] = RootAlignment{true, root_spans[i].offset + tres[i].begin, tres[i].nm, tres[i].cigar_off, tres[i].cigar_len};
    }

    prof.mark("root-align");
    // ---- records (alignment.cpp:37-79, output.cpp:49-108): per reference in id order, alignments in verification order
    hvec<hvec<u32>> roots_of_read(reads.size());
    for (u32 i = 0; i < root_anchor.size(); ++i) roots_of_read[A[root_anchor[i]].read].push_back(i);   // already in verification order
    for (size_t r = 0; r < reads.size(); ++r) {
        bool have_best = false;
        u32 best = 0;
        for (u32 i : roots_of_read[r]) if (root_res[i].exists && (!have_best || root_res[i].nm < best)) { best = root_res[i].nm; have_best = true; }
        bool primary_written = false;
        for (u32 ref = 0; ref < H.seq_len.size(); ++ref)
            for (u32 i : roots_of_read[r]) {
                AnchorState const& a = A[root_anchor[i]];
                if (a.ref_id != ref || !root_res[i].exists) continue;
                u32 flag = a.orientation ? 16u : 0u;
                bool const primary = !primary_written && root_res[i].nm == best;
                if (primary) primary_written = true;
                else flag |= 256u;
                run->records.push_back(flx_record{reads[r].read_index, flag, (int32_t)ref, saturate_i32(root_res[i].start), root_res[i].nm,
                                                  root_res[i].cigar_off, root_res[i].cigar_len, 0});
            }
        if (!primary_written) run->records.push_back(flx_record{reads[r].read_index, 4u, -1, 0, 0, 0, 0, 0});
        if (st_local) {                                                                                  // parallelization.cpp:262-268
            u64 n_al = 0;
            for (u32 i : roots_of_read[r]) if (root_res[i].exists) { ++n_al; st_local->at(Stats::EDIT_DISTANCE).add(root_res[i].nm); }
            st_local->at(Stats::ALIGNMENTS_PER_QUERY).add(n_al);
        }
    }
    if (st_local) {
        double const total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_slice).count();
        for (size_t r = 0; r < reads.size(); ++r) st_local->at(Stats::MS_VERIFICATION).add((u64)((total_ms - search_ms) / (double)std::max<size_t>(1, reads.size())));
        stats_merge_locked(ctx->read_stats, *st_local);
    }
    run->cigars = std::move(cig);
    {
        u64 found = 0;
        for (auto const& rr : root_res) found += rr.exists;
        std::lock_guard<std::mutex> g(ctx->mu);
        flx_path_counters& pc = ctx->path;
        pc.inner_tests_requested += n_inner_requested; pc.root_alignments_requested += root_reqs.size(); pc.root_alignments_found += found;
        pc.records += run->records.size(); pc.reads += end_read - first_read;
    }
    prof.mark("records");
    return FLX_OK;
}

}  // namespace

extern "C" int flx_align_reads_resident(flx_ctx* ctx, const flx_params* P, const flx_reads* RD, flx_run** out) {
    if (!ctx || !P || !out || !RD || RD->ctx != ctx) { set_error("flx_align_reads_resident: null argument or reads of another context"); return FLX_ERR_INVALID; }
    FLX_HIP(hipSetDevice(ctx->device));
    if (P->query_error_probability < 0 && P->query_num_errors < P->pex_seed_num_errors) { set_error("query errors must be >= seed errors (floxer_cli.cpp:180)"); return FLX_ERR_INVALID; }
    if (P->pex_seed_num_errors > 3 || P->seed_sampling_step_size == 0 || P->num_anchors_per_verification_task == 0) { set_error("invalid parameters"); return FLX_ERR_INVALID; }
    if (P->search.max_num_anchors_hard < P->search.max_num_anchors_soft) { set_error("max-anchors-hard must not be smaller than max-anchors-soft"); return FLX_ERR_INVALID; }
    u64 const n_reads = RD->n_reads;
    PhaseTimer dprof("dispatch");
    auto run = std::make_unique<flx_run>();
    run->skipped.assign(n_reads, 0);
    if (n_reads == 0) { *out = run.release(); return FLX_OK; }       // an empty batch is an empty run
    // reads are independent units (parallelization.cpp:77-87): the batch is cut into contiguous chunks and every lane (a host
    // thread with its own stream and workspaces) takes the next chunk when it is done with its last one.
    size_t n_lanes = ctx->external_stream ? 1 : ctx->lanes.size();
    n_lanes = std::max<size_t>(1, std::min<size_t>(n_lanes, (n_reads + 63) / 64));
    // a chunk's kernels last as long as their longest job whatever the number of jobs, and a launch is the more efficient the
    // more jobs it has, so chunks are large: one per lane up to 2048 reads, more than one per lane beyond that
    // (a batch that gives every lane 1024 reads or more is cut into 2048-read chunks, batches overlap, see acquire_lane; with the
    // interval optimisation 1024-read chunks were better while the rounds' bookkeeping was host work: 2048 now, 78 k -> 83 k reads/s)
    u64 const big_chunk = 2048;
    u64 chunk_reads = n_reads >= 1024 * n_lanes ? big_chunk : std::max<u64>(64, (n_reads + n_lanes - 1) / n_lanes);
    if (const char* env = getenv("FLX_CHUNK_READS")) { u64 const v = strtoull(env, nullptr, 10); if (v >= 1) chunk_reads = v; }
    if (n_lanes == 1) chunk_reads = std::max<u64>(n_reads, 1);
    // A chunk's workspaces grow with its bases, not its reads (seeds, hits, requests, trace arena; with first_reported also the DFS
    // stacks of the ordered K1, ~100 bytes per read base): chunks are also cut at FLX_CHUNK_BASES read bases (default 48 M), so a
    // batch of 100-kb reads gets more, smaller chunks instead of workspaces of tens of GB per lane.
    u64 chunk_bases = 48ull << 20;
    if (const char* env = getenv("FLX_CHUNK_BASES")) { u64 const v = strtoull(env, nullptr, 10); if (v >= 1) chunk_bases = v; }
    hvec<u64> chunk_first{0};                                // chunk c = reads [chunk_first[c], chunk_first[c + 1])
    {
        u64 reads_in = 0, bases_in = 0;
        for (u64 i = 0; i < n_reads; ++i) {
            u64 const len = RD->lens[i];
            if (reads_in > 0 && (reads_in >= chunk_reads || bases_in + len > chunk_bases)) { chunk_first.push_back(i); reads_in = 0; bases_in = 0; }
            ++reads_in;
            bases_in += len;
        }
        chunk_first.push_back(n_reads);
    }
    size_t const n_chunks = chunk_first.size() - 1;
    run->parts.resize(n_chunks);
    hvec<flx_run>& parts = run->parts;
    hvec<int> rcs(n_chunks, FLX_OK);
    std::vector<std::string> errs(n_chunks);
    {
        std::lock_guard<std::mutex> g(RD->peq_mu);
        if (!RD->peq_built && n_reads) {
            LaneLease lease(ctx, ctx->external_stream ? 0 : -1);
            take_spare_read_buffer(ctx, 2, RD->d_peq);
            take_spare_read_buffer(ctx, 1, RD->d_pack);
            int const rc = build_peq(lease.lane, RD->d_pool.as<u8>(), RD->pool.size(), RD->d_peq);
            if (rc) return rc;
            if (ctx->didx.filter) {
                int const rc2 = RD->d_pack.ensure(pack_words_for(RD->pool.size()) * 4 + 64);
                if (rc2) return rc2;
                int const e = DeviceApi::pack_pool(lease.lane->stream, RD->d_pool.as<u8>(), RD->pool.size(), RD->d_pack.as<u32>());
                if (e) { set_error(std::string("pack_pool: ") + hipGetErrorString((hipError_t)e)); return FLX_ERR_NO_DEVICE; }
            }
            if (!RD->peq_event) FLX_HIP(hipEventCreateWithFlags(&RD->peq_event, hipEventDisableTiming));
            FLX_HIP(hipEventRecord(RD->peq_event, lease.lane->stream));
            RD->peq_built = true;
        }
    }
    std::atomic<size_t> next_chunk{0};
    std::atomic<bool> failed{false};
    auto work = [&]() {
        for (size_t c; (c = next_chunk.fetch_add(1)) < n_chunks && !failed.load();) {
            u64 const a = chunk_first[c], b = chunk_first[c + 1];
            parts[c].skipped.assign(n_reads, 0);
            LaneLease lease(ctx, ctx->external_stream ? 0 : -1);      // waits while other calls on this context hold all lanes
            rcs[c] = align_slice(lease.lane, P, RD, a, b, &parts[c]);
            if (rcs[c]) { errs[c] = flx_last_error(); failed.store(true); }
            else { lease.lane->has_run = true; if (!ctx->external_stream) ctx->warm_one_cold_lane(lease.lane); }
        }
    };
    size_t const n_workers = std::min(n_lanes, n_chunks);
    if (n_workers == 1) work();
    else {
        std::vector<std::thread> threads;
        for (size_t l = 0; l < n_workers; ++l) threads.emplace_back(work);
        for (auto& t : threads) t.join();
    }
    dprof.mark("lanes");
    for (size_t c = 0; c < n_chunks; ++c)
        if (rcs[c]) { set_error(errs[c]); return rcs[c]; }
    for (auto& p : parts)
        for (u64 i = 0; i < n_reads; ++i) run->skipped[i] |= p.skipped[i];
    *out = run.release();
    dprof.mark("merge");
    return FLX_OK;
}

extern "C" uint64_t flx_run_num_records(const flx_run* run) {
    if (!run) return 0;
    uint64_t n = run->records.size();
    for (auto const& p : run->parts) n += p.records.size();
    return n;
}
extern "C" uint64_t flx_run_num_cigar_words(const flx_run* run) {
    if (!run) return 0;
    uint64_t n = run->cigars.size();
    for (auto const& p : run->parts) n += p.cigars.size();
    return n;
}
extern "C" int flx_run_copy(const flx_run* run, flx_record* records, uint32_t* cigar_words, uint8_t* skipped) {
    if (!run) { set_error("null run"); return FLX_ERR_INVALID; }
    PhaseTimer cprof("run_copy");
    // the run itself plus its per-lane parts, each copied by its own thread (the CIGAR pools are tens of MB per part)
    hvec<const flx_run*> pieces{run};
    for (auto const& p : run->parts) pieces.push_back(&p);
    hvec<uint64_t> rec_base(pieces.size()), cig_base(pieces.size());
    uint64_t rb = 0, cb = 0;
    for (size_t i = 0; i < pieces.size(); ++i) { rec_base[i] = rb; cig_base[i] = cb; rb += pieces[i]->records.size(); cb += pieces[i]->cigars.size(); }
    auto emit = [&](size_t i) {
        flx_run const& part = *pieces[i];
        if (records) for (size_t r = 0; r < part.records.size(); ++r) { records[rec_base[i] + r] = part.records[r]; records[rec_base[i] + r].cigar_offset += cig_base[i]; }
        if (cigar_words && !part.cigars.empty()) memcpy(cigar_words + cig_base[i], part.cigars.data(), part.cigars.size() * 4);
    };
    if (pieces.size() <= 2) for (size_t i = 0; i < pieces.size(); ++i) emit(i);
    else {
        size_t const n_threads = std::min<size_t>(8, pieces.size());
        std::vector<std::thread> threads;
        for (size_t t = 0; t < n_threads; ++t)
            threads.emplace_back([&, t] { for (size_t i = t; i < pieces.size(); i += n_threads) emit(i); });
        for (auto& t : threads) t.join();
    }
    if (skipped && !run->skipped.empty()) memcpy(skipped, run->skipped.data(), run->skipped.size());
    cprof.mark("copy");
    return FLX_OK;
}
extern "C" void flx_run_free(flx_run* run) {
    PhaseTimer fprof("run_free");
    delete run;
    fprof.mark("free");
}
