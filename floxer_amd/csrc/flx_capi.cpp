// C ABI: host arithmetic, PEX trees, index lifetime, device context, kernel accounting.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <malloc.h>
#include <memory>
#include <thread>

#include "flx_context.hpp"

namespace flx {
const char* last_error_cstr();
}
using namespace flx;

extern "C" {

const char* flx_last_error(void) { return last_error_cstr(); }
const char* flx_version(void) { return "floxer_amd 0.1.0 (gfx950)"; }

uint64_t flx_ceil_div(uint64_t a, uint64_t b) { return ceil_div(a, b); }
uint64_t flx_floating_point_error_aware_ceil(double value) { return fp_aware_ceil(value); }
int32_t flx_saturate_value_to_int32_max(uint64_t value) { return saturate_i32(value); }
void flx_chars_to_rank_sequence(const char* chars, uint64_t n, uint8_t* out) { for (uint64_t i = 0; i < n; ++i) out[i] = char_to_rank(chars[i]); }
void flx_reverse_complement_rank(const uint8_t* ranks, uint64_t n, uint8_t* out) { reverse_complement(ranks, n, out); }

int flx_pex_tree_build(uint64_t query_length, uint64_t query_num_errors, uint64_t leaf_max_num_errors, int bottom_up,
                       flx_pex_node* nodes, uint64_t capacity, uint64_t* n_inner, uint64_t* n_leaves) {
    if (!n_inner || !n_leaves) { set_error("flx_pex_tree_build: null argument"); return FLX_ERR_INVALID; }
    if (query_length == 0 || query_length > SCH_POS_MASK || query_num_errors >= query_length) { set_error("flx_pex_tree_build: invalid length / errors"); return FLX_ERR_INVALID; }
    PexTree const t = build_pex_tree(query_length, query_num_errors, leaf_max_num_errors, bottom_up != 0);
    *n_inner = t.inner.size();
    *n_leaves = t.leaves.size();
    if (t.inner.size() + t.leaves.size() > capacity || !nodes) { set_error("node buffer too small"); return FLX_ERR_CAPACITY; }
    if (!t.inner.empty()) memcpy(nodes, t.inner.data(), t.inner.size() * sizeof(flx_pex_node));
    memcpy(nodes + t.inner.size(), t.leaves.data(), t.leaves.size() * sizeof(flx_pex_node));
    return FLX_OK;
}

// ---------------------------------------------------------------- index
int flx_index_build(const uint8_t* concat, const uint64_t* lens, uint32_t n_refs, flx_index** out) {
    if (!concat || !lens || !out || n_refs == 0) { set_error("flx_index_build: null argument or no reference"); return FLX_ERR_INVALID; }
    HostIndex* h = build_host_index(concat, lens, n_refs);
    if (!h) return FLX_ERR_INVALID;
    *out = new flx_index{h};
    return FLX_OK;
}
int flx_index_build_on_device(int hip_device, const uint8_t* concat, const uint64_t* lens, uint32_t n_refs, flx_index** out) {
    if (!concat || !lens || !out || n_refs == 0) { set_error("flx_index_build_on_device: null argument or no reference"); return FLX_ERR_INVALID; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); set_error("flx_index_build_on_device: no HIP device"); return FLX_ERR_NO_DEVICE; }
    if (hip_device < 0 || hip_device >= count) { set_error("flx_index_build_on_device: device ordinal out of range"); return FLX_ERR_INVALID; }
    HostIndex* h = build_host_index(concat, lens, n_refs, hip_device);
    if (!h) return FLX_ERR_INVALID;
    *out = new flx_index{h};
    return FLX_OK;
}
int flx_index_save(const flx_index* index, const char* path) {
    if (!index || !path) { set_error("flx_index_save: null argument"); return FLX_ERR_INVALID; }
    return save_host_index(*index->host, path);
}
int flx_index_load(const char* path, flx_index** out) {
    if (!path || !out) { set_error("flx_index_load: null argument"); return FLX_ERR_INVALID; }
    HostIndex* h = load_host_index(path);
    if (!h) return FLX_ERR_IO;
    *out = new flx_index{h};
    return FLX_OK;
}
void flx_index_free(flx_index* index) {
    if (!index) return;
    delete index->host;
    delete index;
}
uint64_t flx_index_text_length(const flx_index* index) { return index ? index->host->n : 0; }
uint32_t flx_index_num_references(const flx_index* index) { return index ? (uint32_t)index->host->seq_len.size() : 0; }
uint64_t flx_index_device_bytes(const flx_index* index) {
    if (!index) return 0;
    HostIndex const& h = *index->host;
    return (h.occ[0].size() + h.occ[1].size()) * sizeof(OccBlock) + h.sa.size() * 4 + h.text.size() + 2 * TEXT_PAD + h.kmer_table.size() * 4;
}
int flx_index_matches_reference(const flx_index* index, const uint8_t* concat, const uint64_t* lens, uint32_t n_refs) {
    if (!index || !concat || !lens) { set_error("flx_index_matches_reference: null argument"); return FLX_ERR_INVALID; }
    HostIndex const& h = *index->host;
    if (h.seq_len.size() != n_refs) { set_error("the index holds " + std::to_string(h.seq_len.size()) + " sequences, the reference " + std::to_string(n_refs)); return FLX_ERR_INVALID; }
    uint64_t off = 0;
    for (uint32_t r = 0; r < n_refs; ++r) {
        if (h.seq_len[r] != lens[r]) { set_error("sequence " + std::to_string(r) + " has another length in the index"); return FLX_ERR_INVALID; }
        if (lens[r] && memcmp(h.text.data() + h.seq_start[r], concat + off, lens[r]) != 0) { set_error("sequence " + std::to_string(r) + " differs from the indexed text"); return FLX_ERR_INVALID; }
        off += lens[r];
    }
    return FLX_OK;
}
int flx_index_copy_sa(const flx_index* index, uint64_t* out) {
    if (!index || !out) { set_error("null argument"); return FLX_ERR_INVALID; }
    for (size_t i = 0; i < index->host->sa.size(); ++i) out[i] = index->host->sa[i];
    return FLX_OK;
}
int flx_index_copy_sa_u32(const flx_index* index, uint32_t* out) {
    if (!index || !out) { set_error("null argument"); return FLX_ERR_INVALID; }
    memcpy(out, index->host->sa.data(), index->host->sa.size() * 4);
    return FLX_OK;
}
int flx_index_copy_bwt(const flx_index* index, int reversed, uint8_t* out) {
    if (!index || !out) { set_error("null argument"); return FLX_ERR_INVALID; }
    auto const& b = index->host->bwt[reversed ? 1 : 0];
    memcpy(out, b.data(), b.size());
    return FLX_OK;
}

// ---------------------------------------------------------------- context
// ROCm multiplexes a process's HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); kernels of lanes that share a
// queue do not overlap. The HIP runtime reads the variable when it initialises, so it is set when this library is loaded (an
// existing value wins). A host program that initialises HIP before loading the library sets it itself.
__attribute__((constructor)) static void flx_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

int flx_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return count;
}

int flx_ctx_create(int hip_device, const flx_index* index, flx_ctx** out) {
    if (!index || !out) { set_error("flx_ctx_create: null argument"); return FLX_ERR_INVALID; }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        set_error(std::string("no HIP device available (") + hipGetErrorString(e) + "); floxer_amd has no CPU fallback");
        return FLX_ERR_NO_DEVICE;
    }
    if (hip_device < 0 || hip_device >= count) { set_error("flx_ctx_create: device ordinal out of range"); return FLX_ERR_INVALID; }
    FLX_HIP(hipSetDevice(hip_device));
    auto ctx = std::make_unique<flx_ctx>();
    ctx->device = hip_device;
    ctx->hidx = index->host;
    // Every batch builds and drops some hundred MB of host-side lists per lane. With glibc's defaults those go back to the
    // kernel on free and fault in again on the next batch; keep them in the heap instead (FLX_KEEP_MALLOC_DEFAULTS=1 leaves the
    // process-wide malloc settings alone).
    if (!getenv("FLX_KEEP_MALLOC_DEFAULTS")) {
        mallopt(M_MMAP_THRESHOLD, 32 << 20);
        mallopt(M_TRIM_THRESHOLD, 1 << 30);
        mallopt(M_TOP_PAD, 256 << 20);
    }
    // one lane per host thread the process may use, at most 16 (FLX_LANES overrides)
    size_t n_lanes = std::max<size_t>(4, std::min<size_t>(16, std::thread::hardware_concurrency()));
    if (const char* env = getenv("FLX_LANES")) { size_t const v = strtoull(env, nullptr, 10); if (v >= 1 && v <= 64) n_lanes = v; }
    for (size_t l = 0; l < n_lanes; ++l) {
        auto lane = std::make_unique<Lane>();
        lane->ctx = ctx.get();
        lane->id = (int)l;
        FLX_HIP(hipStreamCreateWithFlags(&lane->own_stream, hipStreamNonBlocking));
        lane->stream = lane->own_stream;
        ctx->lanes.push_back(std::move(lane));
        ctx->free_lanes.push_back((int)l);
    }
    FLX_HIP(hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
    hipStream_t const s0 = ctx->lanes[0]->stream;
    HostIndex const& H = *index->host;
    int rc;
    auto up = [&](DeviceBuffer& b, const void* src, size_t bytes) -> int {
        if ((rc = b.ensure(bytes))) return rc;
        FLX_HIP(hipMemcpyAsync(b.ptr, src, bytes, hipMemcpyHostToDevice, s0));
        return FLX_OK;
    };
    if ((rc = up(ctx->occ0, H.occ[0].data(), H.occ[0].size() * sizeof(OccBlock)))) return rc;
    if ((rc = up(ctx->occ1, H.occ[1].data(), H.occ[1].size() * sizeof(OccBlock)))) return rc;
    if ((rc = up(ctx->sa, H.sa.data(), H.sa.size() * 4))) return rc;
    if ((rc = up(ctx->kmer, H.kmer_table.data(), H.kmer_table.size() * 4))) return rc;
    if ((rc = up(ctx->seq_start, H.seq_start.data(), H.seq_start.size() * 8))) return rc;
    if ((rc = ctx->text.ensure(H.n + 2 * TEXT_PAD + 16))) return rc;
    FLX_HIP(hipMemsetAsync(ctx->text.ptr, 0, ctx->text.cap, s0));
    FLX_HIP(hipMemcpyAsync((char*)ctx->text.ptr + TEXT_PAD, H.text.data(), H.n, hipMemcpyHostToDevice, s0));
    FLX_HIP(hipStreamSynchronize(s0));
    ctx->didx.occ[0] = ctx->occ0.as<OccBlock>();
    ctx->didx.occ[1] = ctx->occ1.as<OccBlock>();
    ctx->didx.sa = ctx->sa.as<u32>();
    ctx->didx.kmer = ctx->kmer.as<u32>();
    ctx->didx.text = ctx->text.as<u8>() + TEXT_PAD;
    for (int c = 0; c < 7; ++c) ctx->didx.C[c] = (u32)H.C[c];
    ctx->didx.n = (u32)H.n;
    // trace arena budget: FLX_TRACE_ARENA_MB (whole context), default 40% of the free HBM, at least 256 MB; split over the lanes
    size_t free_b = 0, total_b = 0;
    FLX_HIP(hipMemGetInfo(&free_b, &total_b));
    size_t budget = free_b / 10 * 4;
    if (const char* env = getenv("FLX_TRACE_ARENA_MB")) { size_t const mb = strtoull(env, nullptr, 10); if (mb) budget = mb << 20; }
    budget = std::max<size_t>(budget, (size_t)256 << 20);
    for (auto& lane : ctx->lanes) lane->trace_budget_bytes = std::max<size_t>(budget / n_lanes, (size_t)128 << 20);
    *out = ctx.release();
    return FLX_OK;
}

void flx_ctx_destroy(flx_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (auto& lane : ctx->lanes) { (void)hipStreamSynchronize(lane->stream); lane->release_all(); }
    for (DeviceBuffer* b : {&ctx->occ0, &ctx->occ1, &ctx->sa, &ctx->text, &ctx->text_rev, &ctx->kmer, &ctx->seq_start}) b->release();
    if (ctx->upload_stream) (void)hipStreamDestroy(ctx->upload_stream);
    delete ctx;
}

int flx_ctx_set_stream(flx_ctx* ctx, void* hip_stream) {
    if (!ctx) { set_error("null context"); return FLX_ERR_INVALID; }
    int rc = ctx->sync_all();
    if (rc) return rc;
    Lane* l0 = ctx->lane0();
    l0->stream = hip_stream ? (hipStream_t)hip_stream : l0->own_stream;
    ctx->external_stream = hip_stream != nullptr;      // with a caller-owned stream every launch goes to that stream only
    return FLX_OK;
}

int flx_ctx_enable_kernel_timing(flx_ctx* ctx, int enable) {
    if (!ctx) { set_error("null context"); return FLX_ERR_INVALID; }
    int rc = ctx->sync_all();
    if (rc) return rc;
    ctx->timing = enable != 0;
    return FLX_OK;
}
int flx_ctx_reset_kernel_stats(flx_ctx* ctx) {
    if (!ctx) { set_error("null context"); return FLX_ERR_INVALID; }
    int rc = ctx->sync_all();
    if (rc) return rc;
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->stats.clear();
    ctx->stat_order.clear();
    return FLX_OK;
}
int flx_ctx_get_kernel_stats(flx_ctx* ctx, flx_kernel_stat* out, uint32_t* n) {
    if (!ctx || !n) { set_error("null argument"); return FLX_ERR_INVALID; }
    int rc = ctx->sync_all();
    if (rc) return rc;
    std::lock_guard<std::mutex> g(ctx->mu);
    uint32_t const cap = *n;
    *n = (uint32_t)ctx->stat_order.size();
    if (ctx->stat_order.size() > cap || (!out && cap)) { set_error("stat buffer too small"); return FLX_ERR_CAPACITY; }
    for (size_t i = 0; i < ctx->stat_order.size(); ++i) out[i] = ctx->stats[ctx->stat_order[i]];
    return FLX_OK;
}

int flx_ctx_set_stats(flx_ctx* ctx, flx_stats* stats) {
    if (!ctx) { set_error("null context"); return FLX_ERR_INVALID; }
    int const rc = ctx->sync_all();
    if (rc) return rc;
    ctx->read_stats = stats;
    return FLX_OK;
}
int flx_ctx_get_path_counters(flx_ctx* ctx, flx_path_counters* out) {
    if (!ctx || !out) { set_error("null argument"); return FLX_ERR_INVALID; }
    std::lock_guard<std::mutex> g(ctx->mu);
    *out = ctx->path;
    return FLX_OK;
}
int flx_ctx_reset_path_counters(flx_ctx* ctx) {
    if (!ctx) { set_error("null context"); return FLX_ERR_INVALID; }
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->path = flx_path_counters{};
    return FLX_OK;
}

}  // extern "C"
