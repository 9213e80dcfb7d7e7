// C ABI: host arithmetic, PEX trees, index lifetime, device context, kernel accounting.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <malloc.h>
#include <memory>
#include <mutex>
#include <thread>

#include "flx_context.hpp"

namespace flx {
const char* last_error_cstr();
}
using namespace flx;

extern "C" {

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

// ---------------------------------------------------------------- the HBM image of an index
namespace {
void image_sizes(HostIndex const& H, uint64_t bytes[5]) {
    u64 const nb = H.n / OCC_BLOCK_POS + 1;
    bytes[0] = nb * sizeof(OccBlock);
    bytes[1] = nb * sizeof(OccBlock);
    bytes[2] = H.n * 4;
    bytes[3] = H.n + 2 * TEXT_PAD + 16;
    bytes[4] = (((u64)1 << (2 * KMER_Q)) * 3) * 4;
}
bool has_arrays(HostIndex const& H) { return index_has_arrays(H); }
int upload_image(HostIndex const& H, void* const buf[5], hipStream_t s) {
    FLX_HIP(hipMemcpyAsync(buf[0], H.occ[0].data(), H.occ[0].size() * sizeof(OccBlock), hipMemcpyHostToDevice, s));
    FLX_HIP(hipMemcpyAsync(buf[1], H.occ[1].data(), H.occ[1].size() * sizeof(OccBlock), hipMemcpyHostToDevice, s));
    FLX_HIP(hipMemcpyAsync(buf[2], H.sa.data(), H.sa.size() * 4, hipMemcpyHostToDevice, s));
    FLX_HIP(hipMemsetAsync(buf[3], 0, H.n + 2 * TEXT_PAD + 16, s));
    FLX_HIP(hipMemcpyAsync((char*)buf[3] + TEXT_PAD, H.text.data(), H.n, hipMemcpyHostToDevice, s));
    FLX_HIP(hipMemcpyAsync(buf[4], H.kmer_table.data(), H.kmer_table.size() * 4, hipMemcpyHostToDevice, s));
    FLX_HIP(hipStreamSynchronize(s));
    return FLX_OK;
}
int check_device(int hip_device, const char* who) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        (void)hipGetLastError();
        set_error(std::string("no HIP device available (") + hipGetErrorString(e) + "); floxer_amd has no CPU fallback");
        return FLX_ERR_NO_DEVICE;
    }
    if (hip_device < 0 || hip_device >= count) { set_error(std::string(who) + ": device ordinal out of range"); return FLX_ERR_INVALID; }
    return FLX_OK;
}

// a context whose index image lives in `image` (owned by the context when own_image)
int make_context(int hip_device, const flx_index* index, void* const image[5], bool own_image, flx_ctx** out) {
    FLX_HIP(hipSetDevice(hip_device));
    auto ctx = std::make_unique<flx_ctx>();      // (its destructor releases whatever has been set up when a step below fails)
    ctx->device = hip_device;
    ctx->hidx = index->host;
    HostIndex const& H = *index->host;
    // Every batch builds and drops some hundred MB of host-side lists per lane. With glibc's defaults those go back to the
    // kernel on free and fault in again on the next batch; keep them in the heap instead (FLX_KEEP_MALLOC_DEFAULTS=1 leaves the
    // process-wide malloc settings alone).
    if (!getenv("FLX_KEEP_MALLOC_DEFAULTS")) {
        mallopt(M_MMAP_THRESHOLD, 32 << 20);
        mallopt(M_TRIM_THRESHOLD, 1 << 30);
        mallopt(M_TOP_PAD, 256 << 20);
    }
    // 16 lanes, fewer only on hosts with fewer than 4 hardware threads (FLX_LANES overrides): a lane's thread sleeps while its chunk
    // is on the GPU, and the GPU wants chunks in every stage (16 lanes on 4 cores: 96 % of the throughput on 16 cores; 4 lanes: 68 %)
    size_t n_lanes = std::thread::hardware_concurrency() >= 4 ? 16 : 8;
    if (const char* env = getenv("FLX_LANES")) { size_t const v = strtoull(env, nullptr, 10); if (v >= 1 && v <= 64) n_lanes = v; }
    for (size_t l = 0; l < n_lanes; ++l) {
        auto lane = std::make_unique<Lane>();
        lane->ctx = ctx.get();
        lane->id = (int)l;
        // (FLX_STREAMS=N, experiment: lanes beyond the N-th share the streams of the first N; a lane's waits then cover its partner's launches)
        static size_t const n_streams = [] { const char* e = getenv("FLX_STREAMS"); return e ? (size_t)std::max(1, atoi(e)) : (size_t)64; }();
        if (l < n_streams) { FLX_HIP(hipStreamCreateWithFlags(&lane->own_stream, hipStreamNonBlocking)); lane->stream = lane->own_stream; }
        else lane->stream = ctx->lanes[l % n_streams]->own_stream;
        ctx->lanes.push_back(std::move(lane));
        ctx->free_lanes.push_back((int)l);
    }
    // the host lists' blocks page-locked for DMA (FLX_NO_PIN=1: pageable, the runtime stages every copy). Process-wide and installed
    // once, by the first context: blocks the pool took before that stay pageable (the unregister of such a block fails quietly)
    static std::once_flag pin_once;
    std::call_once(pin_once, [] {
        if (getenv("FLX_NO_PIN")) return;
        host_pool_pin_hook.store([](void* p, size_t bytes, int pin) {
            if (pin) (void)hipHostRegister(p, bytes, hipHostRegisterPortable);
            else (void)hipHostUnregister(p);
            (void)hipGetLastError();      // (a block that could not be locked is simply pageable)
        }, std::memory_order_release);
    });
    // K1 launches in flight at a time: with every lane free to start its search the GPU swings between phases where K1's waves
    // (memory-bound, long-lived) hold most wave slots and phases where only the VALU-bound DP kernels run; about 3072 K1 waves at a
    // time keep the mix steady (16 lanes, 3.1 Gb / 10 kb: 74.7 k reads/s unlimited, 73-75 k with 4 x 512 waves, 78-79.5 k with 6 x 512,
    // 75-76 k with 7 or 8; with the rounds' job lists still on the host 4 x 512 was the best). FLX_K1_CONCURRENT overrides (0 = unlimited).
    {
        size_t const share = 4096 / std::max<size_t>(1, std::min<size_t>(n_lanes, 8));
        ctx->k1_tokens = n_lanes >= 2 ? (int)std::max<size_t>(1, 3072 / share) : 0;
        if (const char* env = getenv("FLX_K1_CONCURRENT")) ctx->k1_tokens = atoi(env);
    }
    FLX_HIP(hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
    hipStream_t const s0 = ctx->lanes[0]->stream;
    int rc;
    void* img[5];
    if (own_image) {
        uint64_t bytes[5];
        image_sizes(H, bytes);
        DeviceBuffer* bufs[5] = {&ctx->occ0, &ctx->occ1, &ctx->sa, &ctx->text, &ctx->kmer};
        for (int i = 0; i < 5; ++i) { if ((rc = bufs[i]->ensure(bytes[i], true))) return rc; img[i] = bufs[i]->ptr; }
        if ((rc = upload_image(H, img, s0))) return rc;
    } else for (int i = 0; i < 5; ++i) img[i] = image[i];
    if ((rc = ctx->seq_start.ensure(H.seq_start.size() * 8))) return rc;
    FLX_HIP(hipMemcpyAsync(ctx->seq_start.ptr, H.seq_start.data(), H.seq_start.size() * 8, hipMemcpyHostToDevice, s0));
    FLX_HIP(hipStreamSynchronize(s0));
    ctx->didx.occ[0] = reinterpret_cast<const OccBlock*>(img[0]);
    ctx->didx.occ[1] = reinterpret_cast<const OccBlock*>(img[1]);
    ctx->didx.sa = reinterpret_cast<const u32*>(img[2]);
    ctx->didx.text = reinterpret_cast<const u8*>(img[3]) + TEXT_PAD;
    ctx->didx.kmer = reinterpret_cast<const u32*>(img[4]);
    for (int c = 0; c < 7; ++c) ctx->didx.C[c] = (u32)H.C[c];
    ctx->didx.n = (u32)H.n;
    // The tables the seeding kernels use beyond the image: the inverse suffix array and the presence filter, made here from the
    // text and the suffix array in HBM (hg38 size: 12.4 GB + 8.6 GB, a fraction of a second). FLX_NO_DERIVED=1: neither (the walk
    // then uses rank queries only); a filter that does not fit is left out.
    ctx->didx.isa = nullptr; ctx->didx.filter = nullptr; ctx->didx.filter_m = nullptr; ctx->didx.filter_k = 0; ctx->didx.filter_tmin = 0;
    if (!getenv("FLX_NO_DERIVED") && H.n > 0) {
        u32 fk = 0;
        (void)DeviceApi::derived_bytes(H.n, &fk);
        // (neither table is needed for correct results: a device that cannot hold one - shared with other contexts or processes - gets a
        // context without it: no text walk without the inverse suffix array, no pruning without the filter)
        bool have_isa = false;
        {
            size_t free_i = 0, total_i = 0;
            FLX_HIP(hipMemGetInfo(&free_i, &total_i));
            size_t const want = (size_t)H.n * 4 + 64;
            have_isa = want < free_i / 2 && ctx->isa.ensure(want, true) == FLX_OK;
        }
        bool have_filter = false, have_filter_m = false;
        if (fk) {
            size_t free_f = 0, total_f = 0;
            FLX_HIP(hipMemGetInfo(&free_f, &total_f));
            size_t const want = (size_t)(((1ull << (2 * fk)) + 63) / 64) * 8;
            have_filter = want < free_f / 2 && ctx->filter.ensure(want, true) == FLX_OK;
            // the mirrored twin (the children of a leftward extension in one word; FLX_MIRRORED_FILTER=1): measured at 3.1 Gb it takes 8 % of
            // the filter walk's lines away (91.5 -> 84.0 GB per 16384 reads) and none of its time, for another 8.6 GB: off by default
            if (have_filter && getenv("FLX_MIRRORED_FILTER")) {
                FLX_HIP(hipMemGetInfo(&free_f, &total_f));
                have_filter_m = want < free_f / 2 && ctx->filter_m.ensure(want, true) == FLX_OK;
            }
        }
        int const e = DeviceApi::derive_index(s0, ctx->didx, have_isa ? ctx->isa.as<u32>() : nullptr, have_filter ? ctx->filter.as<u64>() : nullptr, have_filter_m ? ctx->filter_m.as<u64>() : nullptr);
        if (e) { set_error(std::string("derive_index: ") + hipGetErrorString((hipError_t)e)); return FLX_ERR_NO_DEVICE; }
        FLX_HIP(hipStreamSynchronize(s0));
    }
    if (getenv("FLX_ALLOC_DEBUG"))
        fprintf(stderr, "[flx alloc] context: occ0 %p occ1 %p sa %p text %p kmer %p (n %llu) isa %p filter %p (K %u, tmin %u) seq_start %p\n", img[0], img[1], img[2], img[3],
                img[4], (unsigned long long)H.n, ctx->isa.ptr, ctx->filter.ptr, ctx->didx.filter_k, ctx->didx.filter_tmin, ctx->seq_start.ptr);
    for (auto& lane : ctx->lanes) {
        int const e = DeviceApi::warm_scratch(lane->stream);
        if (e) { set_error(std::string("warm_scratch: ") + hipGetErrorString((hipError_t)e)); return FLX_ERR_NO_DEVICE; }
    }
    for (auto& lane : ctx->lanes) FLX_HIP(hipStreamSynchronize(lane->stream));
    // trace arena budget: FLX_TRACE_ARENA_MB (whole context), default 40% of the free HBM but not more than 4 GB per lane (a 2048-read
    // chunk of 10-kb reads needs 2-4 GB; what is not taken stays free for other contexts and processes on the GPU), at least 256 MB;
    // split over the lanes
    size_t free_b = 0, total_b = 0;
    FLX_HIP(hipMemGetInfo(&free_b, &total_b));
    size_t budget = std::min<size_t>(free_b / 10 * 4, n_lanes * ((size_t)4 << 30));
    if (const char* env = getenv("FLX_TRACE_ARENA_MB")) { size_t const mb = strtoull(env, nullptr, 10); if (mb) budget = mb << 20; }
    budget = std::max<size_t>(budget, (size_t)256 << 20);
    for (auto& lane : ctx->lanes) lane->trace_budget_bytes = std::max<size_t>(budget / n_lanes, (size_t)128 << 20);
    *out = ctx.release();
    return FLX_OK;
}
}  // namespace

int flx_ctx_create(int hip_device, const flx_index* index, flx_ctx** out) {
    if (!index || !out) { set_error("flx_ctx_create: null argument"); return FLX_ERR_INVALID; }
    if (int const rc = check_device(hip_device, "flx_ctx_create")) return rc;
    if (!has_arrays(*index->host)) { set_error("flx_ctx_create: this index holds no arrays (flx_index_meta_import): use flx_ctx_create_on_image"); return FLX_ERR_INVALID; }
    return make_context(hip_device, index, nullptr, true, out);
}

int flx_ctx_create_on_image(int hip_device, const flx_index* index, void* const device_buffers[5], const flx_index_image* sizes, flx_ctx** out) {
    if (!index || !out || !device_buffers || !sizes) { set_error("flx_ctx_create_on_image: null argument"); return FLX_ERR_INVALID; }
    for (int i = 0; i < 5; ++i) if (!device_buffers[i]) { set_error("flx_ctx_create_on_image: null device buffer"); return FLX_ERR_INVALID; }
    {
        uint64_t want[5];
        image_sizes(*index->host, want);
        for (int i = 0; i < 5; ++i)
            if (sizes->bytes[i] != want[i]) { set_error("flx_ctx_create_on_image: the buffers' sizes are not this index's image layout (flx_index_image_layout)"); return FLX_ERR_INVALID; }
    }
    if (int const rc = check_device(hip_device, "flx_ctx_create_on_image")) return rc;
    return make_context(hip_device, index, device_buffers, false, out);
}

int flx_index_image_layout(const flx_index* index, flx_index_image* out) {
    if (!index || !out) { set_error("flx_index_image_layout: null argument"); return FLX_ERR_INVALID; }
    image_sizes(*index->host, out->bytes);
    return FLX_OK;
}

int flx_index_image_upload(const flx_index* index, int hip_device, void* const device_buffers[5]) {
    if (!index || !device_buffers) { set_error("flx_index_image_upload: null argument"); return FLX_ERR_INVALID; }
    if (!has_arrays(*index->host)) { set_error("flx_index_image_upload: this index holds no arrays"); return FLX_ERR_INVALID; }
    if (int const rc = check_device(hip_device, "flx_index_image_upload")) return rc;
    FLX_HIP(hipSetDevice(hip_device));
    return upload_image(*index->host, device_buffers, nullptr);
}

flx_ctx::~flx_ctx() {
    (void)hipSetDevice(device);
    for (auto& lane : lanes) { if (lane->stream) (void)hipStreamSynchronize(lane->stream); lane->release_all(); }
    for (DeviceBuffer* b : {&occ0, &occ1, &sa, &text, &text_rev, &kmer, &seq_start, &isa, &filter}) b->release();
    if (upload_stream) (void)hipStreamDestroy(upload_stream);
}

void flx_ctx_destroy(flx_ctx* ctx) { delete ctx; }

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
