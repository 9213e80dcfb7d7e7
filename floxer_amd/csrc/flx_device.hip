// HIP kernels of the seed-and-verify path for gfx950 (MI355X, wave64). No MFMA: the path is integer rank lookups
// (random 128-byte lines) and bit-parallel edit-distance DP walked as skewed anti-diagonals across the lanes of a wave.
//
//   K0 peq_build      query bytes -> per-64-row equality bit masks (6 symbols), wave ballots
//   K1 fm_search      search_ng21::search_n per seed (search.cpp:173-188): DFS over the expanded optimum search scheme
//   K2 fm_locate      index.locate(row) (search.cpp:253, 284) as an SA gather
//   K3/K4 ed_align    seqan3 edit-distance semi-global DP (alignment.cpp:89-125, 160): score + end column, optional trace
//   K5 ed_traceback   trace walk + CIGAR (alignment.cpp:166-180)
#include <hip/hip_runtime.h>

#include <type_traits>
#include <hipcub/hipcub.hpp>
#include <rocprim/rocprim.hpp>

#include <cstdlib>

#include "flx_internal.hpp"
#include "flx_stdsort.hpp"

namespace flx {

// ================================================================================================ helpers
__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63u; }

// value of the previous lane (lane-1); lane 0 receives 0. wave_shr:1 DPP is a single VALU move on gfx9-family ISAs.
__device__ __forceinline__ u32 from_prev_lane(u32 v) {
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

__device__ __forceinline__ u32 wave_max_u32(u32 v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        u32 const o = (u32)__shfl_xor((int)v, off);
        v = v > o ? v : o;
    }
    return (u32)__builtin_amdgcn_readfirstlane((int)v);
}

// ================================================================================================ K0: Peq planes
// peq[(word * 6) + sym] bit r = (seq[64*word + r] == sym). One wave per 64 query bytes: six ballots.
__global__ void __launch_bounds__(256) peq_build_kernel(const u8* __restrict__ seq, u64 len, u64* __restrict__ peq, u64 n_words) {
    u64 const wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wave >= n_words) return;
    u64 const pos = wave * 64 + lane_id();
    u32 const sym = pos < len ? seq[pos] : 7u;
#pragma unroll
    for (u32 s = 0; s < 6; ++s) {
        u64 const m = __ballot(sym == s);
        if (lane_id() == s) peq[wave * 6 + s] = m;
    }
}

int DeviceApi::build_peq(void* stream, const u8* d_seq, u64 len, u64* d_peq) {
    u64 const n_words = len / 64 + 2;    // +1 partial word, +1 so that the funnel shift may read word+1
    u64 const threads = n_words * 64;
    hipLaunchKernelGGL(peq_build_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_seq, len,
                       d_peq, n_words);
    return (int)hipGetLastError();
}

// ================================================================================================ suffix array (index construction)
// Prefix doubling with radix sorts: ranks of the first 10 symbols, then h = 10, 20, 40, ...: suffixes sorted by (rank[i], rank[i+h])
// until all ranks differ. A suffix that is a prefix of another sorts first (positions past the end rank 0), as the host's SA-IS
// orders them. 36 bytes of HBM per text symbol while it runs.
__global__ void __launch_bounds__(256) sa_init_kernel(const u8* __restrict__ text, u64 n, u64* __restrict__ keys, u32* __restrict__ sa) {
    u64 const i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 key = 0;
    for (u32 j = 0; j < 10; ++j) key = (key << 3) | (i + j < n ? (u64)text[i + j] + 1u : 0u);
    keys[i] = key;
    sa[i] = (u32)i;
}
__global__ void __launch_bounds__(256) sa_flag_kernel(const u64* __restrict__ keys, u64 n, u32* __restrict__ flags) {
    u64 const j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    flags[j] = (j == 0 || keys[j] != keys[j - 1]) ? 1u : 0u;
}
__global__ void __launch_bounds__(256) sa_rank_kernel(const u32* __restrict__ sa, const u32* __restrict__ r, u64 n, u32* __restrict__ rank) {
    u64 const j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    rank[sa[j]] = r[j];
}
__global__ void __launch_bounds__(256) sa_key_kernel(const u32* __restrict__ sa, const u32* __restrict__ rank, u64 n, u64 h, u64* __restrict__ keys) {
    u64 const j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    u64 const i = sa[j];
    keys[j] = ((u64)rank[i] << 32) | (i + h < n ? (u64)rank[i + h] : 0ull);
}

namespace {
// workspaces of one suffix-array construction (36 bytes per text symbol)
struct SaWork {
    u64 *keys = nullptr, *keys2 = nullptr;
    u32 *sa = nullptr, *sa2 = nullptr, *rank = nullptr, *flags = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    hipError_t alloc(u64 n, hipStream_t s) {
        hipError_t e;
        if ((e = hipMalloc(&keys, n * 8)) != hipSuccess) return e;
        if ((e = hipMalloc(&keys2, n * 8)) != hipSuccess) return e;
        if ((e = hipMalloc(&sa, n * 4)) != hipSuccess) return e;
        if ((e = hipMalloc(&sa2, n * 4)) != hipSuccess) return e;
        if ((e = hipMalloc(&rank, n * 4)) != hipSuccess) return e;
        if ((e = hipMalloc(&flags, n * 4)) != hipSuccess) return e;
        size_t sort_bytes = 0, scan_bytes = 0;
        if ((e = rocprim::radix_sort_pairs(nullptr, sort_bytes, keys, keys2, sa, sa2, (size_t)n, 0u, 64u, s)) != hipSuccess) return e;
        if ((e = rocprim::inclusive_scan(nullptr, scan_bytes, flags, flags, (size_t)n, rocprim::plus<u32>(), s)) != hipSuccess) return e;
        tmp_bytes = std::max(sort_bytes, scan_bytes);
        return hipMalloc(&tmp, tmp_bytes);
    }
    void release() {
        for (void* p : {(void*)keys, (void*)keys2, (void*)sa, (void*)sa2, (void*)rank, (void*)flags, tmp}) if (p) (void)hipFree(p);
        keys = keys2 = nullptr; sa = sa2 = rank = flags = nullptr; tmp = nullptr;
    }
};

// suffix array of d_text[0, n) into w.sa (device)
hipError_t sa_on_device(hipStream_t s, const u8* d_text, u64 n, SaWork& w) {
    hipError_t e;
    unsigned const blocks = (unsigned)((n + 255) / 256);
    u32 top = 0;
    hipLaunchKernelGGL(sa_init_kernel, dim3(blocks), dim3(256), 0, s, d_text, n, w.keys, w.sa);
    for (u64 h = 10;; h *= 2) {
        // sort the suffixes by their keys; ranks = number of distinct keys up to and including each position
        if ((e = rocprim::radix_sort_pairs(w.tmp, w.tmp_bytes, w.keys, w.keys2, w.sa, w.sa2, (size_t)n, 0u, 64u, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(sa_flag_kernel, dim3(blocks), dim3(256), 0, s, w.keys2, n, w.flags);
        if ((e = rocprim::inclusive_scan(w.tmp, w.tmp_bytes, w.flags, w.flags, (size_t)n, rocprim::plus<u32>(), s)) != hipSuccess) return e;
        hipLaunchKernelGGL(sa_rank_kernel, dim3(blocks), dim3(256), 0, s, w.sa2, w.flags, n, w.rank);
        if ((e = hipMemcpyAsync(&top, w.flags + (n - 1), 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
        std::swap(w.sa, w.sa2);
        if ((u64)top == n || h >= n) break;                    // all suffixes distinct
        hipLaunchKernelGGL(sa_key_kernel, dim3(blocks), dim3(256), 0, s, w.sa, w.rank, n, h, w.keys);
    }
    return hipGetLastError();
}
}  // namespace

int DeviceApi::suffix_array(int hip_device, const u8* text, u64 n, u32* out) {
    if (n == 0) return 0;
    hipError_t e;
    u8* d_text = nullptr;
    hipStream_t s = nullptr;
    SaWork w;
#define SA_HIP(x) do { e = (x); if (e != hipSuccess) goto done; } while (0)
    SA_HIP(hipSetDevice(hip_device));
    SA_HIP(hipStreamCreate(&s));
    SA_HIP(hipMalloc(&d_text, n));
    SA_HIP(w.alloc(n, s));
    SA_HIP(hipMemcpyAsync(d_text, text, n, hipMemcpyHostToDevice, s));
    SA_HIP(sa_on_device(s, d_text, n, w));
    SA_HIP(hipMemcpyAsync(out, w.sa, n * 4, hipMemcpyDeviceToHost, s));
    SA_HIP(hipStreamSynchronize(s));
    e = hipGetLastError();
done:
    w.release();
    if (d_text) (void)hipFree(d_text);
    if (s) (void)hipStreamDestroy(s);
    return (int)e;
}

// ------------------------------------------------------------------------------------------------ BWT + occurrence blocks on the device
__global__ void __launch_bounds__(256) bwt_kernel(const u8* __restrict__ text, const u32* __restrict__ sa, u64 n, u8* __restrict__ bwt) {
    u64 const i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 const p = sa[i];
    bwt[i] = text[p ? p - 1 : n - 1];
}
__global__ void __launch_bounds__(256) reverse_kernel(const u8* __restrict__ text, u64 n, u8* __restrict__ rev) {
    u64 const i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rev[i] = text[n - 1 - i];
}
// one wave per two 32-position blocks: the three bit-planes by ballot, every block's own symbol counts into cnt[c * nb + b]
__global__ void __launch_bounds__(256) occ_planes_kernel(const u8* __restrict__ bwt, u64 n, u64 nb, OccBlock* __restrict__ blocks, u32* __restrict__ cnt) {
    u64 const pair = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (pair * 2 >= nb) return;
    u32 const lane = lane_id();
    u64 const pos = pair * 64 + lane;
    u32 const sym = pos < n ? bwt[pos] : 7u;
    u64 const p0 = __ballot(sym & 1u), p1 = __ballot(sym & 2u), p2 = __ballot(sym & 4u);
    u32 const half = lane >> 5, l = lane & 31u;                 // lanes 0..31 write block 2*pair, lanes 32..63 block 2*pair + 1
    u64 const b = pair * 2 + half;
    if (b >= nb) return;
    u32 const q0 = (u32)(half ? p0 >> 32 : p0), q1 = (u32)(half ? p1 >> 32 : p1), q2 = (u32)(half ? p2 >> 32 : p2);
    if (l < 5) {
        u32 const m = (l & 1u ? q0 : ~q0) & (l & 2u ? q1 : ~q1) & (l & 4u ? q2 : ~q2);
        cnt[(u64)l * nb + b] = (u32)__popc(m);
    } else if (l < 8) blocks[b].w[l] = l == 5 ? q0 : l == 6 ? q1 : q2;
}
__global__ void __launch_bounds__(256) occ_counts_kernel(const u32* __restrict__ cnt, u64 nb, OccBlock* __restrict__ blocks) {
    u64 const i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb * 5) return;
    u64 const c = i / nb, b = i - c * nb;
    blocks[b].w[c] = cnt[i];
}

// Suffix array, both BWTs and both occurrence tables of text[0, n) on the device; results land in host memory. The counts of the
// symbols 0..4 per block are made absolute by five exclusive scans over the blocks.
int DeviceApi::index_arrays(int hip_device, const u8* text, u64 n, u32* out_sa, u8* out_bwt0, u8* out_bwt1, OccBlock* out_occ0, OccBlock* out_occ1) {
    if (n == 0) return 0;
    hipError_t e;
    u8 *d_text = nullptr, *d_rev = nullptr, *d_bwt = nullptr;
    OccBlock* d_occ = nullptr;
    u32* d_cnt = nullptr;
    hipStream_t s = nullptr;
    SaWork w;
    u64 const nb = n / OCC_BLOCK_POS + 1;
    unsigned const blocks_n = (unsigned)((n + 255) / 256);
    SA_HIP(hipSetDevice(hip_device));
    SA_HIP(hipStreamCreate(&s));
    SA_HIP(hipMalloc(&d_text, n));
    SA_HIP(hipMalloc(&d_rev, n));
    SA_HIP(hipMalloc(&d_bwt, n));
    SA_HIP(hipMalloc(&d_occ, nb * sizeof(OccBlock)));
    SA_HIP(hipMalloc(&d_cnt, nb * 5 * 4));
    SA_HIP(w.alloc(std::max<u64>(n, nb), s));
    {
        size_t need = 0;
        SA_HIP(rocprim::exclusive_scan(nullptr, need, d_cnt, d_cnt, 0u, (size_t)nb, rocprim::plus<u32>(), s));
        if (need > w.tmp_bytes) { (void)hipFree(w.tmp); w.tmp = nullptr; w.tmp_bytes = need; SA_HIP(hipMalloc(&w.tmp, need)); }
    }
    SA_HIP(hipMemcpyAsync(d_text, text, n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(reverse_kernel, dim3(blocks_n), dim3(256), 0, s, d_text, n, d_rev);
    for (int dir = 0; dir < 2; ++dir) {
        const u8* t = dir ? d_rev : d_text;
        SA_HIP(sa_on_device(s, t, n, w));
        if (dir == 0) SA_HIP(hipMemcpyAsync(out_sa, w.sa, n * 4, hipMemcpyDeviceToHost, s));
        hipLaunchKernelGGL(bwt_kernel, dim3(blocks_n), dim3(256), 0, s, t, w.sa, n, d_bwt);
        SA_HIP(hipMemcpyAsync(dir ? out_bwt1 : out_bwt0, d_bwt, n, hipMemcpyDeviceToHost, s));
        hipLaunchKernelGGL(occ_planes_kernel, dim3((unsigned)(((nb + 1) / 2 * 64 + 255) / 256)), dim3(256), 0, s, d_bwt, n, nb, d_occ, d_cnt);
        for (u32 c = 0; c < 5; ++c)
            SA_HIP(rocprim::exclusive_scan(w.tmp, w.tmp_bytes, d_cnt + (u64)c * nb, d_cnt + (u64)c * nb, 0u, (size_t)nb, rocprim::plus<u32>(), s));
        hipLaunchKernelGGL(occ_counts_kernel, dim3((unsigned)((nb * 5 + 255) / 256)), dim3(256), 0, s, d_cnt, nb, d_occ);
        SA_HIP(hipMemcpyAsync(dir ? out_occ1 : out_occ0, d_occ, nb * sizeof(OccBlock), hipMemcpyDeviceToHost, s));
        SA_HIP(hipStreamSynchronize(s));
    }
    e = hipGetLastError();
done:
#undef SA_HIP
    w.release();
    for (void* p : {(void*)d_text, (void*)d_rev, (void*)d_bwt, (void*)d_occ, (void*)d_cnt}) if (p) (void)hipFree(p);
    if (s) (void)hipStreamDestroy(s);
    return (int)e;
}

// ================================================================================================ K1: FM search, the reference's order
// The default walk (error children first, stack in LDS, presence filter, one-row subtrees against the text) is in flx_fm_core.hpp /
// flx_search.hip. fm_search_ordered_kernel below walks the DFS of search_ng21 in the reference's own order with an explicit stack in
// HBM: for first_reported (the first n rows in emission order) and the raw-emission test hook, where the order of discovery itself
// is the result. One lane serves one seed; a rank query reads one 32-byte block (32 BWT positions: five absolute counters + three
// bit-planes) with two 16-byte loads and pop-counts the positions below the offset.

// r[c] = number of symbol c in bwt[0, pos) for c = 0..4
__device__ __forceinline__ void rank5(const OccBlock* __restrict__ tab, u32 pos, u32 r[5]) {
    const uint4* __restrict__ q = reinterpret_cast<const uint4*>(tab + (pos >> 5));
    uint4 const a = q[0], b = q[1];
    u32 const mask = (1u << (pos & 31u)) - 1u;
    u32 const p0 = b.y, p1 = b.z, p2 = b.w;
    u32 const n2 = ~p2 & mask;
    r[0] = a.x + (u32)__popc(n2 & ~(p1 | p0));
    r[1] = a.y + (u32)__popc(n2 & ~p1 & p0);
    r[2] = a.z + (u32)__popc(n2 & p1 & ~p0);
    r[3] = a.w + (u32)__popc(n2 & p1 & p0);
    r[4] = b.x + (u32)__popc(p2 & mask & ~(p1 | p0));
}

// both ends of the interval [lo, lo + nlen): cl[c] = rows of the child of symbol c (c = 0..5), ab[c] = its lower bound on the
// extended side (symbol 0, the sequence delimiter, is only ever a match child: a read holding the character '$', input.cpp:165-176)
__device__ __forceinline__ void extend_all(const DevIndex& idx, const OccBlock* __restrict__ tab, u32 lo, u32 nlen, u32 ab[6], u32 cl[6]) {
    u32 ra[5], rb[5];
    rank5(tab, lo, ra);
    rank5(tab, lo + nlen, rb);
    u32 sum_a = 0, sum_l = 0;
#pragma unroll
    for (u32 c = 0; c < 5; ++c) { cl[c] = rb[c] - ra[c]; sum_a += ra[c]; sum_l += cl[c]; }
    cl[5] = nlen - sum_l;
    ab[0] = ra[0];                                                    // C[0] = 0
#pragma unroll
    for (u32 c = 1; c < 5; ++c) ab[c] = idx.C[c] + ra[c];
    ab[5] = idx.C[5] + (lo - sum_a);
}

// frame state word: x:20 | e:3 | linfo:2 | rinfo:2 | next_sym:3 | right:1
enum : u32 { INFO_M = 0, INFO_I = 1, INFO_D = 2, INFO_S = 3 };
__device__ __forceinline__ u32 st_pack(u32 x, u32 e, u32 li, u32 ri, u32 sym, u32 right) {
    return x | (e << 20) | (li << 23) | (ri << 25) | (sym << 27) | (right << 30);
}
#define ST_X(s) ((s) & 0xFFFFFu)
#define ST_E(s) (((s) >> 20) & 7u)
#define ST_LI(s) (((s) >> 23) & 3u)
#define ST_RI(s) (((s) >> 25) & 3u)
#define ST_SYM(s) (((s) >> 27) & 7u)
#define ST_RIGHT(s) (((s) >> 30) & 1u)

__device__ __forceinline__ u32 wave_sum_u32(u32 v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += (u32)__shfl_xor((int)v, off);
    return v;
}

// counters: [0] hit slots reserved, [1] stack overflow flag, [2] cursor extensions (rank pairs), [3] unused,
//           [4] wave-iterations, [5] max iterations of a wave, [6] busy lane-iterations, [7] seed queue head,
//           [8] wave-iterations after the seed queue ran dry, [9] their maximum over the waves
//
// DFS sizes differ by orders of magnitude between seeds, so neither lanes nor waves are bound to seeds: the launch is a fixed
// number of waves, a wave takes FM_GRAB consecutive seeds at a time from a global counter (counters[7]) and hands them to its lanes
// as they finish (wave-uniform bookkeeping in scalar registers). One loop iteration = one DFS step of every busy lane (at most one
// rank pair), which keeps the divergent part of the loop short.
constexpr u32 FM_GRAB = 64;
constexpr u32 FM_HIT_GRAB = 64;
constexpr u32 FM_MAX_WAVES = 4096;
constexpr u32 FM_SEEDS_PER_WAVE = 256;      // a launch has at most n_seeds / this many waves, so that every wave gets several ranges
constexpr u32 FM_KEY_MAX_X = 0x3FFFu;

// hit slots for the hits the lanes found in the last iteration. Slots are reserved FM_HIT_GRAB at a time per wave (one global
// atomic per range instead of one per hit, all on one address); the unused rest of a range is filled with entries of seed
// 0xFFFFFFFF, which the consumers skip. The hit's ordinal within its seed (the order the kernel found them in) rides in the upper
// bits of the error count (errors <= 3): the hits of a seed are put into one segment without a sort.
#define FM_EMIT_HITS()                                                                                                              \
    do {                                                                                                                            \
        u64 const emit = __ballot(hit_pending);                                                                                     \
        if (emit) {                                                                                                                 \
            u32 const n_emit = (u32)__popcll(emit);                                                                                 \
            if (h_end - h_next < n_emit) {                                                                                          \
                { u32 const at = h_next + lane; if (at < h_end && at < hit_cap) hits[at] = DevHit{0xFFFFFFFFu, 0u, 0u, 0u, 0ull}; } \
                u32 b = 0;                                                                                                          \
                if (lane == 0) b = atomicAdd(&counters[0], FM_HIT_GRAB);                                                            \
                h_next = (u32)__builtin_amdgcn_readfirstlane((int)b);                                                               \
                h_end = h_next + FM_HIT_GRAB;                                                                                       \
            }                                                                                                                       \
            if (hit_pending) {                                                                                                      \
                u32 const slot = h_next + (u32)__popcll(emit & lanes_below);                                                        \
                if (slot < hit_cap) hits[slot] = DevHit{sid, nlb, hit_rep, seed_cnt ? ne | (min(hit_idx, 0xFFFFFFu) << 8) : ne, hit_key}; \
                if (seed_cnt) seed_cnt[sid] = hit_idx + 1u;                                                                         \
                ++hit_idx;                                                                                                          \
            }                                                                                                                       \
            h_next += n_emit;                                                                                                       \
            hit_pending = false;                                                                                                    \
        }                                                                                                                           \
    } while (0)

// seeds for the idle lanes: k = index of this lane's new seed or 0xFFFFFFFF (wave-uniform bookkeeping of the grabbed range)
#define FM_ASSIGN_SEEDS(k)                                                                                                          \
    do {                                                                                                                            \
        u32 const n_idle = (u32)__popcll(idle);                                                                                     \
        u32 const avail = q_end - q_next;                                                                                           \
        u32 new_base = 0;                                                                                                           \
        bool grabbed = false;                                                                                                       \
        if (avail < n_idle && !queue_done) {                                                                                        \
            u32 b = 0;                                                                                                              \
            if (lane == 0) b = atomicAdd(&counters[7], FM_GRAB);                                                                    \
            new_base = (u32)__builtin_amdgcn_readfirstlane((int)b);                                                                 \
            grabbed = true;                                                                                                         \
        }                                                                                                                           \
        u32 const r = (u32)__popcll(idle & lanes_below);                                                                            \
        if (want) {                                                                                                                 \
            if (r < avail) k = q_next + r;                                                                                          \
            else if (grabbed && new_base + (r - avail) < n_seeds) k = new_base + (r - avail);                                       \
        }                                                                                                                           \
        if (grabbed) {                                                                                                              \
            if (new_base >= n_seeds) { q_next = 0; q_end = 0; queue_done = true; }                                                  \
            else {                                                                                                                  \
                q_end = min(new_base + FM_GRAB, n_seeds);                                                                           \
                q_next = min(new_base + (n_idle - avail), q_end);                                                                   \
                queue_done = new_base + FM_GRAB >= n_seeds;                                                                         \
            }                                                                                                                       \
        } else q_next += min(n_idle, avail);                                                                                        \
    } while (0)

// start of search `srch` of the seed: the root cursor, or the cursor of the seed's first KMER_Q characters when the search begins
// with an exact, rightward part that long and free of N. false: the search finds nothing.
__device__ __forceinline__ bool fm_begin_search(DevIndex const& idx, const u64* __restrict__ ex, const u8* __restrict__ q, u32 len,
                                                u32& nlb, u32& nlbr, u32& nlen, u32& nx) {
    nlb = 0; nlbr = 0; nlen = idx.n; nx = 0;
    if (len >= KMER_Q && ((ex[KMER_Q - 1] >> 27) & 1u)) {
        u32 const p0 = (u32)ex[0] & SCH_POS_MASK;
        u32 w[2];
        __builtin_memcpy(w, q + p0, 8);                                  // eight ranks, first character in the low byte
        u32 const t0 = w[0] - 0x01010101u, t1 = w[1] - 0x01010101u;      // A,C,G,T -> 0..3; anything else leaves bits 2..7 set
        if (((t0 | t1) & 0xFCFCFCFCu) == 0u) {
            // gather the four 2-bit fields of a word, first character most significant: b0<<6 | b1<<4 | b2<<2 | b3
            u32 const code = (((t0 * 0x40100401u) >> 24) << 8) | ((t1 * 0x40100401u) >> 24);
            const u32* __restrict__ e = idx.kmer + 3u * code;
            nlb = e[0]; nlbr = e[1]; nlen = e[2];
            nx = KMER_Q;
            if (nlen == 0) return false;
        }
    }
    return true;
}

// the children of a branching node that exist: bit 0 match, bits 2c-1 / 2c deletion / substitution of symbol c, bit 11 insertion
__device__ __forceinline__ u32 fm_child_mask(const u32 cl[6], u32 next_sym, bool match_allowed, bool deletion, bool insertion) {
    u32 mask = 0;
#pragma unroll
    for (u32 c = 1; c < 6; ++c) {
        if (cl[c] > 0u) {
            if (deletion) mask |= 1u << (2u * c - 1u);
            if (c != next_sym) mask |= 1u << (2u * c);
            else if (match_allowed) mask |= 1u;
        }
    }
    if (next_sym == 0u && match_allowed && cl[0] > 0u) mask |= 1u;      // a '$' of the query matches a sequence delimiter
    if (insertion) mask |= 1u << 11;
    return mask;
}

// The DFS in the reference's own order (match child first): frames are written to the seed's stack in HBM when they are made
// (64 B = four 16-byte stores) and read back when the DFS returns to them; the children of the top frame are in LDS.
__global__ void __launch_bounds__(64) fm_search_ordered_kernel(DevIndex idx, const u8* __restrict__ seq, const u64* __restrict__ scheme,
                                                               const DevSeed* __restrict__ seeds, u32 n_seeds, u32 max_hits,
                                                               DevFrame* __restrict__ stack, DevHit* __restrict__ hits, u32 hit_cap,
                                                               u32* __restrict__ counters, u32* __restrict__ seed_cnt) {
    __shared__ uint4 child[6][64];              // top frame: {abs, oth, len, -} of the child cursor of symbol s+1 (entry 5: symbol 0), per lane
    u32 q_next = 0, q_end = 0;
    bool queue_done = false;
    u32 h_next = 0, h_end = 0;
    u32 const lane = threadIdx.x & 63u;
    u64 const lanes_below = (1ull << lane) - 1ull;

    u32 n_ext = 0, n_iter = 0, n_busy_iter = 0, n_tail_iter = 0;
    bool busy = false, exhausted = false;
    u32 sid = 0, srch = 0, num_searches = 0, len = 0, ct = 0, stack_frames = 0;
    const u8* __restrict__ q = seq;
    uint4* __restrict__ stk = reinterpret_cast<uint4*>(stack);
    const u64* __restrict__ ex_base = scheme;
    bool in_search = false;
    const u64* __restrict__ ex = scheme;
    u32 l_last = 0, u_last = 0;
    u32 nlb = 0, nlbr = 0, nlen = 0, nx = 0, ne = 0, nli = INFO_M, nri = INFO_M;
    // top frame (frame depth-1 of the stack): its node and the mask of children not taken yet
    u32 f_lb = 0, f_lbr = 0, f_len = 0, f_state = 0, f_mask = 0;
    u32 depth = 0;                              // frames on the stack, the top one included
    bool need_child = false;
    bool hit_pending = false;
    u32 hit_rep = 0, hit_idx = 0;
    u64 const hit_key = 0;                      // the ordinals of this kernel's hits are the emission order

    while (true) {
        FM_EMIT_HITS();
        bool const want = !busy && !exhausted;
        u64 const idle = __ballot(want);
        if (idle) {                                                     // wave-uniform
            u32 k = 0xFFFFFFFFu;
            FM_ASSIGN_SEEDS(k);
            if (want) {
                if (k != 0xFFFFFFFFu) {
                    DevSeed const seed = seeds[k];
                    sid = seed.id;
                    q = seq + seed.seq_off;
                    stk = reinterpret_cast<uint4*>(stack + seed.stack_off);
                    len = seed.length;
                    num_searches = seed.frames_searches >> 24;
                    stack_frames = seed.frames_searches & 0xFFFFFFu;
                    ex_base = scheme + seed.scheme_off;
                    srch = 0; ct = 0; hit_idx = 0;
                    busy = true;
                    in_search = false;
                } else exhausted = true;
            }
        }
        if (__all(exhausted && !busy)) break;
        ++n_iter;
        if (queue_done && q_next == q_end) ++n_tail_iter;
        if (!busy) continue;
        ++n_busy_iter;

        if (!in_search) {
            if (srch >= num_searches) { busy = false; continue; }
            ex = ex_base + (u64)srch * len;
            u32 const last_entry = (u32)ex[len - 1];
            l_last = (last_entry >> 20) & 7u;
            u_last = (last_entry >> 23) & 7u;
            ne = 0; nli = INFO_M; nri = INFO_M;
            f_mask = 0;
            depth = 0;
            need_child = false;
            in_search = true;
            if (!fm_begin_search(idx, ex, q, len, nlb, nlbr, nlen, nx)) { in_search = false; ++srch; continue; }
        }

        // ---- one DFS step
        if (need_child) {
            if (f_mask == 0u) {
                // the top frame has no child left (or there is no frame): back to the frame below it
                if (depth <= 1u) { in_search = false; ++srch; continue; }    // search exhausted
                --depth;
                const uint4* __restrict__ g = stk + (depth - 1u) * 4u;
                uint4 const v0 = g[0], v1 = g[1], v2 = g[2], v3 = g[3];
                f_lb = v2.w; f_lbr = v3.x; f_state = v3.z;
                f_mask = v3.w;                                             // never empty: see where frames are made
                // bounds of the children on the other side: prefix sums of their lengths, symbol 0 first
                u32 const o0 = ST_RIGHT(f_state) ? f_lb : f_lbr;
                u32 const o1 = o0 + v1.y;
                u32 const o2 = o1 + v1.z, o3 = o2 + v1.w, o4 = o3 + v2.x, o5 = o4 + v2.y;
                f_len = o5 + v2.z - o0;                                    // the children's rows are the node's
                child[0][lane] = uint4{v0.x, o1, v1.z, 0u};
                child[1][lane] = uint4{v0.y, o2, v1.w, 0u};
                child[2][lane] = uint4{v0.z, o3, v2.x, 0u};
                child[3][lane] = uint4{v0.w, o4, v2.y, 0u};
                child[4][lane] = uint4{v1.x, o5, v2.z, 0u};
                child[5][lane] = uint4{v3.y, o0, v1.y, 0u};
            }
            u32 const ci = (u32)__ffs((int)f_mask) - 1u;
            f_mask &= f_mask - 1u;
            u32 const st = f_state;
            u32 const right = ST_RIGHT(st);
            u32 const px = ST_X(st), pe = ST_E(st);
            u32 info, sym;
            if (ci == 0) { sym = ST_SYM(st); nx = px + 1; ne = pe; info = INFO_M; }
            else if (ci == 11) { sym = 1; nx = px + 1; ne = pe + 1; info = INFO_I; }
            else {
                sym = (ci + 1) >> 1;
                bool const del = ci & 1u;
                nx = del ? px : px + 1;
                ne = pe + 1;
                info = del ? INFO_D : INFO_S;
            }
            uint4 const c = child[sym ? sym - 1u : 5u][lane];              // sym is 1..5 for every child but the match of a '$'
            if (ci == 11) { nlb = f_lb; nlbr = f_lbr; nlen = f_len; }
            else { nlen = c.z; nlb = right ? c.y : c.x; nlbr = right ? c.x : c.y; }
            nli = right ? ST_LI(st) : info;
            nri = right ? info : ST_RI(st);
            need_child = false;
        }

        // ---- inspect node (nlb, nlbr, nlen, nx, ne, nli, nri); nlen > 0 by construction
        if (nx == len) {
            bool const ok_l = nli == INFO_M || nli == INFO_I, ok_r = nri == INFO_M || nri == INFO_I;
            if (ok_l && ok_r && l_last <= ne && ne <= u_last) {
                u32 rep = nlen;
                if (ct + rep > max_hits) rep = max_hits - ct;        // search_n truncates the last cursor
                ct += rep;
                hit_pending = true;                                  // written at the top of the next iteration
                hit_rep = rep;
                if (ct == max_hits) { busy = false; continue; }      // search_n aborts all remaining searches of the seed
            }
            need_child = true;
            continue;
        }
        u32 const sch = (u32)ex[nx];
        u32 const lower = (sch >> 20) & 7u, upper = (sch >> 23) & 7u, right = (sch >> 26) & 1u;
        if (ne > upper) { need_child = true; continue; }
        bool const mismatch_allowed = lower <= ne + 1 && ne + 1 <= upper;
        bool const match_allowed = lower <= ne && ne <= upper;
        if (!mismatch_allowed && !match_allowed) { need_child = true; continue; }

        u32 const next_sym = q[sch & SCH_POS_MASK];
        u32 const lo = right ? nlbr : nlb, other = right ? nlb : nlbr;
        u32 ab[6], cl[6];
        extend_all(idx, idx.occ[right], lo, nlen, ab, cl);
        ++n_ext;

        if (mismatch_allowed) {
            // this node branches: it becomes the top frame. The frame below keeps its place on the stack if it still has children
            // (its mask is brought up to date), else its place is taken.
            if (depth > 0u) {
                if (f_mask != 0u) reinterpret_cast<u32*>(stk + (depth - 1u) * 4u)[15] = f_mask;
                else --depth;
            }
            if (depth >= stack_frames) { atomicOr(&counters[1], 1u); busy = false; continue; }
            u32 const tinfo = right ? nri : nli;
            f_lb = nlb; f_lbr = nlbr; f_len = nlen;
            f_state = st_pack(nx, ne, nli, nri, next_sym, right);
            f_mask = fm_child_mask(cl, next_sym, match_allowed, tinfo == INFO_M || tinfo == INFO_D, tinfo == INFO_M || tinfo == INFO_I);
            uint4* __restrict__ g = stk + depth * 4u;
            g[0] = uint4{ab[1], ab[2], ab[3], ab[4]};
            g[1] = uint4{ab[5], cl[0], cl[1], cl[2]};
            g[2] = uint4{cl[3], cl[4], cl[5], nlb};
            g[3] = uint4{nlbr, ab[0], f_state, f_mask};
            ++depth;
            u32 const o1 = other + cl[0], o2 = o1 + cl[1], o3 = o2 + cl[2], o4 = o3 + cl[3], o5 = o4 + cl[4];
            child[0][lane] = uint4{ab[1], o1, cl[1], 0u};
            child[1][lane] = uint4{ab[2], o2, cl[2], 0u};
            child[2][lane] = uint4{ab[3], o3, cl[3], 0u};
            child[3][lane] = uint4{ab[4], o4, cl[4], 0u};
            child[4][lane] = uint4{ab[5], o5, cl[5], 0u};
            child[5][lane] = uint4{ab[0], other, cl[0], 0u};
            need_child = true;
        } else {
            // only an exact extension is possible: continue in place (no frame)
            if (next_sym > 5u) { need_child = true; continue; }
            u32 clen = cl[0], cabs = ab[0], coth = other;
#pragma unroll
            for (u32 c = 1; c < 6; ++c) {
                coth += c <= next_sym ? cl[c - 1u] : 0u;
                bool const take = c == next_sym;
                clen = take ? cl[c] : clen;
                cabs = take ? ab[c] : cabs;
            }
            if (clen == 0) { need_child = true; continue; }
            nlb = right ? coth : cabs;
            nlbr = right ? cabs : coth;
            if (right) nri = INFO_M; else nli = INFO_M;
            nlen = clen;
            nx = nx + 1;
        }
    }
    { u32 const at = h_next + lane; if (at < h_end && at < hit_cap) hits[at] = DevHit{0xFFFFFFFFu, 0u, 0u, 0u, 0ull}; }
    n_ext = wave_sum_u32(n_ext);
    n_busy_iter = wave_sum_u32(n_busy_iter);
    if (lane == 0) {
        atomicAdd(&counters[2], n_ext); atomicAdd(&counters[6], n_busy_iter);
        atomicAdd(&counters[4], n_iter); atomicMax(&counters[5], n_iter); atomicAdd(&counters[8], n_tail_iter); atomicMax(&counters[9], n_tail_iter);
    }
}
#undef FM_EMIT_HITS
#undef FM_ASSIGN_SEEDS

static u32 fm_seeds_per_wave() {
    static u32 const v = [] { const char* e = getenv("FLX_FM_SEEDS_PER_WAVE"); u32 const x = e ? (u32)strtoul(e, nullptr, 10) : 0u; return x ? x : FM_SEEDS_PER_WAVE; }();
    return v;
}

u32 fm_search_max_keyed_length() { return FM_KEY_MAX_X; }

// the walk in the reference's order (fm_search_ordered_kernel); the default walk is DeviceApi::search_filtered (flx_search.hip)
int DeviceApi::search(void* stream, const DevIndex& idx, const u8* d_seq, const u64* d_scheme, const DevSeed* d_seeds, u32 n_seeds,
                      u32 max_hits_per_seed, DevFrame* d_stack, DevHit* d_hits, u32 hit_cap, u32* d_counters, u32* d_seed_cnt) {
    if (n_seeds == 0) return 0;
    if (!d_stack) return (int)hipErrorInvalidValue;
    u32 const spw = fm_seeds_per_wave();
    dim3 const grid(std::min<u32>((n_seeds + spw - 1) / spw, FM_MAX_WAVES));
    hipLaunchKernelGGL(fm_search_ordered_kernel, grid, dim3(64), 0, (hipStream_t)stream, idx, d_seq, d_scheme, d_seeds, n_seeds,
                       max_hits_per_seed, d_stack, d_hits, hit_cap, d_counters, d_seed_cnt);
    return (int)hipGetLastError();
}

// Scans (exclusive sums) of the anchor selection's count arrays in two launches without any waiting between blocks: every block reduces
// its tile, then every block scans its tile again behind the reduction of the tiles before it (a few hundred words it adds up
// itself). The library scans are single-pass with decoupled look-back: their blocks spin on their predecessors' results, which on a
// GPU filled with other lanes' kernels made a 1.2 M-element scan take a millisecond and burn issue slots meanwhile (11 % of the
// kernel time of a run went into them); an onesweep radix sort in place of rocprim's merge sort for the same reason cost 10 % of the
// throughput.
constexpr u32 SCAN_ITEMS = 8, SCAN_TILE = 256 * SCAN_ITEMS;
template <bool MAX> __device__ __forceinline__ u32 scan_op(u32 a, u32 b) { return MAX ? max(a, b) : a + b; }
template <bool MAX>
__device__ __forceinline__ u32 block_reduce_256(u32 v, u32* __restrict__ lds4) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = scan_op<MAX>(v, (u32)__shfl_xor((int)v, off));
    if (lane_id() == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    u32 const r = scan_op<MAX>(scan_op<MAX>(lds4[0], lds4[1]), scan_op<MAX>(lds4[2], lds4[3]));
    __syncthreads();
    return r;
}
template <bool MAX>
__global__ void __launch_bounds__(256) vr_scan_reduce_kernel(const u32* __restrict__ in, u32 n, u32* __restrict__ tile_total) {
    __shared__ u32 lds4[4];
    u32 const base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    u32 v = 0;
#pragma unroll
    for (u32 j = 0; j < SCAN_ITEMS; ++j) if (base + j < n) v = scan_op<MAX>(v, in[base + j]);
    u32 const total = block_reduce_256<MAX>(v, lds4);
    if (threadIdx.x == 0) tile_total[blockIdx.x] = total;
}
template <bool MAX, bool EXCLUSIVE = false>
__global__ void __launch_bounds__(256) vr_scan_apply_kernel(const u32* __restrict__ in, u32 n, const u32* __restrict__ tile_total, u32* __restrict__ out) {
    __shared__ u32 lds4[4];
    __shared__ u32 wave_total[4];
    u32 before = 0;                                       // the tiles before this one
    for (u32 t = threadIdx.x; t < blockIdx.x; t += 256u) before = scan_op<MAX>(before, tile_total[t]);
    before = block_reduce_256<MAX>(before, lds4);
    u32 const base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    u32 item[SCAN_ITEMS];
    u32 run = 0;
#pragma unroll
    for (u32 j = 0; j < SCAN_ITEMS; ++j) {
        if (EXCLUSIVE) item[j] = run;
        run = scan_op<MAX>(run, base + j < n ? in[base + j] : 0u);
        if (!EXCLUSIVE) item[j] = run;
    }
    // exclusive scan of the threads' totals: within the wave by shuffles, across the four waves through LDS
    u32 incl = run;
#pragma unroll
    for (u32 off = 1; off < 64u; off <<= 1) {
        u32 const up = (u32)__shfl_up((int)incl, off);
        if (lane_id() >= off) incl = scan_op<MAX>(incl, up);
    }
    if (lane_id() == 63u) wave_total[threadIdx.x >> 6] = incl;
    __syncthreads();
    u32 prefix = before;
    for (u32 w = 0; w < (threadIdx.x >> 6); ++w) prefix = scan_op<MAX>(prefix, wave_total[w]);
    u32 const excl = (u32)__shfl_up((int)incl, 1);
    if (lane_id() > 0) prefix = scan_op<MAX>(prefix, excl);
#pragma unroll
    for (u32 j = 0; j < SCAN_ITEMS; ++j) if (base + j < n) out[base + j] = scan_op<MAX>(prefix, item[j]);
}
static void exclusive_sum(hipStream_t s, const u32* in, u32* out, u32 n, u32* tile_total) {
    unsigned const tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL((vr_scan_reduce_kernel<false>), dim3(tiles), dim3(256), 0, s, in, n, tile_total);
    hipLaunchKernelGGL((vr_scan_apply_kernel<false, true>), dim3(tiles), dim3(256), 0, s, in, n, tile_total, out);
}

// ================================================================================================ K1b: anchor selection
// hits -> per-seed segments in emission order (a scan over the seeds' hit counts + a scatter by the ordinal each hit carries),
// then one thread per seed does what search.cpp:190-318 does with the seed's groups: hard cap, group order, rows round robin,
// locate through the suffix array, buckets per reference sorted by position, useless anchors erased (search.cpp:352-389).
// Handled here: seeds with at most SEL_MAX groups whose rows all fit under the soft cap and SEL_MAX. The two std::sort calls of
// the reference (groups by (count, errors), a bucket's anchors by position) are reproduced step for step (std_sort_emulated):
// their comparators tie (the same row reached through two groups gives two anchors of equal position) and the order of equal
// elements shows in the result. Every seed not handled is flagged and goes through the host code.
constexpr u32 SEL_MAX = 64;
struct SelStat { u8 useful, raw, flag, excluded; u32 excluded_soft; };      // flag 1: the host selects this seed's anchors; = DevSelStat

__global__ void __launch_bounds__(256) hit_scatter_kernel(const DevHit* __restrict__ hits, const u32* __restrict__ counters, u32 hit_cap,
                                                          const u32* __restrict__ offset, DevHit* __restrict__ grouped) {
    u32 const n_slots = min(counters[0], hit_cap);
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += gridDim.x * blockDim.x) {
        DevHit h = hits[i];
        if (h.seed == 0xFFFFFFFFu) continue;
        u32 const ordinal = h.errors >> 8;
        h.errors &= 0xFFu;
        grouped[offset[h.seed] + ordinal] = h;
    }
}

struct SelGroup { u32 lb, len, errors; };
struct SelAnchor { u32 pos; u32 ref; u32 errors; };           // pos within its reference sequence (the text has fewer than 2^32 symbols)
struct SelKey { u32 lo, hi; };
__device__ __forceinline__ bool sel_key_less(u64 k, SelKey const& o) { return k < ((u64)o.lo | ((u64)o.hi << 32)); }

// the selection of one seed whose groups (cnt <= CAP) hold `total` <= CAP rows; returns false when the seed has to go to the host.
// Working storage from the caller (a thread's indexed private arrays would live in scratch memory: round 3 had 720 B per lane there):
// g: CAP groups; w: CAP SelKeys while the groups are put into emission order, CAP SelAnchors afterwards (the two do not overlap in
// time); stacks: 48 ints when CAP > 16 (std::sort's partitions)
template <u32 CAP, bool WRITE>
__device__ __forceinline__ bool select_seed(const DevHit* __restrict__ groups, u32 cnt, u32 total, const u32* __restrict__ sa, u32 n_text,
                                            const u64* __restrict__ seq_start, u32 n_ref, u32 erase, u32 sid, SelStat& st, u32& produced,
                                            DevOutAnchor* __restrict__ out, u32 at, u32 out_cap, SelGroup* g, void* w, int* stacks) {
    // the groups in search_n's emission order (the keys of fm_search_kernel; all 0 from the ordered kernel, whose hits are in it already:
    // the insertion sort is stable), then ordered by (count, errors) (search.cpp:200-212)
    {
        SelKey* const key = static_cast<SelKey*>(w);
        for (u32 i = 0; i < cnt; ++i) {
            DevHit const h = groups[i];
            u32 j = i;
            while (j > 0 && sel_key_less(h.key, key[j - 1])) { key[j] = key[j - 1]; g[j] = g[j - 1]; --j; }
            key[j] = SelKey{(u32)h.key, (u32)(h.key >> 32)};
            g[j] = SelGroup{h.lb, h.len, h.errors};
        }
    }
    auto less_g = [](SelGroup const& x, SelGroup const& y) { return x.len != y.len ? x.len < y.len : x.errors < y.errors; };
    if (CAP <= 16u) insertion_sort_emulated(g, (int)cnt, less_g);
    else if (!std_sort_emulated(g, (int)cnt, less_g, stacks)) return false;
    // rows round robin over the groups (search.cpp:239-272), located
    SelAnchor* const an = static_cast<SelAnchor*>(w);
    u32 kept = 0;
    bool bad = false;
    // (`total` = the rows to keep: all of them, or the soft cap when the seed has more: the cycle then stops in the middle of a round)
    for (u32 round = 0; kept < total; ++round)
        for (u32 gi = 0; gi < cnt && kept < total; ++gi) {
            if (g[gi].len <= round) continue;
            u32 const row = g[gi].lb + round;
            u64 const p = row < n_text ? sa[row] : 0xFFFFFFFFull;
            if (p >= n_text) bad = true;
            u32 r = 0;
            if (n_ref > 1) {                                   // last sequence that starts at or before p
                u32 lo = 0, hi = n_ref;
                while (hi - lo > 1) { u32 const mid = (lo + hi) >> 1; if (seq_start[mid] <= p) lo = mid; else hi = mid; }
                r = lo;
            }
            an[kept++] = SelAnchor{(u32)(p - seq_start[r]), r, g[gi].errors};
        }
    if (bad) return false;                                  // the host reports the error
    // buckets per reference in id order, each keeping the order of selection (search.cpp:78-100, 304-318)
    for (u32 i = 1; i < kept; ++i) {
        SelAnchor const v = an[i];
        u32 j = i;
        while (j > 0 && v.ref < an[j - 1].ref) { an[j] = an[j - 1]; --j; }
        an[j] = v;
    }
    u64 gone = 0;                                           // bit i: anchor i erased
    if (erase) {
        u32 b0 = 0;
        while (b0 < kept) {                                 // one bucket = one reference (search.cpp:352-389)
            u32 b1 = b0;
            while (b1 < kept && an[b1].ref == an[b0].ref) ++b1;
            auto less_p = [](SelAnchor const& x, SelAnchor const& y) { return x.pos < y.pos; };
            if (CAP <= 16u) insertion_sort_emulated(an + b0, (int)(b1 - b0), less_p);
            else if (!std_sort_emulated(an + b0, (int)(b1 - b0), less_p, stacks)) return false;
            // an erased anchor compares with "infinitely many" errors
            auto better = [&](u32 a, u32 b) {
                u64 const ea = (gone >> a) & 1 ? ~0ull : (u64)an[a].errors, eb = (gone >> b) & 1 ? ~0ull : (u64)an[b].errors;
                u64 const d = an[a].pos < an[b].pos ? an[b].pos - an[a].pos : an[a].pos - an[b].pos;
                return ea <= eb && d <= eb - ea;
            };
            for (u32 cur = b0; cur + 1 < b1;) {
                u32 other = cur + 1;
                while (other < b1 && better(cur, other)) { gone |= 1ull << other; ++other; }
                if (other < b1 && better(other, cur)) gone |= 1ull << cur;
                cur = other;
            }
            b0 = b1;
        }
    }
    st.raw = (u8)kept;
    for (u32 i = 0; i < kept; ++i)
        if (!((gone >> i) & 1)) {
            if (WRITE && at + produced < out_cap) out[at + produced] = DevOutAnchor{sid, 0u, an[i].ref, an[i].errors, (u64)an[i].pos};
            ++produced;
        }
    st.useful = (u8)produced;
    return true;
}

// Every seed's class: nothing to select (no hit / over the hard cap / left to the host: its statistics are final here), light (at
// most SEL_LIGHT groups and rows: one thread per seed, seed_select_kernel) or heavy (up to SELW_MAX_GROUPS groups, any number of
// rows up to the hard cap of which the soft cap's worth, at most SEL_MAX, is kept: one wave per seed, seed_select_wave_kernel).
// rows[sid] = the slots the seed gets in the sparse anchor list. Light and heavy seeds go on two lists (wave-aggregated appends;
// the order of a list does not matter, every seed writes to its own slots).
constexpr u32 SEL_LIGHT = 8;
constexpr u32 SELW_MAX_GROUPS = 512, SELW_FEW_GROUPS = 64;
__global__ void __launch_bounds__(256) seed_rows_kernel(const DevHit* __restrict__ grouped, const u32* __restrict__ hit_offset, u32 n_seeds,
                                                        u32 hard_cap, u32 soft_cap, u32* __restrict__ rows, SelStat* __restrict__ stat,
                                                        u32* __restrict__ n_out, u32* __restrict__ lists, u32* __restrict__ list_counts) {
    u32 const sid = blockIdx.x * blockDim.x + threadIdx.x;
    u32 cls = 0;                                             // 1 light, 2 heavy (a wave, up to SELW_FEW_GROUPS groups), 3 heavy with more groups
    if (sid < n_seeds) {
        u32 const g0 = hit_offset[sid], cnt = hit_offset[sid + 1] - g0;
        SelStat st{0, 0, 0, 0, 0};
        u32 total = 0;
        if (cnt > hard_cap) st.excluded = 1;                // every group has at least one row: over the hard cap whatever the rows are
        else if (cnt > SELW_MAX_GROUPS) st.flag = 1;        // more groups than the wave kernel's arrays hold: the host
        else if (cnt > 0) {
            u32 all = 0;
            for (u32 i = 0; i < cnt; ++i) all += min(grouped[g0 + i].len, 0x1000000u);
            total = min(all, soft_cap);                      // rows kept (search.cpp:239-272 stops at the soft cap)
            if (all > hard_cap) st.excluded = 1;
            else if (total > SEL_MAX) st.flag = 1;           // a soft cap beyond the anchor arrays: the host
            else { cls = (cnt <= SEL_LIGHT && total <= SEL_LIGHT) ? 1u : cnt <= SELW_FEW_GROUPS ? 2u : 3u; st.excluded_soft = all - total; }
        }
        rows[sid] = cls ? total : 0u;
        if (!cls) { stat[sid] = st; n_out[sid] = 0; }
        else stat[sid].excluded_soft = st.excluded_soft;     // (the select kernels fill in the rest)
    }
#pragma unroll
    for (u32 c = 1; c <= 3; ++c) {
        u64 const m = __ballot(cls == c);
        if (!m) continue;
        u32 base = 0;
        if (lane_id() == 0) base = atomicAdd(&list_counts[c - 1], (u32)__popcll(m));
        base = (u32)__builtin_amdgcn_readfirstlane((int)base);
        if (cls == c) lists[(c - 1) * n_seeds + base + (u32)__popcll(m & ((1ull << lane_id()) - 1ull))] = sid;
    }
}

// the seeds of one list: their anchors to their slots of the sparse list (row_offset), n_out says how many. A thread's arrays are in LDS
// for the light seeds (CAP groups + CAP keys / anchors, 12 B each, a word of padding per thread against bank conflicts); a seed with one
// group of one row - most seeds of a read with one locus - takes neither.
template <u32 CAP>
__global__ void __launch_bounds__(64, 4) seed_select_kernel(const u32* __restrict__ list, const u32* __restrict__ list_count,
                                                         const DevHit* __restrict__ grouped, const u32* __restrict__ hit_offset,
                                                         const u32* __restrict__ sa, u32 n_text, const u64* __restrict__ seq_start, u32 n_ref,
                                                         u32 erase, SelStat* __restrict__ stat, u32* __restrict__ n_out,
                                                         const u32* __restrict__ row_offset, const u32* __restrict__ rows,
                                                         DevOutAnchor* __restrict__ sparse, u32 sparse_cap) {
    constexpr u32 IN_LDS = CAP <= 16u ? 1u : 0u;
    constexpr u32 STRIDE = 6u * CAP + 1u;                       // words per thread
    __shared__ u32 s_pool[IN_LDS ? 64u * STRIDE : 1u];
    SelGroup g_priv[IN_LDS ? 1u : CAP];
    SelAnchor w_priv[IN_LDS ? 1u : CAP];
    int stacks_priv[IN_LDS ? 1 : 48];
    SelGroup* const g = IN_LDS ? reinterpret_cast<SelGroup*>(s_pool + threadIdx.x * STRIDE) : g_priv;
    void* const w = IN_LDS ? static_cast<void*>(s_pool + threadIdx.x * STRIDE + 3u * CAP) : static_cast<void*>(w_priv);
    u32 const n = *list_count;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        u32 const sid = list[i];
        u32 const g0 = hit_offset[sid], cnt = hit_offset[sid + 1] - g0;
        SelStat st{0, 0, 0, 0, stat[sid].excluded_soft};
        u32 produced = 0;
        bool ok;
        if (cnt == 1u && rows[sid] == 1u) {
            // one group, one row: every order and strategy keeps exactly it
            DevHit const h = grouped[g0];
            u64 const p = h.lb < n_text ? sa[h.lb] : 0xFFFFFFFFull;
            ok = p < n_text;
            if (ok) {
                u32 r = 0;
                if (n_ref > 1) { u32 lo = 0, hi = n_ref; while (hi - lo > 1) { u32 const mid = (lo + hi) >> 1; if (seq_start[mid] <= p) lo = mid; else hi = mid; } r = lo; }
                u32 const at = row_offset[sid];
                if (at < sparse_cap) sparse[at] = DevOutAnchor{sid, 0u, r, h.errors, p - seq_start[r]};
                produced = 1; st.raw = 1; st.useful = 1;
            }
        } else ok = select_seed<CAP, true>(grouped + g0, cnt, rows[sid], sa, n_text, seq_start, n_ref, erase, sid, st, produced, sparse, row_offset[sid], sparse_cap, g, w, stacks_priv);
        if (!ok) {
            st = SelStat{0, 0, 1, 0, 0};
            produced = 0;
        }
        stat[sid] = st;
        n_out[sid] = st.flag ? 0u : produced;
    }
}

// One wave per heavy seed: what select_seed does, with the parts that parallelise spread over the lanes - the emission order and
// the stable orders as ranks (element i goes to the number of elements in front of it), the rows' round robin as one ballot per
// round, SA and reference lookups one per lane - and the parts that are std::sort's own (more than 16 elements: introsort, whose
// order of equal elements has to be reproduced step by step) and the erase sweep on one lane over LDS arrays. A round-2 profile had
// the thread-per-seed form of this at 3 ms per launch on 2.3 KB of scratch per thread (profiles/r03_k1v2_kernel_stats.csv).
// (MAXG: groups a seed of the list may have; 64 groups keep the block at 2.7 KB of LDS, which finds room on a CU next to the DP
// kernels of other lanes; the few seeds with up to 512 groups take the 17-KB form)
template <u32 MAXG>
__global__ void __launch_bounds__(64, MAXG <= 64 ? 4 : 2) seed_select_wave_kernel(const u32* __restrict__ list, const u32* __restrict__ list_count,
                                                              const DevHit* __restrict__ grouped, const u32* __restrict__ hit_offset,
                                                              const u32* __restrict__ sa, u32 n_text, const u64* __restrict__ seq_start, u32 n_ref,
                                                              u32 erase, SelStat* __restrict__ stat, u32* __restrict__ n_out,
                                                              const u32* __restrict__ row_offset, const u32* __restrict__ rows,
                                                              DevOutAnchor* __restrict__ sparse, u32 sparse_cap) {
    __shared__ u64 s_key[MAXG];
    __shared__ SelGroup s_a[MAXG], s_b[MAXG];
    __shared__ u32 s_row[SEL_MAX], s_err[SEL_MAX];
    __shared__ SelAnchor s_an[SEL_MAX];
    __shared__ u32 s_flag[4];                 // [0] a sort gave up (host), [1..2] erased anchors (bits)
    __shared__ int s_stacks[48];              // std::sort's partitions still to do (one lane sorts)
    u32 const lane = lane_id();
    u64 const below = (1ull << lane) - 1ull;
    u32 const n = *list_count;
    for (u32 li = blockIdx.x; li < n; li += gridDim.x) {
        u32 const sid = list[li];
        u32 const g0 = hit_offset[sid], cnt = hit_offset[sid + 1] - g0, total = rows[sid];
        __syncthreads();
        for (u32 i = lane; i < cnt; i += 64u) { DevHit const h = grouped[g0 + i]; s_key[i] = h.key; s_a[i] = SelGroup{h.lb, h.len, h.errors}; }
        if (lane < 4u) s_flag[lane] = 0u;
        __syncthreads();
        // ---- emission order (the keys of fm_search; equal keys keep their order), then std::sort by (count, errors): up to 16
        //      elements that is an insertion sort, i.e. stable
        for (u32 i = lane; i < cnt; i += 64u) {
            u64 const k = s_key[i];
            u32 r = 0;
            for (u32 j = 0; j < cnt; ++j) { u64 const kj = s_key[j]; r += (kj < k || (kj == k && j < i)) ? 1u : 0u; }
            s_b[r] = s_a[i];
        }
        __syncthreads();
        auto less_g = [](SelGroup const& x, SelGroup const& y) { return x.len != y.len ? x.len < y.len : x.errors < y.errors; };
        if (cnt <= 16u) {
            if (lane < cnt) {
                SelGroup const me = s_b[lane];
                u32 r = 0;
                for (u32 j = 0; j < cnt; ++j) { SelGroup const o = s_b[j]; r += (less_g(o, me) || (!less_g(me, o) && j < lane)) ? 1u : 0u; }
                s_a[r] = me;
            }
        } else {
            if (lane == 0u && !std_sort_emulated(s_b, (int)cnt, less_g, s_stacks)) s_flag[0] = 1u;
            __syncthreads();
            for (u32 i = lane; i < cnt; i += 64u) s_a[i] = s_b[i];
        }
        __syncthreads();
        // ---- rows round robin over the groups (search.cpp:239-272): row lb + round of every group that still has one, until
        //      `total` are kept
        u32 kept = 0;
        for (u32 round = 0; kept < total; ++round) {
            bool any = false;
            for (u32 base = 0; base < cnt && kept < total; base += 64u) {
                u32 const i = base + lane;
                bool const alive = i < cnt && s_a[i].len > round;
                u64 const m = __ballot(alive);
                if (!m) continue;
                any = true;
                u32 const slot = kept + (u32)__popcll(m & below);
                if (alive && slot < total) { s_row[slot] = s_a[i].lb + round; s_err[slot] = s_a[i].errors; }
                kept = min(total, kept + (u32)__popcll(m));
            }
            if (!any) break;
        }
        __syncthreads();
        // ---- locate, reference, position; buckets per reference in id order, each keeping the order of selection
        bool const mine = lane < kept;
        u64 p = 0;
        u32 ref = 0, err = 0;
        bool bad = false;
        if (mine) {
            u32 const row = s_row[lane];
            err = s_err[lane];
            p = row < n_text ? sa[row] : 0xFFFFFFFFull;
            bad = p >= n_text;
            if (!bad && n_ref > 1) {
                u32 lo = 0, hi = n_ref;
                while (hi - lo > 1) { u32 const mid = (lo + hi) >> 1; if (seq_start[mid] <= p) lo = mid; else hi = mid; }
                ref = lo;
            }
        }
        if (__any(bad)) s_flag[0] = 1u;
        u32 r3 = 0;
        for (u32 j = 0; j < kept; ++j) { u32 const rj = (u32)__shfl((int)ref, (int)j); r3 += (rj < ref || (rj == ref && j < lane)) ? 1u : 0u; }
        if (mine && !bad) s_an[r3] = SelAnchor{(u32)(p - seq_start[ref]), ref, err};
        __syncthreads();
        // ---- erase_useless_anchors (search.cpp:352-389) bucket by bucket: std::sort by position, then the sweep
        if (erase && lane == 0u && s_flag[0] == 0u) {
            u64 gone = 0;
            u32 b0 = 0;
            while (b0 < kept) {
                u32 b1 = b0;
                while (b1 < kept && s_an[b1].ref == s_an[b0].ref) ++b1;
                if (!std_sort_emulated(s_an + b0, (int)(b1 - b0), [](SelAnchor const& x, SelAnchor const& y) { return x.pos < y.pos; }, s_stacks)) { s_flag[0] = 1u; break; }
                auto better = [&](u32 a, u32 b) {          // an erased anchor compares with "infinitely many" errors
                    u64 const ea = (gone >> a) & 1 ? ~0ull : (u64)s_an[a].errors, eb = (gone >> b) & 1 ? ~0ull : (u64)s_an[b].errors;
                    u64 const d = s_an[a].pos < s_an[b].pos ? s_an[b].pos - s_an[a].pos : s_an[a].pos - s_an[b].pos;
                    return ea <= eb && d <= eb - ea;
                };
                for (u32 cur = b0; cur + 1 < b1;) {
                    u32 other = cur + 1;
                    while (other < b1 && better(cur, other)) { gone |= 1ull << other; ++other; }
                    if (other < b1 && better(other, cur)) gone |= 1ull << cur;
                    cur = other;
                }
                b0 = b1;
            }
            s_flag[1] = (u32)gone;
            s_flag[2] = (u32)(gone >> 32);
        }
        __syncthreads();
        bool const to_host = s_flag[0] != 0u;
        u64 const gone = (u64)s_flag[1] | ((u64)s_flag[2] << 32);
        bool const keep = mine && !to_host && !((gone >> lane) & 1ull);
        u64 const km = __ballot(keep);
        u32 const produced = (u32)__popcll(km);
        if (keep) {
            u32 const at = row_offset[sid] + (u32)__popcll(km & below);
            SelAnchor const a = s_an[lane];
            if (at < sparse_cap) sparse[at] = DevOutAnchor{sid, 0u, a.ref, a.errors, (u64)a.pos};
        }
        if (lane == 0u) {
            SelStat st{(u8)produced, (u8)kept, 0, 0, stat[sid].excluded_soft};
            if (to_host) st = SelStat{0, 0, 1, 0, 0};
            stat[sid] = st;
            n_out[sid] = to_host ? 0u : produced;
        }
    }
}

__global__ void __launch_bounds__(256) seed_compact_kernel(const DevOutAnchor* __restrict__ sparse, const u32* __restrict__ row_offset,
                                                           const u32* __restrict__ n_out, const u32* __restrict__ out_offset, u32 n_seeds,
                                                           DevOutAnchor* __restrict__ out, u32 out_cap) {
    u32 const sid = blockIdx.x * blockDim.x + threadIdx.x;
    if (sid >= n_seeds) return;
    u32 const n = n_out[sid], from = row_offset[sid], to = out_offset[sid];
    for (u32 i = 0; i < n; ++i) if (to + i < out_cap) out[to + i] = sparse[from + i];
}

size_t DeviceApi::select_scan_bytes(u32 n_seeds) {
    return ((size_t)(n_seeds + 1) / SCAN_TILE + 1) * sizeof(u32);        // tile totals of exclusive_sum
}

int DeviceApi::select(void* stream, const DevHit* d_hits, const u32* d_counters, u32 hit_cap, u32* d_seed_cnt, u32* d_hit_offset,
                      DevHit* d_grouped, u32 n_seeds, const DevIndex& idx, const u64* d_seq_start, u32 n_ref, u32 hard_cap, u32 soft_cap,
                      bool erase, void* d_stat, u32* d_n_out, u32* d_out_offset, DevOutAnchor* d_out, u32 out_cap, u32* d_rows,
                      u32* d_row_offset, DevOutAnchor* d_sparse, u32 sparse_cap, void* d_scan_tmp, size_t scan_bytes, u32* d_lists) {
    if (n_seeds == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    // d_seed_cnt, d_rows and d_n_out have n_seeds + 1 entries, the last one zero: the scans end with the totals
    hipError_t e = hipSuccess;
    exclusive_sum(s, d_seed_cnt, d_hit_offset, n_seeds + 1, (u32*)d_scan_tmp);
    hipLaunchKernelGGL(hit_scatter_kernel, dim3(2048), dim3(256), 0, s, d_hits, d_counters, hit_cap, d_hit_offset, d_grouped);
    SelStat* const stat = reinterpret_cast<SelStat*>(d_stat);
    u32* const list_counts = d_lists + 3 * (size_t)n_seeds;
    if ((e = hipMemsetAsync(list_counts, 0, 12, s)) != hipSuccess) return (int)e;
    hipLaunchKernelGGL(seed_rows_kernel, dim3((n_seeds + 255) / 256), dim3(256), 0, s, d_grouped, d_hit_offset, n_seeds, hard_cap, soft_cap, d_rows,
                       stat, d_n_out, d_lists, list_counts);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    exclusive_sum(s, d_rows, d_row_offset, n_seeds + 1, (u32*)d_scan_tmp);
    // grids sized for the usual shares (a quarter of the seeds light, a per cent heavy); the kernels loop over their lists
    hipLaunchKernelGGL((seed_select_kernel<SEL_LIGHT>), dim3(std::max(1u, (n_seeds / 4 + 63) / 64)), dim3(64), 0, s, d_lists, list_counts, d_grouped, d_hit_offset,
                       idx.sa, idx.n, d_seq_start, n_ref, erase ? 1u : 0u, stat, d_n_out, d_row_offset, d_rows, d_sparse, sparse_cap);
    // (one wave per heavy seed)
    hipLaunchKernelGGL((seed_select_wave_kernel<SELW_FEW_GROUPS>), dim3(std::max(1u, std::min(n_seeds / 8u + 1u, 16384u))), dim3(64), 0, s, d_lists + n_seeds, list_counts + 1, d_grouped,
                       d_hit_offset, idx.sa, idx.n, d_seq_start, n_ref, erase ? 1u : 0u, stat, d_n_out, d_row_offset, d_rows, d_sparse, sparse_cap);
    hipLaunchKernelGGL((seed_select_wave_kernel<SELW_MAX_GROUPS>), dim3(std::max(1u, std::min(n_seeds / 256u + 1u, 2048u))), dim3(64), 0, s, d_lists + 2 * (size_t)n_seeds, list_counts + 2, d_grouped,
                       d_hit_offset, idx.sa, idx.n, d_seq_start, n_ref, erase ? 1u : 0u, stat, d_n_out, d_row_offset, d_rows, d_sparse, sparse_cap);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    exclusive_sum(s, d_n_out, d_out_offset, n_seeds + 1, (u32*)d_scan_tmp);
    hipLaunchKernelGGL(seed_compact_kernel, dim3((n_seeds + 255) / 256), dim3(256), 0, s, d_sparse, d_row_offset, d_n_out, d_out_offset, n_seeds,
                       d_out, out_cap);
    return (int)hipGetLastError();
}

// ================================================================================================ K2: locate
__global__ void __launch_bounds__(256) fm_locate_kernel(const u32* __restrict__ sa, u32 n_text, const u32* __restrict__ rows, u32 n,
                                                        u32* __restrict__ out) {
    u32 const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 const r = rows[i];
    out[i] = r < n_text ? sa[r] : 0xFFFFFFFFu;
}

int DeviceApi::locate(void* stream, const DevIndex& idx, const u32* d_rows, u32 n, u32* d_out) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(fm_locate_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, idx.sa, idx.n, d_rows, n, d_out);
    return (int)hipGetLastError();
}

// launch shapes of the DP kernels (K3/K4 below): words per lane a kernel is instantiated for; FLX_NO_BAND=1 computes whole matrices
static const u32 kWordsPerLane[] = {1, 2, 3, 4, 5, 6, 8, 13, 25};

static bool use_band() {
    static int const v = getenv("FLX_NO_BAND") ? 0 : 1;
    return v != 0;
}

// ================================================================================================ K3/K4: edit-distance DP
// Myers/Hyyro bit-vector columns, semi-global (free reference ends). One job occupies G = lanes_per_job consecutive lanes,
// lane g owns W consecutive 64-row words of the column. Lanes run skewed: at step t lane g computes reference column
// t - g, so the carries of column c travel lane g -> g+1 between step t and t+1 (wave_shr DPP) and every lane is busy
// after the G-step fill. With TRACE the horizontal-positive and vertical-positive delta words of every (column, word) are
// stored in step-major ("skewed") order so that each step's stores of a job are one contiguous, fully coalesced run.
template <int W, bool TRACE>
__global__ void __launch_bounds__(64) ed_align_kernel(const u8* __restrict__ text, const u64* __restrict__ peq,
                                                      const DevAlignJob* __restrict__ jobs, u32 n_jobs, u32 log2_g,
                                                      u64* __restrict__ trace, DevAlignOut* __restrict__ out, u16* __restrict__ lastrow) {
    extern __shared__ __attribute__((aligned(16))) u64 lds_eq[];     // [6 symbols][64 lanes][W words]
    u32 const lane = lane_id();
    u32 const G = 1u << log2_g;
    u32 const lg = lane & (G - 1u);
    u32 const jobs_per_wave = 64u >> log2_g;
    u32 const job_id = blockIdx.x * jobs_per_wave + (lane >> log2_g);
    bool const valid = job_id < n_jobs;
    DevAlignJob job;
    if (valid) job = jobs[job_id];
    else { job.ref_off = 0; job.q_off = 0; job.trace_off = 0; job.n = 0; job.m = 1; job.k = 0; job.out_index = 0; }

    u32 const nw = (job.m + 63u) >> 6;                  // words in a column
    u32 const L = (nw + W - 1u) / W;                    // lanes that own words
    bool const owner = valid && lg < L;

    // ---- equality masks of this lane's words: funnel-shift of the pool-wide Peq planes to the job's row origin
    {
        u64 const a = job.q_off >> 6;
        u32 const sh = (u32)(job.q_off & 63u);
#pragma unroll
        for (int w = 0; w < W; ++w) {
            u32 const gw = lg * W + w;
#pragma unroll
            for (u32 s = 0; s < 6; ++s) {
                u64 v = 0;
                if (owner && gw < nw) {
                    u64 const lo = peq[(a + gw) * 6 + s];
                    u64 const hi = peq[(a + gw + 1) * 6 + s];
                    v = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
                    u32 const rows_left = job.m - gw * 64u;             // rows of this word that belong to the query
                    if (rows_left < 64u) v &= (1ull << rows_left) - 1ull;
                }
                lds_eq[(s * 64u + lane) * W + w] = v;
            }
        }
    }
    __syncthreads();

    u64 vp[W], vn[W];
#pragma unroll
    for (int w = 0; w < W; ++w) { vp[w] = ~0ull; vn[w] = 0ull; }

    bool const last_lane = owner && lg == L - 1u;
    u32 const w_last = (nw - 1u) - (L - 1u) * W;        // local index of the word that holds row m-1
    u64 const last_bit = 1ull << ((job.m - 1u) & 63u);
    u32 score = job.m, best = job.m, best_col = 0;

    u32 const my_steps = owner ? job.n + L - 1u : 0u;
    u32 const t_max = wave_max_u32(my_steps);

    // per-lane reference stream: this lane needs p[t] at step t (column t - lg)
    const u8* __restrict__ p = text + job.ref_off - lg;
    auto load8 = [&](u32 t) -> u64 {
        // unaligned 8-byte read assembled from two aligned ones (never faults: text has TEXT_PAD bytes on both sides)
        const u8* const addr = p + t;
        uintptr_t const ai = (uintptr_t)addr;
        const u64* const base = reinterpret_cast<const u64*>(ai & ~(uintptr_t)7);
        u32 const shb = (u32)(ai & 7u) * 8u;
        u64 const lo = base[0], hi = base[1];
        return shb ? (lo >> shb) | (hi << (64u - shb)) : lo;
    };
    u32 const my_last = owner ? job.n + lg : 0u;         // steps [lg, n+lg) are this lane's columns
    u64 queue = 0, next_queue = 0;
    if (owner) { queue = load8(0); if (8 < my_last) next_queue = load8(8); }

    u32 cout = 0;
    u64 const trace_lane_base = job.trace_off + (u64)lg * W;
    u64 const trace_step_stride = (u64)L * W;

    for (u32 t = 0; t < t_max; ++t) {
        if ((t & 7u) == 0u && t > 0u) {
            queue = next_queue;
            if (owner && t + 8u < my_last) next_queue = load8(t + 8u);
        }
        u32 const sym = (u32)(queue & 7ull);
        queue >>= 8;
        u32 cin = from_prev_lane(cout);
        if (lg == 0u) cin = 0u;
        bool const active = owner && t >= lg && t < my_last;
        if (active) {
            u64 c_d0 = cin & 1u, c_hp = (cin >> 1) & 1u, c_hn = (cin >> 2) & 1u;
            const u64* __restrict__ eqp = &lds_eq[(sym * 64u + lane) * W];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                u64 const eq = eqp[w];
                u64 const pv = vp[w], mv = vn[w];
                u64 const x = eq | mv;
                u64 const t1 = pv + (x & pv);
                u64 const tt = t1 + c_d0;
                u64 const carry = (u64)(t1 < pv) | (u64)(tt < t1);
                u64 const d0 = (tt ^ pv) | x;
                u64 const hn = pv & d0;
                u64 const hp = mv | ~(pv | d0);
                u64 const xh = (hp << 1) | c_hp;
                u64 const nvn = xh & d0;
                u64 const nvp = (hn << 1) | ~(xh | d0) | c_hn;
                c_d0 = carry;
                c_hp = hp >> 63;
                c_hn = hn >> 63;
                vn[w] = nvn;
                vp[w] = nvp;
                if (TRACE) {
                    ulonglong2 v;
                    v.x = hp;
                    v.y = nvp;
                    *reinterpret_cast<ulonglong2*>(trace + 2ull * (trace_lane_base + (u64)t * trace_step_stride + (u64)w)) = v;
                }
                if (last_lane && (u32)w == w_last) {
                    score += (hp & last_bit) ? 1u : 0u;
                    score -= (hn & last_bit) ? 1u : 0u;
                }
            }
            cout = (u32)c_d0 | ((u32)c_hp << 1) | ((u32)c_hn << 2);
            if (last_lane && score <= best) { best = score; best_col = t - lg + 1u; }
        }
    }
    if (last_lane) {
        DevAlignOut o;
        o.score = best <= job.k ? best : 0xFFFFFFFFu;
        o.end_col = best_col;
        out[job.out_index] = o;
    }
}

// ------------------------------------------------------------------------------------------------ banded, ring-scheduled form
// Only cells on diagonals -k <= col - row <= (n - m) + k can lie on an alignment of the whole query inside the window with at
// most k errors (Ukkonen). The query's 64*W-row word groups g = 0..Lg-1 are therefore only computed for the columns
// [64W*g - k, 64W*(g+1) - 1 + (n-m) + k]; group g runs on lane (g mod R) of the job's R-lane ring, skewed by g steps, and a lane
// moves on to group g+R when its window ends (host guarantees the windows of g and g+R do not overlap in time). A group that
// starts late starts from the all-(+1) column, a group whose predecessor has finished receives horizontal delta +1: both only
// over-estimate cells outside the band, every cell on a valid path (and the trace bits of its predecessors) stays exact.
// Each carry word also hands the predecessor's bottom-row value down so that the last group knows D[m][c] absolutely.
template <int W>
__global__ void __launch_bounds__(64) ed_band_kernel(const u8* __restrict__ text, const u64* __restrict__ peq,
                                                     const DevAlignJob* __restrict__ jobs, u32 n_jobs, u32 log2_r,
                                                     u64* __restrict__ trace, DevAlignOut* __restrict__ out, u16* __restrict__ lastrow) {
    // LDS: [6 symbols][64 lanes][W words] equality masks, then one 256-byte ring of reference symbols per job of the wave
    extern __shared__ __attribute__((aligned(16))) u64 lds_eq[];
    u8* const lds_sym = reinterpret_cast<u8*>(lds_eq + 6 * 64 * W);
    u32 const lane = lane_id();
    u32 const R = 1u << log2_r;
    u32 const p = lane & (R - 1u);
    u32 const jobs_per_wave = 64u >> log2_r;
    u32 const job_slot = lane >> log2_r;
    u32 const job_id = blockIdx.x * jobs_per_wave + job_slot;
    bool const valid = job_id < n_jobs;
    DevAlignJob job;
    if (valid) job = jobs[job_id];
    else { job.ref_off = 0; job.q_off = 0; job.trace_off = 0; job.n = 0; job.m = 1; job.k = 0; job.out_index = 0; }

    int const n = (int)job.n, m = (int)job.m, k = (int)job.k;
    int const nw = (m + 63) >> 6;
    int const Lg = (nw + W - 1) / W;                      // word groups
    int const band_hi = n - m + k;                        // largest useful diagonal (col - row, 1-based)
    u32 const src_lane = (lane & ~(R - 1u)) | ((lane - 1u) & (R - 1u));
    // (jobs of at most four lanes: the lanes of a job are at most a few columns apart, half the ring does; with 16 to 64 jobs per
    // wave the rings are most of the wave's LDS, and LDS is what limits how many DP waves fit next to the search kernel's)
    u32 const ring_mask = log2_r <= 2u ? 127u : 255u;
    int const ring_lead = log2_r <= 2u ? 40 : 88;
    u8* const ring = lds_sym + job_slot * (ring_mask + 1u);

    int g = (int)p;                                       // current group of this lane
    int c_lo = 0, c_hi = -1;
    u64 vp[W], vn[W];
    int rows_g = 0;
    auto enter_group = [&]() {
        // window of columns (0-based) and equality masks of group g
        int const r0 = 64 * W * g;
        int const r1 = min(m, r0 + 64 * W);
        rows_g = r1 - r0;
        c_lo = max(0, r0 - k);
        c_hi = min(n - 1, r1 - 1 + band_hi);
        u64 const a = job.q_off >> 6;
        u32 const sh = (u32)(job.q_off & 63u);
#pragma unroll
        for (int w = 0; w < W; ++w) {
            int const gw = g * W + w;
#pragma unroll
            for (u32 s = 0; s < 6; ++s) {
                u64 v = 0;
                if (gw < nw) {
                    u64 const lo = peq[(a + gw) * 6 + s];
                    u64 const hi = peq[(a + gw + 1) * 6 + s];
                    v = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
                    int const rows_left = m - gw * 64;
                    if (rows_left < 64) v &= (1ull << rows_left) - 1ull;
                }
                lds_eq[(s * 64u + lane) * W + w] = v;
            }
            vp[w] = ~0ull;
            vn[w] = 0ull;
        }
    };
    bool has_group = valid && g < Lg;
    if (has_group) enter_group();
    else {
#pragma unroll
        for (int w = 0; w < W; ++w) { vp[w] = ~0ull; vn[w] = 0ull; }
    }

    u32 const my_steps = valid ? (u32)(n + Lg - 1) : 0u;
    u32 const t_max = wave_max_u32(my_steps);

    // Reference symbols travel through the LDS ring (column c at ring[c & ring_mask]) so that the step loop issues no global
    // loads: a load in the loop would make every step wait for the previous step's trace stores (loads and stores share
    // vmcnt). The R lanes of a job refill 64 columns at a time, far ahead of the newest column any of them needs.
    const u8* __restrict__ ref = text + job.ref_off;
    int loaded = 0;                                       // columns [0, loaded) have been written to the ring (job-uniform)
    int g_front = 0;                                      // oldest group that is not finished (job-uniform): needs the newest column
    auto refill = [&]() {
        u32 const per_lane = 64u >> log2_r;               // R lanes x per_lane bytes = 64 columns
        for (u32 i = 0; i < per_lane; ++i) {
            int const c = loaded + (int)(p * per_lane + i);
            u8 const v = (valid && c < n) ? ref[c] : (u8)7;
            ring[(u32)c & ring_mask] = v;
        }
        loaded += 64;
    };
    refill();
    refill();
    __syncthreads();

    u32 cout = 2u;                                        // inactive lanes hand down "horizontal +1"
    int bot = 0;                                          // D[last row of the group][current column]
    bool started = false;
    int best = m, best_col = 0;
    u32 const last_shift = (u32)(m - 1) & 63u;
    int const w_last_of_last = (nw - 1) - (Lg - 1) * W;   // word of the last group that holds row m-1

    for (u32 t = 0; t < t_max; ++t) {
        if ((t & 15u) == 0u) {
            // every 16 steps: retire finished front groups and keep the symbol ring >= 72 columns ahead of the front group
            while (g_front + 1 < Lg) {
                int const fr1 = min(m, 64 * W * (g_front + 1));
                int const f_hi = min(n - 1, fr1 - 1 + band_hi);
                if ((int)t - g_front > f_hi) ++g_front; else break;
            }
            if ((int)t - g_front + ring_lead > loaded && loaded < n) { refill(); __builtin_amdgcn_s_waitcnt(0); }
        }
        int c = (int)t - g;
        if (has_group && c > c_hi && g + (int)R < Lg) {
            // this lane's group is finished: take over group g + R (its window starts strictly later)
            g += (int)R;
            enter_group();
            started = false;
            c = (int)t - g;
        }
        u32 const cin_raw = (u32)__shfl((int)cout, (int)src_lane);
        bool const active = has_group && c >= c_lo && c <= c_hi;
        if (active) {
            u32 const sym = ring[(u32)c & ring_mask];
            u32 const cin = g == 0 ? 0u : cin_raw;
            u64 c_hp = (cin >> 1) & 1u, c_hn = (cin >> 2) & 1u;
            if (!started) {
                // column just left of the window: all vertical deltas +1 below the predecessor's bottom value
                int const top_prev = g == 0 ? 0 : (int)(cin >> 3) - (int)c_hp + (int)c_hn;
                bot = top_prev + rows_g;
                started = true;
            }
            const u64* __restrict__ eqp = &lds_eq[(sym * 64u + lane) * W];
            u64 hp_keep = 0, hn_keep = 0;                 // horizontal deltas of the word that holds row m-1 (last group only)
#pragma unroll
            for (int w = 0; w < W; ++w) {
                u64 const eq = eqp[w];
                u64 const pv = vp[w], mv = vn[w];
                u64 const x = eq | mv;
                u64 const tt = pv + (x & pv) + c_hn;      // the adder's carry-in is the predecessor word's top horizontal-negative bit
                u64 const d0 = (tt ^ pv) | x;
                u64 const hn = pv & d0;
                u64 const hp = mv | ~(pv | d0);
                u64 const xh = (hp << 1) | c_hp;
                u64 const nvn = xh & d0;
                u64 const nvp = (hn << 1) | ~(xh | d0) | c_hn;
                c_hp = hp >> 63;
                c_hn = hn >> 63;
                vn[w] = nvn;
                vp[w] = nvp;
                if (w == w_last_of_last) { hp_keep = hp; hn_keep = hn; }
            }
            if (g != Lg - 1) bot += (int)c_hp - (int)c_hn;
            else {
                bot += (int)((hp_keep >> last_shift) & 1ull) - (int)((hn_keep >> last_shift) & 1ull);
                if (bot <= best) { best = bot; best_col = c + 1; }
            }
            cout = ((u32)c_hp << 1) | ((u32)c_hn << 2) | ((u32)bot << 3);
        } else {
            cout = 2u;
        }
    }
    if (valid && has_group && g == Lg - 1) {
        DevAlignOut o;
        o.score = best <= k ? (u32)best : 0xFFFFFFFFu;
        o.end_col = (u32)best_col;
        out[job.out_index] = o;
    }
}


// ------------------------------------------------------------------------------------------------ banded existence test, 16 columns per step
// The same band, groups, ring of lanes and skew as ed_band_kernel, for launches that want no trace: a lane takes 16 columns of its
// group per step (the reference symbols of the block sit in four registers, the carries of 16 columns cross to the next lane as
// one word), so the per-step work of ed_band_kernel (ring read, lane shuffle, window tests, start logic) is paid once per 16
// columns and the only LDS access per column and word is the equality mask. Windows are widened to whole blocks (cells outside
// the band may be computed, from exact or over-estimated inputs: both are over-estimates there, as in ed_band_kernel); a group
// keeps going for the block in which the next group starts, whose start value D[last row of this group][column before that
// block] travels with the carries.
// (the jobs of one wave: group `blk` of 64 >> log2_r jobs)
// TRACE (K4): per block-step T = b + g, ring lane and word the block's 16 pairs of carry bits that enter the word from above (one
// u32) and the word's {vp, vn} before the block (one 16-byte slot) are written out (TraceLayout; ed_traceback_wave_kernel recomputes
// any word's trace bits over any block from those), and the last group stores D[m][c] of its columns.
// queue: hand-over slots per job behind the equality masks in LDS (a power of two, or 0: no job of the launch has a ring that waits, ring_delay);
// err: set when a job's delay does not fit the queue (null: the host chose the shape per job and knows it fits)
template <int W, bool TRACE>
__device__ __forceinline__ void ed_block_body(const u8* __restrict__ text, const u64* __restrict__ peq, const DevAlignJob* __restrict__ jobs, u32 n_jobs,
                                              u32 log2_r, DevAlignOut* __restrict__ out, u32 blk, u64* __restrict__ lds_eq,
                                              u64* __restrict__ trace, u16* __restrict__ lastrow, u32 queue, u32* __restrict__ err) {
    u32 const lane = lane_id();
    u32 const R = 1u << log2_r;
    u32 const p = lane & (R - 1u);
    u32 const jobs_per_wave = 64u >> log2_r;
    u32 const job_id = blk * jobs_per_wave + (lane >> log2_r);
    bool valid = job_id < n_jobs;
    DevAlignJob job;
    if (valid) job = jobs[job_id];
    else { job.ref_off = 0; job.q_off = 0; job.trace_off = 0; job.n = 0; job.m = 1; job.k = 0; job.out_index = 0; }
    int const n = (int)job.n, m = (int)job.m, k = (int)job.k;
    if (valid && (n == 0 || n + k < m)) {
        // no column at all: all m rows are insertions; fewer columns than m - k: no alignment within k
        if (p == 0u) { DevAlignOut o; o.score = (n == 0 && m <= k) ? (u32)m : 0xFFFFFFFFu; o.end_col = 0u; out[job.out_index] = o; }
        valid = false;
    }
    int const nw = (m + 63) >> 6;
    int const Lg = (nw + W - 1) / W;
    int const band_hi = n - m + k;
    u32 const src_lane = (lane & ~(R - 1u)) | ((lane - 1u) & (R - 1u));
    // the ring's schedule (flx_internal.hpp): group g takes block b at block-step b + g + (g / R) delay
    int delay = valid ? (int)ring_delay((u32)n, (u32)m, (u32)k, (u32)W, R) : 0;
    if (delay > 0 && (u32)delay + 1u > queue) {          // (a shape that does not hold the job: reported, never computed wrongly)
        if (err && p == 0u) atomicOr(err, 1u);
        if (p == 0u) { DevAlignOut o; o.score = 0xFFFFFFFFu; o.end_col = 0u; out[job.out_index] = o; }
        valid = false;
        delay = 0;
    }
    bool const any_delay = __any(delay > 0);
    uint2* __restrict__ const hand = reinterpret_cast<uint2*>(lds_eq + 7u * 64u * W) + (size_t)(lane >> log2_r) * queue;      // this job's hand-over slots
    u32 const qmask = queue - 1u;
    auto offset_of = [&](int gg) { return gg + (gg >> log2_r) * delay; };
#pragma unroll
    for (int w = 0; w < W; ++w) lds_eq[(6u * 64u + lane) * W + w] = 0ull;

    // The query is right-aligned in its Lg groups: `pad` rows in front of row 0 that match every symbol and start with vertical delta 0
    // keep D = 0 (what row 0's boundary is) down to the first real row, and the last real row is bit 63 of every group's last word: the
    // value that travels down (and the score in the last group) follows from the word's carries, no row has to be picked out of a word.
    int const pad = Lg * 64 * W - m;                      // 0 <= pad < 64 W: only group 0 holds padding
    int g = (int)p;
    int b_lo = 0, b_hi = -1, rows_g = 0;
    int above_b_hi = 0;                                   // last block of the group above (its lane may run a later group after that)
    int off_g = 0;                                        // offset_of(g)
    u64 vp[W], vn[W];
    auto enter_group = [&]() {
        int const r0 = max(0, 64 * W * g - pad);
        int const r1 = 64 * W * (g + 1) - pad;
        rows_g = r1 - r0;
        b_lo = max(0, r0 - k) >> 4;
        b_hi = min(n - 1, r1 - 1 + band_hi) >> 4;
        if (g + 1 < Lg) b_hi = max(b_hi, max(0, r1 - k) >> 4);
        above_b_hi = max(min(n - 1, r0 - 1 + band_hi) >> 4, b_lo);      // (group g - 1: rows up to r0, kept going for this group's first block)
        off_g = offset_of(g);
#pragma unroll
        for (int w = 0; w < W; ++w) {
            int const rs = 64 * (g * W + w) - pad;        // real row of the word's bit 0 (negative: that many padding rows first)
            u64 const padmask = rs <= -64 ? ~0ull : rs < 0 ? (1ull << (u32)(-rs)) - 1ull : 0ull;
            i64 const off = (i64)job.q_off + rs;          // pool position of the word's bit 0
#pragma unroll
            for (u32 s = 0; s < 6; ++s) {
                u64 v = 0;
                if (rs > -64) {
                    if (off >= 0) {
                        u64 const a = (u64)off >> 6;
                        u32 const sh = (u32)off & 63u;
                        u64 const lo = peq[a * 6 + s];
                        u64 const hi = peq[(a + 1) * 6 + s];
                        v = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
                    } else v = peq[s] << (u32)(-off);     // (the pool starts inside the word: the bits in front of it are padding rows)
                }
                lds_eq[(s * 64u + lane) * W + w] = v | padmask;
            }
            lds_eq[(6u * 64u + lane) * W + w] = padmask;  // symbol 6 (columns past the end of the window) matches nothing but the padding
            vp[w] = ~padmask;
            vn[w] = 0ull;
        }
    };
    bool const has_group = valid && g < Lg;
    if (has_group) enter_group();
    else {
#pragma unroll
        for (int w = 0; w < W; ++w) { vp[w] = ~0ull; vn[w] = 0ull; }
    }
    u32 const my_steps = valid ? (u32)(((n - 1) >> 4) + offset_of(Lg - 1) + 1) : 0u;
    u32 const t_max = wave_max_u32(my_steps);
    const u8* __restrict__ ref = text + job.ref_off;
    u32* __restrict__ carry_out = nullptr;
    ulonglong2* __restrict__ ckpt_out = nullptr;
    if (TRACE) {
        TraceLayout const tl = ckpt_trace_layout(job.n, job.m, job.k, (u32)W, R);
        carry_out = reinterpret_cast<u32*>(reinterpret_cast<ulonglong2*>(trace) + job.trace_off);
        ckpt_out = reinterpret_cast<ulonglong2*>(trace) + job.trace_off + tl.carry_slots;
    }

    u32 cw_out = 0x55555555u;                             // what an idle lane hands down: horizontal +1 in every column
    int botv_out = 0;
    int bot = 0, best = m, best_col = 0;
    // The reference symbols of a block are loaded one block-step ahead, and (TRACE) what a block writes is stored at the start of the
    // lane's next block, in front of that load: the wait for the symbols at the top of a step then only covers memory operations issued a
    // whole block of computation earlier (loads and stores share one counter and come back in order).
    uint4 tq_pre = make_uint4(0, 0, 0, 0);
    int pre_b = -1;                                       // tq_pre holds the symbols of block pre_b of this job (-1: nothing)
    bool pend = false, pend_last = false;                 // TRACE: the carries (and last-row values) of the lane's previous block are still in registers
    u64 pend_slot = 0;
    int pend_b = 0;
    u32 cbits[W];
    u32 rowv[8];                                          // last group: D[m][c] of the block's columns, two per word
    auto flush = [&]() {
        if (TRACE && pend) {
#pragma unroll
            for (int w = 0; w < W; ++w) carry_out[pend_slot + w] = cbits[w];
            if (pend_last && lastrow) {
                // (a job's last-row region starts at a multiple of 16 entries and covers whole blocks: 0xFFFF past column n)
                uint4* __restrict__ dst = reinterpret_cast<uint4*>(lastrow + job.lastrow_off + 16 * (u64)pend_b);
                dst[0] = make_uint4(rowv[0], rowv[1], rowv[2], rowv[3]);
                dst[1] = make_uint4(rowv[4], rowv[5], rowv[6], rowv[7]);
            }
            pend = false;
        }
    };
    for (u32 T = 0; T < t_max; ++T) {
        int b = (int)T - off_g;
        if (has_group && b > b_hi && g + (int)R < Lg) {   // this lane's group is finished: group g + R starts no earlier than now (ring_delay)
            g += (int)R;
            enter_group();
            b = (int)T - off_g;
        }
        u32 cw_in = (u32)__shfl((int)cw_out, (int)src_lane);
        int botv_in = __shfl(botv_out, (int)src_lane);
        if (any_delay) {
            // the last lane of a ring leaves what it handed down at step T - 1 in the job's queue; the first lane, in a later revolution than
            // the group above it, takes what that group handed down `delay` steps before that: the same block of the group above
            if (delay > 0 && p == R - 1u) hand[(T - 1u) & qmask] = make_uint2(cw_out, (u32)botv_out);
            if (delay > 0 && p == 0u && g >= (int)R) { uint2 const h = hand[(T - 1u - (u32)delay) & qmask]; cw_in = h.x; botv_in = (int)h.y; }
        }
        bool const active = has_group && b >= b_lo && b <= b_hi;
        if (active) {
            if (g == 0) { cw_in = 0u; botv_in = 0; }      // the row above the matrix: D = 0 in every column
            else if (b > above_b_hi) cw_in = 0x55555555u;      // the group above has ended (what its lane hands down now belongs to a later group): +1 per column
            if (b == b_lo) bot = botv_in + rows_g;        // column left of the window: all vertical deltas +1 below the group above
            int const bot_start = bot;
            bool const last = g == Lg - 1;
            uint4 tq = tq_pre;
            if (pre_b != b) __builtin_memcpy(&tq, ref + 16 * b, 16);
            u32 const quad[4] = {tq.x, tq.y, tq.z, tq.w};
            u32 cw = 0;
            u64 const slot = ((u64)T * R + p) * W;        // this lane's words at this block-step
            flush();
            if (TRACE) {
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    ulonglong2 v;
                    v.x = vp[w];
                    v.y = vn[w];
                    ckpt_out[slot + w] = v;
                    cbits[w] = 0;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) rowv[q] = 0xFFFFFFFFu;
            }
            if (b < b_hi) { __builtin_memcpy(&tq_pre, ref + 16 * (b + 1), 16); pre_b = b + 1; }
            // The block's 16 columns, in two forms. Only a job's last group looks at its bottom row column by column (the score and its
            // rightmost column; K4: the last row itself): a wave none of whose lanes is in a last group - most block-steps of a launch, the
            // jobs of a wave start together and are of one size class - runs the form without that, takes the group's bottom value across
            // the block from the carries' bit counts, and shifts the outgoing carries into their word instead of placing each pair.
            auto block16 = [&](auto track_tag) {
                constexpr bool TRACK = decltype(track_tag)::value;
                // (the block's inputs through an empty asm: values of this form alone. Otherwise the compiler computes what the two forms have in
                // common - sixteen columns' symbols, addresses, carry bits, end-of-window tests - once, in front of the branch, and keeps it all
                // in registers through the block: 186 of them for one word per lane instead of 79)
                u32 qv[4] = {quad[0], quad[1], quad[2], quad[3]};
                u32 cwi = cw_in;
                int bb = b;
                asm volatile("" : "+v"(qv[0]), "+v"(qv[1]), "+v"(qv[2]), "+v"(qv[3]), "+v"(cwi), "+v"(bb));
                u32 acc = 0;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    __builtin_amdgcn_sched_barrier(0);    // (nothing moves across four columns: the form without comparisons would have all sixteen columns' loads in flight)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int const j = 4 * qd + i;
                        int const c = 16 * bb + j;
                        u32 sym = (qv[qd] >> (8 * i)) & 0xFFu;
                        sym = c < n ? sym : 6u;
                        u64 c_hp = (cwi >> (2 * j)) & 1u, c_hn = (cwi >> (2 * j + 1)) & 1u;
                        const u64* __restrict__ eqp = &lds_eq[(sym * 64u + lane) * W];
                        u64 hp_last = 0, hn_last = 0;
#pragma unroll
                        for (int w = 0; w < W; ++w) {
                            if (TRACE) cbits[w] |= ((u32)c_hp | ((u32)c_hn << 1)) << (2 * j);
                            u64 const eq = eqp[w];
                            u64 const pv = vp[w], mv = vn[w];
                            u64 const x = eq | mv;
                            u64 const tt = pv + (x & pv) + c_hn;
                            u64 const d0 = (tt ^ pv) | x;
                            u64 const hn = pv & d0;
                            u64 const hp = mv | ~(pv | d0);
                            u64 const xh = (hp << 1) | c_hp;
                            vn[w] = xh & d0;
                            vp[w] = (hn << 1) | ~(xh | d0) | c_hn;
                            if (w + 1 < W || TRACK) { c_hp = hp >> 63; c_hn = hn >> 63; }
                            if (w + 1 == W) { hp_last = hp; hn_last = hn; }
                        }
                        if (TRACK) {
                            acc |= ((u32)c_hp | ((u32)c_hn << 1)) << (2 * j);
                            bot += (int)(u32)c_hp - (int)(u32)c_hn;       // the group's last row is its last word's bit 63
                            if (last && c < n && bot <= best) { best = bot; best_col = c + 1; }
                            if (TRACE && c < n) {
                                u32 const v16 = (u32)min(bot, 0xFFFF);
                                rowv[j >> 1] = (j & 1) ? (rowv[j >> 1] & 0xFFFFu) | (v16 << 16) : (rowv[j >> 1] & 0xFFFF0000u) | v16;
                            }
                        } else {
                            // (column j's pair ends at bits 2j, 2j + 1 once the fifteen columns after it have pushed it down)
                            acc = (acc >> 2) | (((u32)(hp_last >> 32) >> 1) & 0x40000000u) | ((u32)(hn_last >> 32) & 0x80000000u);
                        }
                    }
                }
                if (!TRACK) bot += (int)__popc(acc & 0x55555555u) - (int)__popc(acc & 0xAAAAAAAAu);
                return acc;
            };
            cw = __any(last) ? block16(std::true_type{}) : block16(std::false_type{});
            if (TRACE) { pend = true; pend_last = last; pend_slot = slot; pend_b = b; }
            cw_out = cw;
            botv_out = bot_start;
        } else {
            cw_out = 0x55555555u;
            botv_out = 0;
        }
    }
    flush();
    if (valid && has_group && g == Lg - 1) {
        DevAlignOut o;
        o.score = best <= k ? (u32)best : 0xFFFFFFFFu;
        o.end_col = (u32)best_col;
        out[job.out_index] = o;
    }
}


// A launch is a grid over the groups of jobs, or (n_jobs_dev: the job count is on the device, verification rounds of flx_rounds.hip)
// a fixed grid whose waves take the groups in turn until the count is reached.
// (waves per SIMD the register allocation is held to: what round 3's single-form block had - the two forms of the block made the
// scheduler keep every equality mask of a block in flight, 186 registers for one word per lane)
template <int W>
__global__ void __launch_bounds__(64, (W <= 2 ? 4 : W <= 5 ? 3 : W <= 8 ? 2 : 1)) ed_exists_block_kernel(const u8* __restrict__ text, const u64* __restrict__ peq,
                                                             const DevAlignJob* __restrict__ jobs, u32 n_jobs, u32 log2_r,
                                                             DevAlignOut* __restrict__ out, const u32* __restrict__ n_jobs_dev, u32 queue, u32* __restrict__ err) {
    // LDS: [7 symbols][64 lanes][W words] equality masks; symbol 6 (columns past the end of the window) matches nothing; then `queue`
    // hand-over slots per job of the wave
    extern __shared__ __attribute__((aligned(16))) u64 lds_eq[];
    if (n_jobs_dev) n_jobs = min(n_jobs, *n_jobs_dev);
    u32 const jobs_per_wave = 64u >> log2_r;
    for (u32 blk = blockIdx.x; blk * jobs_per_wave < n_jobs; blk += gridDim.x)
        ed_block_body<W, false>(text, peq, jobs, n_jobs, log2_r, out, blk, lds_eq, nullptr, nullptr, queue, err);
}

// K4: the same body with the checkpointed trace and the last rows written out, one wave per group of jobs
template <int W>
__global__ void __launch_bounds__(64, (W <= 4 ? 3 : W <= 6 ? 2 : 1)) ed_trace_block_kernel(const u8* __restrict__ text, const u64* __restrict__ peq,
                                                            const DevAlignJob* __restrict__ jobs, u32 n_jobs, u32 log2_r,
                                                            u64* __restrict__ trace, DevAlignOut* __restrict__ out, u16* __restrict__ lastrow, u32 queue) {
    extern __shared__ __attribute__((aligned(16))) u64 lds_eq[];
    __builtin_amdgcn_s_setprio(2);                   // (few waves, long chains, 18 KB of LDS each: they go first on a shared SIMD)
    ed_block_body<W, true>(text, peq, jobs, n_jobs, log2_r, out, blockIdx.x, lds_eq, trace, lastrow, queue, nullptr);
}

static bool exists_block_form() {            // FLX_EXISTS_STEPWISE=1: existence tests through ed_band_kernel (one column per step)
    static int const v = getenv("FLX_EXISTS_STEPWISE") ? 0 : 1;
    return v != 0;
}
// hand-over slots a job with this delay needs (a power of two; 0: none)
static u32 ring_queue_for(u32 delay) {
    if (delay == 0) return 0;
    u32 q = 32;
    while (q < delay + 1u) q *= 2;
    return q;
}
// the widest band (diagonals - 1 = n - m + 2k) a shape holds: any when every group has a lane; else the one whose ring_delay still fits
// `queue` hand-over slots (three blocks of slack for the roundings of ring_group_blocks; a job beyond it is reported by the kernel)
static u64 shape_width_cap(u32 nw, AlignShape sh) {
    u32 const w = sh.words_per_lane, r = sh.lanes_per_job;
    if ((nw + w - 1) / w <= r) return 0xFFFFFFFFull;
    u64 const no_wait = (u64)64 * w * (r - 1) + r;
    if (sh.queue < 8) return no_wait;
    return no_wait + 16ull * (sh.queue - 5u);
}
u64 DeviceApi::shape_width_cap(u32 nw, AlignShape sh) { return flx::shape_width_cap(nw, sh); }

// n, m, k: the job the shape is for (shape_holding: a job as wide as its class allows)
static AlignShape choose_align_shape_uncached(u32 n, u32 m, u32 k, bool band, bool parallel) {
    u32 const nw = (m + 63) / 64;
    i64 const width = (i64)n - (i64)m + 2 * (i64)k;
    // rings that wait (ring_delay): block kernels only, throughput form only; FLX_RING_STRETCH = how much longer than the shortest schedule a
    // job's block-steps may get, in percent (default 135; 100 = round 3's shapes)
    static int const stretch = [] { const char* e = getenv("FLX_RING_STRETCH"); int const v = e ? atoi(e) : 135; return v < 100 ? 100 : v; }();
    bool const may_wait = band && !parallel && exists_block_form() && stretch > 100;
    AlignShape best{0, 0, 0};
    u64 best_cost = ~0ull;
    u64 shortest = ~0ull, shortest_key = ~0ull;           // block-steps of the shape round 3 chose (fewest lanes x words among the rings that never wait)
    for (int pass = 0; pass < (may_wait ? 2 : 1); ++pass)
    for (u32 w : kWordsPerLane)
        for (u32 r = 1; r <= 64; r *= 2) {
            u32 const groups = (nw + w - 1) / w;
            bool ok = groups <= r;                        // every group has its own lane
            if (!ok && band) ok = (i64)64 * w * (r - 1) + r + 1 > width;   // group g + r starts after group g has ended
            u32 delay = 0;
            if (!ok && may_wait && pass == 1) {
                delay = ring_delay(n, m, k, w, r);
                ok = delay + 1u <= RING_QUEUE_MAX;
            }
            if (!ok) continue;
            u64 const steps = band ? ring_steps(n, m, k, w, r) : (u64)n + groups;
            if (pass == 0 && may_wait) { u64 const key = (u64)w * r * 1000 + w; if (key < shortest_key) { shortest_key = key; shortest = steps; } continue; }
            // throughput form: cost ~ wave slots consumed (words per lane times lanes reserved), fewer words per lane on ties;
            // parallel form: fewest words per lane first (shortest dependent chain per step, most waves), then fewest lanes.
            // (A cost by instructions issued, r * (35 + 25 w), which prefers four words per lane over one at the same w * r, made
            // the existence tests a third slower: more distinct shapes per round = more launches, and fewer resident waves per CU
            // with the larger LDS tables; measured in round 2, gpurun_out r02u.)
            // (FLX_SHAPE_MODEL=a,b: throughput form by instructions issued instead, r * (a + b * w): per column a lane pays `a` whatever its
            // words and `b` per word)
            // Round 4, rings that wait: cost = the block-steps the job's lanes sit through, lanes x steps x words - what the launch issues for
            // the job whether a lane has a block to compute or not - among the shapes whose schedule is at most `stretch` percent of the one round 3 chose
            static int const model_a = [] { const char* e = getenv("FLX_SHAPE_MODEL"); int a = 0, b = 0; return e && sscanf(e, "%d,%d", &a, &b) == 2 ? a : 0; }();
            static int const model_b = [] { const char* e = getenv("FLX_SHAPE_MODEL"); int a = 0, b = 0; return e && sscanf(e, "%d,%d", &a, &b) == 2 ? b : 0; }();
            u64 cost;
            if (may_wait) {
                if (steps * 100 > shortest * (u64)stretch) continue;
                cost = steps * r * (8 * w + 1);
            } else cost = parallel ? (u64)w * 1000 + r : model_b ? (u64)r * (u64)(model_a + model_b * (int)w) * 16 + w : (u64)w * r * 1000 + w;
            if (cost < best_cost) { best_cost = cost; best = AlignShape{w, r, band ? 1u : 0u, ring_queue_for(delay)}; }
        }
    return best;
}

AlignShape choose_align_shape(u32 n, u32 m, u32 k, bool parallel) {
    u32 const nw = (m + 63) / 64;
    bool const band = use_band();
    i64 const width = (i64)n - (i64)m + 2 * (i64)k;       // diagonals that matter, minus one
    if (const char* forced = getenv("FLX_FORCE_SHAPE")) {   // "W,R": measurements of one launch shape (scripts/shape_cost.py)
        u32 w = 0, r = 0;
        if (sscanf(forced, "%u,%u", &w, &r) == 2 && w && r) {
            if ((nw + w - 1) / w <= r || (band && (i64)64 * w * (r - 1) + r + 1 > width)) return AlignShape{w, r, band ? 1u : 0u};
            u32 const delay = band && exists_block_form() ? ring_delay(n, m, k, w, r) : RING_QUEUE_MAX;
            if (delay + 1u <= RING_QUEUE_MAX) return AlignShape{w, r, 1u, ring_queue_for(delay)};
        }
    }
    // the jobs of one verification level repeat a handful of (rows, columns, errors) triples: small direct-mapped memo per thread
    struct Entry { u32 n, m, k; AlignShape shape; bool valid; };
    thread_local Entry memo[2][256] = {};
    Entry& e = memo[parallel ? 1 : 0][(m * 31u + n * 7u + k) & 255u];
    if (e.valid && e.n == n && e.m == m && e.k == k) return e.shape;
    e = Entry{n, m, k, choose_align_shape_uncached(n, m, k, band, parallel), true};
    return e.shape;
}
u32 align_supported_max_query() { return 25u * 64u * 64u; }

u64 align_trace_slots(u32 n, u32 m, u32 k, AlignShape sh) {
    if (sh.banded) {
        TraceLayout const tl = ckpt_trace_layout(n, m, k, sh.words_per_lane, sh.lanes_per_job);
        return tl.carry_slots + tl.ckpt_slots;
    }
    // full trace, step-major: (n + groups - 1) steps x groups lanes x W words of {hp, vp}
    u32 const nw = (m + 63) / 64;
    u64 const groups = (nw + sh.words_per_lane - 1) / sh.words_per_lane;
    return ((u64)n + groups - 1) * groups * sh.words_per_lane;
}

template <int W>
static int launch_align(hipStream_t s, const u8* d_text, const u64* d_peq, const DevAlignJob* d_jobs, u32 n_jobs, u32 log2_g, bool trace,
                        bool banded, u64* d_trace, DevAlignOut* d_out, u16* d_lastrow, u32 queue) {
    u32 const jobs_per_wave = 64u >> log2_g;
    u32 const blocks = (n_jobs + jobs_per_wave - 1) / jobs_per_wave;
    size_t const lds = (size_t)6 * 64 * W * sizeof(u64) + (banded ? (size_t)jobs_per_wave * (log2_g <= 2u ? 128 : 256) : 0);
#define FLX_LAUNCH(KERNEL)                                                                                                           \
    do {                                                                                                                             \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
        hipLaunchKernelGGL((KERNEL), dim3(blocks), dim3(64), lds, s, d_text, d_peq, d_jobs, n_jobs, log2_g, d_trace, d_out, d_lastrow); \
    } while (0)
    if (banded && !trace && !d_lastrow && exists_block_form()) {
        size_t const lds_b = (size_t)7 * 64 * W * sizeof(u64) + (size_t)jobs_per_wave * queue * sizeof(uint2);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ed_exists_block_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
        hipLaunchKernelGGL((ed_exists_block_kernel<W>), dim3(blocks), dim3(64), lds_b, s, d_text, d_peq, d_jobs, n_jobs, log2_g, d_out, (const u32*)nullptr, queue, (u32*)nullptr);
        return (int)hipGetLastError();
    }
    if (banded && trace) {
        size_t const lds_b = (size_t)7 * 64 * W * sizeof(u64) + (size_t)jobs_per_wave * queue * sizeof(uint2);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ed_trace_block_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
        hipLaunchKernelGGL((ed_trace_block_kernel<W>), dim3(blocks), dim3(64), lds_b, s, d_text, d_peq, d_jobs, n_jobs, log2_g, d_trace, d_out, d_lastrow, queue);
        return (int)hipGetLastError();
    }
    if (banded) FLX_LAUNCH((ed_band_kernel<W>));
    else { if (trace) FLX_LAUNCH((ed_align_kernel<W, true>)); else FLX_LAUNCH((ed_align_kernel<W, false>)); }
#undef FLX_LAUNCH
    return (int)hipGetLastError();
}

// existence tests whose number is on the device (*d_n_jobs, at most max_jobs): a grid of at most `max_waves` waves that take the
// groups of jobs in turn
template <int W>
static int launch_exists_counted(hipStream_t s, const u8* d_text, const u64* d_peq, const DevAlignJob* d_jobs, u32 max_jobs, u32 log2_g, DevAlignOut* d_out,
                                 const u32* d_n_jobs, u32 max_waves, u32 queue, u32* d_err) {
    u32 const jobs_per_wave = 64u >> log2_g;
    u32 const blocks = std::max(1u, std::min((max_jobs + jobs_per_wave - 1) / jobs_per_wave, max_waves));
    size_t const lds_b = (size_t)7 * 64 * W * sizeof(u64) + (size_t)jobs_per_wave * queue * sizeof(uint2);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ed_exists_block_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
    hipLaunchKernelGGL((ed_exists_block_kernel<W>), dim3(blocks), dim3(64), lds_b, s, d_text, d_peq, d_jobs, max_jobs, log2_g, d_out, d_n_jobs, queue, d_err);
    return (int)hipGetLastError();
}
int DeviceApi::align_exists_counted(void* stream, const u8* d_text, const u64* d_peq, const DevAlignJob* d_jobs, u32 max_jobs, const u32* d_n_jobs,
                                    AlignShape shape, u32 max_waves, DevAlignOut* d_out, u32* d_err) {
    if (max_jobs == 0) return 0;
    u32 log2_g = 0;
    while ((1u << log2_g) < shape.lanes_per_job) ++log2_g;
    hipStream_t s = (hipStream_t)stream;
    switch (shape.words_per_lane) {
        case 1: return launch_exists_counted<1>(s, d_text, d_peq, d_jobs, max_jobs, log2_g, d_out, d_n_jobs, max_waves, shape.queue, d_err);
        case 2: return launch_exists_counted<2>(s, d_text, d_peq, d_jobs, max_jobs, log2_g, d_out, d_n_jobs, max_waves, shape.queue, d_err);
        case 3: return launch_exists_counted<3>(s, d_text, d_peq, d_jobs, max_jobs, log2_g, d_out, d_n_jobs, max_waves, shape.queue, d_err);
        case 4: return launch_exists_counted<4>(s, d_text, d_peq, d_jobs, max_jobs, log2_g, d_out, d_n_jobs, max_waves, shape.queue, d_err);
        case 5: return launch_exists_counted<5>(s, d_text, d_peq, d_jobs, max_jobs, log2_g, d_out, d_n_jobs, max_waves, shape.queue, d_err);
        case 6: return launch_exists_counted<6>(s, d_text, d_peq, d_jobs, max_jobs, log2_g, d_out, d_n_jobs, max_waves, shape.queue, d_err);
        case 8: return launch_exists_counted<8>(s, d_text, d_peq, d_jobs, max_jobs, log2_g, d_out, d_n_jobs, max_waves, shape.queue, d_err);
        case 13: return launch_exists_counted<13>(s, d_text, d_peq, d_jobs, max_jobs, log2_g, d_out, d_n_jobs, max_waves, shape.queue, d_err);
        case 25: return launch_exists_counted<25>(s, d_text, d_peq, d_jobs, max_jobs, log2_g, d_out, d_n_jobs, max_waves, shape.queue, d_err);
        default: return (int)hipErrorInvalidValue;
    }
}
// the cheapest shape (parallel: the one with the fewest words per lane) that holds every job of at most nw query words and `width` diagonals
// the shape for a round's size class: a job of nw words whose band is `width` diagonals wide (a window of its own: n - m + 2k = 4k + 1)
AlignShape DeviceApi::shape_holding(u32 nw, i64 width, bool parallel) {
    u32 const m = 64u * nw, k = (u32)(std::max<i64>(width, 1) / 4), n = (u32)((i64)m + std::max<i64>(width, 1) - 2 * (i64)k);
    AlignShape sh = choose_align_shape_uncached(n, m, k, use_band(), parallel);
    // room for the unions of a cluster's windows: slots for a band a quarter wider than the class's own, when that costs no more than the next size
    if (sh.queue) { u32 const wider = ring_queue_for(ring_delay(n + (u32)(width / 4), m, k, sh.words_per_lane, sh.lanes_per_job) + 3u); if (wider <= RING_QUEUE_MAX) sh.queue = std::max(sh.queue, wider); }
    return sh;
}

int DeviceApi::align(void* stream, const u8* d_text, const u64* d_peq, const DevAlignJob* d_jobs, u32 n_jobs, AlignShape shape, bool trace,
                     u64* d_trace, DevAlignOut* d_out, u16* d_lastrow) {
    if (n_jobs == 0) return 0;
    u32 log2_g = 0;
    while ((1u << log2_g) < shape.lanes_per_job) ++log2_g;
    hipStream_t s = (hipStream_t)stream;
    bool const b = shape.banded != 0;
    // (jobs of one launch share its words and lanes, not their delays: the most slots a shape may ask for, for any ring that may wait)
    u32 const queue = b ? RING_QUEUE_MAX : 0u;
    switch (shape.words_per_lane) {
        case 1: return launch_align<1>(s, d_text, d_peq, d_jobs, n_jobs, log2_g, trace, b, d_trace, d_out, d_lastrow, queue);
        case 2: return launch_align<2>(s, d_text, d_peq, d_jobs, n_jobs, log2_g, trace, b, d_trace, d_out, d_lastrow, queue);
        case 3: return launch_align<3>(s, d_text, d_peq, d_jobs, n_jobs, log2_g, trace, b, d_trace, d_out, d_lastrow, queue);
        case 4: return launch_align<4>(s, d_text, d_peq, d_jobs, n_jobs, log2_g, trace, b, d_trace, d_out, d_lastrow, queue);
        case 5: return launch_align<5>(s, d_text, d_peq, d_jobs, n_jobs, log2_g, trace, b, d_trace, d_out, d_lastrow, queue);
        case 6: return launch_align<6>(s, d_text, d_peq, d_jobs, n_jobs, log2_g, trace, b, d_trace, d_out, d_lastrow, queue);
        case 8: return launch_align<8>(s, d_text, d_peq, d_jobs, n_jobs, log2_g, trace, b, d_trace, d_out, d_lastrow, queue);
        case 13: return launch_align<13>(s, d_text, d_peq, d_jobs, n_jobs, log2_g, trace, b, d_trace, d_out, d_lastrow, queue);
        case 25: return launch_align<25>(s, d_text, d_peq, d_jobs, n_jobs, log2_g, trace, b, d_trace, d_out, d_lastrow, queue);
        default: return (int)hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------------ rightmost minimum of a last row
// One wave per window: the best end column of a window inside a job's column range is the last column with the minimal
// last-row value (alignment.cpp: seqan3 reports the rightmost best end), 1-based like ed_band_kernel's own result.
__global__ void __launch_bounds__(64) lastrow_min_kernel(const u16* __restrict__ lastrow, const DevRowWindow* __restrict__ windows,
                                                         u32 n_windows, DevAlignOut* __restrict__ out) {
    u32 const id = blockIdx.x;
    if (id >= n_windows) return;
    DevRowWindow const w = windows[id];
    u32 const lane = threadIdx.x & 63u;
    const u16* __restrict__ row = lastrow + w.first;
    u32 best = 0xFFFFu, col = 0;
    for (u32 c = lane; c < w.n; c += 64u) {
        u32 const v = row[c];
        if (v <= best) { best = v; col = c + 1u; }
    }
    // wave reduction on (value ascending, column descending): key = value << 32 | ~column
    u64 key = ((u64)best << 32) | (u64)(0xFFFFFFFFu - col);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        u64 const other = __shfl_xor(key, off);
        key = other < key ? other : key;
    }
    if (lane == 0) {
        u32 const v = (u32)(key >> 32), c = 0xFFFFFFFFu - (u32)key;
        DevAlignOut o;
        o.score = (v != 0xFFFFu && v <= w.k) ? v : 0xFFFFFFFFu;
        o.end_col = c;
        out[w.out_index] = o;
    }
}

int DeviceApi::lastrow_min(void* stream, const u16* d_lastrow, const DevRowWindow* d_windows, u32 n_windows, DevAlignOut* d_out) {
    if (n_windows == 0) return 0;
    hipLaunchKernelGGL(lastrow_min_kernel, dim3(n_windows), dim3(64), 0, (hipStream_t)stream, d_lastrow, d_windows, n_windows, d_out);
    return (int)hipGetLastError();
}

// ================================================================================================ K5: traceback + CIGAR
// One wave walks one job's path from (m, end_col) to row 0 with seqan3's preference up (I) > left (D) > diagonal (=/X).
// The 64 lanes fetch the trace words and symbols of the next 64 cells along the current diagonal in one go (a path is mostly
// diagonal: ~2/3 of 8 % errors are indels, i.e. a diagonal change every ~19 cells), wave ballots turn them into three 64-bit
// masks and the run-length encoding of the diagonal stretch up to the first indel is done on those masks. The CIGAR is written
// backwards into the job's slab so that it ends up in forward order without a reversal pass.
__global__ void __launch_bounds__(64) ed_traceback_kernel(const u8* __restrict__ text, const u8* __restrict__ query,
                                                          const u64* __restrict__ trace, const DevTraceJob* __restrict__ jobs, u32 n_jobs,
                                                          u32* __restrict__ cigar, DevTraceOut* __restrict__ out) {
    u32 const id = blockIdx.x;
    if (id >= n_jobs) return;
    u32 const lane = lane_id();
    DevTraceJob const job = jobs[id];
    const u8* __restrict__ r = text + job.ref_off;
    const u8* __restrict__ q = query + job.q_off;
    u32* __restrict__ slab = cigar + job.cigar_off;
    u32 const W = job.words_per_lane, L = job.lanes;
    u32 wpos = job.cigar_cap;
    u32 i = job.m, j = job.end_col;                 // wave-uniform walker position
    u32 cur_op = 0xFFu, cur_len = 0;
    bool overflow = false;
    auto emit = [&](u32 op, u32 len) {              // wave-uniform run-length merge; lane 0 stores
        if (len == 0) return;
        if (op == cur_op) { cur_len += len; return; }
        if (cur_len) {
            if (wpos == 0) overflow = true;
            else { --wpos; if (lane == 0) slab[wpos] = (cur_len << 4) | cur_op; }
        }
        cur_op = op;
        cur_len = len;
    };
    while (i > 0 && !overflow) {
        if (j == 0) { emit(1u, i); i = 0; break; }                      // only insertions remain
        // lane l looks at cell (i - l, j - l)
        bool const in_range = lane < i && lane < j;
        bool up = false, left = false, eq = false;
        if (in_range) {
            u32 const ci = i - lane, cj = j - lane;
            u32 const gw = (ci - 1u) >> 6, bit = (ci - 1u) & 63u;
            u32 const g = gw / W, w = gw - g * W;
            u64 const slot = job.trace_off + ((u64)(cj - 1u + g) * L + (g % L)) * W + w;
            ulonglong2 const v = *reinterpret_cast<const ulonglong2*>(trace + 2ull * slot);
            up = (v.y >> bit) & 1ull;
            left = (v.x >> bit) & 1ull;
            eq = q[ci - 1u] == r[cj - 1u];
        }
        u64 const m_range = __ballot(in_range);
        u64 const m_indel = __ballot(up || left) & m_range;
        u64 const m_eq = __ballot(eq);
        u32 const n_range = (u32)__popcll(m_range);                     // cells available on this diagonal (contiguous from lane 0)
        u32 const n_diag = m_indel ? (u32)__builtin_ctzll(m_indel) : n_range;   // diagonal cells before the first indel
        // run-length encode the diagonal stretch [0, n_diag)
        u32 pos = 0;
        while (pos < n_diag) {
            bool const is_eq = (m_eq >> pos) & 1ull;
            u64 const same = is_eq ? m_eq : ~m_eq;
            u64 const rest = ~(same >> pos);                            // first position (relative) where the kind changes
            u32 run = rest ? (u32)__builtin_ctzll(rest) : 64u - pos;
            if (run > n_diag - pos) run = n_diag - pos;
            emit(is_eq ? 7u : 8u, run);
            pos += run;
        }
        i -= n_diag;
        j -= n_diag;
        if (m_indel) {
            // the cell at lane n_diag takes an indel: up (I) has priority over left (D)
            bool const is_up = __shfl((int)up, (int)n_diag) != 0;
            if (is_up) { emit(1u, 1u); --i; }
            else { emit(2u, 1u); --j; }
        }
    }
    if (!overflow && cur_len) {
        if (wpos == 0) overflow = true;
        else { --wpos; if (lane == 0) slab[wpos] = (cur_len << 4) | cur_op; }
    }
    if (lane == 0) {
        DevTraceOut o;
        o.begin = j;
        o.cigar_start = wpos;
        o.cigar_len = overflow ? 0xFFFFFFFFu : job.cigar_cap - wpos;
        o.pad = 0;
        out[job.out_index] = o;
    }
}

// ------------------------------------------------------------------------------------------------ K5: traceback over a checkpointed trace, one wave per job
// The walk is serial, the recomputation of the trace is not: a path moves up its diagonal and drifts from it by one column per
// indel only, so the (word, 16-column block) windows it is going to cross are known in advance. A round therefore recomputes 64
// windows at once, one per lane: for each of the 8 words at and above the walker the 8 blocks around the columns the path's
// current diagonal crosses in that word (exactly one checkpoint + one carry word + 16 columns each). Then the wave walks: lane l
// looks at cell (i - l, j - l), ballots give the stretch of diagonal moves up to the first indel, and the walk goes on until it
// needs a window the round does not hold (the path drifted further than foreseen, or left the 8 words), which starts the next round
// from where the walker stands. ~20 rounds for a 10-kb path.
// Rows are in K4's coordinates: the query right-aligned in its groups, `pad` rows in front of row 1 (ed_block_body).
constexpr u32 TBW_WORDS = 8, TBW_BLOCKS = 8;     // windows of a round: words x blocks = 64 lanes
constexpr u32 TBW_REF = 1024;                    // reference symbols cached per round (columns)

__global__ void __launch_bounds__(64) ed_traceback_wave_kernel(const u8* __restrict__ text, const u64* __restrict__ peq,
                                                               const u64* __restrict__ trace, const DevTraceJob* __restrict__ jobs,
                                                               u32 n_jobs, u32* __restrict__ cigar, DevTraceOut* __restrict__ out) {
    __shared__ ulonglong2 win[64 * 17];              // [window * 17 + column % 16] = {hp, vp} of the window's word after that column (17: no bank conflicts)
    __shared__ u32 win_valid[64];                    // non-zero: the window was computed
    __shared__ u64 eqm[TBW_WORDS][6];                // equality masks of the round's words
    __shared__ u8 refs[TBW_REF];                     // reference symbols of columns [ref_base, ref_base + TBW_REF)
    u32 const lane = threadIdx.x & 63u;
    __builtin_amdgcn_s_setprio(2);                   // (short launches that hold much LDS go first on a shared SIMD: they leave sooner)
    // (a grid smaller than the job list: a wave takes jobs in turn, so that a launch holds no more LDS than its grid's waves)
    for (u32 id = blockIdx.x; id < n_jobs; id += gridDim.x) {
    __syncthreads();
    DevTraceJob const job = jobs[id];
    const u8* __restrict__ r = text + job.ref_off;
    u32* __restrict__ slab = cigar + job.cigar_off;
    int const W = (int)job.words_per_lane, R = (int)job.lanes;
    int const n = (int)job.n, m = (int)job.m, k = (int)job.k;
    int const band_hi = n - m + k;
    int const nw = (max(m, 1) + 63) >> 6;
    int const Lg = (nw + W - 1) / W;
    int const pad = Lg * 64 * W - max(m, 1);
    TraceLayout const tl = ckpt_trace_layout(job.n, job.m ? job.m : 1u, job.k, (u32)W, (u32)R);
    int const ring_wait = (int)ring_delay(job.n, job.m ? job.m : 1u, job.k, (u32)W, (u32)R);      // (K4's schedule: block b of group g at block-step b + g + (g / R) wait)
    const u32* __restrict__ carry = reinterpret_cast<const u32*>(reinterpret_cast<const ulonglong2*>(trace) + job.trace_off);
    const ulonglong2* __restrict__ ckpt = reinterpret_cast<const ulonglong2*>(trace) + job.trace_off + tl.carry_slots;

    u32 wpos = job.cigar_cap;
    int i = m, j = (int)job.end_col;                 // wave-uniform walker position
    u32 cur_op = 0xFFu, cur_len = 0;
    bool overflow = false;
    auto emit = [&](u32 op, u32 len) {               // wave-uniform run-length merge; lane 0 stores
        if (len == 0) return;
        if (op == cur_op) { cur_len += len; return; }
        if (cur_len) {
            if (wpos == 0) overflow = true;
            else { --wpos; if (lane == 0) slab[wpos] = (cur_len << 4) | cur_op; }
        }
        cur_op = op;
        cur_len = len;
    };
    // first block of (padded) word w's windows in a round that started on diagonal `diag` (column - row): the path crosses the word's
    // rows at columns 64w - pad + diag .. 64w - pad + 63 + diag (0-based)
    auto first_block = [&](int w, int diag) {
        int const c_lo = 64 * w - pad + diag;
        return (c_lo >= 0 ? c_lo / 16 : -((-c_lo + 15) / 16)) - 1;
    };

    while (i > 0 && !overflow) {
        if (j == 0) { emit(1u, (u32)i); i = 0; break; }                 // only insertions remain
        // ---- a round: windows of words gw_top, gw_top-1, ... around the walker's diagonal
        int const gw_top = (i - 1 + pad) >> 6;
        int const diag = j - i;
        // columns the round can touch: from 64 * TBW_WORDS + 32 below the walker's to 16 * TBW_BLOCKS above it
        int const ref_base = max(0, j - 640) & ~15;
        __syncthreads();                                                // (one wave: orders this round's LDS writes after the last round's reads)
        for (u32 x = lane; x < TBW_REF; x += 64u) { int const c = ref_base + (int)x; refs[x] = c < n ? r[c] : (u8)7; }
        __syncthreads();
        {
            int const w = gw_top - (int)(lane / TBW_BLOCKS);
            u32 valid = 0;
            if (w >= 0) {
                int const g = w / W, ww = w - g * W, p = g % R;
                int const b = first_block(w, diag) + (int)(lane % TBW_BLOCKS);
                // the blocks group g is computed for (ed_block_body's enter_group)
                int const r0 = max(0, 64 * W * g - pad), r1 = 64 * W * (g + 1) - pad;
                int const b_lo = max(0, r0 - k) >> 4;
                int b_hi = min(n - 1, r1 - 1 + band_hi) >> 4;
                if (g + 1 < Lg) b_hi = max(b_hi, max(0, r1 - k) >> 4);
                // equality masks of the word, padding rows included (the lane of the word's first window also keeps them for the walk)
                u64 eq[6];
                int const rs = 64 * w - pad;
                u64 const padmask = rs <= -64 ? ~0ull : rs < 0 ? (1ull << (u32)(-rs)) - 1ull : 0ull;
                {
                    i64 const off = (i64)job.q_off + rs;
#pragma unroll
                    for (u32 sy = 0; sy < 6; ++sy) {
                        u64 v = 0;
                        if (rs > -64) {
                            if (off >= 0) {
                                u64 const a = (u64)off >> 6;
                                u32 const sh = (u32)off & 63u;
                                u64 const lo = peq[a * 6 + sy];
                                u64 const hi = peq[(a + 1) * 6 + sy];
                                v = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
                            } else v = peq[sy] << (u32)(-off);
                        }
                        eq[sy] = v | padmask;
                    }
                    if (lane % TBW_BLOCKS == 0) {
#pragma unroll
                        for (u32 sy = 0; sy < 6; ++sy) eqm[lane / TBW_BLOCKS][sy] = eq[sy];
                    }
                }
                // all 16 columns of the block must be in the symbol cache
                if (b >= b_lo && b <= b_hi && 16 * b >= ref_base && 16 * b + 16 <= ref_base + (int)TBW_REF) {
                    u64 const slot = ((u64)(b + g + (g / R) * ring_wait) * R + p) * W + ww;
                    ulonglong2 const v = ckpt[slot];
                    u64 pv = v.x, mv = v.y;
                    u32 const cw = carry[slot];
#pragma unroll 4
                    for (u32 sidx = 0; sidx < 16u; ++sidx) {
                        u32 const cb = (cw >> (2u * sidx)) & 3u;
                        u64 const c_hp = cb & 1u, c_hn = cb >> 1;
                        u32 const rsym = refs[16 * b + (int)sidx - ref_base] & 7u;
                        u64 const e = rsym == 0 ? eq[0] : rsym == 1 ? eq[1] : rsym == 2 ? eq[2] : rsym == 3 ? eq[3] : rsym == 4 ? eq[4] : rsym == 5 ? eq[5] : padmask;
                        u64 const x_ = e | mv;
                        u64 const sum = pv + (x_ & pv) + c_hn;
                        u64 const d0 = (sum ^ pv) | x_;
                        u64 const hn = pv & d0;
                        u64 const hp = mv | ~(pv | d0);
                        u64 const xh = (hp << 1) | c_hp;
                        mv = xh & d0;
                        pv = (hn << 1) | ~(xh | d0) | c_hn;
                        ulonglong2 o;
                        o.x = hp;
                        o.y = pv;
                        win[lane * 17u + sidx] = o;
                    }
                    valid = 1u;
                }
            }
            win_valid[lane] = valid;
        }
        __syncthreads();
        // ---- walk while the round's windows cover the walker
        bool progressed = false;
        while (i > 0 && !overflow) {
            if (j == 0) break;
            // lane l looks at cell (i - l, j - l)
            bool const in_range = (int)lane < i && (int)lane < j;
            bool have = false, up = false, left = false, same = false;
            if (in_range) {
                int const ci = i - (int)lane, cj = j - (int)lane;
                int const w = (ci - 1 + pad) >> 6;
                u32 const bit = (u32)(ci - 1 + pad) & 63u;
                int const c = cj - 1;
                int const wslot = gw_top - w;
                int const bslot = (c >> 4) - first_block(w, diag);
                int const col = c - ref_base;
                if (wslot < (int)TBW_WORDS && bslot >= 0 && bslot < (int)TBW_BLOCKS && col >= 0) {
                    u32 const slot = (u32)wslot * TBW_BLOCKS + (u32)bslot;
                    if (win_valid[slot]) {
                        have = true;
                        ulonglong2 const v = win[slot * 17u + ((u32)c & 15u)];
                        up = (v.y >> bit) & 1ull;
                        left = (v.x >> bit) & 1ull;
                        u32 const rsym = refs[col] & 7u;
                        same = rsym < 6u && ((eqm[wslot][rsym] >> bit) & 1ull);
                    }
                }
            }
            u64 const m_range = __ballot(in_range);
            u64 const m_have = __ballot(have);
            u64 const m_miss = m_range & ~m_have;
            u32 const n_range = (u32)__popcll(m_range);                     // cells on this diagonal (contiguous from lane 0)
            u32 const n_have = m_miss ? (u32)__builtin_ctzll(m_miss) : n_range;   // cells from lane 0 up to the first one without a window
            if (n_have == 0) break;                                         // the walker's own cell is not covered: next round
            u64 const m_indel = __ballot(up || left) & m_have;
            u64 const m_eq = __ballot(same);
            u32 n_diag = m_indel ? (u32)__builtin_ctzll(m_indel) : 64u;     // diagonal cells before the first indel
            bool const take_indel = n_diag < n_have;
            if (n_diag > n_have) n_diag = n_have;
            // run-length encode the diagonal stretch [0, n_diag)
            u32 pos = 0;
            while (pos < n_diag) {
                bool const is_eq = (m_eq >> pos) & 1ull;
                u64 const sm = is_eq ? m_eq : ~m_eq;
                u64 const rest = ~(sm >> pos);                              // first position (relative) where the kind changes
                u32 run = rest ? (u32)__builtin_ctzll(rest) : 64u - pos;
                if (run > n_diag - pos) run = n_diag - pos;
                emit(is_eq ? 7u : 8u, run);
                pos += run;
            }
            i -= (int)n_diag;
            j -= (int)n_diag;
            if (take_indel) {
                // the cell at lane n_diag takes an indel: up (I) has priority over left (D)
                bool const is_up = __shfl((int)up, (int)n_diag) != 0;
                if (is_up) { emit(1u, 1u); --i; }
                else { emit(2u, 1u); --j; }
            }
            progressed = true;
        }
        if (!progressed && i > 0 && j > 0 && !overflow) { overflow = true; }   // (cannot happen: the walker's window is always in its own round)
    }
    if (!overflow && cur_len) {
        if (wpos == 0) overflow = true;
        else { --wpos; if (lane == 0) slab[wpos] = (cur_len << 4) | cur_op; }
    }
    if (lane == 0) {
        DevTraceOut o;
        o.begin = (u32)j;
        o.cigar_start = wpos;
        o.cigar_len = overflow ? 0xFFFFFFFFu : job.cigar_cap - wpos;
        o.pad = 0;
        out[job.out_index] = o;
    }
    }
}

int DeviceApi::traceback(void* stream, const u8* d_text, const u8* d_query, const u64* d_peq, const u64* d_trace, const DevTraceJob* d_jobs,
                         u32 n_jobs, bool checkpointed, u32* d_cigar, DevTraceOut* d_out) {
    if (n_jobs == 0) return 0;
    static u32 const tb_waves = [] { const char* e = getenv("FLX_TRACEBACK_WAVES"); return (u32)(e ? std::max(1, atoi(e)) : 1u << 30); }();
    if (checkpointed)
        hipLaunchKernelGGL(ed_traceback_wave_kernel, dim3(std::min(n_jobs, tb_waves)), dim3(64), 0, (hipStream_t)stream, d_text, d_peq, d_trace, d_jobs, n_jobs,
                           d_cigar, d_out);
    else
        hipLaunchKernelGGL(ed_traceback_kernel, dim3(n_jobs), dim3(64), 0, (hipStream_t)stream, d_text, d_query, d_trace, d_jobs,
                           n_jobs, d_cigar, d_out);
    return (int)hipGetLastError();
}

}  // namespace flx
