// K1, the seeding kernels for gfx950 (wave64), and the tables they need beyond the index image. The per-lane walks are in
// flx_fm_core.hpp; this file holds what a wave does around them: handing seeds / queued subtrees to idle lanes from a global
// counter, reserving output slots 64 at a time, and the builders of the inverse suffix array, the presence filter and the 2-bit
// form of a sequence pool.
//
//   fm_search_filter_kernel   one lane per seed: the walk over intervals of more than one row (rank queries on the occurrence
//                             tables), children of forced runs tested against the presence filter first, one-row subtrees queued
//   fm_search_text_kernel     one lane per queued subtree: the walk against the text itself
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "flx_fm_core.hpp"

namespace flx {

namespace {

__device__ __forceinline__ u32 s_lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ u32 s_wave_sum(u32 v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += (u32)__shfl_xor((int)v, off);
    return v;
}

// ================================================================================================ derived tables
__global__ void __launch_bounds__(256) isa_kernel(const u32* __restrict__ sa, u64 n, u32* __restrict__ isa) {
    u64 const i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) isa[sa[i]] = (u32)i;
}

constexpr u32 FILTER_SPAN = 64;             // window ends per thread
__global__ void __launch_bounds__(256) filter_build_kernel(const u8* __restrict__ text, u64 n, u32 K, u32 tmin, u64* __restrict__ bits) {
    u64 const t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    i64 const q0 = (i64)(t * FILTER_SPAN);
    if (q0 >= (i64)n) return;
    i64 const q1 = q0 + (i64)FILTER_SPAN < (i64)n ? q0 + (i64)FILTER_SPAN : (i64)n;
    filter_add_range(text, (i64)n, q0, q1, K, tmin, [&](u64 word, u64 mask) { atomicOr((unsigned long long*)&bits[word], (unsigned long long)mask); });
}

// 16 symbols per thread -> one word
__global__ void __launch_bounds__(256) pack_pool_kernel(const u8* __restrict__ seq, u64 len, u32* __restrict__ qpack, u64 n_words) {
    u64 const w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    qpack[w] = pack_word(seq, len, w);
}

}  // namespace

static u32 filter_k_for(u64 n) {        // FLX_FILTER_K overrides the default (0 = no filter)
    if (const char* env = getenv("FLX_FILTER_K")) {
        u32 const k = (u32)atoi(env);
        return k == 0 ? 0u : std::max(FILTER_MIN_K, std::min(FILTER_MAX_K, k));
    }
    return filter_k_default(n);
}

size_t DeviceApi::derived_bytes(u64 n, u32* k_out) {
    u32 const k = filter_k_for(n);
    if (k_out) *k_out = k;
    return (size_t)n * 4 + (k ? (size_t)filter_words(k) * 8 : 0);
}

// isa and the presence filter from idx.text / idx.sa into d_isa (n words) and d_filter (filter_words(k) 64-bit words, k = filter_k_for(n);
// may be null with k == 0); sets the four derived fields of idx
int DeviceApi::derive_index(void* stream, DevIndex& idx, u32* d_isa, u64* d_filter) {
    hipStream_t s = (hipStream_t)stream;
    u64 const n = idx.n;
    u32 const k = d_filter ? filter_k_for(n) : 0u;
    idx.isa = d_isa;
    idx.filter = k ? d_filter : nullptr;
    idx.filter_k = k;
    idx.filter_tmin = k ? filter_tmin_for(n, k) : 0u;
    if (n == 0) return 0;
    hipLaunchKernelGGL(isa_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, idx.sa, n, d_isa);
    if (k) {
        hipError_t e = hipMemsetAsync(d_filter, 0, (size_t)filter_words(k) * 8, s);
        if (e != hipSuccess) return (int)e;
        u64 const threads = (n + FILTER_SPAN - 1) / FILTER_SPAN;
        hipLaunchKernelGGL(filter_build_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, idx.text, n, k, idx.filter_tmin, d_filter);
    }
    return (int)hipGetLastError();
}

// A kernel that needs scratch (private segment) makes the runtime reserve it for the kernel's hardware queue when it is first launched
// there; when HBM is full by then the runtime aborts the process. Launched once per lane when the context is made, this reserves what
// the largest scratch user of the pipeline needs (seed_select_kernel<8>: 400 B per lane) while memory is still free.
// one block per read: its 2 x n_leaves seeds (forward then reverse complement, leaf by leaf: the caller's order, which the ids follow)
__global__ void __launch_bounds__(256) seed_build_kernel(const DevSeedRead* __restrict__ reads, u32 n_reads, const DevSeedLeaf* __restrict__ leaves,
                                                         const DevSeedClass* __restrict__ classes, DevSeed* __restrict__ out) {
    u32 const r = blockIdx.x;
    if (r >= n_reads) return;
    DevSeedRead const rd = reads[r];
    for (u32 t = threadIdx.x; t < 2u * rd.n_leaves; t += blockDim.x) {
        u32 const o = t >= rd.n_leaves ? 1u : 0u, l = t - o * rd.n_leaves;
        DevSeedLeaf const lf = leaves[rd.leaf_first + l];
        DevSeedClass const c = classes[rd.class_first + lf.cls];
        DevSeed d;
        d.seq_off = (o ? rd.pool_rev : rd.pool_fwd) + lf.from;
        d.length = lf.length;
        d.scheme_off = c.scheme_off;
        d.frames_searches = c.frames_searches;
        d.stack_off = 0;
        d.id = rd.seed_base + t;
        d.flags = rd.flags;
        d.pad = 0;
        out[c.pos_base + o * c.count + lf.rank] = d;
    }
}
int DeviceApi::build_seeds(void* stream, const DevSeedRead* reads, u32 n_reads, const DevSeedLeaf* leaves, const DevSeedClass* classes, DevSeed* out) {
    if (n_reads == 0) return 0;
    hipLaunchKernelGGL(seed_build_kernel, dim3(n_reads), dim3(256), 0, (hipStream_t)stream, reads, n_reads, leaves, classes, out);
    return (int)hipGetLastError();
}

__global__ void __launch_bounds__(64) scratch_warm_kernel(u32* __restrict__ sink) {
    volatile u32 a[128];
    a[threadIdx.x & 127u] = threadIdx.x;
    a[(threadIdx.x * 5u + 1u) & 127u] = 1u;
    if (sink) *sink = a[(threadIdx.x * 7u) & 127u];
}
int DeviceApi::warm_scratch(void* stream) {
    hipLaunchKernelGGL(scratch_warm_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (u32*)nullptr);
    return (int)hipGetLastError();
}

int DeviceApi::pack_pool(void* stream, const u8* d_seq, u64 len, u32* d_qpack) {
    u64 const n_words = pack_words_for(len);
    hipLaunchKernelGGL(pack_pool_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_seq, len, d_qpack, n_words);
    return (int)hipGetLastError();
}

// ================================================================================================ the search kernels
// counters (32 words): [0] hit slots reserved, [1] frame overflow flag, [2] cursor extensions (rank pairs), [3] subtrees queued (records
//   written, without the unused ends of the slot ranges), [13] hits written (both kernels), [14] subtrees handed from lane to lane,
//   [4] wave-iterations, [5] their maximum over the waves, [6] busy lane-iterations, [7] seed queue head, [8] wave-iterations after the
//   seed queue ran dry, [9] their maximum, [10] filter lookups, [11] children dropped by the filter, [12] searches ended by the prefix
//   lookup, [16] item slots reserved, [17] item queue head, [18] text-mode lane-steps, [19] text-mode wave-iterations
namespace {
constexpr u32 FM_GRAB = 64;
constexpr u32 FM_OUT_GRAB = 64;

// wave-uniform bookkeeping of a grabbed range of a global queue (seeds or items)
struct WaveQueue {
    u32 next = 0, end = 0;
    bool done = false;
};
// k = this lane's new queue position or 0xFFFFFFFF. `want`: the lane is idle and wants one; `idle` = ballot of want (not 0).
__device__ __forceinline__ u32 wave_queue_take(WaveQueue& Q, u32* __restrict__ head, u32 n_total, bool want, u64 idle, u32 lane, u64 lanes_below) {
    u32 const n_idle = (u32)__popcll(idle);
    u32 const avail = Q.end - Q.next;
    u32 new_base = 0;
    bool grabbed = false;
    if (avail < n_idle && !Q.done) {
        u32 b = 0;
        if (lane == 0) b = atomicAdd(head, FM_GRAB);
        new_base = (u32)__builtin_amdgcn_readfirstlane((int)b);
        grabbed = true;
    }
    u32 const r = (u32)__popcll(idle & lanes_below);
    u32 k = 0xFFFFFFFFu;
    if (want) {
        if (r < avail) k = Q.next + r;
        else if (grabbed && new_base + (r - avail) < n_total) k = new_base + (r - avail);
    }
    if (grabbed) {
        if (new_base >= n_total) { Q.next = 0; Q.end = 0; Q.done = true; }
        else {
            Q.end = min(new_base + FM_GRAB, n_total);
            Q.next = min(new_base + (n_idle - avail), Q.end);
            Q.done = new_base + FM_GRAB >= n_total;
        }
    } else Q.next += min(n_idle, avail);
    return k;
}

// output slots, reserved FM_OUT_GRAB at a time per wave (one global atomic per range instead of one per record); the unused rest of
// a range is filled with records of seed 0xFFFFFFFF, which the consumers skip
struct WaveSlots { u32 next = 0, end = 0; };
__device__ __forceinline__ void wave_slots_close(WaveSlots const& S, DevHit* __restrict__ buf, u32 cap, u32 lane) {
    u32 const at = S.next + lane;
    if (at < S.end && at < cap) buf[at] = DevHit{0xFFFFFFFFu, 0u, 0u, 0u, 0ull};
}
// returns this lane's slot (valid when `mine`); `emit` = ballot of mine, not 0
__device__ __forceinline__ u32 wave_slots_take(WaveSlots& S, DevHit* __restrict__ buf, u32 cap, u32* __restrict__ counter, u64 emit, u32 lane, u64 lanes_below) {
    u32 const n_emit = (u32)__popcll(emit);
    if (S.end - S.next < n_emit) {
        wave_slots_close(S, buf, cap, lane);
        u32 b = 0;
        if (lane == 0) b = atomicAdd(counter, FM_OUT_GRAB);
        S.next = (u32)__builtin_amdgcn_readfirstlane((int)b);
        S.end = S.next + FM_OUT_GRAB;
    }
    u32 const slot = S.next + (u32)__popcll(emit & lanes_below);
    S.next += n_emit;
    return slot;
}
}  // namespace

// STATS: the diagnostic counters [11], [12] are kept (two more registers per lane). 111 VGPRs, no scratch (round 3: 128 + 72 B per lane
// of spills inside the DFS loop, which went through HBM).
//
// Work sharing inside a wave (seed_rows != null). On a text with repeat families a launch used to be the tail of its heaviest seeds: a
// seed inside a diverged family walks tens of thousands of steps while the other lanes of its wave have run out of seeds (round 3: 97 %
// of the wave-iterations after the queue ran dry, a tenth of the lanes busy). Once the wave's part of the seed queue is dry, an idle
// lane takes the upper half of the not yet visited error children of a busy lane's BOTTOM frame (the shallowest one: the largest
// subtrees): it copies the frame (18 words, LDS to LDS) with that half as its mask and carries on as if it had got there itself, under
// the donor's seed, search and keys; the donor keeps the lower half and the match child. Hits carry keys that restore the reference's
// emission order whoever emits them, so nothing downstream changes. What the walk of one seed may stop on - more rows than the hard
// cap - is counted per seed in global memory (seed_rows), so that the lanes sharing a seed stop together.
template <bool STATS>
__global__ void __launch_bounds__(64, 4) fm_search_filter_kernel(FmConst C, u32 n_seeds, DevHit* __restrict__ hits,
                                                              u32 hit_cap, DevHit* __restrict__ items, u32 item_cap, u32* __restrict__ counters,
                                                              u32* __restrict__ seed_cnt, u32* __restrict__ seed_rows, u32 refill, u32 prio, u32 steal_min) {
    extern __shared__ u32 lds[];                // frames: [level][FM_FRAME_WORDS][64 lanes], then 64 words for the pairing of lanes
    if (prio == 3u) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2u) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1u) __builtin_amdgcn_s_setprio(1);
    u32 const lane = s_lane_id();
    u64 const lanes_below = (1ull << lane) - 1ull;
    auto fr = [&](u32 level, u32 word) -> u32& { return lds[(level * FM_FRAME_WORDS + word) * 64u + lane]; };
    const DevSeed* __restrict__ seeds = C.seeds;
    WaveQueue Q;
    WaveSlots HS, IS;
    FmLane L;
    bool exhausted = false;
    u32 n_iter = 0, n_busy_iter = 0, n_tail_iter = 0, n_hits = 0, n_items = 0, n_steals = 0;      // (wave-uniform but n_busy_iter)
    u32* const pair = lds + C.levels * FM_FRAME_WORDS * 64u;

    while (true) {
        // ---- what the lanes produced in the last iteration (read from the node's registers, see FmLane)
        u32 const out = L.out();
        u64 const emit_h = __ballot(out == FM_OUT_HIT);
        if (emit_h) {
            n_hits += (u32)__popcll(emit_h);
            u32 const slot = wave_slots_take(HS, hits, hit_cap, &counters[0], emit_h, lane, lanes_below);
            if (out == FM_OUT_HIT) {
                u32 const sid = seeds[L.pos].id;
                bool keep = true;
                if (seed_rows) {
                    // rows of this seed over all the lanes that walk it (one lane unless children were handed over: then the lane's own count
                    // in fm_step is only a part). A seed with max_hits rows is excluded downstream whatever else it has (search.cpp:190-202).
                    u32 const before = atomicAdd(&seed_rows[sid], L.nlen);
                    keep = before < C.max_hits;
                    if (before + L.nlen >= C.max_hits) L.wn &= ~(WN_BUSY | WN_IN_SEARCH);
                }
                u32 const ord = seed_cnt && keep ? atomicAdd(&seed_cnt[sid], 1u) : 0u;
                if (slot < hit_cap) hits[slot] = keep ? DevHit{sid, L.nlb, L.nlen, seed_cnt ? L.ne() | (min(ord, 0xFFFFFFu) << 8) : L.ne(), L.nkey}
                                                      : DevHit{0xFFFFFFFFu, 0u, 0u, 0u, 0ull};
            }
        }
        u64 const emit_i = __ballot(out == FM_OUT_ITEM);
        if (emit_i) {
            n_items += (u32)__popcll(emit_i);
            u32 const slot = wave_slots_take(IS, items, item_cap, &counters[16], emit_i, lane, lanes_below);
            if (out == FM_OUT_ITEM && slot < item_cap) items[slot] = DevHit{L.pos, L.nlb, L.item_word(), 0u, L.nkey};
        }
        if (out == 3u) { atomicOr(&counters[1], 1u); L.wn = 0; }        // frames ran out (the caller repeats the launch another way)
        else L.clear_out();
        // ---- seeds for the idle lanes, and the start of the next search for the lanes between two searches - in batches. Taking a seed
        //      and starting a search are chains of dependent loads (seed record, scheme entries, packed symbols, filter word, k-mer
        //      table: ~7 us) that every lane of the wave waits for; a lane gets there every dozen iterations, so with 64 lanes some lane
        //      is there in every iteration. Lanes at that point therefore wait until `refill` of them are, or no lane is inside a search.
        bool const busy = L.busy(), in_search = L.in_search();
        bool const want = !busy && !exhausted;
        bool const boundary = want || (busy && !in_search);
        u64 const at_boundary = __ballot(boundary);
        bool const go = (u32)__popcll(at_boundary) >= refill || !__any(busy && in_search);
        u64 const idle = go ? __ballot(want) : 0ull;
        if (idle) {                                                     // wave-uniform
            u32 const k = wave_queue_take(Q, &counters[7], n_seeds, want, idle, lane, lanes_below);
            if (want) {
                if (k != 0xFFFFFFFFu) fm_take_seed(C, L, seeds[k], k);
                else exhausted = true;
            }
        }
        bool const tail = Q.done && Q.next == Q.end;                   // wave-uniform: this wave gets no more seeds
        if (tail && seed_rows) {
            if (!L.busy()) exhausted = true;                            // (a lane that waited for a batch of seeds: there is none)
            bool const idle_lane = !L.busy();
            u64 const thieves = __ballot(idle_lane);
            if ((u32)__popcll(thieves) >= steal_min) {
                // donors: lanes inside a search whose bottom frame has error children to spare
                u32 fmask = 0;
                if (L.busy() && L.in_search() && L.depth() >= 1u) fmask = fr(0, 14);
                u32 const costly = fmask & ~1u;
                u32 const n_costly = fm_popc(costly);
                bool const donor = n_costly >= 2u || (n_costly == 1u && (fmask & 1u));
                u64 const donors = __ballot(donor);
                if (donors) {
                    u32 const n_pairs = min((u32)__popcll(thieves), (u32)__popcll(donors));
                    u32 const my_rank = (u32)__popcll((donor ? donors : thieves) & lanes_below);
                    if (donor) pair[my_rank] = lane;
                    __syncthreads();
                    bool const takes = idle_lane && my_rank < n_pairs, gives = donor && my_rank < n_pairs;
                    u32 const from = takes ? pair[my_rank] : lane;
                    // the donor's seed and search
                    u32 const d_pos = (u32)__shfl((int)L.pos, (int)from), d_exo = (u32)__shfl((int)L.exo, (int)from), d_ws = (u32)__shfl((int)L.ws, (int)from);
                    u32 const d_qlo = (u32)__shfl((int)(u32)L.qoff, (int)from), d_qhi = (u32)__shfl((int)(u32)(L.qoff >> 32), (int)from);
                    u32 w[FM_FRAME_WORDS];
#pragma unroll
                    for (u32 i = 0; i < FM_FRAME_WORDS; ++i) w[i] = lds[i * 64u + from];          // frame 0 of lane `from`
                    __syncthreads();
                    // the upper half of the error children goes (all of them but one when there is no match child to stay behind)
                    u32 const m = w[14], mc = m & ~1u, nc = fm_popc(mc), g = (m & 1u) ? (nc + 1u) / 2u : nc / 2u;
                    u32 give = mc;
                    for (u32 i = g; i < nc; ++i) give &= give - 1u;                                 // drop the lowest nc - g
                    if (gives) fr(0, 14) = m & ~give;
                    if (takes) {
#pragma unroll
                        for (u32 i = 0; i < FM_FRAME_WORDS; ++i) if (i != 14u) fr(0, i) = w[i];
                        fr(0, 14) = give;
                        L.pos = d_pos; L.exo = d_exo; L.ws = d_ws | (1u << 28); L.qoff = (u64)d_qlo | ((u64)d_qhi << 32);
                        L.ct = 0;
                        L.wn = WN_BUSY | WN_IN_SEARCH | WN_NEED_CHILD | WN_DEPTH1;
                    }
                    n_steals += n_pairs;
                }
            }
        }
        if (__all(exhausted && !L.busy())) break;
        ++n_iter;
        if (tail) ++n_tail_iter;
        if (!L.busy() || (!go && !L.in_search())) continue;
        ++n_busy_iter;
        fm_step<STATS>(C, L, fr);
    }
    wave_slots_close(HS, hits, hit_cap, lane);
    wave_slots_close(IS, items, item_cap, lane);
    u32 const s_ext = s_wave_sum(L.n_ext), s_busy = s_wave_sum(n_busy_iter), s_look = s_wave_sum(L.n_lookup);
    if (lane == 0) {
        atomicAdd(&counters[2], s_ext); atomicAdd(&counters[6], s_busy);
        atomicAdd(&counters[4], n_iter); atomicMax(&counters[5], n_iter); atomicAdd(&counters[8], n_tail_iter); atomicMax(&counters[9], n_tail_iter);
        atomicAdd(&counters[10], s_look); atomicAdd(&counters[13], n_hits); atomicAdd(&counters[3], n_items); atomicAdd(&counters[14], n_steals);
    }
    if (STATS) {
        u32 const s_pruned = s_wave_sum(L.n_pruned), s_kills = s_wave_sum(L.n_prefix_kills);
        if (lane == 0) { atomicAdd(&counters[11], s_pruned); atomicAdd(&counters[12], s_kills); }
    }
}

__global__ void __launch_bounds__(64) fm_search_text_kernel(FmConst C, const DevSeed* __restrict__ seeds, const DevHit* __restrict__ items, u32 item_cap,
                                                            DevHit* __restrict__ hits, u32 hit_cap, u32* __restrict__ counters, u32* __restrict__ seed_cnt,
                                                            u32 refill, u32 prio) {
    extern __shared__ u32 lds[];                // frames: [level][TX_FRAME_WORDS][64 lanes]
    if (prio == 3u) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2u) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1u) __builtin_amdgcn_s_setprio(1);
    u32 const lane = s_lane_id();
    u64 const lanes_below = (1ull << lane) - 1ull;
    auto fr = [&](u32 level, u32 word) -> u32& { return lds[(level * TX_FRAME_WORDS + word) * 64u + lane]; };
    u32 const n_slots = min(counters[16], item_cap);      // (the filter kernel has finished: same stream)
    WaveQueue Q;
    WaveSlots HS;
    TxLane L;
    bool exhausted = false;
    u32 n_iter = 0, n_hits = 0;

    while (true) {
        u64 const emit_h = __ballot(L.out == FM_OUT_HIT);
        if (emit_h) {
            n_hits += (u32)__popcll(emit_h);
            u32 const slot = wave_slots_take(HS, hits, hit_cap, &counters[0], emit_h, lane, lanes_below);
            if (L.out == FM_OUT_HIT) {
                u32 const ord = seed_cnt ? atomicAdd(&seed_cnt[L.sid], 1u) : 0u;
                if (slot < hit_cap) hits[slot] = DevHit{L.sid, L.out_lb, 1u, seed_cnt ? L.out_e | (min(ord, 0xFFFFFFu) << 8) : L.out_e, L.out_key};
                L.out = FM_OUT_NONE;
            }
        }
        // (subtrees are handed out in batches too: a hand-out is three dependent loads - item, seed record, suffix array - for the whole wave)
        bool const want = !L.busy && !exhausted;
        u64 const wanting = __ballot(want);
        u64 const idle = ((u32)__popcll(wanting) >= refill || !__any(L.busy)) ? wanting : 0ull;
        if (idle) {
            u32 const k = wave_queue_take(Q, &counters[17], n_slots, want, idle, lane, lanes_below);
            if (want) {
                if (k != 0xFFFFFFFFu) {
                    DevHit const item = items[k];
                    if (item.seed != 0xFFFFFFFFu) tx_take_item(C, L, item, seeds[item.seed]);      // (an unused slot: ask again)
                } else exhausted = true;
            }
        }
        if (__all(exhausted && !L.busy)) break;
        ++n_iter;
        if (!L.busy) continue;
        tx_step(C, L, fr);
    }
    wave_slots_close(HS, hits, hit_cap, lane);
    if (__any(L.overflow) && lane == 0) atomicOr(&counters[1], 1u);
    u32 const s_nodes = s_wave_sum(L.n_nodes);
    if (lane == 0) { atomicAdd(&counters[18], s_nodes); atomicAdd(&counters[19], n_iter); atomicAdd(&counters[13], n_hits); }
}

static u32 env_u32(const char* name, u32 dflt) {
    const char* e = getenv(name);
    return e ? (u32)strtoul(e, nullptr, 10) : dflt;
}

// The search with the stack in LDS, the presence filter and the text walk. d_items: item_cap records for queued subtrees (0: none are
// queued). frame_levels = largest error count of a seed. d_counters: 32 words, zeroed by the caller.
int DeviceApi::search_filtered(void* stream, const DevIndex& idx, const u8* d_seq, const u32* d_qpack, const u64* d_scheme, const DevSeed* d_seeds,
                               u32 n_seeds, u32 max_hits_per_seed, u32 frame_levels, DevHit* d_hits, u32 hit_cap, DevHit* d_items, u32 item_cap,
                               u32* d_counters, u32* d_seed_cnt, u32* d_seed_rows, u32 concurrent_launches) {
    if (n_seeds == 0) return 0;
    static u32 const spw = env_u32("FLX_FM_SEEDS_PER_WAVE", 256);
    static u32 const forced_waves = env_u32("FLX_FM_MAX_WAVES", 0);
    u32 const no_filter = env_u32("FLX_FM_NO_FILTER", 0), no_text = env_u32("FLX_FM_NO_TEXT", 0);      // (read per call: tests switch them)
    u32 const text_min = env_u32("FLX_FM_TEXT_MIN", 2);
    static u32 const refill_a = std::max(1u, std::min(64u, env_u32("FLX_FM_REFILL", 16))), refill_t = std::max(1u, std::min(64u, env_u32("FLX_FM_TEXT_REFILL", 16)));
    u32 const max_waves = forced_waves ? forced_waves : 4096u / std::max(1u, std::min(concurrent_launches, 8u));
    FmConst C{};
    C.idx = idx;
    C.seq = d_seq;
    C.qpack = d_qpack;
    C.scheme = d_scheme;
    C.seeds = d_seeds;
    C.max_hits = max_hits_per_seed;
    C.levels = std::max(1u, frame_levels);
    static u32 const looks = env_u32("FLX_FM_LOOKS", 2);
    C.use_filter = (idx.filter && d_qpack && !no_filter) ? std::max(1u, std::min(2u, looks)) : 0u;
    C.text_min_remain = (d_items && item_cap && idx.isa && !no_text) ? std::max(1u, text_min) : 0u;
    static u32 const fm_prio = env_u32("FLX_FM_PRIO", 0);         // wave priority of the walks (0..3) on SIMDs shared with other kernels
    hipStream_t s = (hipStream_t)stream;
    dim3 const grid(std::min<u32>((n_seeds + spw - 1) / spw, max_waves));
    static u32 const extra_lds = env_u32("FLX_FM_EXTRA_LDS", 0);      // (experiments: bytes of LDS a wave holds without using them)
    size_t const lds_bytes = (size_t)C.levels * FM_FRAME_WORDS * 64 * sizeof(u32) + 64 * sizeof(u32) + extra_lds;
    u32 const steal_min = env_u32("FLX_FM_STEAL_MIN", 1);              // idle lanes a wave waits for before it shares work (0: never)
    u32* const seed_rows = steal_min ? d_seed_rows : nullptr;
    static u32 const stats = env_u32("FLX_SEARCH_DEBUG", 0);           // the diagnostic counters [11], [12] cost two registers per lane
    auto const kernel = stats ? fm_search_filter_kernel<true> : fm_search_filter_kernel<false>;
    hipLaunchKernelGGL(kernel, grid, dim3(64), lds_bytes, s, C, n_seeds, d_hits, hit_cap, d_items, item_cap, d_counters, d_seed_cnt, seed_rows, refill_a, fm_prio, std::max(1u, steal_min));
    if (C.text_min_remain) {
        // (the number of queued subtrees is only known on the device: a fixed grid, waves without work leave at once; the walk is a
        // chain of dependent loads from the L2, so it wants every wave slot: 8 per SIMD)
        static u32 const text_waves = env_u32("FLX_FM_TEXT_WAVES", 8192);
        u32 const tw = std::max(1u, text_waves / std::max(1u, std::min(concurrent_launches, 8u)));
        hipLaunchKernelGGL(fm_search_text_kernel, dim3(std::min<u32>(std::max<u32>(grid.x * 4u, 64u), tw)), dim3(64), (size_t)C.levels * TX_FRAME_WORDS * 64 * sizeof(u32), s, C,
                           d_seeds, d_items, item_cap, d_hits, hit_cap, d_counters, d_seed_cnt, refill_t, fm_prio);
    }
    return (int)hipGetLastError();
}

}  // namespace flx
