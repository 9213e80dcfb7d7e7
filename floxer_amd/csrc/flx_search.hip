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

struct FrameItem { u32 w[24]; };               // [0..17] the frame (word 14: the children that go), [18] seed position, [19] exo, [20] ws, [21..22] qoff, [23] rows of the seed so far
// Hand-over of subtrees between the waves of one launch (see fm_search_filter_kernel). All in HBM, zeroed before the launch but `items`:
//   ctrl: waves started (MBC_STARTED), waves waiting or gone (MBC_WAITING), tail / head of the hungry list (MBC_TAIL, MBC_HEAD), a flag set once a
//   wave's queue has been dry for long (MBC_LONG: the launch has heavy seeds: finished waves then wait for work instead of leaving) - each in
//   a 128-byte line of its own: thousands of waiting waves poll MBC_WAITING, and the busy waves' looks at the list must not queue behind them
//   state[w]: a wave's mailbox (MB_*); count[w]: the subtrees in it; hungry[]: waves waiting, in the order they asked (wave + 1)
enum : u32 { MB_INACTIVE = 0, MB_WAITING = 1, MB_FILLING = 2, MB_FILLED = 3, MB_CLOSED = 4 };
struct Mailboxes {
    u32* ctrl; u32* state; u32* count; u32* hungry; FrameItem* items;      // null ctrl: no hand-over between waves
    u32 n_hungry;                                                           // entries of the hungry list
};
constexpr u32 MBC_STARTED = 0, MBC_WAITING = 32, MBC_TAIL = 64, MBC_HEAD = 96, MBC_LONG = 128;
constexpr u32 MB_CTRL_WORDS = 160, MB_HUNGRY_PER_WAVE = 4;

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
__global__ void __launch_bounds__(256) filter_build_kernel(const u8* __restrict__ text, u64 n, u32 K, u32 tmin, u64* __restrict__ bits, u64* __restrict__ bits_m) {
    u64 const t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    i64 const q0 = (i64)(t * FILTER_SPAN);
    if (q0 >= (i64)n) return;
    i64 const q1 = q0 + (i64)FILTER_SPAN < (i64)n ? q0 + (i64)FILTER_SPAN : (i64)n;
    filter_add_range(text, (i64)n, q0, q1, K, tmin, [&](u64 word, u64 mask) { atomicOr((unsigned long long*)&bits[word], (unsigned long long)mask); },
                     [&](u64 word, u64 mask) { if (bits_m) atomicOr((unsigned long long*)&bits_m[word], (unsigned long long)mask); });
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

// one launch's mailboxes for up to `waves` waves: the words that are zeroed per launch first, then the subtree slots
size_t DeviceApi::mailbox_bytes(u32 waves) { return ((size_t)MB_CTRL_WORDS + (size_t)waves * (2 + MB_HUNGRY_PER_WAVE)) * 4 + 256 + (size_t)waves * 64 * sizeof(FrameItem); }

size_t DeviceApi::derived_bytes(u64 n, u32* k_out) {
    u32 const k = filter_k_for(n);
    if (k_out) *k_out = k;
    return (size_t)n * 4 + (k ? (size_t)filter_words(k) * 8 : 0);          // the inverse suffix array and the filter
}

// isa and the presence filter from idx.text / idx.sa into d_isa (n words; may be null: no text walk) and d_filter (filter_words(k) 64-bit words, k = filter_k_for(n);
// may be null with k == 0); sets the four derived fields of idx
int DeviceApi::derive_index(void* stream, DevIndex& idx, u32* d_isa, u64* d_filter, u64* d_filter_mirrored) {
    hipStream_t s = (hipStream_t)stream;
    u64 const n = idx.n;
    u32 const k = d_filter ? filter_k_for(n) : 0u;
    idx.isa = d_isa;
    idx.filter = k ? d_filter : nullptr;
    idx.filter_m = k ? d_filter_mirrored : nullptr;
    idx.filter_k = k;
    idx.filter_tmin = k ? filter_tmin_for(n, k) : 0u;
    if (n == 0) return 0;
    if (d_isa) hipLaunchKernelGGL(isa_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, idx.sa, n, d_isa);
    if (k) {
        hipError_t e = hipMemsetAsync(d_filter, 0, (size_t)filter_words(k) * 8, s);
        if (e != hipSuccess) return (int)e;
        if (d_filter_mirrored && (e = hipMemsetAsync(d_filter_mirrored, 0, (size_t)filter_words(k) * 8, s)) != hipSuccess) return (int)e;
        u64 const threads = (n + FILTER_SPAN - 1) / FILTER_SPAN;
        hipLaunchKernelGGL(filter_build_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, idx.text, n, k, idx.filter_tmin, d_filter, d_filter_mirrored);
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
//   written, without the unused ends of the slot ranges), [13] hits written (both kernels), [14] subtrees handed from lane to lane, [15] subtrees handed from wave to wave,
//   [20] walks abandoned because their seed had passed the hard cap elsewhere,
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

// STATS: the diagnostic counters [11], [12] are kept (two more registers per lane). No scratch (round 3: 128 VGPRs + 72 B per lane of
// spills inside the DFS loop, which went through HBM).
//
// Work sharing (seed_rows != null). On a text with repeat families a launch used to be the tail of its heaviest seeds: a seed inside a
// diverged family walks tens of thousands of steps while the other lanes of its wave have run out of seeds (round 3: 97 % of the
// wave-iterations after the queue ran dry, a tenth of the lanes busy).
//  (1) inside a wave: once the wave's part of the seed queue is dry, an idle lane takes work from a busy lane's BOTTOM frame (the
//      shallowest one: the largest subtrees): the match child when there is one, else the upper half of the error children. It copies the
//      frame (18 words, LDS to LDS) with those children as its mask and carries on as if it had got there itself, under the donor's
//      seed, search and keys.
//  (2) between waves (MB.ctrl != null): a wave with nothing left asks for work - it puts itself on a list in HBM and polls its mailbox;
//      a busy wave that sees the list non-empty hands one subtree per lane that can spare one (FrameItem: the frame + the seed's
//      identity, 96 B) into the asking wave's mailbox. No wave waits without a bound: a waiting wave leaves when every wave of the
//      launch is waiting or gone, when not all waves of the launch are resident yet (it would hold their room), or after MB_POLLS polls;
//      a mailbox changes hands by compare-and-swap (WAITING -> FILLING by the giver, WAITING -> CLOSED by its owner), so a subtree is
//      never written into a mailbox nobody reads. Waves only wait in launches that have shown a long tail (ctrl[6]): on a uniform text
//      a launch's waves finish within a hundred iterations of each other and leave at once.
// Hits carry keys that restore the reference's emission order whoever emits them, so nothing downstream changes. What the walk of one
// seed may stop on - more rows than the hard cap - is counted per seed in global memory (seed_rows), so that the lanes and waves that
// share a seed stop together.
constexpr u32 MB_POLLS = 100000;               // x (s_sleep of 3.4 us + one load) ~ 0.5 s: the bound, not the expected wait
// (relaxed: a look at a control word must not cost the compute unit its vector cache, which an acquire would invalidate - the busy waves
// look every few iterations, the waiting ones every few microseconds; the two places where data follows a flag fence explicitly)
__device__ __forceinline__ u32 mb_load(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mb_store(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mb_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
__device__ __forceinline__ void mb_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }

template <bool STATS>
__global__ void __launch_bounds__(64, 4) fm_search_filter_kernel(FmConst C, u32 n_seeds, DevHit* __restrict__ hits,
                                                              u32 hit_cap, DevHit* __restrict__ items, u32 item_cap, u32* __restrict__ counters,
                                                              u32* __restrict__ seed_cnt, u32* __restrict__ seed_rows, Mailboxes MB, u32 refill, u32 prio, u32 steal_min, u32 steal_after, u32 cap_look) {
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
    u32 n_iter = 0, n_busy_iter = 0, n_tail_iter = 0, n_hits = 0, n_items = 0, n_steals = 0, n_given = 0;      // (wave-uniform but n_busy_iter, n_capped)
    u32 n_capped = 0;
    bool counted_waiting = false, flagged_long = false;                                                         // (wave-uniform)
    u32* const pair = lds + C.levels * FM_FRAME_WORDS * 64u;
    u32 const me = blockIdx.x;
    if (MB.ctrl && lane == 0) atomicAdd(&MB.ctrl[MBC_STARTED], 1u);

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
            bool const long_tail = n_tail_iter >= steal_after;          // (a short tail - a uniform text: 100 iterations - is over before sharing it pays)
            if (MB.ctrl && !flagged_long && n_tail_iter >= 4u * steal_after) { flagged_long = true; if (lane == 0) mb_store(&MB.ctrl[MBC_LONG], 1u); }
            bool const idle_lane = !L.busy();
            u64 const thieves = long_tail ? __ballot(idle_lane) : 0ull;
            // is a wave asking for work? (one look every fourth iteration of a long tail)
            bool asked = false;
            if (MB.ctrl && long_tail && (n_iter & 3u) == 0u) {
                u32 a = 0;
                if (lane == 0) { u32 const t = min(mb_load(&MB.ctrl[MBC_TAIL]), MB.n_hungry); a = mb_load(&MB.ctrl[MBC_HEAD]) < t ? 1u : 0u; }
                asked = __builtin_amdgcn_readfirstlane((int)a) != 0;
            }
            if ((thieves && (u32)__popcll(thieves) >= steal_min) || asked) {
                // donors: lanes inside a search whose bottom frame has something to spare
                u32 fmask = 0;
                if (L.busy() && L.in_search() && L.depth() >= 1u) fmask = fr(0, 14);
                u32 const costly = fmask & ~1u;
                u32 const n_costly = fm_popc(costly);
                bool const donor = n_costly >= 2u || (n_costly == 1u && (fmask & 1u));      // (something goes and something stays)
                u64 const donors = __ballot(donor);
                // What goes: the MATCH child when there is one - the continuation of the path without a further error, i.e. everything the
                // seed still has to do at its later positions, by far the largest subtree (the error children of one node are at most eleven
                // chains); the lane that takes it becomes a donor itself one step later, so a heavy seed's spine spreads over the wave in a
                // few iterations. Without a match child: the upper half of the error children.
                u32 give = 1u;
                if (!(fmask & 1u)) {
                    give = costly;
                    for (u32 i = n_costly / 2u; i < n_costly; ++i) give &= give - 1u;                // drop the lowest n_costly - n_costly / 2
                }
                bool gives = false;
                if (donors && thieves) {
                    u32 const n_pairs = min((u32)__popcll(thieves), (u32)__popcll(donors));
                    u32 const my_rank = (u32)__popcll((donor ? donors : thieves) & lanes_below);
                    if (donor) pair[my_rank] = lane;
                    __syncthreads();
                    bool const takes = idle_lane && my_rank < n_pairs;
                    gives = donor && my_rank < n_pairs;
                    u32 const from = takes ? pair[my_rank] : lane;
                    // the donor's seed and search
                    u32 const d_pos = (u32)__shfl((int)L.pos, (int)from), d_exo = (u32)__shfl((int)L.exo, (int)from), d_ws = (u32)__shfl((int)L.ws, (int)from);
                    u32 const d_qlo = (u32)__shfl((int)(u32)L.qoff, (int)from), d_qhi = (u32)__shfl((int)(u32)(L.qoff >> 32), (int)from);
                    u32 const d_give = (u32)__shfl((int)give, (int)from), d_ct = (u32)__shfl((int)L.ct, (int)from);
                    u32 w[FM_FRAME_WORDS];
#pragma unroll
                    for (u32 i = 0; i < FM_FRAME_WORDS; ++i) w[i] = lds[i * 64u + from];          // frame 0 of lane `from`
                    __syncthreads();
                    if (gives) fr(0, 14) = fmask & ~give;
                    if (takes) {
#pragma unroll
                        for (u32 i = 0; i < FM_FRAME_WORDS; ++i) if (i != 14u) fr(0, i) = w[i];
                        fr(0, 14) = d_give;
                        L.pos = d_pos; L.exo = d_exo; L.ws = d_ws | (1u << 28); L.qoff = (u64)d_qlo | ((u64)d_qhi << 32);
                        L.ct = d_ct;                                // (the seed's rows as the donor has counted them: both stop at the cap from there)
                        L.wn = WN_BUSY | WN_IN_SEARCH | WN_NEED_CHILD | WN_DEPTH1;
                    }
                    n_steals += n_pairs;
                }
                // what no lane of this wave took goes to a wave that asked
                bool const sends = asked && donor && !gives;
                u64 const sending = __ballot(sends);
                if (sending) {
                    // the first asking wave whose mailbox is still open (lane 0 negotiates)
                    u32 to = 0xFFFFFFFFu;
                    if (lane == 0) {
                        u32 const h = mb_load(&MB.ctrl[MBC_HEAD]);
                        if (h < min(mb_load(&MB.ctrl[MBC_TAIL]), MB.n_hungry) && atomicCAS(&MB.ctrl[MBC_HEAD], h, h + 1u) == h) {
                            u32 id = 0;
                            for (u32 spin = 0; spin < 100000u && (id = mb_load(&MB.hungry[h])) == 0u; ++spin) __builtin_amdgcn_s_sleep(1);      // (written right after the slot was taken)
                            if (id && atomicCAS(&MB.state[id - 1u], (u32)MB_WAITING, (u32)MB_FILLING) == MB_WAITING) to = id - 1u;      // (a read-modify-write sees the latest value)
                        }
                    }
                    to = (u32)__builtin_amdgcn_readfirstlane((int)to);
                    if (to != 0xFFFFFFFFu) {
                        if (sends) {
                            uint4* __restrict__ dst = reinterpret_cast<uint4*>(MB.items[(size_t)to * 64u + (u32)__popcll(sending & lanes_below)].w);
                            dst[0] = uint4{fr(0, 0), fr(0, 1), fr(0, 2), fr(0, 3)};
                            dst[1] = uint4{fr(0, 4), fr(0, 5), fr(0, 6), fr(0, 7)};
                            dst[2] = uint4{fr(0, 8), fr(0, 9), fr(0, 10), fr(0, 11)};
                            dst[3] = uint4{fr(0, 12), fr(0, 13), give, fr(0, 15)};
                            dst[4] = uint4{fr(0, 16), fr(0, 17), L.pos, L.exo};
                            dst[5] = uint4{L.ws, (u32)L.qoff, (u32)(L.qoff >> 32), L.ct};
                            fr(0, 14) = fmask & ~give;
                        }
                        if (lane == 0) MB.count[to] = (u32)__popcll(sending);
                        mb_release();                                   // the subtrees and their number before the flag
                        if (lane == 0) mb_store(&MB.state[to], (u32)MB_FILLED);
                        n_given += (u32)__popcll(sending);
                    }
                }
            }
        }
        if (__all(exhausted && !L.busy())) {
            // ---- nothing left in this wave: leave, or ask the waves that are still busy for work
            if (!MB.ctrl) break;
            u32 verdict = 0;                                            // 0 leave, 1 a mailbox with work
            if (lane == 0) {
                bool const all_started = mb_load(&MB.ctrl[MBC_STARTED]) >= gridDim.x, long_launch = mb_load(&MB.ctrl[MBC_LONG]) != 0u;
                if (all_started && long_launch) {
                    mb_store(&MB.state[me], (u32)MB_WAITING);
                    mb_release();                                       // (the mailbox is open before anyone can find it on the list)
                    u32 const slot = atomicAdd(&MB.ctrl[MBC_TAIL], 1u);
                    if (slot < MB.n_hungry) mb_store(&MB.hungry[slot], me + 1u);
                    if (!counted_waiting) atomicAdd(&MB.ctrl[MBC_WAITING], 1u);
                    u32 st = MB_WAITING;
                    if (slot < MB.n_hungry)
                        for (u32 p = 0; p < MB_POLLS; ++p) {
                            st = mb_load(&MB.state[me]);
                            if (st == MB_FILLED) break;
                            if ((p & 7u) == 7u && mb_load(&MB.ctrl[MBC_WAITING]) >= gridDim.x) break;      // nobody is left to give any
                            __builtin_amdgcn_s_sleep(127);
                        }
                    if (st != MB_FILLED) {
                        u32 const old = atomicCAS(&MB.state[me], (u32)MB_WAITING, (u32)MB_CLOSED);
                        if (old == MB_FILLING || old == MB_FILLED) {                              // a giver got in first: it is writing (it never waits)
                            for (u32 p = 0; p < MB_POLLS && (st = mb_load(&MB.state[me])) != MB_FILLED; ++p) __builtin_amdgcn_s_sleep(8);
                            if (st != MB_FILLED) atomicOr(&counters[1], 2u);                      // (cannot happen; if it does the call fails instead of losing a subtree)
                        }
                    }
                    if (st == MB_FILLED) { verdict = 1; atomicSub(&MB.ctrl[MBC_WAITING], 1u); }
                    else verdict = 2;                                   // (counted as waiting for good)
                }
            }
            verdict = (u32)__builtin_amdgcn_readfirstlane((int)verdict);
            if (verdict == 2u) counted_waiting = true;
            if (verdict != 1u) break;
            mb_acquire();                                               // the flag before the subtrees
            u32 const n_in = min(MB.count[me], 64u);
            if (lane < n_in) {
                // a subtree another wave handed over: its frame becomes this lane's bottom frame
                const uint4* __restrict__ src = reinterpret_cast<const uint4*>(MB.items[(size_t)me * 64u + lane].w);
                uint4 const q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3], q4 = src[4], q5 = src[5];
                fr(0, 0) = q0.x; fr(0, 1) = q0.y; fr(0, 2) = q0.z; fr(0, 3) = q0.w; fr(0, 4) = q1.x; fr(0, 5) = q1.y; fr(0, 6) = q1.z; fr(0, 7) = q1.w;
                fr(0, 8) = q2.x; fr(0, 9) = q2.y; fr(0, 10) = q2.z; fr(0, 11) = q2.w; fr(0, 12) = q3.x; fr(0, 13) = q3.y; fr(0, 14) = q3.z; fr(0, 15) = q3.w;
                fr(0, 16) = q4.x; fr(0, 17) = q4.y;
                L.pos = q4.z; L.exo = q4.w; L.ws = q5.x | (1u << 28); L.qoff = (u64)q5.y | ((u64)q5.z << 32);
                L.ct = q5.w;
                L.wn = WN_BUSY | WN_IN_SEARCH | WN_NEED_CHILD | WN_DEPTH1;
            }
            if (lane == 0) mb_store(&MB.state[me], (u32)MB_INACTIVE);
        }
        ++n_iter;
        if (tail) ++n_tail_iter;
        if (!L.busy() || (!go && !L.in_search())) continue;
        // In a long tail a lane looks now and then whether its seed has passed the hard cap meanwhile (the other lanes and waves that walk the
        // seed's subtrees count into the same word): such a seed is excluded downstream whatever else it has, and the lane is better used on
        // work another lane can spare. (A lane on its own seed notices at its next hit anyway; this is for the long stretches without one.)
        if (seed_rows && n_tail_iter >= steal_after && (n_iter & cap_look) == 0u && L.in_search()) {
            if (seed_rows[seeds[L.pos].id] >= C.max_hits) { L.wn = 0; ++n_capped; continue; }
        }
        ++n_busy_iter;
        fm_step<STATS>(C, L, fr);
    }
    wave_slots_close(HS, hits, hit_cap, lane);
    wave_slots_close(IS, items, item_cap, lane);
    u32 const s_ext = s_wave_sum(L.n_ext), s_busy = s_wave_sum(n_busy_iter), s_look = s_wave_sum(L.n_lookup), s_capped = s_wave_sum(n_capped);
    if (lane == 0) {
        atomicAdd(&counters[20], s_capped);
        if (MB.ctrl && !counted_waiting) atomicAdd(&MB.ctrl[MBC_WAITING], 1u);       // gone: nothing more to give
        atomicAdd(&counters[2], s_ext); atomicAdd(&counters[6], s_busy);
        atomicAdd(&counters[4], n_iter); atomicMax(&counters[5], n_iter); atomicAdd(&counters[8], n_tail_iter); atomicMax(&counters[9], n_tail_iter);
        atomicAdd(&counters[10], s_look); atomicAdd(&counters[13], n_hits); atomicAdd(&counters[3], n_items); atomicAdd(&counters[14], n_steals);
        atomicAdd(&counters[15], n_given);
    }
    if (STATS) {
        u32 const s_pruned = s_wave_sum(L.n_pruned), s_kills = s_wave_sum(L.n_prefix_kills);
        if (lane == 0) { atomicAdd(&counters[11], s_pruned); atomicAdd(&counters[12], s_kills); }
    }
}

// The text walk reads the seed and the text next to its string symbol by symbol, each read a dependent load that used to miss the L2
// (round 3: 17.7 HBM requests per queued subtree, 48 GB per 16384 reads). A lane now copies both once, when it takes a subtree: the
// seed (up to TX_WIN_MAXLEN symbols) and the text from 4 symbols left of the leftmost position the subtree can reach to 4 right of the
// rightmost (string + remaining seed symbols + one per error either side), four bits per symbol, into LDS; the walk then reads LDS.
// Longer seeds (the one leaf of a PEX tree that takes the remainder, 20-kb reads at 2 %) keep reading memory.
// Two sizes: seeds of up to 64 symbols (10-kb reads at 8 %: leaves of 24 and 36) or of up to 160 (20-kb reads at 2 %: leaves of 98 .. 147;
// 11 KB of windows per wave). MAXLEN seed symbols need (MAXLEN + 3 + 8 + 3) / 8 words of text window and (MAXLEN + 3) / 8 of seed.
template <u32 MAXLEN> struct TxWin { static constexpr u32 T = (MAXLEN + 14u + 7u) / 8u, Q = (MAXLEN + 3u + 7u) / 8u; };
__device__ __forceinline__ u32 nibbles_of(u32 lo, u32 hi) {          // eight bytes (values 0..15) -> eight nibbles, first byte lowest
    u32 a = (lo | (lo >> 4)) & 0x00FF00FFu, b = (hi | (hi >> 4)) & 0x00FF00FFu;
    a = (a | (a >> 8)) & 0xFFFFu; b = (b | (b >> 8)) & 0xFFFFu;
    return a | (b << 16);
}
// one flag (bit 0 of the nibble) per nibble of v that is not 1 .. 5 (not A, C, G, T, N)
__device__ __forceinline__ u32 nibbles_not_acgtn(u32 v) {
    u32 nz = v | (v >> 1);
    nz |= nz >> 2;                                                    // bit 0 of a nibble: the nibble is not zero
    u32 const ge6 = (v >> 3) | ((v >> 2) & (v >> 1));                 // bit 0 of a nibble: 8 or more, or 6 / 7
    return (~nz | ge6) & 0x11111111u;
}
__device__ __forceinline__ u32 nibbles_any(u32 v) { v |= v >> 1; v |= v >> 2; return v & 0x11111111u; }      // bit 0 of every nibble that is not zero
struct TxWinAccess {
    const u32* t; const u32* q;                 // this lane's windows in LDS, [word][lane]
    i64 w0; u32 qshift; bool win;               // text position of the window's first symbol; seed symbol i is window symbol i + qshift
    const u8* text; const u8* seq;
    u32 nt, nq;                                 // words of the two windows
    // eight window symbols from symbol i upwards (nibble k = symbol i + k) / downwards (nibble 7 = symbol i, nibble 6 = symbol i - 1, ...; zeros in
    // front of the window's first symbol); words past the window read as zero
    __device__ __forceinline__ u32 up8(const u32* w, u32 n, u32 i) const {
        u32 const k = i >> 3;
        u32 const lo = k < n ? w[k * 64u] : 0u, hi = k + 1u < n ? w[(k + 1u) * 64u] : 0u;
        return __builtin_amdgcn_alignbit(hi, lo, (i & 7u) * 4u);
    }
    __device__ __forceinline__ u32 down8(const u32* w, u32 n, u32 i) const { return i >= 7u ? up8(w, n, i - 7u) : (n ? w[0] << ((7u - i) * 4u) : 0u); }
    // a forced run in one go: eight symbols per comparison instead of one (the walk's instructions were these loops: 6900 per wave-iteration)
    __device__ __forceinline__ u32 run_right(u32 qp, i64 tpos, u32 run) const {
        if (!win) { u32 i = 0; for (; i < run; ++i) { u32 const c = seq[qp + i]; if (c != text[tpos + (i64)i] || c - 1u >= 5u) break; } return i; }
        u32 const qi = qp + qshift, ti = (u32)(tpos - w0);
        for (u32 i = 0; i < run; i += 8u) {
            u32 const a = up8(q, nq, qi + i), b = up8(t, nt, ti + i);
            u32 bad = nibbles_any(a ^ b) | nibbles_not_acgtn(a);
            u32 const m = run - i;
            if (m < 8u) bad &= (1u << (4u * m)) - 1u;
            if (bad) return i + ((u32)__builtin_ctz(bad) >> 2);
        }
        return run;
    }
    __device__ __forceinline__ u32 run_left(u32 qp, i64 tpos, u32 run) const {
        if (!win) { u32 i = 0; for (; i < run; ++i) { u32 const c = seq[qp - i]; if (c != text[tpos - (i64)i] || c - 1u >= 5u) break; } return i; }
        u32 const qi = qp + qshift, ti = (u32)(tpos - w0);
        for (u32 i = 0; i < run; i += 8u) {
            u32 const a = down8(q, nq, qi - i), b = down8(t, nt, ti - i);
            u32 bad = nibbles_any(a ^ b) | nibbles_not_acgtn(a);
            u32 const m = run - i;
            if (m < 8u) bad &= ~((1u << (4u * (8u - m))) - 1u);
            if (bad) return i + (7u - ((31u - (u32)__builtin_clz(bad)) >> 2));
        }
        return run;
    }
    __device__ __forceinline__ u32 text_at(i64 pos) const {
        if (!win) return text[pos];
        u32 const i = (u32)(pos - w0);
        return (t[(i >> 3) * 64u] >> ((i & 7u) * 4u)) & 15u;
    }
    __device__ __forceinline__ u32 q_at(u32 qp) const {
        if (!win) return seq[qp];
        u32 const i = qp + qshift;
        return (q[(i >> 3) * 64u] >> ((i & 7u) * 4u)) & 15u;
    }
};

template <u32 TX_WIN_MAXLEN>
__global__ void __launch_bounds__(64) fm_search_text_kernel(FmConst C, const DevSeed* __restrict__ seeds, const DevHit* __restrict__ items, u32 item_cap,
                                                            DevHit* __restrict__ hits, u32 hit_cap, u32* __restrict__ counters, u32* __restrict__ seed_cnt,
                                                            u32 refill, u32 prio, u32 windows) {
    extern __shared__ u32 lds[];                // frames: [level][TX_FRAME_WORDS][64 lanes]; then the text windows and the seed windows, [word][lane]
    constexpr u32 TX_WIN_T = TxWin<TX_WIN_MAXLEN>::T, TX_WIN_Q = TxWin<TX_WIN_MAXLEN>::Q;
    if (prio == 3u) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2u) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1u) __builtin_amdgcn_s_setprio(1);
    u32 const lane = s_lane_id();
    u64 const lanes_below = (1ull << lane) - 1ull;
    auto fr = [&](u32 level, u32 word) -> u32& { return lds[(level * TX_FRAME_WORDS + word) * 64u + lane]; };
    u32* const win_t = lds + C.levels * TX_FRAME_WORDS * 64u + lane;
    u32* const win_q = win_t + TX_WIN_T * 64u;
    u32 const n_slots = min(counters[16], item_cap);      // (the filter kernel has finished: same stream)
    WaveQueue Q;
    WaveSlots HS;
    TxLane L;
    TxWinAccess ac{win_t, win_q, 0, 0u, false, C.idx.text, C.seq, TX_WIN_T, TX_WIN_Q};
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
                    if (item.seed != 0xFFFFFFFFu) {                                                  // (an unused slot: ask again)
                        tx_take_item(C, L, item, seeds[item.seed]);
                        ac.seq = C.seq + L.qoff;
                        ac.win = windows && L.len <= TX_WIN_MAXLEN && L.nx >= 1u;
                        if (ac.win) {
                            // the leftmost seed position the string covers (the scheme entry knows it): that many symbols are still to come on
                            // the left, the rest of the seed on the right, plus a text symbol per error (at most 3) and the one read that fails
                            u32 const a = sch_lo(L.ex[L.nx]);
                            ac.w0 = ((i64)L.pL - (i64)a - 4) & ~(i64)3;
                            const u32* __restrict__ tsrc = reinterpret_cast<const u32*>(C.idx.text + ac.w0);
#pragma unroll
                            for (u32 j = 0; j < TX_WIN_T; ++j) win_t[j * 64u] = nibbles_of(tsrc[2u * j], tsrc[2u * j + 1u]);
                            ac.qshift = (u32)(L.qoff & 3ull);
                            const u32* __restrict__ qsrc = reinterpret_cast<const u32*>(C.seq + (L.qoff & ~3ull));
                            u32 const q_words = (L.len + ac.qshift + 7u) >> 3;
#pragma unroll
                            for (u32 j = 0; j < TX_WIN_Q; ++j) if (j < q_words) win_q[j * 64u] = nibbles_of(qsrc[2u * j], qsrc[2u * j + 1u]);
                        }
                    }
                } else exhausted = true;
            }
        }
        if (__all(exhausted && !L.busy)) break;
        ++n_iter;
        if (!L.busy) continue;
        tx_step(C, L, fr, ac);
    }
    wave_slots_close(HS, hits, hit_cap, lane);
    if (__any(L.overflow) && lane == 0) atomicOr(&counters[1], 1u);
    u32 const s_nodes = s_wave_sum(L.n_nodes);
    if (lane == 0) { atomicAdd(&counters[18], s_nodes); atomicAdd(&counters[19], n_iter); atomicAdd(&counters[13], n_hits); atomicMax(&counters[21], n_iter); atomicAdd(&counters[22], n_iter ? 1u : 0u); }
}

static u32 env_u32(const char* name, u32 dflt) {
    const char* e = getenv(name);
    return e ? (u32)strtoul(e, nullptr, 10) : dflt;
}

// The search with the stack in LDS, the presence filter and the text walk. d_items: item_cap records for queued subtrees (0: none are
// queued). frame_levels = largest error count of a seed. d_counters: 32 words, zeroed by the caller.
int DeviceApi::search_filtered(void* stream, const DevIndex& idx, const u8* d_seq, const u32* d_qpack, const u64* d_scheme, const DevSeed* d_seeds,
                               u32 n_seeds, u32 max_hits_per_seed, u32 frame_levels, DevHit* d_hits, u32 hit_cap, DevHit* d_items, u32 item_cap,
                               u32* d_counters, u32* d_seed_cnt, u32* d_seed_rows, void* d_mailboxes, u32 mailbox_waves, u32 concurrent_launches, bool long_seeds) {
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
    u32 const cap_look = env_u32("FLX_FM_CAP_LOOK", 64) - 1u;         // a lane in a long tail looks at its seed's row count every so many iterations (a power of two)
    u32 const steal_after = env_u32("FLX_FM_STEAL_AFTER", 64);         // iterations a wave's queue has been dry before its lanes share work
    static u32 const stats = env_u32("FLX_SEARCH_DEBUG", 0);           // the diagnostic counters [11], [12] cost two registers per lane
    // the mailboxes of this launch's waves: control words, states, counts and the hungry list zeroed, then the subtree slots
    Mailboxes MB{};
    if (seed_rows && d_mailboxes && mailbox_waves >= grid.x && env_u32("FLX_FM_NO_MAILBOXES", 0) == 0) {
        u32* const words = (u32*)d_mailboxes;
        size_t const zeroed = ((size_t)MB_CTRL_WORDS + (size_t)grid.x * (2 + MB_HUNGRY_PER_WAVE)) * 4;
        hipError_t const e = hipMemsetAsync(words, 0, zeroed, s);
        if (e != hipSuccess) return (int)e;
        MB.ctrl = words; MB.state = words + MB_CTRL_WORDS; MB.count = MB.state + grid.x; MB.hungry = MB.count + grid.x;
        MB.n_hungry = grid.x * MB_HUNGRY_PER_WAVE;
        MB.items = reinterpret_cast<FrameItem*>((char*)d_mailboxes + (zeroed + 255) / 256 * 256);
    }
    auto const kernel = stats ? fm_search_filter_kernel<true> : fm_search_filter_kernel<false>;
    hipLaunchKernelGGL(kernel, grid, dim3(64), lds_bytes, s, C, n_seeds, d_hits, hit_cap, d_items, item_cap, d_counters, d_seed_cnt, seed_rows, MB, refill_a, fm_prio, std::max(1u, steal_min), steal_after, cap_look);
    if (C.text_min_remain) {
        // (the number of queued subtrees is only known on the device: a fixed grid, waves without work leave at once; the walk is a
        // chain of dependent loads from the L2, so it wants every wave slot: 8 per SIMD)
        static u32 const text_waves = env_u32("FLX_FM_TEXT_WAVES", 8192);
        u32 const tw = std::max(1u, text_waves / std::max(1u, std::min(concurrent_launches, 8u)));
        // (the windows need 4-byte aligned bases: the text is, a sequence pool handed in at an odd address is read byte by byte as before)
        u32 const windows = env_u32("FLX_FM_NO_WINDOWS", 0) == 0 && ((uintptr_t)d_seq & 3u) == 0 && ((uintptr_t)idx.text & 3u) == 0;
        dim3 const tgrid(std::min<u32>(std::max<u32>(grid.x * 4u, 64u), tw));
        if (long_seeds)
            hipLaunchKernelGGL(fm_search_text_kernel<160>, tgrid, dim3(64), ((size_t)C.levels * TX_FRAME_WORDS + TxWin<160>::T + TxWin<160>::Q) * 64 * sizeof(u32), s, C,
                               d_seeds, d_items, item_cap, d_hits, hit_cap, d_counters, d_seed_cnt, refill_t, fm_prio, windows);
        else
            hipLaunchKernelGGL(fm_search_text_kernel<64>, tgrid, dim3(64), ((size_t)C.levels * TX_FRAME_WORDS + TxWin<64>::T + TxWin<64>::Q) * 64 * sizeof(u32), s, C,
                               d_seeds, d_items, item_cap, d_hits, hit_cap, d_counters, d_seed_cnt, refill_t, fm_prio, windows);
    }
    return (int)hipGetLastError();
}

}  // namespace flx
