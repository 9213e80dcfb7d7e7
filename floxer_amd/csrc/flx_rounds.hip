// Verification rounds on the device, three launches per round (verification.cpp:44-117 for all anchors of a chunk at once).
//
// The anchors of a chunk climb their PEX trees together; a round tests the anchors whose current node is in the smallest node-size
// class still pending. Identical (window, node) tests run once and the windows of one locus form a cluster that is tested on the
// intersection and the union of its windows (an alignment inside the intersection is one inside every member's window, none inside
// the union is none inside any; only when the union holds one and the intersection does not are the members tested one by one, each
// alone in the next round). All of that only saves existence tests: which anchors pass does not depend on how they were grouped.
//
//   vr2_request_kernel   one block per query (read x orientation; its anchors are contiguous): the requests of its anchors in the
//                        round's class, sorted by (node, window start) in LDS, distinct windows and clusters marked by walks over
//                        the sorted run, one or two jobs per cluster appended to the round's job list, every anchor told the
//                        slots of its cluster's jobs
//   ed_exists_block      (flx_device.hip) the round's jobs in one common launch shape chosen by the host from the node tables; the
//                        job count stays on the device
//   vr2_apply_kernel     per anchor: its cluster's results -> up to the parent, dead, or alone next round; what is left to climb
//
// Round 2 of this build had the same steps as ~45 launches per round (one global radix sort of all anchors, scans, scatters, a
// plan the host read back to launch one existence kernel per shape class): on a GPU full of long-running waves each small launch
// waits ~200 us for a slot, 9 rounds of them were a quarter of a chunk's wall time (profiles/r03_pipeline_before_rounds.txt).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "flx_internal.hpp"

namespace flx {

namespace {

// anchors of a query looked at per pass of its block. 256: 4.75 KB of LDS per block, so that the block finds room on a CU whose LDS the DP
// kernels of other lanes have taken (at 1024, 19 KB, a launch waited 3 ms for its blocks to be placed); a query with more anchors is
// handled tile by tile (the anchors of a node's subtree are contiguous, few nodes are cut; windows cut apart are simply tested twice)
constexpr u32 VR2_TILE = 256;
constexpr u32 VR2_THREADS = 64;
constexpr u64 VR2_NO_KEY = ~0ull;

__device__ __forceinline__ u64 vr2_word_steps(u32 n, u32 m, u32 k, u32 W) {          // job_word_steps of flx_pipeline.cpp, banded
    u64 const nw = (m + 63u) / 64u;
    i64 const band_hi = (i64)n - (i64)m + (i64)k;
    u64 total = 0;
    for (i64 g = 0; g * W < (i64)nw; ++g) {
        i64 const r0 = 64 * (i64)W * g, r1 = min((i64)m, r0 + 64 * (i64)W);
        i64 const lo = max((i64)0, r0 - (i64)k), hi = min((i64)n - 1, r1 - 1 + band_hi);
        if (hi >= lo) total += (u64)(hi - lo + 1) * (u64)min((i64)W, (i64)nw - g * W);
    }
    return total;
}

// sort key of a request: node index | window start in the text | alone. An anchor that is to be tested alone (VR_SOLO) never shares a
// cluster: its key differs from the same window asked for by others and the walks below cut clusters around it.
__device__ __forceinline__ u64 vr2_key(u32 node, u64 start, u32 solo) { return ((u64)node << 33) | (start << 1) | solo; }

}  // namespace

// One wave per query: a block of 64 threads finds a free wave slot on a GPU full of other lanes' kernels far sooner than four waves
// and their LDS at once, and the steps below are short (a query has a few hundred anchors).
__global__ void __launch_bounds__(VR2_THREADS) vr2_request_kernel(Vr2Buffers B, u32 limit, u32 acct_words, u32 width_cap, u32 parity, u32 check_pending) {
    __shared__ u64 s_key[VR2_TILE];
    __shared__ u32 s_len[VR2_TILE];           // window length of the tile's anchor (by its place in the tile)
    __shared__ u16 s_val[VR2_TILE];
    __shared__ u8 s_flag[VR2_TILE];           // bit 0: first of its distinct window, bit 1: first of its cluster
    __shared__ u32 s_jobs[VR2_TILE];          // jobs of the cluster that starts here (0 elsewhere); then their exclusive sums
    __shared__ u32 s_count, s_base;
    u32 const tid = threadIdx.x;
    u32 const q = blockIdx.x;
    if (q == 0 && tid == 0) {
        B.scalars[VR2_N_CLIMBING] = 0u; B.scalars[VR2_SMALLEST] = 0xFFFFFFFFu; B.scalars[VR2_DONE] = 0u;      // vr2_apply counts into them
        B.scalars[VR2_N_JOBS + (parity ^ 1u)] = 0u;                                                         // the next round's job counter
        B.scalars[VR2_QUEUE] = 0u;                                                                          // this round's existence kernel starts at job 0
        B.scalars[VR2_PENDING + (parity ^ 1u)] = 0u;                                                        // vr2_apply counts the next round's anchors
    }
    // (a round of a queued series for which the round before counted no anchor: written by that round's vr2_apply, stable during this kernel)
    if (check_pending && B.scalars[VR2_PENDING + parity] == 0u) return;
    u32 const a0 = B.q_first[q], a1 = B.q_first[q + 1];
    u32* const n_jobs = &B.scalars[VR2_N_JOBS + parity];
    for (u32 tile = a0; tile < a1; tile += VR2_TILE) {
        u32 const n_here = min(VR2_TILE, a1 - tile);
        u32 size = 64;
        while (size < n_here) size <<= 1;
        __syncthreads();
        if (tid == 0) s_count = 0;
        __syncthreads();
        // ---- the requests of this tile's anchors that are in the round
        u32 mine = 0;
        u32 tree_base = 0;
        for (u32 t = tid; t < size; t += VR2_THREADS) {
            u64 key = VR2_NO_KEY;
            if (t < n_here) {
                u32 const i = tile + t;
                u8 const st = B.status[i];
                if (st == VR_CLIMBING || st == VR_SOLO) {
                    DevVrAnchor const a = B.anchors[i];
                    u32 const nd_i = B.node[i];
                    DevVrNode const nd = B.nodes[a.tree_base + nd_i];
                    if (nd.rows <= limit) {
                        i64 const start_signed = a.diag_rel + (i64)nd.from - (i64)nd.errors;
                        u64 const start = start_signed > 0 ? (u64)start_signed : 0ull;
                        u64 const base = (u64)nd.rows + 2ull * nd.errors + 1ull;
                        key = vr2_key(nd_i, a.seq_start + start, st == VR_SOLO ? 1u : 0u);
                        s_len[t] = (u32)min(base, a.seq_len - start);
                        ++mine;
                    }
                }
            }
            s_key[t] = key;
            s_val[t] = (u16)t;
        }
        if (mine) atomicAdd(&s_count, mine);
        __syncthreads();
        u32 const cnt = s_count;
        if (cnt == 0) continue;
        DevVrAnchor const a_first = B.anchors[tile];                  // (the tree and the read are the query's)
        tree_base = a_first.tree_base;
        // ---- bitonic sort of (key, anchor) over `size` slots; slots without a request sort to the end
        for (u32 k = 2; k <= size; k <<= 1)
            for (u32 j = k >> 1; j > 0; j >>= 1) {
                for (u32 t = tid; t < size; t += VR2_THREADS) {
                    u32 const p = t ^ j;
                    if (p > t) {
                        u64 const x = s_key[t], y = s_key[p];
                        bool const up = (t & k) == 0;
                        if ((x > y) == up) { s_key[t] = y; s_key[p] = x; u16 const v = s_val[t]; s_val[t] = s_val[p]; s_val[p] = v; }
                    }
                }
                __syncthreads();
            }
        // ---- one thread per node run: distinct windows and clusters. A cluster = the distinct windows of a node whose starts fall
        //      into the same bucket of max(8, rows / 8) columns counted from the node's first window - as long as their union stays
        //      within the diagonals the round's launch shape holds (width_cap; the host chose the shape for single windows)
        for (u32 t = tid; t < cnt; t += VR2_THREADS) {
            u64 const key = s_key[t];
            u32 const node = (u32)(key >> 33);
            if (t > 0 && (u32)(s_key[t - 1] >> 33) == node) continue;          // not the head of a run
            DevVrNode const nd = B.nodes[tree_base + node];
            u64 const d = max((u64)8, (u64)nd.rows / 8ull);
            u64 const first = (key >> 1) & 0xFFFFFFFFull;
            u64 prev_key = VR2_NO_KEY, prev_bucket = ~0ull, c_lo = 0, c_hi = 0;
            bool prev_solo = false;
            for (u32 j = t; j < cnt && (u32)(s_key[j] >> 33) == node; ++j) {
                u64 const kj = s_key[j];
                bool const solo = kj & 1ull;
                u64 const st = (kj >> 1) & 0xFFFFFFFFull;
                u64 const bucket = (st - first) / d;
                u8 f = 0;
                if (kj != prev_key) {
                    u64 const en = st + s_len[s_val[j]];
                    f = 1;
                    u64 const u_hi = max(c_hi, en);
                    bool const too_wide = (u_hi - c_lo) + 2ull * nd.errors > (u64)nd.rows + width_cap;      // union width = n - m + 2k
                    if (j == t || solo || prev_solo || bucket != prev_bucket || too_wide) { f = 3; c_lo = st; c_hi = en; }
                    else c_hi = u_hi;
                }
                s_flag[j] = f;
                prev_key = kj; prev_bucket = bucket; prev_solo = solo;
            }
        }
        __syncthreads();
        // ---- one thread per cluster: one job, or two when it has several windows that share columns (intersection and union)
        for (u32 t = tid; t < size; t += VR2_THREADS) {
            u32 nj = 0;
            if (t < cnt && (s_flag[t] & 2u)) {
                u32 distinct = 0;
                u64 hi_start = 0, lo_end = 0;
                for (u32 j = t; j < cnt && (j == t || !(s_flag[j] & 2u)); ++j) {
                    if (!(s_flag[j] & 1u)) continue;
                    u64 const st = (s_key[j] >> 1) & 0xFFFFFFFFull, en = st + s_len[s_val[j]];
                    if (distinct == 0) { hi_start = st; lo_end = en; } else { hi_start = max(hi_start, st); lo_end = min(lo_end, en); }
                    ++distinct;
                }
                nj = distinct > 1 && lo_end > hi_start ? 2u : 1u;
            }
            s_jobs[t] = nj;
        }
        __syncthreads();
        // ---- exclusive sums of the clusters' job counts (a thread's consecutive slots, then the wave by shuffles), one slot range of
        //      the round's job list for the query
        {
            u32 const per = size / VR2_THREADS;                                   // size >= 64: 1 .. 16 slots per thread
            u32 const first = tid * per;
            u32 run = 0;
            for (u32 x = 0; x < per; ++x) run += s_jobs[first + x];
            u32 incl = run;
#pragma unroll
            for (u32 off = 1; off < 64u; off <<= 1) { u32 const up = (u32)__shfl_up((int)incl, off); if (tid >= off) incl += up; }
            u32 at = incl - run;
            for (u32 x = 0; x < per; ++x) { u32 const v = s_jobs[first + x]; s_jobs[first + x] = at; at += v; }
            if (tid == VR2_THREADS - 1u) s_base = incl ? atomicAdd(n_jobs, incl) : 0u;
            __syncthreads();
        }
        u32 const base_slot = s_base;
        // ---- the jobs, and every anchor's way to them: slot << 2 | has the union job << 1 | has the own-window / intersection job
        u64 steps = 0, bytes = 0;
        u32 n_req = 0;
        for (u32 t = tid; t < cnt; t += VR2_THREADS) {
            if (!(s_flag[t] & 2u)) continue;
            u32 const slot = base_slot + s_jobs[t];
            u32 distinct = 0;
            u64 lo_start = 0, hi_start = 0, lo_end = 0, hi_end = 0;
            u32 last = t;
            for (u32 j = t; j < cnt && (j == t || !(s_flag[j] & 2u)); ++j) {
                last = j;
                if (!(s_flag[j] & 1u)) continue;
                u64 const st = (s_key[j] >> 1) & 0xFFFFFFFFull, en = st + s_len[s_val[j]];
                if (distinct == 0) { lo_start = hi_start = st; lo_end = hi_end = en; }
                else { hi_start = max(hi_start, st); lo_start = min(lo_start, st); lo_end = min(lo_end, en); hi_end = max(hi_end, en); }
                ++distinct;
            }
            DevVrNode const nd = B.nodes[tree_base + (u32)(s_key[t] >> 33)];
            u64 const q_off = a_first.q_base + nd.from;
            u32 code;
            if (distinct == 1) {
                B.jobs[slot] = DevAlignJob{lo_start, q_off, 0, (u32)(hi_end - lo_start), nd.rows, nd.errors, slot, 0};
                code = (slot << 2) | 1u;
                steps += vr2_word_steps((u32)(hi_end - lo_start), nd.rows, nd.errors, acct_words); bytes += (hi_end - lo_start) + nd.rows;
            } else if (lo_end > hi_start) {
                B.jobs[slot] = DevAlignJob{hi_start, q_off, 0, (u32)(lo_end - hi_start), nd.rows, nd.errors, slot, 0};
                B.jobs[slot + 1u] = DevAlignJob{lo_start, q_off, 0, (u32)(hi_end - lo_start), nd.rows, nd.errors, slot + 1u, 0};
                code = (slot << 2) | 3u;
                steps += vr2_word_steps((u32)(lo_end - hi_start), nd.rows, nd.errors, acct_words) + vr2_word_steps((u32)(hi_end - lo_start), nd.rows, nd.errors, acct_words);
                bytes += (lo_end - hi_start) + (hi_end - lo_start) + 2ull * nd.rows;
            } else {
                B.jobs[slot] = DevAlignJob{lo_start, q_off, 0, (u32)(hi_end - lo_start), nd.rows, nd.errors, slot, 0};
                code = (slot << 2) | 2u;
                steps += vr2_word_steps((u32)(hi_end - lo_start), nd.rows, nd.errors, acct_words); bytes += (hi_end - lo_start) + nd.rows;
            }
            for (u32 j = t; j <= last; ++j) B.a_slot[tile + s_val[j]] = code;
            n_req += last - t + 1u;
        }
        // accounting (one atomic per wave)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { steps += __shfl_xor(steps, off); bytes += __shfl_xor(bytes, off); n_req += (u32)__shfl_xor((int)n_req, off); }
        if (tid == 0u && n_req) {
            atomicAdd(&B.scalars[VR2_N_REQ], n_req);
            atomicAdd((unsigned long long*)&B.scalars[VR2_WORD_STEPS], (unsigned long long)steps);
            atomicAdd((unsigned long long*)&B.scalars[VR2_BYTES], (unsigned long long)bytes);
        }
    }
}

// host_scalars (page-locked, mapped): where the last block to finish leaves the round's scalars, so that the host reads them behind
// an event without a copy of its own
__global__ void __launch_bounds__(256) vr2_apply_kernel(Vr2Buffers B, u32 n, u32* __restrict__ host_scalars, u32 next_limit, u32 parity) {
    __shared__ u32 s_cnt[4], s_min[4], s_pend[4];
    u32 count = 0, smallest = 0xFFFFFFFFu, pending = 0;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        u8 st = B.status[i];
        u32 const code = B.a_slot[i];
        if (code != 0xFFFFFFFFu) {
            B.a_slot[i] = 0xFFFFFFFFu;
            u32 const slot = code >> 2;
            bool const has_a = code & 1u, has_b = code & 2u;
            // own window / intersection holds an alignment -> pass; the union (or the one window there is) holds none -> dead; else the
            // members one by one: alone in the next round
            u8 dec;
            if (has_a && B.outs[slot].score != 0xFFFFFFFFu) dec = 1;
            else if (!has_b) dec = 2;
            else if (B.outs[slot + (has_a ? 1u : 0u)].score == 0xFFFFFFFFu) dec = 2;
            else dec = 0;
            if (dec == 1) {
                u32 const tb = B.anchors[i].tree_base;
                u32 const parent = B.nodes[tb + B.node[i]].parent;
                B.node[i] = parent;
                st = B.nodes[tb + parent].parent == 0xFFFFFFFFu ? VR_AT_ROOT : VR_CLIMBING;
            } else st = dec == 2 ? VR_DEAD : VR_SOLO;
            B.status[i] = st;
        }
        if (st == VR_CLIMBING || st == VR_SOLO) {
            ++count;
            u32 const rows = B.nodes[B.anchors[i].tree_base + B.node[i]].rows;
            smallest = min(smallest, rows);
            pending += rows <= next_limit ? 1u : 0u;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        count += (u32)__shfl_xor((int)count, off);
        pending += (u32)__shfl_xor((int)pending, off);
        smallest = min(smallest, (u32)__shfl_xor((int)smallest, off));
    }
    if ((threadIdx.x & 63u) == 0u) { s_cnt[threadIdx.x >> 6] = count; s_min[threadIdx.x >> 6] = smallest; s_pend[threadIdx.x >> 6] = pending; }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 const c = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3], m = min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3]));
        u32 const pd = s_pend[0] + s_pend[1] + s_pend[2] + s_pend[3];
        if (c) { atomicAdd(&B.scalars[VR2_N_CLIMBING], c); atomicMin(&B.scalars[VR2_SMALLEST], m); }
        if (pd) atomicAdd(&B.scalars[VR2_PENDING + (parity ^ 1u)], pd);
        __threadfence();
        if (atomicAdd(&B.scalars[VR2_DONE], 1u) == gridDim.x - 1u && host_scalars) {
            __threadfence();
            for (u32 w = 0; w < VR2_SCALARS; ++w) host_scalars[w] = __hip_atomic_load(&B.scalars[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
        }
    }
}

// everything up to the round's job list (B.jobs, B.scalars[VR2_N_JOBS + parity]; the other parity's counter is zeroed for the next round); acct_words: words per lane of the launch shape (accounting);
// width_cap: most diagonals (n - m + 2k) a job of the round may have (what the launch shapes hold)
int DeviceApi::vr2_request(void* stream, Vr2Buffers const& B, u32 n_queries, u32 limit, u32 acct_words, u32 width_cap, u32 parity, bool check_pending) {
    if (n_queries == 0) return 0;
    hipLaunchKernelGGL(vr2_request_kernel, dim3(n_queries), dim3(VR2_THREADS), 0, (hipStream_t)stream, B, limit, std::max(1u, acct_words), width_cap, parity & 1u,
                       check_pending ? 1u : 0u);
    return (int)hipGetLastError();
}
int DeviceApi::vr2_apply(void* stream, Vr2Buffers const& B, u32 n_anchors, u32* host_scalars, u32 next_limit, u32 parity) {
    if (n_anchors == 0) return 0;
    hipLaunchKernelGGL(vr2_apply_kernel, dim3(std::min<u32>((n_anchors + 255) / 256, 1024u)), dim3(256), 0, (hipStream_t)stream, B, n_anchors, host_scalars, next_limit,
                       parity & 1u);
    return (int)hipGetLastError();
}

}  // namespace flx
