// floxer_amd — K3 in lane-per-job form: existence tests with Ukkonen's cutoff (alignment.cpp:147-164 decides "is there an alignment
// of the whole query inside this window with at most k errors"; verification.cpp:64-117 asks it for every inner node of the PEX tree).
//
// The ring form (flx_device.hip, ed_exists_block_kernel) gives a job R lanes that walk its 64-row word groups in lockstep, skewed, over
// the static band -k <= col - row <= (n - m) + k. Most of that band is dead: a cell can only lie on an alignment within k if its own
// value is <= k, and below the first ~2k rows only the cells around a true occurrence are (the stripe narrows from 2k + 1 diagonals to
// nothing as the errors of the occurrence use the budget up). Lanes in lockstep cannot skip what one of them still needs, so here a lane
// owns a whole job and walks its word groups one after the other, each only over the blocks (16 columns) that can hold a value <= k:
//
//   * group g hands its bottom row down through a per-lane buffer in LDS: per block the 16 pairs of horizontal-delta bits; it remembers
//     the first and the last block (fl, ll) in which that row had a value <= k and the row's value in front of the block the next group
//     will start at;
//   * group g + 1 starts at block max(its static first block, fl) from the column "all +1 below the row above" (every cell to the left
//     of it in these rows is > k: paths enter the rows through the row above, and never move left), reads the deltas of the row above
//     while that row was computed and "+1 per column" after it (an over-estimate: the row above was > k from there on), and ends at the
//     first block b > ll whose left column holds no value <= k (by the bound (top + bottom - rows) / 2 on a column with steps of at most
//     1): everything in these rows from there on is > k. No group above a dead one is ever needed again: the job ends as "no alignment".
//
// Values > k are over-estimates, values <= k are exact (every cell on a path to a cell <= k is itself <= k and computed), so score and
// end column (the rightmost minimum of the last row, alignment.cpp) are those of the full matrix whenever the score is within k.
// Lanes take jobs from a queue (a job's length is data dependent), one block of one group per lane and iteration.
#include <hip/hip_runtime.h>

#include "flx_internal.hpp"

namespace flx {

namespace {

constexpr int LB_NONE = 0x7FFFFFFF;
enum : u32 { PH_NEED_JOB = 0, PH_GROUP_START = 1, PH_BLOCK = 2, PH_DONE = 3 };

__device__ __forceinline__ u32 lane_index() { return threadIdx.x & 63u; }

}  // namespace

// LDS: [7 symbols][64 lanes] equality masks of the lanes' current word groups (symbol 6 = past the window), then per lane `cap` words: the
// carries (16 x {hp, hn}) of a bottom row, block b at b mod cap. A group reads the word of the row above at b and then puts its own there:
// the blocks of a row that are ever read are those it was computed for, fewer than cap.
// counters (optional, 8 x u64): [0] blocks computed, [1] wave iterations, [2] lane-iterations in the block phase, [3] groups entered
__global__ void __launch_bounds__(64) ed_exists_lane_kernel(const u8* __restrict__ text, const u64* __restrict__ peq,
                                                            const DevAlignJob* __restrict__ jobs, u32 n_jobs, const u32* __restrict__ n_jobs_dev,
                                                            u32* __restrict__ queue, u32 cap, DevAlignOut* __restrict__ out,
                                                            unsigned long long* __restrict__ stats, u32 prio) {
    extern __shared__ __attribute__((aligned(16))) u64 lds_eq[];
    u32 const lane = lane_index();
    u32* const row = reinterpret_cast<u32*>(lds_eq + 7 * 64) + lane * cap;
    // (prio: a wave's chain of blocks is long and nothing hides it; with priority its instructions go first on a SIMD it shares with other
    // kernels' waves)
    if (prio == 3u) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2u) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1u) __builtin_amdgcn_s_setprio(1);
    if (n_jobs_dev) n_jobs = min(n_jobs, *n_jobs_dev);

    u32 phase = PH_NEED_JOB;
    // the job
    const u8* ref = text;
    u64 q_off = 0;
    int n = 0, m = 1, k = 0, Lg = 1, pad = 0, band_hi = 0;
    u32 out_index = 0;
    // the group
    int g = 0, b = 0, b_hi = -1, rows_g = 64, b_lo_next = 0;
    u32 slot = 0;                                           // b mod cap
    u64 vp = ~0ull, vn = 0ull;
    int bot = 0, top_cur = 0, best = 0, best_col = 0;
    // the group above: computed up to block le (exclusive), first / last block with a value <= k in its bottom row, that row's value in front of the
    // block the next group starts at (if computed) and in its last computed column (16 * le - 1)
    int p_le = 0, p_fl = LB_NONE, p_ll = -1, p_anchor = 0, p_bot_end = 0;
    int c_ls = 0, c_fl = LB_NONE, c_ll = -1, c_anchor = 0;
    uint4 tq0 = make_uint4(0, 0, 0, 0), tq1 = tq0;          // reference symbols of blocks b and b + 1
    u32 cw_next = 0x55555555u;                              // the row above at block b (read one block ahead)
    u64 eq_lo[6] = {0, 0, 0, 0, 0, 0}, eq_hi[6] = {0, 0, 0, 0, 0, 0};      // Peq words of the next group to start (loaded while the one before it runs)
    bool overflow = false;
    u64 n_blocks = 0, n_iter = 0, n_lane_iter = 0, n_groups = 0;

    // masks of group gg of the lane's job: the twelve loads now, the words put together when the group starts (padding rows match everything)
    i64 eq_off = 0;                                         // pool position of the word's bit 0 (negative: the pool starts inside the word)
    auto issue_eq = [&](int gg) {
        eq_off = (i64)q_off + 64 * gg - pad;
        u64 const a = eq_off >= 0 ? (u64)eq_off >> 6 : 0ull;
#pragma unroll
        for (u32 sy = 0; sy < 6; ++sy) {
            eq_lo[sy] = peq[a * 6 + sy];
            eq_hi[sy] = peq[(a + 1) * 6 + sy];
        }
    };
    auto finish_eq = [&](u32 sy, u64 padmask) -> u64 {
        u32 const sh = (u32)eq_off & 63u;
        u64 const joined = sh ? (eq_lo[sy] >> sh) | (eq_hi[sy] << (64u - sh)) : eq_lo[sy];
        u64 const v = eq_off >= 0 ? joined : eq_lo[sy] << (u32)(-eq_off);      // (the bits in front of the pool are padding rows)
        return v | padmask;
    };

    for (;;) {
        // ---- lanes without a job take the next ones of the queue
        u64 const m_need = __ballot(phase == PH_NEED_JOB);
        if (m_need) {
            u32 const leader = (u32)__builtin_ctzll(m_need);
            u32 base = 0;
            if (lane == leader) base = atomicAdd(queue, (u32)__popcll(m_need));
            base = (u32)__shfl((int)base, (int)leader);
            if (phase == PH_NEED_JOB) {
                u32 const id = base + (u32)__popcll(m_need & ((1ull << lane) - 1ull));
                if (id >= n_jobs) phase = PH_DONE;
                else {
                    DevAlignJob const job = jobs[id];
                    n = (int)job.n; m = (int)job.m; k = (int)job.k;
                    out_index = job.out_index;
                    if (n == 0 || n + k < m) {
                        // no column at all: all m rows are insertions; fewer columns than m - k: no alignment within k
                        DevAlignOut o;
                        o.score = (n == 0 && m <= k) ? (u32)m : 0xFFFFFFFFu;
                        o.end_col = 0u;
                        out[out_index] = o;
                    } else {
                        ref = text + job.ref_off;
                        q_off = job.q_off;
                        Lg = (m + 63) >> 6;
                        pad = Lg * 64 - m;           // the query right-aligned in its groups (ed_block_body): only group 0 holds padding
                        band_hi = n - m + k;
                        g = 0;
                        best = m;
                        best_col = 0;
                        overflow = false;
                        p_fl = LB_NONE; p_ll = -1; p_le = 0;
                        issue_eq(0);
                        phase = PH_GROUP_START;
                    }
                }
            }
        }
        if (!__any(phase != PH_DONE)) break;

        // ---- lanes at the start of a group: its window, where it starts, its equality masks
        if (phase == PH_GROUP_START) {
            int const r0 = max(0, 64 * g - pad);
            int const r1 = 64 * (g + 1) - pad;
            rows_g = r1 - r0;
            int const b_lo = max(0, r0 - k) >> 4;
            b_hi = min(n - 1, r1 - 1 + band_hi) >> 4;
            b_lo_next = max(0, r1 - k) >> 4;
            bool go = true;
            if (g == 0) { b = 0; top_cur = 0; }
            else if (p_fl == LB_NONE || p_ll < b_lo - 1) go = false;        // nothing <= k reaches these rows: no alignment
            else {
                // (p_ll == b_lo - 1: the value <= k may sit in the column just left of block b_lo, whose diagonal neighbour is in these rows)
                b = max(b_lo, p_fl);
                if (b > b_hi) go = false;
            }
            if (!go) {
                DevAlignOut o;
                o.score = 0xFFFFFFFFu;
                o.end_col = 0u;
                out[out_index] = o;
                phase = PH_NEED_JOB;
            } else {
                u64 const padmask = g == 0 && pad ? (1ull << (u32)pad) - 1ull : 0ull;
#pragma unroll
                for (u32 sy = 0; sy < 6; ++sy) lds_eq[sy * 64u + lane] = finish_eq(sy, padmask);
                lds_eq[6u * 64u + lane] = padmask;
                vp = ~padmask;
                vn = 0ull;
                __builtin_memcpy(&tq0, ref + 16 * (i64)b, 16);
                __builtin_memcpy(&tq1, ref + 16 * (i64)min(b + 1, b_hi), 16);
                if (g + 1 < Lg) issue_eq(g + 1);
                slot = (u32)b % cap;
                if (g > 0) {
                    // the row above in front of block b: computed there (p_fl <= b, so that is the block its anchor was taken at), or
                    // behind its last block, where it goes on with +1 per column
                    top_cur = b < p_le ? p_anchor : p_bot_end + 16 * (b - p_le);
                    cw_next = b < p_le ? row[slot] : 0x55555555u;
                }
                bot = top_cur + rows_g;                                    // the column left of the first block: all vertical deltas +1
                c_ls = b;
                c_fl = LB_NONE;
                c_ll = -1;
                phase = PH_BLOCK;
                ++n_groups;
            }
        }

        // ---- one block of the lane's group
        ++n_iter;
        if (phase == PH_BLOCK) {
            ++n_lane_iter;
            // the group ends behind its static window, or (below group 0) once the row above has had its last value <= k and the column
            // left of this block holds none either
            bool const ended = b > b_hi || (g > 0 && b > p_ll && top_cur + bot - rows_g > 2 * k);
            if (!ended) {
                ++n_blocks;
                u32 const cw_in = g == 0 ? 0u : cw_next;
                u32 const quad[4] = {tq0.x, tq0.y, tq0.z, tq0.w};
                tq0 = tq1;
                __builtin_memcpy(&tq1, ref + 16 * (i64)min(b + 2, b_hi), 16);      // two blocks ahead
                int const bot_start = bot;
                bool const last = g == Lg - 1;
                int minb = LB_NONE;
                u32 cw = 0;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int const j = 4 * qd + i;
                        int const c = 16 * b + j;
                        u32 sym = (quad[qd] >> (8 * i)) & 0xFFu;
                        sym = c < n ? sym : 6u;
                        u64 const c_hp = (cw_in >> (2 * j)) & 1u, c_hn = (cw_in >> (2 * j + 1)) & 1u;
                        u64 const eq = lds_eq[sym * 64u + lane];
                        u64 const x = eq | vn;
                        u64 const tt = vp + (x & vp) + c_hn;
                        u64 const d0 = (tt ^ vp) | x;
                        u64 const hn = vp & d0;
                        u64 const hp = vn | ~(vp | d0);
                        u64 const xh = (hp << 1) | c_hp;
                        vn = xh & d0;
                        vp = (hn << 1) | ~(xh | d0) | c_hn;
                        u32 const o_hp = (u32)(hp >> 63), o_hn = (u32)(hn >> 63);
                        cw |= (o_hp | (o_hn << 1)) << (2 * j);
                        bot += (int)o_hp - (int)o_hn;                      // the group's last row is its word's bit 63
                        minb = min(minb, bot);
                        if (last && c < n && bot <= best) { best = bot; best_col = c + 1; }
                    }
                }
                if (b - c_ls >= (int)cap) overflow = true;                 // (the launch's rows do not hold this window: reported at the end)
                row[slot] = cw;
                top_cur += __popc(cw_in & 0x55555555u) - __popc(cw_in & 0xAAAAAAAAu);
                if (minb <= k) { if (c_fl == LB_NONE) c_fl = b; c_ll = b; }
                if (b == max(b_lo_next, c_fl)) c_anchor = bot_start;       // where the next group starts (c_fl unknown yet: no block matches)
                ++b;
                slot = slot + 1u == cap ? 0u : slot + 1u;
                cw_next = b < p_le ? row[slot] : 0x55555555u;
            } else if (g == Lg - 1 || overflow) {
                DevAlignOut o;
                o.score = overflow ? 0xFFFFFFFDu : best <= k ? (u32)best : 0xFFFFFFFFu;
                o.end_col = (u32)best_col;
                out[out_index] = o;
                if (overflow) atomicAdd(&queue[1], 1u);
                phase = PH_NEED_JOB;
            } else {
                p_le = b; p_fl = c_fl; p_ll = c_ll; p_anchor = c_anchor; p_bot_end = bot;
                ++g;
                phase = PH_GROUP_START;
            }
        }
    }
    if (stats) {
        atomicAdd(&stats[0], n_blocks);
        if (lane == 0) atomicAdd(&stats[1], n_iter);
        atomicAdd(&stats[2], n_lane_iter);
        atomicAdd(&stats[3], n_groups);
    }
}

// ------------------------------------------------------------------------------------------------ P lanes per job
// One lane per job walks a job's groups one after the other: 9 x fewer blocks than the ring form computes, but the large nodes of a PEX tree
// are few and long (10 kb @ 8 %: the top round has 8 k jobs of 3 k blocks each per 2048 reads: 124 waves with a chain of 3450 blocks).
// Here a job has a team of P consecutive lanes; lane t of the team takes the groups t, t + P, t + 2P, ... and runs one block behind the lane
// that computes the group above (the team's lanes are in one wave: "behind" is a matter of which lanes sit an iteration out). What a group
// needs from the one above - the carries of its bottom row block by block, the first block with a value <= k, the row's value in front of
// the start block - is read where the lane above keeps it: its row buffer in LDS and, while that lane is still at work on the group, its
// registers (shuffles); the values of a finished group stay in LDS (fin). The rules of the one-lane form hold unchanged, with two of them
// turned into waits: a group starts once the group above has computed its start block and has shown a value <= k in reach of it (or has
// ended: then the decision of the one-lane form), and a group that would end where the row above has had no value <= k so far waits for
// the group above to end or to show one. The blocks computed are the same as in the one-lane form, hence the same results.
namespace {
enum : u32 { TP_NEED_JOB = 0, TP_GROUP_START = 1, TP_BLOCK = 2, TP_DONE = 3 };
constexpr u32 FIN_WORDS = 6;                    // group, blocks computed up to (exclusive), first / last block with a value <= k, anchor, value at the end
}

template <u32 P>
__global__ void __launch_bounds__(64) ed_exists_team_kernel(const u8* __restrict__ text, const u64* __restrict__ peq,
                                                            const DevAlignJob* __restrict__ jobs, u32 n_jobs, const u32* __restrict__ n_jobs_dev,
                                                            u32* __restrict__ queue, u32 cap, DevAlignOut* __restrict__ out,
                                                            unsigned long long* __restrict__ stats, u32 prio) {
    extern __shared__ __attribute__((aligned(16))) u64 lds_eq[];
    u32 const lane = lane_index();
    u32 const tl = lane & (P - 1u), team_base = lane & ~(P - 1u), above = team_base + ((tl + P - 1u) & (P - 1u));
    u64 const team_mask = (P == 64u ? ~0ull : ((1ull << P) - 1ull)) << team_base;
    u32* const rows = reinterpret_cast<u32*>(lds_eq + 7 * 64);
    u32* const row = rows + lane * cap;                       // this lane's bottom rows, block b at b mod cap
    const u32* const row_above = rows + above * cap;
    int* const fin_all = reinterpret_cast<int*>(rows + 64u * cap);
    int* const fin = fin_all + lane * FIN_WORDS;
    const int* const fin_above = fin_all + above * FIN_WORDS;
    if (prio == 3u) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2u) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1u) __builtin_amdgcn_s_setprio(1);
    if (n_jobs_dev) n_jobs = min(n_jobs, *n_jobs_dev);

    u32 phase = TP_NEED_JOB;
    bool job_over = false, job_dead = false, job_overflow = false;      // this lane saw the job's end (its last group ended) / its death
    const u8* ref = text;
    u64 q_off = 0;
    int n = 0, m = 1, k = 0, Lg = 1, pad = 0, band_hi = 0;
    u32 out_index = 0;
    int g = 0x3FFFFFFF, b = 0, b_hi = -1, b_lo = 0, rows_g = 64, b_lo_next = 0;      // g: the group at work or about to start; past the last: 0x3FFFFFFF
    bool running = false;                                                              // between the group's start and its end
    u32 slot = 0;
    u64 vp = ~0ull, vn = 0ull;
    int bot = 0, top_cur = 0, best = 0, best_col = 0;
    int c_ls = 0, c_fl = LB_NONE, c_ll = -1, c_anchor = 0;
    uint4 tq0 = make_uint4(0, 0, 0, 0), tq1 = tq0;
    u64 eq_lo[6] = {0, 0, 0, 0, 0, 0}, eq_hi[6] = {0, 0, 0, 0, 0, 0};
    u64 n_blocks = 0, n_iter = 0, n_lane_iter = 0, n_groups = 0;

    i64 eq_off = 0;
    auto issue_eq = [&](int gg) {
        eq_off = (i64)q_off + 64 * gg - pad;
        u64 const a = eq_off >= 0 ? (u64)eq_off >> 6 : 0ull;
#pragma unroll
        for (u32 sy = 0; sy < 6; ++sy) {
            eq_lo[sy] = peq[a * 6 + sy];
            eq_hi[sy] = peq[(a + 1) * 6 + sy];
        }
    };
    auto finish_eq = [&](u32 sy, u64 padmask) -> u64 {
        u32 const sh = (u32)eq_off & 63u;
        u64 const joined = sh ? (eq_lo[sy] >> sh) | (eq_hi[sy] << (64u - sh)) : eq_lo[sy];
        u64 const v = eq_off >= 0 ? joined : eq_lo[sy] << (u32)(-eq_off);
        return v | padmask;
    };

    for (;;) {
        // ---- a team whose lanes are all free takes the next job of the queue (its lanes read the same record)
        u64 const m_free = __ballot(phase == TP_NEED_JOB);
        bool const team_free = (m_free & team_mask) == team_mask;
        u64 const m_lead = __ballot(team_free && tl == 0u);
        if (m_lead) {
            u32 const leader = (u32)__builtin_ctzll(m_lead);
            u32 base = 0;
            if (lane == leader) base = atomicAdd(queue, (u32)__popcll(m_lead));
            base = (u32)__shfl((int)base, (int)leader);
            if (team_free) {
                u32 const id = base + (u32)__popcll(m_lead & ((1ull << team_base) - 1ull));
                if (id >= n_jobs) phase = TP_DONE;
                else {
                    DevAlignJob const job = jobs[id];
                    n = (int)job.n; m = (int)job.m; k = (int)job.k;
                    out_index = job.out_index;
                    job_over = job_dead = job_overflow = false;
                    running = false;
                    if (n == 0 || n + k < m) {
                        if (tl == 0u) {
                            DevAlignOut o;
                            o.score = (n == 0 && m <= k) ? (u32)m : 0xFFFFFFFFu;
                            o.end_col = 0u;
                            out[out_index] = o;
                        }
                        // (phase stays TP_NEED_JOB: the team asks again)
                    } else {
                        ref = text + job.ref_off;
                        q_off = job.q_off;
                        Lg = (m + 63) >> 6;
                        pad = Lg * 64 - m;
                        band_hi = n - m + k;
                        best = m;
                        best_col = 0;
                        fin[0] = -1;
                        if ((int)tl < Lg) { g = (int)tl; issue_eq(g); phase = TP_GROUP_START; }
                        else { g = 0x3FFFFFFF; phase = TP_BLOCK; }      // (more lanes than groups: this one waits for the job's end)
                    }
                }
            }
        }
        if (!__any(phase != TP_DONE)) break;
        ++n_iter;

        // ---- the lane above: where it is (registers), what it left of its last group (LDS)
        int const a_g = __shfl(g, (int)above), a_b = __shfl(b, (int)above), a_fl = __shfl(c_fl, (int)above), a_ll = __shfl(c_ll, (int)above);
        int const a_anchor = __shfl(c_anchor, (int)above), a_running = __shfl((int)running, (int)above);
        bool const above_done = g > 0 && g < 0x3FFFFFFF && a_g > g - 1;       // it has left group g - 1 behind: fin holds that group
        bool const above_live = g > 0 && g < 0x3FFFFFFF && a_g == g - 1 && a_running != 0;
        int p_le = 0, p_fl = LB_NONE, p_ll = -1, p_anchor = 0, p_bot_end = 0;
        if (above_done) { p_le = fin_above[1]; p_fl = fin_above[2]; p_ll = fin_above[3]; p_anchor = fin_above[4]; p_bot_end = fin_above[5]; }
        else if (above_live) { p_le = a_b; p_fl = a_fl; p_ll = a_ll; p_anchor = a_anchor; }

        // ---- lanes at the start of a group
        if (phase == TP_GROUP_START && (g == 0 || above_done || above_live)) {
            int const r0 = max(0, 64 * g - pad);
            int const r1 = 64 * (g + 1) - pad;
            rows_g = r1 - r0;
            b_lo = max(0, r0 - k) >> 4;
            b_hi = min(n - 1, r1 - 1 + band_hi) >> 4;
            b_lo_next = max(0, r1 - k) >> 4;
            int go = 1;                                           // 1 start, 0 no alignment, -1 not yet known
            if (g == 0) { b = 0; top_cur = 0; }
            else if (above_done) {
                if (p_fl == LB_NONE || p_ll < b_lo - 1) go = 0;
                else { b = max(b_lo, p_fl); if (b > b_hi) go = 0; }
            } else {
                // the group above is at work: start once it has shown a value <= k in reach and has computed the start block
                if (p_fl == LB_NONE || p_ll < b_lo - 1) go = -1;
                else { b = max(b_lo, p_fl); if (b > b_hi) go = 0; else if (p_le <= b) go = -1; }
            }
            if (go == 0) job_dead = true;
            else if (go == 1) {
                u64 const padmask = g == 0 && pad ? (1ull << (u32)pad) - 1ull : 0ull;
#pragma unroll
                for (u32 sy = 0; sy < 6; ++sy) lds_eq[sy * 64u + lane] = finish_eq(sy, padmask);
                lds_eq[6u * 64u + lane] = padmask;
                vp = ~padmask;
                vn = 0ull;
                __builtin_memcpy(&tq0, ref + 16 * (i64)b, 16);
                __builtin_memcpy(&tq1, ref + 16 * (i64)min(b + 1, b_hi), 16);
                if (g + (int)P < Lg) issue_eq(g + (int)P);
                slot = (u32)b % cap;
                if (g > 0) top_cur = b < p_le ? p_anchor : p_bot_end + 16 * (b - p_le);
                bot = top_cur + rows_g;
                c_ls = b;
                c_fl = LB_NONE;
                c_ll = -1;
                running = true;
                phase = TP_BLOCK;
                ++n_groups;
            }
        } else if (phase == TP_BLOCK && g < 0x3FFFFFFF) {
            // ---- one block of the lane's group - or an iteration's wait for the group above
            bool const above_known = g == 0 || above_done;        // (the row above is final)
            bool const dry = g > 0 && b > p_ll && top_cur + bot - rows_g > 2 * k;      // nothing <= k in the row above from here on, as far as it is known
            bool const ended = b > b_hi || (above_known && dry);
            bool const have_row = g == 0 || above_done || b < p_le;      // the row above at block b is there (or known to be "+1 per column")
            if (!ended && !(dry && !above_known) && have_row) {
                ++n_lane_iter;
                ++n_blocks;
                u32 const cw_in = g == 0 ? 0u : (b < p_le ? row_above[(u32)b % cap] : 0x55555555u);
                u32 const quad[4] = {tq0.x, tq0.y, tq0.z, tq0.w};
                tq0 = tq1;
                __builtin_memcpy(&tq1, ref + 16 * (i64)min(b + 2, b_hi), 16);
                int const bot_start = bot;
                bool const last = g == Lg - 1;
                int minb = LB_NONE;
                u32 cw = 0;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int const j = 4 * qd + i;
                        int const c = 16 * b + j;
                        u32 sym = (quad[qd] >> (8 * i)) & 0xFFu;
                        sym = c < n ? sym : 6u;
                        u64 const c_hp = (cw_in >> (2 * j)) & 1u, c_hn = (cw_in >> (2 * j + 1)) & 1u;
                        u64 const eq = lds_eq[sym * 64u + lane];
                        u64 const x = eq | vn;
                        u64 const tt = vp + (x & vp) + c_hn;
                        u64 const d0 = (tt ^ vp) | x;
                        u64 const hn = vp & d0;
                        u64 const hp = vn | ~(vp | d0);
                        u64 const xh = (hp << 1) | c_hp;
                        vn = xh & d0;
                        vp = (hn << 1) | ~(xh | d0) | c_hn;
                        u32 const o_hp = (u32)(hp >> 63), o_hn = (u32)(hn >> 63);
                        cw |= (o_hp | (o_hn << 1)) << (2 * j);
                        bot += (int)o_hp - (int)o_hn;
                        minb = min(minb, bot);
                        if (last && c < n && bot <= best) { best = bot; best_col = c + 1; }
                    }
                }
                if (b - c_ls >= (int)cap) job_overflow = true;
                row[slot] = cw;
                top_cur += __popc(cw_in & 0x55555555u) - __popc(cw_in & 0xAAAAAAAAu);
                if (minb <= k) { if (c_fl == LB_NONE) c_fl = b; c_ll = b; }
                if (b == max(b_lo_next, c_fl)) c_anchor = bot_start;
                ++b;
                slot = slot + 1u == cap ? 0u : slot + 1u;
            } else if (ended) {
                running = false;
                if (g == Lg - 1 || job_overflow) job_over = true;
                else {
                    fin[0] = g; fin[1] = b; fin[2] = c_fl; fin[3] = c_ll; fin[4] = c_anchor; fin[5] = bot;
                    g += (int)P;
                    if (g < Lg) phase = TP_GROUP_START;
                    else g = 0x3FFFFFFF;                              // no further group for this lane: it waits for the job's end
                }
            }
        }

        // ---- the job's end: its last group has ended (that lane writes the result), or a group found nothing to start from
        u64 const m_over = __ballot(job_over), m_dead = __ballot(job_dead);
        if ((m_over | m_dead) & team_mask) {
            if (phase != TP_NEED_JOB && phase != TP_DONE) {
                bool const writer = job_over || (!(m_over & team_mask) && job_dead && (u32)__builtin_ctzll(m_dead & team_mask) == lane);
                if (writer) {
                    DevAlignOut o;
                    o.score = job_over ? (job_overflow ? 0xFFFFFFFDu : best <= k ? (u32)best : 0xFFFFFFFFu) : 0xFFFFFFFFu;
                    o.end_col = job_over ? (u32)best_col : 0u;
                    out[out_index] = o;
                    if (job_over && job_overflow) atomicAdd(&queue[1], 1u);
                }
                phase = TP_NEED_JOB;
                g = 0x3FFFFFFF;
                running = false;
                job_over = job_dead = false;
            }
        }
    }
    if (stats) {
        atomicAdd(&stats[0], n_blocks);
        if (lane == 0) atomicAdd(&stats[1], n_iter);
        atomicAdd(&stats[2], n_lane_iter);
        atomicAdd(&stats[3], n_groups);
    }
}

size_t DeviceApi::exists_lane_lds_bytes(u32 cap_blocks) { return (size_t)7 * 64 * 8 + (size_t)64 * cap_blocks * 4 + 64 * FIN_WORDS * 4; }

int DeviceApi::align_exists_lanes(void* stream, const u8* d_text, const u64* d_peq, const DevAlignJob* d_jobs, u32 max_jobs, const u32* d_n_jobs,
                                  u32* d_queue, u32 waves, u32 cap_blocks, DevAlignOut* d_out, unsigned long long* d_stats, u32 lanes_per_job) {
    if (max_jobs == 0) return 0;
    size_t const lds = exists_lane_lds_bytes(cap_blocks);
    if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
    static u32 const prio = [] { const char* e = getenv("FLX_EXISTS_PRIO"); return e ? (u32)atoi(e) : 3u; }();
    u32 const P = lanes_per_job >= 16 ? 16u : lanes_per_job >= 8 ? 8u : lanes_per_job >= 4 ? 4u : lanes_per_job >= 2 ? 2u : 1u;
    u32 const blocks = std::max(1u, std::min((u32)(((u64)max_jobs * P + 63u) / 64u), waves));
    auto launch = [&](auto kernel) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), lds, (hipStream_t)stream, d_text, d_peq, d_jobs, max_jobs, d_n_jobs, d_queue, cap_blocks, d_out, d_stats, prio);
    };
    switch (P) {
        case 16: launch(ed_exists_team_kernel<16>); break;
        case 8: launch(ed_exists_team_kernel<8>); break;
        case 4: launch(ed_exists_team_kernel<4>); break;
        case 2: launch(ed_exists_team_kernel<2>); break;
        default: launch(ed_exists_lane_kernel); break;
    }
    return (int)hipGetLastError();
}

}  // namespace flx
