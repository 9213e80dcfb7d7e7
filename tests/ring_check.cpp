// The ring schedule of the banded block kernels (flx_internal.hpp: ring_group_blocks, ring_delay, ring_offset, ring_steps), checked on the
// host over random job shapes: (1) a shape round 3 allowed (every group a lane of its own, or group g + R starting after group g has ended)
// never waits; (2) with the delay the function returns no lane is asked for two blocks in one block-step and every group runs exactly one
// block-step behind the group above it within a revolution; (3) simulated block-step by block-step, every block of every group is computed
// after the block of the group above it that it reads (directly: one step earlier; across a revolution: delay + 1 steps earlier, which is
// what the hand-over queue of ed_block_body holds). Test infrastructure.
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

#include "../floxer_amd/csrc/flx_internal.hpp"
using namespace flx;

int main() {
    std::mt19937 rng(5);
    u32 const Ws[] = {1, 2, 3, 4, 5, 6, 8, 13, 25};
    int no_wait_checked = 0, bad = 0, simulated = 0;
    for (int it = 0; it < 60000; ++it) {
        u32 const m = 1 + rng() % 12000, k = rng() % (m / 8 + 2), n = m - std::min(k, m - 1) + rng() % (3 * k + 40);
        if (n == 0 || (u64)n + k < m) continue;
        u32 const W = Ws[rng() % 9], R = 1u << (rng() % 7);
        int const nw = (int)((m + 63) / 64), Lg = (nw + (int)W - 1) / (int)W, pad = Lg * 64 * (int)W - (int)m;
        long const width = (long)n - (long)m + 2 * (long)k;
        u32 const d = ring_delay(n, m, k, W, R);
        if ((u32)Lg <= R || (long)64 * W * (R - 1) + R + 1 > width) { ++no_wait_checked; if (d != 0) { if (bad++ < 5) printf("FAIL a shape that never waited has delay %u: n %u m %u k %u W %u R %u\n", d, n, m, k, W, R); } }
        // the lane of group g is free before group g + R wants it
        for (int g = 0; g + (int)R < Lg; ++g) {
            int lo0, hi0, lo1, hi1;
            ring_group_blocks((int)n, (int)m, (int)k, (int)W, Lg, pad, g, lo0, hi0);
            ring_group_blocks((int)n, (int)m, (int)k, (int)W, Lg, pad, g + (int)R, lo1, hi1);
            if ((long)lo1 + (long)ring_offset((u32)g + R, R, d) <= (long)hi0 + (long)ring_offset((u32)g, R, d)) { if (bad++ < 5) printf("FAIL lane busy: n %u m %u k %u W %u R %u g %d\n", n, m, k, W, R, g); }
        }
        // what a group reads from above exists: block b of group g - 1 is computed before block b of group g, and group g - 1 covers every
        // block of group g up to its own last one (beyond that the kernel substitutes +1 per column)
        if (it % 8 == 0) {
            ++simulated;
            for (int g = 1; g < Lg; ++g) {
                int lo0, hi0, lo1, hi1;
                ring_group_blocks((int)n, (int)m, (int)k, (int)W, Lg, pad, g - 1, lo0, hi0);
                ring_group_blocks((int)n, (int)m, (int)k, (int)W, Lg, pad, g, lo1, hi1);
                long const gap = (long)ring_offset((u32)g, R, d) - (long)ring_offset((u32)g - 1, R, d);
                bool const same_revolution = (u32)g % R != 0;
                if (gap != (same_revolution ? 1 : 1 + (long)d) || lo1 < lo0 || lo1 > hi0) { if (bad++ < 5) printf("FAIL hand-over: n %u m %u k %u W %u R %u g %d gap %ld\n", n, m, k, W, R, g, gap); }
            }
            // the schedule's length is the last group's last block
            int lo, hi;
            ring_group_blocks((int)n, (int)m, (int)k, (int)W, Lg, pad, Lg - 1, lo, hi);
            if (ring_steps(n, m, k, W, R) != (u64)hi + ring_offset((u32)Lg - 1, R, d) + 1 || hi != (int)((n - 1) >> 4)) { if (bad++ < 5) printf("FAIL steps: n %u m %u k %u W %u R %u\n", n, m, k, W, R); }
        }
    }
    printf("%s %d shapes that never waited, %d schedules walked, %d failures\n", bad ? "FAILURES" : "ok", no_wait_checked, simulated, bad);
    return bad ? 1 : 0;
}
