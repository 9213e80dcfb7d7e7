// Micro-benchmark: what does a divergent 16-byte-per-lane gather cost on gfx950? Every active lane reads K x 16 B of its own random
// 64-byte block per step (K1's access shape), with different fractions of the lanes active and different table sizes.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/gather_cost.hip -o /tmp/gather_cost && /tmp/gather_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int K, int STRIDE_LANES>
__global__ void __launch_bounds__(64) gather(const uint4* __restrict__ tab, unsigned long long n_blocks, int steps, unsigned active_mod, unsigned* out) {
    unsigned const lane = threadIdx.x, gid = blockIdx.x * 64 + lane;
    unsigned long long x = gid * 0x9E3779B97F4A7C15ull + 12345;
    unsigned acc = 0;
    bool const active = (lane % active_mod) == 0;
    for (int s = 0; s < steps; ++s) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        unsigned long long b = (x >> 8) % n_blocks;
        if (STRIDE_LANES == 2) b = (__shfl((long long)b, lane & ~1) & ~1ull) | (lane & 1);   // pairs of lanes share a 128-byte line
        if (active) {
            const uint4* p = tab + b * 4;
#pragma unroll
            for (int k = 0; k < K; ++k) { uint4 v = p[k]; acc ^= v.x + v.y + v.z + v.w; }
            x += acc & 1;          // the next address depends on the data: one step in flight per lane, like the DFS
        }
    }
    out[gid] = acc;
}

template <int K, int SL>
double run(const uint4* tab, unsigned long long n_blocks, int waves, int steps, unsigned active_mod, unsigned* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((gather<K, SL>), dim3(waves), dim3(64), 0, 0, tab, n_blocks, steps / 4, active_mod, out);
    hipEventRecord(a);
    hipLaunchKernelGGL((gather<K, SL>), dim3(waves), dim3(64), 0, 0, tab, n_blocks, steps, active_mod, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main(int argc, char** argv) {
    unsigned long long bytes = argc > 1 ? strtoull(argv[1], 0, 10) : (2ull << 30);
    unsigned long long n_blocks = bytes / 64;
    uint4* tab; hipMalloc(&tab, bytes); hipMemset(tab, 1, bytes);
    int const steps = 2000;
    if (argc > 2) {
        // calibration run for FETCH_SIZE: 4096 waves x 64 lanes x `steps` random 64-byte blocks, one 16-byte load each (K = 1), then
        // two per block (K = 2): the counter per dispatch / blocks = bytes the counter tallies per block
        unsigned* out; hipMalloc(&out, (size_t)4096 * 64 * 4);
        hipLaunchKernelGGL((gather<1, 1>), dim3(4096), dim3(64), 0, 0, tab, n_blocks, steps, 1u, out);
        hipLaunchKernelGGL((gather<2, 1>), dim3(4096), dim3(64), 0, 0, tab, n_blocks, steps, 1u, out);
        hipDeviceSynchronize();
        printf("calibration: each dispatch reads %.0f random 64-byte blocks of a %.1f GB table (K = 1, then K = 2 loads per block)\n", 4096.0 * 64 * steps, bytes / 1e9);
        return 0;
    }
    for (int waves : {4096, 8192}) {
        unsigned* out; hipMalloc(&out, (size_t)waves * 64 * 4);
        for (unsigned mod : {1u, 2u, 4u}) {
            double lanes = (double)waves * 64 / mod * steps;
            double t1 = run<1, 1>(tab, n_blocks, waves, steps, mod, out), t2 = run<2, 1>(tab, n_blocks, waves, steps, mod, out),
                   t3 = run<3, 1>(tab, n_blocks, waves, steps, mod, out), t4 = run<4, 1>(tab, n_blocks, waves, steps, mod, out);
            printf("table %.1f GB waves %d active 1/%u: blocks/s (G)  K=1 %.1f  K=2 %.1f  K=3 %.1f  K=4 %.1f   | ms %.2f %.2f %.2f %.2f\n", bytes / 1e9, waves, mod,
                   lanes / t1 / 1e6, lanes / t2 / 1e6, lanes / t3 / 1e6, lanes / t4 / 1e6, t1, t2, t3, t4);
        }
        double lanes = (double)waves * 64 * steps;
        double p2 = run<2, 2>(tab, n_blocks, waves, steps, 1, out), p4 = run<4, 2>(tab, n_blocks, waves, steps, 1, out);
        printf("table %.1f GB waves %d pairs sharing a 128-B line: half-lines/s (G)  K=2 %.1f  K=4 %.1f\n", bytes / 1e9, waves, lanes / p2 / 1e6, lanes / p4 / 1e6);
        hipFree(out);
    }
    return 0;
}
