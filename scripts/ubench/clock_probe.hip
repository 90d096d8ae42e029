// Box probe 2: the EFFECTIVE shader clock under light load.
// Round 4, first lease: a box ran the training step in 3.60 ms instead of 2.30 while load_latency.hip read the same
// 87 / 222 / 374 ns as on the fast boxes -- the dependent-load latency does NOT separate the boxes.  The kernels that were
// 2-5x slower there are the LDS / VALU chains on small grids (wopos_small_*, attn_fwd3 at 128 workgroups); the pure
// global-load chains (bn_act_bwd_small, bn_finalize) and the full-chip bandwidth kernels ran at the same speed.  That is the
// signature of a shader clock that follows the load.  This probe times a dependent v_fma_f32 chain and a dependent
// ds_read chain (wall_clock64, constant 100 MHz) at three grid sizes: one wave, 128 workgroups x 256 threads (the step's
// typical grid), 2048 x 256 (whole chip), each as 100 back-to-back launches (first / median / last reported: a clock that
// ramps with sustained load shows up as first > last).
// Build: hipcc --offload-arch=gfx950 -O3 clock_probe.hip -o clock_probe.bin ; prints one JSON line.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

constexpr int N_FMA = 1 << 15, N_LDS = 1 << 12;

__global__ __launch_bounds__(256) void chain_kernel(float a, float b, unsigned long long* out, int slot, float* sink) {
    __shared__ int ring[256];
    ring[threadIdx.x] = (threadIdx.x * 37 + 11) & 255;          // a permutation of 0..255 (37 is odd)
    __syncthreads();
    float x = (float)threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    const unsigned long long c0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N_FMA; ++i) x = __builtin_fmaf(x, a, b);
    const unsigned long long t1 = wall_clock64();
    const unsigned long long c1 = clock64();
    int j = threadIdx.x;
#pragma unroll 16
    for (int i = 0; i < N_LDS; ++i) j = ring[j];
    const unsigned long long t2 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[slot * 3] = t1 - t0;
        out[slot * 3 + 1] = t2 - t1;
        out[slot * 3 + 2] = c1 - c0;
    }
    if (x == 12345.678f || j == 1000) sink[0] = x + j;          // keep both chains alive
}

int main() {
    const int REPS = 100;
    unsigned long long* out;
    float* sink;
    if (hipMalloc(&out, REPS * 3 * 8) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("{\"error\": \"alloc\"}\n"); return 1; }
    const int grids[3] = {1, 128, 2048}, threads[3] = {64, 256, 256};
    const char* names[3] = {"one_wave", "wg128", "wg2048"};
    printf("{");
    for (int g = 0; g < 3; ++g) {
        (void)hipDeviceSynchronize();
        for (int r = 0; r < REPS; ++r) hipLaunchKernelGGL(chain_kernel, dim3(grids[g]), dim3(threads[g]), 0, 0, 1.0000001f, 1e-9f, out, r, sink);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(REPS * 3);
        (void)hipMemcpy(h.data(), out, REPS * 3 * 8, hipMemcpyDeviceToHost);
        std::vector<double> f(REPS), l(REPS);
        for (int r = 0; r < REPS; ++r) { f[r] = h[r * 3] * 10.0 / N_FMA; l[r] = h[r * 3 + 1] * 10.0 / N_LDS; }
        const double f0 = f[0], fl = f[REPS - 1], l0 = l[0], ll = l[REPS - 1];
        const double cyc = (double)h[(REPS - 1) * 3 + 2] / (double)h[(REPS - 1) * 3];      // clock64 ticks per 10 ns
        std::sort(f.begin(), f.end());
        std::sort(l.begin(), l.end());
        printf("%s\"%s\": {\"ns_per_dependent_fma\": {\"first\": %.3f, \"median\": %.3f, \"last\": %.3f}, "
               "\"ns_per_dependent_lds_read\": {\"first\": %.2f, \"median\": %.2f, \"last\": %.2f}, \"clock64_per_wall_tick\": %.3f}",
               g ? ", " : "", names[g], f0, f[REPS / 2], fl, l0, l[REPS / 2], ll, cyc);
    }
    printf("}\n");
    return 0;
}
