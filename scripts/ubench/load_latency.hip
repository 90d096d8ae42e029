// Box probe: latency of a DEPENDENT global load on this GPU box (one lane chasing a random cyclic permutation, 64-byte
// stride) for three footprints -- 1 MB (inside one XCD's L2, warmed), 64 MB (inside the 256 MB Infinity Cache, warmed), 1 GB
// (HBM, never-touched lines).
// The training step of this repo is ~330 kernels per 2.3 ms, most of them chains of a few dependent memory round trips
// per workgroup: its time follows this number, not the bandwidth.  Boxes of the same SKU and the
// same reported clocks were observed with step times of 2.34 and 3.63 ms; bench.py records this probe beside the step
// time so the two can be told apart.  Prints one JSON line.  Build: hipcc --offload-arch=gfx950 -O3 load_latency.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ void chase(const uint32_t* __restrict__ ring, uint32_t start, int steps, unsigned long long* out) {
    uint32_t i = start + (threadIdx.x >> 6);                 // (not provably uniform: vector loads, like the kernels' own)
    const unsigned long long t0 = wall_clock64();
    for (int s = 0; s < steps; ++s) i = ring[(size_t)i * 16];
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = i; }
}

// warm: walk the whole cycle once first (the footprint then sits in the deepest cache that holds it); else every
// measured walk starts a quarter of the cycle further on and only ever touches lines nobody has read
static double probe(size_t bytes, int steps, bool warm) {
    const size_t slots = bytes / 64;
    std::vector<uint32_t> perm(slots), host(slots * 16, 0);
    for (size_t k = 0; k < slots; ++k) perm[k] = (uint32_t)k;
    uint64_t rng = 0x9e3779b97f4a7c15ull;
    for (size_t k = slots - 1; k > 0; --k) {                                  // Fisher-Yates, then link the order into one cycle
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        const size_t j = (size_t)((rng >> 33) % (k + 1));
        std::swap(perm[k], perm[j]);
    }
    for (size_t k = 0; k < slots; ++k) host[(size_t)perm[k] * 16] = perm[(k + 1) % slots];
    uint32_t* d;
    unsigned long long* o;
    if (hipMalloc(&d, slots * 64) != hipSuccess || hipMalloc(&o, 16) != hipSuccess) return -1.0;
    (void)hipMemcpy(d, host.data(), slots * 64, hipMemcpyHostToDevice);
    unsigned long long res[2] = {0, 0};
    if (warm) {
        hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, perm[0], (int)slots, o);
        (void)hipDeviceSynchronize();
    }
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, perm[(slots / 4) * rep + 17], steps, o);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(res, o, 16, hipMemcpyDeviceToHost);
        const double ns = (double)res[0] * 10.0 / steps;                      // wall_clock64: 100 MHz
        if (getenv("MEDT_PROBE_DEBUG")) fprintf(stderr, "bytes %zu rep %d: %.0f ns/step, end index %llu\n", bytes, rep, ns, res[1]);
        if (ns < best) best = ns;
    }
    (void)hipFree(d);
    (void)hipFree(o);
    return best;
}

int main() {
    const double l2 = probe((size_t)1 << 20, 4096, true), mall = probe((size_t)64 << 20, 4096, true),
                 hbm = probe((size_t)1 << 30, 4096, false);
    printf("{\"dependent_load_ns\": {\"1MB\": %.0f, \"64MB\": %.0f, \"1GB\": %.0f}}\n", l2, mall, hbm);
    return 0;
}
