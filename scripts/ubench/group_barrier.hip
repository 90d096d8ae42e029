// Microbenchmark behind the round-4 decision "phases of one launch + a barrier among the workgroups of a BatchNorm group"
// versus "one launch per phase" for the local (patch) branch of MedT.
//
//   A. K dependent launches of a small kernel (128 workgroups x 256 threads; each writes 4 KB and reads the 4 KB its
//      neighbour wrote in the previous launch), captured into ONE hipGraph and replayed: microseconds per launch.
//   B. ONE launch of the same work as K phases separated by a barrier among the 8 workgroups of a group (ticket counter in
//      global memory, agent-scope release before the arrive, agent-scope acquire after the wait): microseconds per phase,
//      with the workgroups of a group (i) on consecutive block ids (round-robin dispatch puts them on 8 different XCDs) and
//      (ii) 16 block ids apart (same XCD under round-robin dispatch).
//   Every phase checks the neighbour's values: a stale read is counted as an error.
// Build: hipcc --offload-arch=gfx950 -O3 group_barrier.hip -o group_barrier.bin ; prints one JSON line.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int GROUPS = 16, PER = 8, WGS = GROUPS * PER, THREADS = 256, WORDS = 1024;      // 4 KB per workgroup and phase

__device__ __forceinline__ void locate(int strided, int& grp, int& mem) {
    if (strided) { mem = blockIdx.x / GROUPS; grp = blockIdx.x % GROUPS; }
    else { grp = blockIdx.x / PER; mem = blockIdx.x % PER; }
}

// one phase of "work": write my slab for phase p, read the slab my neighbour in the group wrote for phase p - 1
__device__ __forceinline__ unsigned phase_work(float* buf, int grp, int mem, int p) {
    unsigned bad = 0;
    float* mine = buf + ((size_t)(p & 1) * WGS + grp * PER + mem) * WORDS;
    const float* other = buf + ((size_t)((p - 1) & 1) * WGS + grp * PER + (mem + 1) % PER) * WORDS;
    for (int e = threadIdx.x; e < WORDS; e += THREADS) {
        if (p > 0) {
            const float v = __builtin_nontemporal_load(other + e) ;
            bad += v != (float)(p - 1 + (mem + 1) % PER);
        }
        mine[e] = (float)(p + mem);
    }
    return bad;
}

__global__ __launch_bounds__(THREADS) void one_phase_kernel(float* buf, unsigned* errs, int p, int strided) {
    int grp, mem;
    locate(strided, grp, mem);
    const unsigned bad = phase_work(buf, grp, mem, p);
    if (bad) atomicAdd(errs, bad);
}

__global__ __launch_bounds__(THREADS) void phases_kernel(float* buf, unsigned* errs, unsigned* tickets, int K, int strided) {
    int grp, mem;
    locate(strided, grp, mem);
    unsigned bad = 0;
    for (int p = 0; p < K; ++p) {
        bad += phase_work(buf, grp, mem, p);
        // ---- barrier among the PER workgroups of the group ----
        __syncthreads();                                            // every thread's stores are issued and acknowledged
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(tickets + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(p + 1) * PER;
            while (__hip_atomic_load(tickets + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    if (bad) atomicAdd(errs, bad);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s at %d\"}\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    const int K = 200, REPS = 20;
    float* buf; unsigned *errs, *tickets;
    CK(hipMalloc(&buf, (size_t)2 * WGS * WORDS * sizeof(float)));
    CK(hipMalloc(&errs, 4));
    CK(hipMalloc(&tickets, GROUPS * 4));
    CK(hipMemset(errs, 0, 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double us_launch[2], us_phase[2];
    for (int strided = 0; strided < 2; ++strided) {
        // A: K dependent launches in one graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int p = 0; p < K; ++p) hipLaunchKernelGGL(one_phase_kernel, dim3(WGS), dim3(THREADS), 0, s, buf, errs, p, strided);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        us_launch[strided] = ms * 1e3 / (REPS * K);
        // B: one launch, K phases
        CK(hipMemsetAsync(tickets, 0, GROUPS * 4, s));
        hipLaunchKernelGGL(phases_kernel, dim3(WGS), dim3(THREADS), 0, s, buf, errs, tickets, K, strided);
        CK(hipStreamSynchronize(s));
        float tot = 0.f;
        for (int r = 0; r < REPS; ++r) {
            CK(hipMemsetAsync(tickets, 0, GROUPS * 4, s));
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(phases_kernel, dim3(WGS), dim3(THREADS), 0, s, buf, errs, tickets, K, strided);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            tot += ms;
        }
        us_phase[strided] = tot * 1e3 / (REPS * K);
    }
    unsigned h = 0;
    CK(hipMemcpy(&h, errs, 4, hipMemcpyDeviceToHost));
    printf("{\"workgroups\": %d, \"phases\": %d, \"us_per_graph_launch\": [%.2f, %.2f], \"us_per_barrier_phase\": {\"consecutive_ids\": %.2f, "
           "\"stride16_ids\": %.2f}, \"stale_reads\": %u}\n", WGS, K, us_launch[0], us_launch[1], us_phase[0], us_phase[1], h);
    return 0;
}
