// ticket.hip -- what does "every workgroup draws a ticket, the last one reads what the others wrote" (csrc/tail.h) cost a launch of
// `blocks` workgroups, and is it coherent across the XCDs' L2s?
//   mode 0: empty kernel; 1: one plain store per workgroup; 2: store + __threadfence() by every thread (buffer_wbl2 sc1);
//   3: mode 2 + the ticket atomic; 4: agent-scope relaxed ATOMIC store (write-through, sc1) + workgroup-scope release (s_waitcnt
//   only) + the ticket atomic, the last workgroup reads every word back with agent-scope atomic loads and counts stale ones;
//   5 / 6: mode 4 with two-level tickets, 16 / 64 sub-counters
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); return 1; } } while (0)
__global__ void ticket_kernel(int mode, float* out, unsigned* ticket, float val, unsigned* stale) {
    __shared__ int last;
    if (mode >= 1 && mode <= 3 && threadIdx.x == 0) out[blockIdx.x] = val;
    if (mode == 2 || mode == 3) __threadfence();
    if (mode >= 4 && threadIdx.x == 0) {
        __hip_atomic_store(out + blockIdx.x, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    __syncthreads();
    if (mode >= 3) {
        if (threadIdx.x == 0 && mode >= 5) {
            // two-level tickets: S sub-counters (one 256-byte line each) take the workgroups b % S, the one that fills a sub-counter
            // draws from the master: the longest same-address chain is blocks / S + S atomics instead of `blocks`
            const unsigned S = mode == 5 ? 16u : 64u, sub = blockIdx.x % S, quota = (gridDim.x - sub + S - 1) / S;
            unsigned* sc = ticket + 64 * (1 + sub);
            last = 0;
            if (__hip_atomic_fetch_add(sc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == quota - 1) {
                __hip_atomic_store(sc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned nsub = gridDim.x < S ? gridDim.x : S;
                if (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsub - 1) {
                    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    last = 1;
                }
            }
        } else if (threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = t == gridDim.x - 1;
            if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (last && mode >= 4) {
            unsigned bad = 0;
            for (unsigned b = threadIdx.x; b < gridDim.x; b += blockDim.x)
                bad += __hip_atomic_load(out + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != val;
            if (bad) atomicAdd(stale, bad);
        }
    }
}
int main() {
    float* out; unsigned *tk, *stale;
    CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&tk, 65 * 256)); CK(hipMemset(tk, 0, 65 * 256)); CK(hipMalloc(&stale, 64)); CK(hipMemset(stale, 0, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int blocks : {64, 512, 2048, 8192})
        for (int mode = 0; mode < 7; ++mode) {
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(ticket_kernel, dim3(blocks), dim3(256), 0, 0, mode, out, tk, (float)i, stale);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(ticket_kernel, dim3(blocks), dim3(256), 0, 0, mode, out, tk, (float)(100 + i), stale);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned h, st; CK(hipMemcpy(&h, tk, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&st, stale, 4, hipMemcpyDeviceToHost));
            printf("blocks %5d mode %d: %.2f us per launch (ticket word after: %u, stale words read by the last workgroups: %u)\n", blocks, mode, ms * 1e3 / 1000, h, st);
        }
    return 0;
}
