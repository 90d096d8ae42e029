// spin.hip -- a latency-bound stand-in for the step's small kernels: `blocks` workgroups of 256 lanes each spin for `ticks` of the
// 100 MHz wall clock (s_memrealtime), then one lane writes a word.  scripts/graph_edge_cost.py launches it on torch's streams.
#include <hip/hip_runtime.h>
__global__ void spin_kernel(unsigned long long ticks, int* out) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
    if (threadIdx.x == 0) out[blockIdx.x] = (int)ticks;
}
extern "C" int spin_launch(void* stream, double us, int blocks, int* out) {
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned long long)(us * 100.0), out);
    return (int)hipGetLastError();
}
