// Microbenchmark: issue cost of fp32 VALU instructions on gfx950 (scalar v_fma_f32, packed v_pk_fma_f32, DPP add/mov, v_exp_f32)
// at 1 / 2 / 4 / 8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define N_ACC 8
#define ITERS 4096

template <int MODE>
__global__ void k(float* out, float a, float b) {
    float x[N_ACC];
    f2 y[N_ACC];
    for (int i = 0; i < N_ACC; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = f2{x[i], x[i] + 0.5f}; }
    const f2 a2 = f2{a, a * 1.01f}, b2 = f2{b, b * 0.99f};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < N_ACC; ++i) {
            if (MODE == 0) x[i] = fmaf(x[i], a, b);
            if (MODE == 1) y[i] = y[i] * a2 + b2;
            if (MODE == 2) x[i] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[i]), 0x121, 0xf, 0xf, true)) * 1e-9f;   // mov_dpp/add fused?
            if (MODE == 3) x[i] = __builtin_amdgcn_exp2f(x[i] * 1e-3f);
            if (MODE == 4) x[i] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(a), __float_as_int(x[i]), 0x111, 0xf, 0xf, false));
            if (MODE == 5) y[i] = y[i] * f2{a, a} + b2;      // splat operand (op_sel?)
        }
    }
    float s = 0.f;
    for (int i = 0; i < N_ACC; ++i) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int wps, float* d) {
    // 256 CUs x 4 SIMDs x wps waves
    const int waves = 256 * 4 * wps, blocks = waves / 4;          // 256 threads = 4 waves, one per SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)ITERS * N_ACC * wps;
    printf("%-14s waves/SIMD %d : %8.3f ms  -> %6.2f ns per wave-instr per SIMD (at 2.4 GHz: %5.2f cycles)\n", name, wps, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}

int main() {
    float* d; hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float) * 2);
    for (int wps : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", wps, d);
        run<1>("v_pk_fma_f32", wps, d);
        run<5>("v_pk_fma splat", wps, d);
        run<2>("dpp add+mul", wps, d);
        run<3>("v_exp_f32+mul", wps, d);
        run<4>("v_mov_dpp shr", wps, d);
    }
    return 0;
}
