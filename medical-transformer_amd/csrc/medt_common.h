// medt_common.h -- shared host/device helpers for libmedt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "medt_abi.h"

#define MEDT_THREADS 256            // 4 wave64 per workgroup everywhere
#define MEDT_WAVES (MEDT_THREADS / 64)
// nothing is scheduled across this point: separates a batch of independent global loads from the arithmetic that
// consumes them, so the loads are all in flight before the first wait (latency-bound small kernels)
#define MEDT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define MEDT_LOG2E 1.4426950408889634f
// Workgroup barrier for phases that hand data over through LDS ONLY (block_small.hip): __syncthreads() is a fence over every
// address space and also waits for the wave's outstanding GLOBAL stores (s_waitcnt vmcnt(0)), which nothing inside that kernel
// depends on.  (Measured: no difference -- 2.2052 vs 2.2106 ms/step with __syncthreads() in every fused small-layer kernel;
// the store round trips overlap with the next phase's issue either way.  Kept in the one kernel that has eight barriers.)
// (tests/lane_emu -- the CPU lane emulator the kernel sources are also compiled for -- supplies its own of these)
#ifndef MEDT_STATIC_SHARED
#define MEDT_STATIC_SHARED __shared__      // a statically sized LDS array declared inside a kernel
#endif
#ifndef MEDT_VEC_TYPES                    // clang's OpenCL-style vectors (swizzles, splat casts, element-wise operators)
typedef float medt_f2 __attribute__((ext_vector_type(2)));
typedef float medt_f4 __attribute__((ext_vector_type(4)));
#endif
// Marks a point where the code relies on the lanes of a wavefront executing in lockstep (data handed from lane to lane through
// LDS inside one wave: the in-order LDS pipe needs no barrier).  Nothing on the GPU; the lane emulator, which runs the lanes of a
// wave one after the other between synchronisation points, makes it one.
#ifndef MEDT_WAVE_LOCKSTEP
#define MEDT_WAVE_LOCKSTEP() do { } while (0)
#endif
#ifndef MEDT_LDS_BARRIER
#define MEDT_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

namespace medt {

// ---- host side ------------------------------------------------------------
void set_error(const char* fmt, ...);
int  launch_status(const char* what);          // hipGetLastError -> MEDT_ELAUNCH
// TIMING EXPERIMENTS ONLY (scripts/r4_skip.sh): MEDT_SKIP=fam1,fam2 makes the host wrappers of those kernel families return
// without launching, so the step time that disappears is the family's share of the critical path.  Results are garbage.
bool abl_skip(const char* family);
// More than 64 KB of dynamic LDS per workgroup needs an opt-in per kernel (gfx950: 160 KB per CU).  Once per (call site, device):
// `done` is the call site's static flag array, one byte per device ordinal (racing first calls repeat an idempotent runtime call).
int lds_opt_in(const void* kernel, unsigned char (&done)[64], const char* what);

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int    cdiv(int a, int b) { return (a + b - 1) / b; }

// Bump allocator over the caller's workspace (256-byte aligned carves).
struct Carver {
    char*  base;
    size_t cap, off;
    Carver(void* p, size_t n) : base((char*)p), cap(n), off(0) {}
    template <class T> T* take(size_t count) {
        size_t o = align_up(off, 256);
        off = o + count * sizeof(T);
        return (T*)(base ? base + o : nullptr);
    }
    bool ok() const { return off <= cap; }
};

// Per-BatchNorm statistics block inside medt_axial_saved.stats: 4 arrays of groups*CH floats.
struct BnStats {
    float *mean, *rstd, *scale, *shift;        // scale = weight*rstd, shift = bias - mean*scale
    __host__ __device__ BnStats() : mean(nullptr), rstd(nullptr), scale(nullptr), shift(nullptr) {}
    __host__ __device__ BnStats(float* p, int n) : mean(p), rstd(p + n), scale(p + 2 * n), shift(p + 3 * n) {}
};

// ---- device side ----------------------------------------------------------
// Activation storage of the attention layer's saved tensors (qkv_raw, stacked): float32, or bfloat16 when
// medt_axial_desc.act_dtype == 1.  Arithmetic is always fp32: bf16 is widened on load (exact) and rounded to
// nearest-even on store.  Pointers keep the type float*; `bf` says what they really point to.
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float ld_act(const float* p, size_t i, int bf) {
    return bf ? bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(p)[i]) : p[i];
}
__device__ __forceinline__ void st_act(float* p, size_t i, float v, int bf) {
    if (bf) reinterpret_cast<unsigned short*>(p)[i] = f32_to_bf16_bits(v);
    else p[i] = v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum K per-thread values over the 256-thread workgroup.  Result valid in threads [0,K)
// of wave 0 as the return of the k-th slot: out[k] for tid==k.  `red` is MEDT_WAVES*K floats of LDS.
template <int K>
__device__ __forceinline__ void block_sum(float (&v)[K], float* red, float* dst, int dst_stride = 1) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float s = wave_sum(v[k]);
        if (lane == 0) red[wave * K + k] = s;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += MEDT_THREADS) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < MEDT_WAVES; ++w) s += red[w * K + k];
        dst[k * dst_stride] = s;
    }
    __syncthreads();
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Forward BatchNorm statistics.  The partial sums [sum x, sum x^2] every producer kernel hands to bn_finalize are
// DOUBLES, reduced in double from the per-thread float values on: the batch variance is finalised as E[x^2] - E[x]^2, and a
// float32 reduction tree rounds sum x^2 at ~1e-7 relative -- 1e-7 * (1 + mean^2 / var) of the variance, visibly above the
// noise of the reference, whose nn.BatchNorm accumulates in double on the CPU (bn_output of the attention layers has
// mean^2 / var ~ 100).  With double partials only the single rounding of each x * x is left (random, averages out).
// `red`: MEDT_WAVES * min(K, 32) * 2 floats of LDS (the doubles are stored as two words: no 8-byte alignment needed).
// (sum x, sum (x - m)^2 about a float shift m) -> (sum x, sum x^2) in double: the exact identity
// sum (x - m)^2 = sum x^2 - 2 m sum x + n m^2, whatever m is.  The kernels that hold a BatchNorm group's values in LDS
// (conv_small.hip, axial_small.hip) sum the squares about m = sum / n (second pass), where a float32 sum is harmless.
__device__ __forceinline__ void centered_to_raw(float s, float m2, float m, double count, double& sum, double& sumsq) {
    sum = (double)s;
    sumsq = (double)m2 + 2.0 * (double)m * sum - count * (double)m * (double)m;
}

template <int K>
__device__ __forceinline__ void block_sum_d(const float (&v)[K], float* red, double* dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int CH = K < 32 ? K : 32;
    int* ri = reinterpret_cast<int*>(red);
#pragma unroll
    for (int k0 = 0; k0 < K; k0 += CH) {
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (k0 + k < K) {
                const double s = wave_sum_d((double)v[k0 + k]);
                if (lane == 0) {
                    ri[(wave * CH + k) * 2] = __double2loint(s);
                    ri[(wave * CH + k) * 2 + 1] = __double2hiint(s);
                }
            }
        }
        __syncthreads();
        for (int k = threadIdx.x; k < CH && k0 + k < K; k += MEDT_THREADS) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < MEDT_WAVES; ++w) s += __hiloint2double(ri[(w * CH + k) * 2 + 1], ri[(w * CH + k) * 2]);
            dst[k0 + k] = s;
        }
        __syncthreads();
    }
}

// Backward finalisation:  dx = A*(d - m1 - xhat*m2), xhat = (x-mean)*rstd, A = weight*rstd
//   => dx = c0*d_raw + c1*x + c2 with d = dscale*d_raw.
__device__ __forceinline__ void bn_bwd_coef(double s1, double s2, double count, float dscale, double mean, double rstd,
                                            double w, int training, float* cf) {
    const double A = w * rstd;
    cf[0] = (float)(A * dscale);
    if (training) {
        const double m1 = s1 / count, m2 = s2 / count;
        cf[1] = (float)(-A * rstd * m2);
        cf[2] = (float)(A * (rstd * mean * m2 - m1));
    } else {
        cf[1] = 0.f;
        cf[2] = 0.f;
    }
}


}  // namespace medt
