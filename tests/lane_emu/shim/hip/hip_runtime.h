// hip/hip_runtime.h -- SHIM for tests/lane_emu (test infrastructure): the HIP vocabulary the block kernels use, mapped onto the
// CPU lane emulator so that the kernels' .hip SOURCE compiles with g++.  Not a HIP implementation: only what
// medical-transformer_amd/csrc/block_small.hip and the headers it includes touch.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "lane_emu.h"

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__                         // extern __shared__ float smem[] -> one global array (emu_common.cpp)
#define MEDT_STATIC_SHARED static          // in-kernel static LDS arrays: one instance, shared by all work-items

using std::max;
using std::min;

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct uint2 { unsigned x, y; };

// clang's OpenCL-style float vectors as far as the kernels use them: splat casts, brace initialisation, [] and .x/.y/.z/.w,
// .lo / .hi, element-wise + - * and their compound forms
#define MEDT_VEC_TYPES 1
struct medt_f2 {
    float x, y;
    medt_f2() = default;
    medt_f2(float s) : x(s), y(s) {}
    medt_f2(float a, float b) : x(a), y(b) {}
    float& operator[](int i) { return i ? y : x; }
    float operator[](int i) const { return i ? y : x; }
};
#define LANE_EMU_V2_OP(op)                                                                                      \
    static inline medt_f2 operator op(medt_f2 a, medt_f2 b) { return medt_f2(a.x op b.x, a.y op b.y); }          \
    static inline medt_f2& operator op##=(medt_f2& a, medt_f2 b) { a.x op## = b.x; a.y op## = b.y; return a; }
LANE_EMU_V2_OP(+) LANE_EMU_V2_OP(-) LANE_EMU_V2_OP(*)
#undef LANE_EMU_V2_OP
static inline medt_f2 operator-(medt_f2 a) { return medt_f2(-a.x, -a.y); }
struct lane_emu_f2_pod {                       // (members of anonymous aggregates may not have constructors)
    float x, y;
    operator medt_f2() const { return medt_f2(x, y); }
};
struct medt_f4 {
    union {
        struct { float x, y, z, w; };
        struct { lane_emu_f2_pod lo, hi; };
    };
    medt_f4() = default;
    medt_f4(float s) : x(s), y(s), z(s), w(s) {}
    medt_f4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};

#define threadIdx lane_emu::g_threadIdx
#define blockIdx lane_emu::g_blockIdx
#define blockDim lane_emu::g_blockDim
#define gridDim lane_emu::g_gridDim

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
// (every emulated launch runs to completion before the call returns: events order nothing)
typedef void* hipEvent_t;
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (void*)1; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
// (one work-item runs at a time: read-modify-write is atomic by construction; the ORDER of float atomics is the emulator's)
#define __HIP_MEMORY_SCOPE_AGENT 0
template <class T> static inline T __hip_atomic_load(const T* p, int, int) { return *p; }
template <class T> static inline T atomicAdd(T* p, T v) { const T o = *p; *p = o + v; return o; }
static inline unsigned atomicOr(unsigned* p, unsigned v) { const unsigned o = *p; *p = o | v; return o; }

#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...)                                                      \
    do {                                                                                                             \
        const dim3 g_ = (grid), b_ = (block);                                                                        \
        (void)(stream);                                                                                              \
        lane_emu::launch(lane_emu::Idx3{g_.x, g_.y, g_.z}, lane_emu::Idx3{b_.x, b_.y, b_.z}, (size_t)(lds),           \
                         [&]() { kern(__VA_ARGS__); });                                                              \
    } while (0)

// ---- bit casts, transcendental builtins ------------------------------------------------------------------------------------
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline int __double2loint(double d) { uint64_t u; memcpy(&u, &d, 8); return (int)(uint32_t)u; }
static inline int __double2hiint(double d) { uint64_t u; memcpy(&u, &d, 8); return (int)(uint32_t)(u >> 32); }
static inline double __hiloint2double(int hi, int lo) {
    const uint64_t u = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
    double d; memcpy(&d, &u, 8); return d;
}
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
#define __log2f(x) log2f(x)          // (glibc declares these names itself)
#define __expf(x) expf(x)
#define __logf(x) logf(x)
#define __exp2f(x) exp2f(x)
#define __fdividef(a, b) ((a) / (b))
#define __frcp_rn(x) (1.f / (x))
static inline float __builtin_amdgcn_rcpf(float x) { return 1.f / x; }
static inline float __builtin_amdgcn_rsqf(float x) { return 1.f / sqrtf(x); }
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }

// ---- synchronisation -------------------------------------------------------------------------------------------------------
static inline void __syncthreads() { lane_emu::block_barrier(); }
static inline void __builtin_amdgcn_wave_barrier() { lane_emu::wave_barrier(); }
static inline void __builtin_amdgcn_s_barrier() { lane_emu::block_barrier(); }
// (the product's LDS-only barrier is inline gfx950 assembly; medt_common.h only defines it when this is not defined)
#define MEDT_LDS_BARRIER() lane_emu::block_barrier()
#define MEDT_WAVE_LOCKSTEP() lane_emu::wave_barrier()

// ---- cross-lane operations (gfx950 semantics) --------------------------------------------------------------------------------
static inline int lane_emu_dpp_source(int lane, int ctrl, bool* valid) {
    const int row = lane & ~15, r = lane & 15;
    *valid = true;
    if (ctrl >= 0 && ctrl <= 0xFF) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);      // quad_perm
    if (ctrl >= 0x101 && ctrl <= 0x10F) { const int s = r + (ctrl & 15); *valid = s < 16; return row | (s & 15); }   // row_shl
    if (ctrl >= 0x111 && ctrl <= 0x11F) { const int s = r - (ctrl & 15); *valid = s >= 0; return row | (s & 15); }   // row_shr
    if (ctrl >= 0x121 && ctrl <= 0x12F) return row | ((r - (ctrl & 15)) & 15);                                     // row_ror
    if (ctrl == 0x138) { *valid = lane >= 1; return (lane - 1) & 63; }                                              // wave_shr:1
    if (ctrl == 0x140) return row | (15 - r);                                                                       // row_mirror
    if (ctrl == 0x141) return (lane & ~7) | (7 - (lane & 7));                                                       // row_half_mirror
    abort();
}
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    if (bank_mask != 0xf) abort();
    bool valid;
    const int lane = lane_emu::lane_id(), from = lane_emu_dpp_source(lane, ctrl, &valid);
    const int v = (int)(uint32_t)lane_emu::exchange((uint32_t)src, from);
    if (!((row_mask >> (lane >> 4)) & 1)) return old;             // rows outside row_mask keep the old value
    return valid ? v : (bound_ctrl ? 0 : old);
}
static inline int __builtin_amdgcn_readlane(int v, int lane) { return (int)(uint32_t)lane_emu::exchange((uint32_t)v, lane); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return (int)(uint32_t)lane_emu::exchange((uint32_t)v, 0); }
struct lane_emu_u2 {
    unsigned v[2];
    unsigned operator[](int i) const { return v[i]; }
};
// v_permlane16_swap: the odd rows of the first operand and the even rows of the second trade places
static inline lane_emu_u2 __builtin_amdgcn_permlane16_swap(unsigned a, unsigned b, bool, bool) {
    const int lane = lane_emu::lane_id(), row = lane >> 4;
    // new a: even rows keep a, odd rows receive b of the row below; new b: odd rows keep b, even rows receive a of the row above
    const uint64_t both = ((uint64_t)a << 32) | b;
    const uint64_t other = lane_emu::exchange(both, lane ^ 16);
    lane_emu_u2 r;
    r.v[0] = (row & 1) ? (unsigned)other : a;
    r.v[1] = (row & 1) ? b : (unsigned)(other >> 32);
    return r;
}
// v_permlane32_swap: the upper half of the first operand and the lower half of the second trade places
static inline lane_emu_u2 __builtin_amdgcn_permlane32_swap(unsigned a, unsigned b, bool, bool) {
    const int lane = lane_emu::lane_id();
    const uint64_t both = ((uint64_t)a << 32) | b;
    const uint64_t other = lane_emu::exchange(both, lane ^ 32);
    lane_emu_u2 r;
    r.v[0] = (lane & 32) ? (unsigned)other : a;
    r.v[1] = (lane & 32) ? b : (unsigned)(other >> 32);
    return r;
}
static inline float __shfl_xor(float v, int mask, int = 64) {
    return __uint_as_float((unsigned)lane_emu::exchange(__float_as_uint(v), lane_emu::lane_id() ^ mask));
}
static inline double __shfl_xor(double v, int mask, int = 64) {
    uint64_t u; memcpy(&u, &v, 8);
    u = lane_emu::exchange(u, lane_emu::lane_id() ^ mask);
    double d; memcpy(&d, &u, 8); return d;
}
static inline int __shfl_xor(int v, int mask, int = 64) { return (int)(uint32_t)lane_emu::exchange((uint32_t)v, lane_emu::lane_id() ^ mask); }
static inline int __shfl(int v, int src, int = 64) { return (int)(uint32_t)lane_emu::exchange((uint32_t)v, src); }
static inline double __shfl(double v, int src, int = 64) {
    uint64_t u; memcpy(&u, &v, 8);
    u = lane_emu::exchange(u, src);
    double d; memcpy(&d, &u, 8); return d;
}
// v_mfma_f32_16x16x4_f32: D (16 x 16) = A (16 x 4) B (4 x 16) + C;  lane l holds a = A[l % 16][l / 16], b = B[l / 16][l % 16],
// c / d[v] = C / D[4 (l / 16) + v][l % 16]
static inline medt_f4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, medt_f4 c, int, int, int) {
    const int lane = lane_emu::lane_id(), j = lane & 15, i0 = 4 * (lane >> 4);
    float A[4][4], B[4];                                              // A[v][k] = A[i0 + v][k], B[k] = B[k][j]
    for (int k = 0; k < 4; ++k) {
        B[k] = __uint_as_float((unsigned)lane_emu::exchange(__float_as_uint(b), k * 16 + j));
        for (int v = 0; v < 4; ++v) A[v][k] = __uint_as_float((unsigned)lane_emu::exchange(__float_as_uint(a), k * 16 + i0 + v));
    }
    medt_f4 d = c;
    for (int v = 0; v < 4; ++v)
        for (int k = 0; k < 4; ++k) d[v] = fmaf(A[v][k], B[k], d[v]);
    return d;
}
static inline float __shfl(float v, int src, int = 64) {
    return __uint_as_float((unsigned)lane_emu::exchange(__float_as_uint(v), src));
}
