// axial_tiles.h -- tile movers shared by the attention kernels: (channel, sequence, position) <-> NCHW.
#pragma once
#include "medt_kernels.h"

namespace medt {

struct TileCtx {
    int L, Bo, W, HW, seq0, nseq;
};

// lds[ls*stride + (lch0+ch)*L + i] = src[n][ch0+ch][pixel(seq0+ls, i)]
template <int AXIS>
__device__ __forceinline__ void tile_load(float* lds, int stride, int lch0, const float* __restrict__ src, int CH,
                                          int ch0, int nch, const TileCtx& t, int bf16 = 0) {
    const int per = t.nseq * t.L;
    for (int e = threadIdx.x; e < nch * per; e += MEDT_THREADS) {
        const int ch = e / per, r = e - ch * per;
        int ls, i;
        if (AXIS == 1) { ls = r / t.L; i = r - ls * t.L; } else { i = r / t.nseq; ls = r - i * t.nseq; }
        const int b = t.seq0 + ls, n = b / t.Bo, s = b - n * t.Bo;
        const size_t off = ((size_t)n * CH + ch0 + ch) * t.HW + (AXIS == 1 ? s * t.W + i : i * t.W + s);
        lds[ls * stride + (lch0 + ch) * t.L + i] = ld_act(src, off, bf16);
    }
}

template <int AXIS>
__device__ __forceinline__ void tile_store(const float* lds, int stride, int lch0, float* __restrict__ dst, int CH,
                                           int ch0, int nch, const TileCtx& t, int bf16 = 0) {
    const int per = t.nseq * t.L;
    for (int e = threadIdx.x; e < nch * per; e += MEDT_THREADS) {
        const int ch = e / per, r = e - ch * per;
        int ls, i;
        if (AXIS == 1) { ls = r / t.L; i = r - ls * t.L; } else { i = r / t.nseq; ls = r - i * t.nseq; }
        const int b = t.seq0 + ls, n = b / t.Bo, s = b - n * t.Bo;
        const size_t off = ((size_t)n * CH + ch0 + ch) * t.HW + (AXIS == 1 ? s * t.W + i : i * t.W + s);
        st_act(dst, off, lds[ls * stride + (lch0 + ch) * t.L + i], bf16);
    }
}

// pooled gradient: lds <- dy[n][ch0+ch][h/stride][w/stride] (0 outside the pooled extent)
template <int AXIS>
__device__ __forceinline__ void tile_load_pooled(float* lds, int stride, int lch0, const float* __restrict__ dy, int C,
                                                 int ch0, int nch, int H, int pool, const TileCtx& t) {
    const int per = t.nseq * t.L;
    const int Ho = H / pool, Wo = t.W / pool;
    for (int e = threadIdx.x; e < nch * per; e += MEDT_THREADS) {
        const int ch = e / per, r = e - ch * per;
        int ls, i;
        if (AXIS == 1) { ls = r / t.L; i = r - ls * t.L; } else { i = r / t.nseq; ls = r - i * t.nseq; }
        const int b = t.seq0 + ls, n = b / t.Bo, s = b - n * t.Bo;
        const int h = AXIS == 1 ? s : i, w = AXIS == 1 ? i : s;
        const int ho = h / pool, wo = w / pool;
        float v = 0.f;
        if (ho < Ho && wo < Wo) v = dy[((size_t)(n * C + ch0 + ch) * Ho + ho) * Wo + wo];
        lds[ls * stride + (lch0 + ch) * t.L + i] = v;
    }
}

__device__ __forceinline__ float gate(const float* p) { return p ? *p : 1.f; }
// gate of sequence `seq` (GatePtrs::stride == 0: the shared scalar)
__device__ __forceinline__ float gate_at(const float* p, int stride, int seq) { return p ? p[(size_t)stride * seq] : 1.f; }


}  // namespace medt
