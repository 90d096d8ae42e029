// axial_tiles.h -- tile movers shared by the attention kernels: (channel, sequence, position) <-> NCHW.
#pragma once
#include "medt_kernels.h"

namespace medt {

struct TileCtx {
    int L, Bo, W, HW, seq0, nseq;
};

// lds[ls*stride + (lch0+ch)*L + i] = src[n][ch0+ch][pixel(seq0+ls, i)]
// A thread owns one (sequence, position) of the tile and walks the channels: the (sequence -> image, offset) index
// arithmetic -- integer divisions by runtime extents -- is done once per thread, not once per element, and consecutive
// lanes still touch consecutive addresses of a channel plane.
template <int AXIS>
__device__ __forceinline__ bool tile_locate(const TileCtx& t, int r, int& ls, int& i, size_t& pix_off, int& n) {
    if (AXIS == 1) { ls = r / t.L; i = r - ls * t.L; } else { i = r / t.nseq; ls = r - i * t.nseq; }
    const int b = t.seq0 + ls;
    n = b / t.Bo;
    const int s = b - n * t.Bo;
    pix_off = (size_t)(AXIS == 1 ? s * t.W + i : i * t.W + s);
    return true;
}

template <int AXIS>
__device__ __forceinline__ void tile_load(float* lds, int stride, int lch0, const float* __restrict__ src, int CH,
                                          int ch0, int nch, const TileCtx& t, int bf16 = 0) {
    const int per = t.nseq * t.L;
    for (int r = threadIdx.x; r < per; r += MEDT_THREADS) {
        int ls, i, n;
        size_t pix;
        tile_locate<AXIS>(t, r, ls, i, pix, n);
        const size_t off = ((size_t)n * CH + ch0) * t.HW + pix;
        float* dst = lds + ls * stride + lch0 * t.L + i;
        for (int ch = 0; ch < nch; ++ch) dst[ch * t.L] = ld_act(src, off + (size_t)ch * t.HW, bf16);
    }
}

template <int AXIS>
__device__ __forceinline__ void tile_store(const float* lds, int stride, int lch0, float* __restrict__ dst, int CH,
                                           int ch0, int nch, const TileCtx& t, int bf16 = 0) {
    const int per = t.nseq * t.L;
    for (int r = threadIdx.x; r < per; r += MEDT_THREADS) {
        int ls, i, n;
        size_t pix;
        tile_locate<AXIS>(t, r, ls, i, pix, n);
        const size_t off = ((size_t)n * CH + ch0) * t.HW + pix;
        const float* s = lds + ls * stride + lch0 * t.L + i;
        for (int ch = 0; ch < nch; ++ch) st_act(dst, off + (size_t)ch * t.HW, s[ch * t.L], bf16);
    }
}

// pooled gradient: lds <- dy[n][ch0+ch][h/stride][w/stride] (0 outside the pooled extent)
template <int AXIS>
__device__ __forceinline__ void tile_load_pooled(float* lds, int stride, int lch0, const float* __restrict__ dy, int C,
                                                 int ch0, int nch, int H, int pool, const TileCtx& t) {
    const int per = t.nseq * t.L;
    const int Ho = H / pool, Wo = t.W / pool;
    for (int r = threadIdx.x; r < per; r += MEDT_THREADS) {
        int ls, i;
        if (AXIS == 1) { ls = r / t.L; i = r - ls * t.L; } else { i = r / t.nseq; ls = r - i * t.nseq; }
        const int b = t.seq0 + ls, n = b / t.Bo, s = b - n * t.Bo;
        const int h = AXIS == 1 ? s : i, w = AXIS == 1 ? i : s;
        const int ho = h / pool, wo = w / pool;
        const bool in = ho < Ho && wo < Wo;
        const size_t off = ((size_t)(n * C + ch0) * Ho + ho) * Wo + wo;
        float* dst = lds + ls * stride + lch0 * t.L + i;
        for (int ch = 0; ch < nch; ++ch) dst[ch * t.L] = in ? dy[off + (size_t)ch * Ho * Wo] : 0.f;
    }
}

__device__ __forceinline__ float gate(const float* p) { return p ? *p : 1.f; }
// gate of sequence `seq` (GatePtrs::stride == 0: the shared scalar)
__device__ __forceinline__ float gate_at(const float* p, int stride, int seq) { return p ? p[(size_t)stride * seq] : 1.f; }


}  // namespace medt
