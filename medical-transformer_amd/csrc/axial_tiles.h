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

// The loads of TILE_RB positions x TILE_CB channels are issued as one batch and only then written to LDS: with a wait
// (or a dtype branch) per element the staging of a tile is a chain of dependent global round trips, which is what the
// small problems of the model's layers spend their time on (a few workgroups, nothing else to hide the latency).
constexpr int TILE_RB = 4, TILE_CB = 8;

template <int AXIS>
__device__ __forceinline__ void tile_load(float* lds, int stride, int lch0, const float* __restrict__ src, int CH,
                                          int ch0, int nch, const TileCtx& t, int bf16 = 0) {
    const int per = t.nseq * t.L;
    const unsigned short* s16 = reinterpret_cast<const unsigned short*>(src);
    for (int r0 = threadIdx.x; r0 < per; r0 += MEDT_THREADS * TILE_RB) {
        size_t off[TILE_RB];
        float* dst[TILE_RB];
        bool ok[TILE_RB];
#pragma unroll
        for (int j = 0; j < TILE_RB; ++j) {
            const int r = r0 + j * MEDT_THREADS;
            ok[j] = r < per;
            int ls, i, n;
            size_t pix;
            tile_locate<AXIS>(t, ok[j] ? r : r0, ls, i, pix, n);
            off[j] = ((size_t)n * CH + ch0) * t.HW + pix;
            dst[j] = lds + ls * stride + lch0 * t.L + i;
        }
        for (int c0 = 0; c0 < nch; c0 += TILE_CB) {
            float v[TILE_RB][TILE_CB];
            if (bf16) {
#pragma unroll
                for (int j = 0; j < TILE_RB; ++j)
#pragma unroll
                    for (int k = 0; k < TILE_CB; ++k)
                        if (c0 + k < nch) v[j][k] = bf16_bits_to_f32(s16[off[j] + (size_t)(c0 + k) * t.HW]);
            } else {
#pragma unroll
                for (int j = 0; j < TILE_RB; ++j)
#pragma unroll
                    for (int k = 0; k < TILE_CB; ++k)
                        if (c0 + k < nch) v[j][k] = src[off[j] + (size_t)(c0 + k) * t.HW];
            }
            MEDT_SCHED_FENCE();
#pragma unroll
            for (int j = 0; j < TILE_RB; ++j)
#pragma unroll
                for (int k = 0; k < TILE_CB; ++k)
                    if (ok[j] && c0 + k < nch) dst[j][(c0 + k) * t.L] = v[j][k];
            MEDT_SCHED_FENCE();
        }
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

// pooled gradient: lds <- dy[n][ch0+ch][h/stride][w/stride] (0 outside the pooled extent); batched like tile_load
template <int AXIS>
__device__ __forceinline__ void tile_load_pooled(float* lds, int stride, int lch0, const float* __restrict__ dy, int C,
                                                 int ch0, int nch, int H, int pool, const TileCtx& t) {
    const int per = t.nseq * t.L;
    const int Ho = H / pool, Wo = t.W / pool;
    for (int r0 = threadIdx.x; r0 < per; r0 += MEDT_THREADS * TILE_RB) {
        size_t off[TILE_RB];
        float* dst[TILE_RB];
        bool ok[TILE_RB], in[TILE_RB];
#pragma unroll
        for (int j = 0; j < TILE_RB; ++j) {
            int r = r0 + j * MEDT_THREADS;
            ok[j] = r < per;
            if (!ok[j]) r = r0;
            int ls, i;
            if (AXIS == 1) { ls = r / t.L; i = r - ls * t.L; } else { i = r / t.nseq; ls = r - i * t.nseq; }
            const int b = t.seq0 + ls, n = b / t.Bo, s = b - n * t.Bo;
            const int h = AXIS == 1 ? s : i, w = AXIS == 1 ? i : s;
            const int ho = h / pool, wo = w / pool;
            in[j] = ho < Ho && wo < Wo;
            off[j] = ((size_t)(n * C + ch0) * Ho + min(ho, Ho - 1)) * Wo + min(wo, Wo - 1);     // clamped: loaded, then discarded
            dst[j] = lds + ls * stride + lch0 * t.L + i;
        }
        for (int c0 = 0; c0 < nch; c0 += TILE_CB) {
            float v[TILE_RB][TILE_CB];
#pragma unroll
            for (int j = 0; j < TILE_RB; ++j)
#pragma unroll
                for (int k = 0; k < TILE_CB; ++k)
                    if (c0 + k < nch) v[j][k] = dy[off[j] + (size_t)(c0 + k) * Ho * Wo];
            MEDT_SCHED_FENCE();
#pragma unroll
            for (int j = 0; j < TILE_RB; ++j)
#pragma unroll
                for (int k = 0; k < TILE_CB; ++k)
                    if (ok[j] && c0 + k < nch) dst[j][(c0 + k) * t.L] = in[j] ? v[j][k] : 0.f;
            MEDT_SCHED_FENCE();
        }
    }
}

__device__ __forceinline__ float gate(const float* p) { return p ? *p : 1.f; }
// gate of sequence `seq` (GatePtrs::stride == 0: the shared scalar)
__device__ __forceinline__ float gate_at(const float* p, int stride, int seq) { return p ? p[(size_t)stride * seq] : 1.f; }


}  // namespace medt
