// axial_fast.hip -- bandwidth-tuned forward kernels of the position-aware attention layers
// (AxialAttention / AxialAttention_dynamic, reference lib/models/axialnet.py:155-178) for L % 4 == 0.
//
// Same decomposition as axial_core.hip (one workgroup = S_T sequences of one head, one query row per
// lane) but the inner loop is organised around 16-byte LDS reads -- on gfx950 ds_read_b128 moves 256 B/clk
// per CU against 128 B/clk for ds_read_b32, and LDS, not HBM or VALU, is what bounds the row-per-lane form:
//   * keys/values: one broadcast ds_read_b128 brings 4 consecutive columns j0..j0+3 of a channel;
//   * relative tables: lane i needs entries [i-j0+L-4, i-j0+L-1] -- a 16-byte window whose alignment is
//     (i mod 4).  Each table row is therefore stored four times, shifted by 0..3 entries, and lane i reads
//     copy (i mod 4) at an aligned offset.  The copies are spaced so that the 16 lanes served together hit
//     16 distinct 16-byte slots (copy stride = 16 floats mod 64): conflict-free.
//   * softmax runs online over the 4-column chunks (running max, one rescale per chunk), so the logits are
//     computed once instead of twice.
//
// This file is compiled twice (medt_amd/build.py): MEDT_FAST_BF16=0 -> float32 storage of qkv_raw / stacked
// (axial_attn_fwd_fast), MEDT_FAST_BF16=1 -> bfloat16 storage (axial_attn_fwd_fast_bf16).  The storage type is a
// compile-time constant of the hot loops (kBF); the two sets of kernels live in distinct namespaces.
#include "axial_tiles.h"
#include "fin_inline.h"
#include <type_traits>

#ifndef MEDT_FAST_BF16
#define MEDT_FAST_BF16 0
#endif
#if MEDT_FAST_BF16
#define MEDT_FAST_NS fast_bf16
#define MEDT_FAST_FN axial_attn_fwd_fast_bf16
#else
#define MEDT_FAST_NS fast_f32
#define MEDT_FAST_FN axial_attn_fwd_fast
#endif

namespace medt {
namespace MEDT_FAST_NS {

constexpr bool kBF = MEDT_FAST_BF16 != 0;

__host__ __device__ static inline int copy_stride(int L) {
    const int two = 2 * L;
    return two + (((16 - two) % 64) + 64) % 64;         // == 16 (mod 64) floats, multiple of 4
}
__host__ __device__ static inline int region4(int nch, int L) { return (nch + 1) * L + 4; }

static size_t fast_lds_bytes(const AxialGeom& g) {
    return ((size_t)g.S_T * region4(2 * g.gp, g.L) + 256 + (size_t)2 * g.gp * 4 * copy_stride(g.L)) * sizeof(float);
}

// Stage the 2*GP table rows (tq | tk reversed | tv), four shifted copies each.  scale_k multiplies the tk rows.
template <int GP>
__device__ __forceinline__ void stage_tables4(float* tab, const float* __restrict__ relative, int L, float scale_k) {
    constexpr int HQ = GP / 2, NCH = 2 * GP;
    const int TL = 2 * L - 1, CS = copy_stride(L), per_row = 8 * L;
    for (int e = threadIdx.x; e < NCH * per_row; e += MEDT_THREADS) {
        const int row = e / per_row, rem = e - row * per_row;
        const int r = rem / (2 * L), x = rem - r * 2 * L;
        const int d = x + r;
        float v = 0.f;
        if (d < TL) {
            if (row < HQ) v = relative[row * TL + d];
            else if (row < GP) v = scale_k * relative[row * TL + (TL - 1 - d)];       // Rk, reversed: index j-i+L-1
            else v = relative[row * TL + d];
        }
        tab[(row * 4 + r) * CS + x] = v;
    }
}

// --------------------------------------------------------------------------- //
// forward main pass
// --------------------------------------------------------------------------- //
template <int GP, int AXIS>
__global__ __launch_bounds__(MEDT_THREADS) void attn_fwd4_kernel(AxialGeom g, const float* __restrict__ qkv_raw,
                                                                 BnStats qs, BnStats ss,
                                                                 const float* __restrict__ relative, GatePtrs gates,
                                                                 float* __restrict__ stacked,
                                                                 float* __restrict__ lse_out,
                                                                 float* __restrict__ out_partials) {
    constexpr int HQ = GP / 2, NCH = 2 * GP, OCG = 2 * GP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = g.L, RS = region4(NCH, L), CS = copy_stride(L);
    float* reg = smem;
    float* red = reg + g.S_T * RS;
    float* tab = red + 256;
    const int grp = blockIdx.x / g.tpg, tile = blockIdx.x - grp * g.tpg, hg = blockIdx.y;
    TileCtx t{L, g.Bo, g.W, g.HW, grp * g.spg + tile * g.S_T, min(g.S_T, g.spg - tile * g.S_T)};
    tile_load<AXIS>(reg, RS, 0, qkv_raw, 2 * g.C, hg * NCH, NCH, t, kBF ? 1 : 0);
    const float f_qr = gate(gates.f_qr), f_kr = gate(gates.f_kr), f_sve = gate(gates.f_sve), f_sv = gate(gates.f_sv);
    const float a_qk = ss.scale[grp * g.SC + hg] * MEDT_LOG2E;
    const float a_qr = ss.scale[grp * g.SC + g.G + hg] * f_qr * MEDT_LOG2E;
    const float a_kr = ss.scale[grp * g.SC + 2 * g.G + hg] * f_kr * MEDT_LOG2E;
    stage_tables4<GP>(tab, relative, L, a_kr);
    __syncthreads();
    const int ls = threadIdx.x / L, i = threadIdx.x - ls * L;
    const bool active = ls < t.nseq;
    const float* sc = qs.scale + grp * 2 * g.C + hg * NCH;
    const float* sh = qs.shift + grp * 2 * g.C + hg * NCH;
    if (active) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int idx = ls * RS + ch * L + i;
            reg[idx] = fmaf(reg[idx], sc[ch], sh[ch]);
        }
    }
    __syncthreads();
    float outv[OCG];
    float lse = 0.f;
#pragma unroll
    for (int k = 0; k < OCG; ++k) outv[k] = 0.f;
    if (active) {
        float qa[HQ], qb[HQ];
#pragma unroll
        for (int c = 0; c < HQ; ++c) {
            const float q = reg[ls * RS + c * L + i];
            qa[c] = q * a_qk;
            qb[c] = q * a_qr;
        }
        const float* kp = reg + ls * RS + HQ * L;
        const float* vp = reg + ls * RS + GP * L;
        const int r = i & 3;
        const float* tabr = tab + r * CS;
        float m = -INFINITY, l = 0.f, accv[GP], acce[GP];
#pragma unroll
        for (int c = 0; c < GP; ++c) { accv[c] = 0.f; acce[c] = 0.f; }
        for (int j0 = 0; j0 < L; j0 += 4) {
            const int xoff = i - j0 + L - 4 - r;                 // aligned start of the window [D-3, D]
            float z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f;
#pragma unroll
            for (int c = 0; c < HQ; ++c) {
                const float4 k4 = *reinterpret_cast<const float4*>(kp + c * L + j0);
                const float4 q4 = *reinterpret_cast<const float4*>(tabr + (c * 4) * CS + xoff);
                const float4 t4 = *reinterpret_cast<const float4*>(tabr + ((HQ + c) * 4) * CS + xoff);
                z0 = fmaf(qa[c], k4.x, fmaf(qb[c], q4.w, fmaf(k4.x, t4.w, z0)));     // column j0   <-> entry D
                z1 = fmaf(qa[c], k4.y, fmaf(qb[c], q4.z, fmaf(k4.y, t4.z, z1)));     // column j0+1 <-> entry D-1
                z2 = fmaf(qa[c], k4.z, fmaf(qb[c], q4.y, fmaf(k4.z, t4.y, z2)));
                z3 = fmaf(qa[c], k4.w, fmaf(qb[c], q4.x, fmaf(k4.w, t4.x, z3)));
            }
            const float mn = fmaxf(m, fmaxf(fmaxf(z0, z1), fmaxf(z2, z3)));
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            m = mn;
            const float p0 = __builtin_amdgcn_exp2f(z0 - m), p1 = __builtin_amdgcn_exp2f(z1 - m);
            const float p2 = __builtin_amdgcn_exp2f(z2 - m), p3 = __builtin_amdgcn_exp2f(z3 - m);
            l = fmaf(l, alpha, (p0 + p1) + (p2 + p3));
#pragma unroll
            for (int c = 0; c < GP; ++c) {
                const float4 v4 = *reinterpret_cast<const float4*>(vp + c * L + j0);
                const float4 e4 = *reinterpret_cast<const float4*>(tabr + ((GP + c) * 4) * CS + xoff);
                accv[c] = fmaf(p0, v4.x, fmaf(p1, v4.y, fmaf(p2, v4.z, fmaf(p3, v4.w, accv[c] * alpha))));
                acce[c] = fmaf(p0, e4.w, fmaf(p1, e4.z, fmaf(p2, e4.y, fmaf(p3, e4.x, acce[c] * alpha))));
            }
        }
        const float inv = 1.f / l;
        lse = m + __log2f(l);
#pragma unroll
        for (int c = 0; c < GP; ++c) {
            outv[2 * c] = f_sv * accv[c] * inv;
            outv[2 * c + 1] = f_sve * acce[c] * inv;
        }
    }
    __syncthreads();                                   // all reads of reg done
    if (active) {
#pragma unroll
        for (int k = 0; k < OCG; ++k) reg[ls * RS + k * L + i] = outv[k];
        reg[ls * RS + NCH * L + i] = lse;
    }
    __syncthreads();
    tile_store<AXIS>(reg, RS, 0, stacked, g.OC, hg * OCG, OCG, t, kBF ? 1 : 0);
    if (lse_out) tile_store<AXIS>(reg, RS, NCH, lse_out, g.G, hg, 1, t);
    if (out_partials) {
        float v[2 * OCG];
#pragma unroll
        for (int k = 0; k < OCG; ++k) { v[2 * k] = outv[k]; v[2 * k + 1] = outv[k] * outv[k]; }
        block_sum_d<2 * OCG>(v, red, reinterpret_cast<double*>(out_partials) + ((size_t)blockIdx.x * g.OC + hg * OCG) * 2);
    }
}

// --------------------------------------------------------------------------- //
// Compile-time-L variants (L in {16,32,64,128}: every layer of the 128- and 256-pixel models).
// On top of the above: (1) the workgroup is persistent over "super-tiles" of SS = NT*S_T adjacent sequences,
// so the relative tables are staged once per workgroup instead of once per S_T sequences and the NCHW rows
// fetched for the height axis are SS*4 bytes long instead of S_T*4; (2) bn_qkv's affine is applied while
// staging; (3) all LDS addresses are base + immediate (RS, CS, L are constants), the j-loop is unrolled.
// --------------------------------------------------------------------------- //
typedef medt_f2 f2;
typedef medt_f4 f4;

// Column-ordered tables for the compile-time-L kernels: U[y] = T[TL-1-y] with T the table indexed by
// d = i-j+L-1, so the four entries a lane needs for columns j0..j0+3 sit at ascending addresses
// y = (L-1-i) + j0 + b and line up component-wise with the 4 keys of a ds_read_b128 (packed-FMA friendly).
// Four shifted copies per row as above: copy r holds U[x + r].
template <int GP>
__device__ __forceinline__ void stage_tables_cols(float* tab, const float* __restrict__ relative, int L, float scale_k) {
    constexpr int HQ = GP / 2, NCH = 2 * GP;
    const int TL = 2 * L - 1, CS = copy_stride(L), per_row = 8 * L;
    for (int e = threadIdx.x; e < NCH * per_row; e += MEDT_THREADS) {
        const int row = e / per_row, rem = e - row * per_row;
        const int r = rem / (2 * L), x = rem - r * 2 * L;
        const int y = x + r;
        float v = 0.f;
        if (y < TL) {
            if (row < HQ) v = relative[row * TL + (TL - 1 - y)];                 // Rq[c][i-j+L-1]
            else if (row < GP) v = scale_k * relative[row * TL + y];            // Rk[c][j-i+L-1]
            else v = relative[row * TL + (TL - 1 - y)];                         // Rv[c][i-j+L-1]
        }
        tab[(row * 4 + r) * CS + x] = v;
    }
}

template <int GP, int L>
struct Fast3 {
    static constexpr int HQ = GP / 2, NCH = 2 * GP, OCG = 2 * GP, LEN = L;
    static constexpr int S_T = MEDT_THREADS / L;              // sequences per sub-tile (one row per lane)
    static constexpr int EPT = 1;                             // tile elements per thread and sub-tile (S_T*L/256)
    static constexpr int RS = (NCH + 1) * L + 4;
    static constexpr int NT0 = 8192 / (S_T * RS);
    static constexpr int NT = NT0 < 1 ? 1 : (NT0 > 8 ? 8 : NT0);
    static constexpr int SS = S_T * NT;
    static constexpr int CS = 2 * L + (((16 - 2 * L) % 64) + 64) % 64;
    static constexpr size_t lds_floats = (size_t)SS * RS + 256 + (size_t)NCH * 4 * CS;
    // sub-tiles a kernel variant may be asked to run: the width axis is coalesced at any tile size and keeps its
    // prefetch registers small; the height axis wants many adjacent sequences per tile (contiguous runs of SS*4 B)
    static constexpr int nta(int axis) { return axis == 1 ? (NT < 2 ? NT : 2) : NT; }
};

}  // namespace MEDT_FAST_NS
#if !MEDT_FAST_BF16
int fast3_max_subtiles(int gp, int L, int axis) {     // host mirror of Fast3<GP,L>::nta(axis) (0 = no compile-time variant)
    if (L != 16 && L != 32 && L != 64 && L != 128) return 0;
    const int S_T = MEDT_THREADS / L, RS = (2 * gp + 1) * L + 4;
    int NT = 8192 / (S_T * RS);
    NT = NT < 1 ? 1 : (NT > 8 ? 8 : NT);
    return axis == 1 && NT > 2 ? 2 : NT;
}
#endif
namespace MEDT_FAST_NS {

// Global accesses of the fast kernels go through a wave-uniform 64-bit base plus a 32-bit per-lane byte offset:
// the compiler then uses the saddr form of global_load/global_store and the phases around the sweep spend no VALU
// issue slots on 64-bit address arithmetic (they used to cost ~40 % of the kernel's VALU instructions).
__device__ __forceinline__ float ldg_u(const float* __restrict__ ubase, unsigned byteoff) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ubase) + byteoff);
}
__device__ __forceinline__ void stg_u(float* __restrict__ ubase, unsigned byteoff, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(ubase) + byteoff) = v;
}
// Activation-storage variants (qkv_raw / stacked): element offsets; BF = stored as bfloat16
template <bool BF>
__device__ __forceinline__ const float* act_base(const float* p, size_t elems) {
    return reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + elems * (BF ? 2 : 4));
}
template <bool BF>
__device__ __forceinline__ float* act_base(float* p, size_t elems) {
    return reinterpret_cast<float*>(reinterpret_cast<char*>(p) + elems * (BF ? 2 : 4));
}
// byte offsets (ES = element size of the storage type), as in ldg_u / stg_u
template <bool BF>
__device__ __forceinline__ float lda_u(const float* __restrict__ ubase, unsigned byteoff) {
    if (BF) return bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(ubase) + byteoff));
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ubase) + byteoff);
}
template <bool BF>
__device__ __forceinline__ void sta_u(float* __restrict__ ubase, unsigned byteoff, float v) {
    if (BF) *reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(ubase) + byteoff) = f32_to_bf16_bits(v);
    else *reinterpret_cast<float*>(reinterpret_cast<char*>(ubase) + byteoff) = v;
}

// Four consecutive elements at once (16 bytes of float32, 8 of bfloat16): byte offset a multiple of the access size
template <bool BF>
__device__ __forceinline__ f4 lda4_u(const float* __restrict__ ubase, unsigned byteoff) {
    if (BF) {
        const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(ubase) + byteoff);
        f4 r;
        r.x = __uint_as_float(w.x << 16); r.y = __uint_as_float(w.x & 0xffff0000u);
        r.z = __uint_as_float(w.y << 16); r.w = __uint_as_float(w.y & 0xffff0000u);
        return r;
    }
    return *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(ubase) + byteoff);
}
template <bool BF>
__device__ __forceinline__ void sta4_u(float* __restrict__ ubase, unsigned byteoff, f4 v) {
    if (BF) {
        uint2 w;
        w.x = (unsigned)f32_to_bf16_bits(v.x) | ((unsigned)f32_to_bf16_bits(v.y) << 16);
        w.y = (unsigned)f32_to_bf16_bits(v.z) | ((unsigned)f32_to_bf16_bits(v.w) << 16);
        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(ubase) + byteoff) = w;
    } else {
        *reinterpret_cast<f4*>(reinterpret_cast<char*>(ubase) + byteoff) = v;
    }
}

// Where the elements of one super-tile live.  (n0, s0, nseq) are wave-uniform (first image, first sequence inside
// it, sequences in the tile); element t of a thread is e = threadIdx.x + 256 t, laid out with lanes along the
// contiguous NCHW direction (width axis: along the sequence; height axis: across the sequences of the tile).
// Small-integer divisions use (x + 0.5) * (1/d): exact while x < 2^20 (x is at most a few thousand here).
template <class F, int AXIS>
struct SuperMap {
    static constexpr int L = F::LEN;
    int n0, s0, nseq;
    float inv_bo, inv_nseq;

    __device__ __forceinline__ void set(const AxialGeom& g, int n0_, int s0_, int nseq_) {
        n0 = n0_; s0 = s0_; nseq = nseq_;
        inv_bo = 1.f / (float)g.Bo;
        inv_nseq = 1.f / (float)nseq_;
    }
    // sequence `ls` of the tile -> image offset from n0 and index inside that image
    __device__ __forceinline__ void image_of(const AxialGeom& g, int ls, int& dn, int& sq) const {
        const int s = s0 + ls;
        dn = (int)(((float)s + 0.5f) * inv_bo);
        sq = s - dn * g.Bo;
    }
    __device__ __forceinline__ bool locate(const AxialGeom& g, int t, int& lo, int& dn, int& pix) const {
        const int e = threadIdx.x + t * MEDT_THREADS;
        if (e >= nseq * L) return false;
        int ls, i;
        if (AXIS == 1) { ls = e / L; i = e % L; }
        else if (nseq == F::S_T) { i = e / F::S_T; ls = e % F::S_T; }
        else { i = (int)(((float)e + 0.5f) * inv_nseq); ls = e - i * nseq; }
        int sq;
        image_of(g, ls, dn, sq);
        pix = AXIS == 1 ? sq * g.W + i : i * g.W + sq;
        lo = ls * F::RS + i;
        return true;
    }
};

// Register prefetch of a super-tile: the raw global loads of tile u+1 are issued before the arithmetic of tile u and
// only land in LDS (normalised by bn_qkv) after it, so the HBM latency of one tile hides under the sweep of the
// previous one.  NCHL = channels fetched (q,k,v for the main pass; q,k for the statistics pass).
template <class F, int AXIS, int NCHL>
struct SuperPrefetch {
    static constexpr int L = F::LEN;
    using Map = SuperMap<F, AXIS>;
    static constexpr int NTA = F::nta(AXIS) * F::EPT;
    static_assert(NTA * NCHL <= 32, "prefetch registers");
    float v[NTA * NCHL];

    template <bool BF>
    __device__ __forceinline__ void issue_t(const float* __restrict__ qkv_raw, const AxialGeom& g, int hg, const Map& m) {
        const float* base = act_base<BF>(qkv_raw, ((size_t)m.n0 * 2 * g.C + hg * F::NCH) * g.HW);      // uniform
        const int img = 2 * g.C * g.HW;
#pragma unroll
        for (int t = 0; t < NTA; ++t) {
            int lo, dn, pix;
            if (m.locate(g, t, lo, dn, pix)) {
                constexpr unsigned ES = BF ? 2u : 4u;
                const unsigned off = (unsigned)(dn * img + pix) * ES;
#pragma unroll
                for (int ch = 0; ch < NCHL; ++ch) v[t * NCHL + ch] = lda_u<BF>(base, off + (unsigned)(ch * g.HW) * ES);
            }
        }
    }
    __device__ __forceinline__ void issue(const float* __restrict__ qkv_raw, const AxialGeom& g, int hg, const Map& m) {
        issue_t<kBF>(qkv_raw, g, hg, m);
    }
    __device__ __forceinline__ void commit(float* reg, const AxialGeom& g, const Map& m, const float* __restrict__ sc,
                                           const float* __restrict__ sh) const {
#pragma unroll
        for (int t = 0; t < NTA; ++t) {
            int lo, dn, pix;
            if (m.locate(g, t, lo, dn, pix)) {
#pragma unroll
                for (int ch = 0; ch < NCHL; ++ch) reg[lo + ch * L] = fmaf(v[t * NCHL + ch], sc[ch], sh[ch]);
            }
        }
    }
};

// LDS -> global, `nch` channels starting at LDS channel lch0 (height axis: re-maps rows to lanes for coalescing).
template <class F, int AXIS, bool BF>
__device__ __forceinline__ void store_super_t(const float* reg, int lch0, float* __restrict__ dst, int CH, int ch0, int nch,
                                              const AxialGeom& g, const SuperMap<F, AXIS>& m) {
    constexpr int L = F::LEN;
    float* base = act_base<BF>(dst, ((size_t)m.n0 * CH + ch0) * g.HW);                      // uniform
    const int img = CH * g.HW;
#pragma unroll
    for (int t = 0; t < F::nta(AXIS) * F::EPT; ++t) {
        int lo, dn, pix;
        if (m.locate(g, t, lo, dn, pix)) {
            constexpr unsigned ES = BF ? 2u : 4u;
            const unsigned off = (unsigned)(dn * img + pix) * ES;
            const float* src = reg + lo + lch0 * L;
            for (int ch = 0; ch < nch; ++ch) sta_u<BF>(base, off + (unsigned)(ch * g.HW) * ES, src[ch * L]);
        }
    }
}
// bf16 != 0: dst is stored as bfloat16 (only ever the `stacked` tensor; lse and the rest stay float32)
template <class F, int AXIS>
__device__ __forceinline__ void store_super(const float* reg, int lch0, float* __restrict__ dst, int CH, int ch0, int nch,
                                            const AxialGeom& g, const SuperMap<F, AXIS>& m, int bf16 = 0) {
    if (kBF && bf16) store_super_t<F, AXIS, true>(reg, lch0, dst, CH, ch0, nch, g, m);
    else store_super_t<F, AXIS, false>(reg, lch0, dst, CH, ch0, nch, g, m);
}

// EXACT = true : online softmax (running max, one rescale per 4-column chunk); if `flag` is given the kernel only
//                runs when *flag != 0 (it is then the repair pass behind the bound-referenced variant).
// EXACT = false: softmax referenced to a cheap per-row upper bound of the logits (no running max, no rescale:
//                27 instead of 38 issues per 4 columns).  Softmax is shift invariant, so any reference >= max_j z is
//                exact; the only failure mode is a bound so loose that every 2^(z-bound) underflows, which is
//                detected (l == 0) and reported through *flag for the EXACT kernel launched right behind.
template <int GP, int AXIS, int L, bool EXACT>
__global__ __launch_bounds__(MEDT_THREADS) void attn_fwd3_kernel(AxialGeom g, const float* __restrict__ qkv_raw,
                                                                 BnStats qs, BnStats ss,
                                                                 const float* __restrict__ relative, GatePtrs gates,
                                                                 float* __restrict__ stacked,
                                                                 float* __restrict__ lse_out,
                                                                 float* __restrict__ out_partials,
                                                                 unsigned* __restrict__ flag, FinSrc simsrc) {
    using F = Fast3<GP, L>;
    constexpr int HQ = F::HQ, NCH = F::NCH, OCG = F::OCG, RS = F::RS, CS = F::CS, S_T = F::S_T;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* reg = smem;
    float* red = reg + S_T * g.nt * RS;                       // LDS is sized for the runtime super-tile (occupancy)
    float* tab = red + 256;
    if (EXACT && flag && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;   // nothing to repair
    const int grp = blockIdx.x / g.fparts, part = blockIdx.x - grp * g.fparts, hg = blockIdx.y;
    const float f_qr = gate(gates.f_qr), f_kr = gate(gates.f_kr), f_sve = gate(gates.f_sve), f_sv = gate(gates.f_sv);
    float s_qk, s_qr, s_kr;
    if (simsrc.on) {
        // bn_similarity finalised HERE from the statistics kernel's partial rows (fin_inline.h: no bn_finalize launch in front of
        // this kernel); the first workgroup of each head also writes what the backward pass and the next step read
        // (lanes 0 / 1 / 2 of every wave take the qk / qr / kr channel of the head: one run of the double arithmetic, then a broadcast)
        const int l3 = min((int)(threadIdx.x & 63), 2), chl = l3 * g.G + hg;
        const FinVals v = fin_channel_lane(simsrc, chl);
        s_qk = fin_bcast(v.scale, 0); s_qr = fin_bcast(v.scale, 1); s_kr = fin_bcast(v.scale, 2);
        if (blockIdx.x == 0 && threadIdx.x < 3) fin_save(simsrc, chl, v);
    } else {
        s_qk = ss.scale[grp * g.SC + hg];
        s_qr = ss.scale[grp * g.SC + g.G + hg];
        s_kr = ss.scale[grp * g.SC + 2 * g.G + hg];
    }
    const float a_qk = s_qk * MEDT_LOG2E;
    const float a_qr = s_qr * f_qr * MEDT_LOG2E;
    const float a_kr = s_kr * f_kr * MEDT_LOG2E;
    stage_tables_cols<GP>(tab, relative, L, a_kr);
    if (threadIdx.x < NCH) {                                  // bn_qkv's affine for this head group: LDS-resident
        red[128 + threadIdx.x] = qs.scale[grp * 2 * g.C + hg * NCH + threadIdx.x];
        red[160 + threadIdx.x] = qs.shift[grp * 2 * g.C + hg * NCH + threadIdx.x];
    }
    __syncthreads();
    // max |entry| of the Rq rows and of the (scaled) Rk rows: ingredients of the per-row logit bound below
    float tqmax[HQ], tkmax[HQ];
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int c = 0; c < HQ; ++c) {
            float a = 0.f, b = 0.f;
            for (int x = threadIdx.x; x < 2 * L - 1; x += MEDT_THREADS) {
                a = fmaxf(a, fabsf(tab[(c * 4) * CS + x]));
                b = fmaxf(b, fabsf(tab[((HQ + c) * 4) * CS + x]));
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { a = fmaxf(a, __shfl_xor(a, o, 64)); b = fmaxf(b, __shfl_xor(b, o, 64)); }
            if (lane == 0) { red[wave * 2 * HQ + 2 * c] = a; red[wave * 2 * HQ + 2 * c + 1] = b; }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < HQ; ++c) {
            tqmax[c] = fmaxf(fmaxf(red[2 * c], red[2 * HQ + 2 * c]), fmaxf(red[4 * HQ + 2 * c], red[6 * HQ + 2 * c]));
            tkmax[c] = fmaxf(fmaxf(red[2 * c + 1], red[2 * HQ + 2 * c + 1]),
                             fmaxf(red[4 * HQ + 2 * c + 1], red[6 * HQ + 2 * c + 1]));
        }
    }
    const float* sc = red + 128;
    const float* sh = red + 160;
    const int lsub = threadIdx.x / L, i = threadIdx.x % L;
    const int r = (3 - i) & 3;
    const float* tabr = tab + r * CS + (L - 1 - i - r);       // + j0 per chunk (immediate offset)
    float st_sum[OCG], st_sq[OCG];
#pragma unroll
    for (int k = 0; k < OCG; ++k) { st_sum[k] = 0.f; st_sq[k] = 0.f; }
    int bad = 0;                                              // any row whose bound-referenced sum underflowed
    const int SSr = S_T * g.nt;                               // runtime super-tile (g.nt <= NT sub-tiles)
    const int nsup = (g.spg + SSr - 1) / SSr;
    using Map = SuperMap<F, AXIS>;
    using PF = SuperPrefetch<F, AXIS, NCH>;
    // tiles of this workgroup: u = part, part + fparts, ...; (image, sequence-in-image) advance by a fixed step
    const unsigned step = (unsigned)g.fparts * SSr;
    const int dn_step = step / (unsigned)g.Bo, ds_step = step - dn_step * g.Bo;
    Map cur, nxt;
    {
        const unsigned q0 = (unsigned)part * SSr;
        const int dn = q0 / (unsigned)g.Bo;
        nxt.set(g, grp * g.npg + dn, q0 - dn * g.Bo, min(SSr, g.spg - (int)q0));
    }
    PF pf;
    if (part < nsup) pf.issue(qkv_raw, g, hg, nxt);
    for (int u = part; u < nsup; u += g.fparts) {
        cur = nxt;
        const int nseq = cur.nseq;
        __syncthreads();                                       // previous super-tile fully consumed / tables staged
        pf.commit(reg, g, cur, sc, sh);
        {
            const int un = u + g.fparts;
            int s0 = cur.s0 + ds_step, n0 = cur.n0 + dn_step;
            if (s0 >= g.Bo) { s0 -= g.Bo; ++n0; }
            if (un < nsup) {
                nxt.set(g, n0, s0, min(SSr, g.spg - un * SSr));
                pf.issue(qkv_raw, g, hg, nxt);                 // in flight during the sweep below
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int sub = 0; sub < g.nt; ++sub) {
            const int ls = sub * S_T + lsub;
            const bool active = ls < nseq;
            float outv[OCG], lse = 0.f;
#pragma unroll
            for (int k = 0; k < OCG; ++k) outv[k] = 0.f;
            // max_j |k[c][j]| of this lane's sequence (lanes of one sequence exchange through shuffles; L = 128 spans
            // two waves and goes through LDS)
            float kmax[HQ];
#pragma unroll
            for (int c = 0; c < HQ; ++c) {
                float a = active ? fabsf(reg[ls * RS + (HQ + c) * L + i]) : 0.f;
#pragma unroll
                for (int o = (L < 64 ? L : 64) / 2; o > 0; o >>= 1) a = fmaxf(a, __shfl_xor(a, o, 64));
                kmax[c] = a;
            }
            if (L > 64) {
                __syncthreads();
                if ((threadIdx.x & 63) == 0)
#pragma unroll
                    for (int c = 0; c < HQ; ++c) red[64 + (threadIdx.x >> 6) * HQ + c] = kmax[c];
                __syncthreads();
#pragma unroll
                for (int c = 0; c < HQ; ++c)
                    kmax[c] = fmaxf(red[64 + ((threadIdx.x >> 6) & ~1) * HQ + c], red[64 + ((threadIdx.x >> 6) | 1) * HQ + c]);
            }
            if (active) {
                f2 qa[HQ], qb[HQ];
                float mb = g.bound_shift;            // upper bound of this row's logits (log2 domain); 0 + test hook
#pragma unroll
                for (int c = 0; c < HQ; ++c) {
                    const float q = reg[ls * RS + c * L + i];
                    qa[c] = (f2)(q * a_qk);
                    qb[c] = (f2)(q * a_qr);
                    mb += fabsf(q * a_qk) * kmax[c] + fabsf(q * a_qr) * tqmax[c] + kmax[c] * tkmax[c];
                }
                const float* kp = reg + ls * RS + HQ * L;
                const float* vp = reg + ls * RS + GP * L;
                float m = mb;
                f2 l2 = (f2)(0.f), accv[GP], acce[GP];         // two partial sums each (even / odd columns)
#pragma unroll
                for (int c = 0; c < GP; ++c) { accv[c] = (f2)(0.f); acce[c] = (f2)(0.f); }
                // Fast path: softmax is shift invariant, so any m >= max_j z works as the reference point.  The bound
                // mb costs nothing per pair (no running max, no rescale: 27 instead of 38 issues per 4 columns); all
                // it can do wrong is sit so far above the true maximum that every 2^(z-mb) underflows -- detected
                // below (l == 0), in which case the wave redoes its rows with the exact online softmax.
                auto chunk = [&](const f4 (&k4)[HQ], const f4 (&q4)[HQ], const f4 (&t4)[HQ], const f4 (&v4)[GP],
                                 const f4 (&e4)[GP]) {
                    f2 zlo = (f2)(-mb), zhi = (f2)(-mb);
#pragma unroll
                    for (int c = 0; c < HQ; ++c) {
                        zlo = qa[c] * k4[c].lo + (qb[c] * q4[c].lo + (k4[c].lo * t4[c].lo + zlo));
                        zhi = qa[c] * k4[c].hi + (qb[c] * q4[c].hi + (k4[c].hi * t4[c].hi + zhi));
                    }
                    f2 plo, phi;
                    plo.x = __builtin_amdgcn_exp2f(zlo.x); plo.y = __builtin_amdgcn_exp2f(zlo.y);
                    phi.x = __builtin_amdgcn_exp2f(zhi.x); phi.y = __builtin_amdgcn_exp2f(zhi.y);
                    l2 += plo + phi;
#pragma unroll
                    for (int c = 0; c < GP; ++c) {
                        accv[c] = plo * v4[c].lo + (phi * v4[c].hi + accv[c]);
                        acce[c] = plo * e4[c].lo + (phi * e4[c].hi + acce[c]);
                    }
                };
                // Exact path (rare): running max with one rescale per chunk.
                auto chunk_exact = [&](const f4 (&k4)[HQ], const f4 (&q4)[HQ], const f4 (&t4)[HQ], const f4 (&v4)[GP],
                                       const f4 (&e4)[GP]) {
                    f2 zlo = (f2)(0.f), zhi = (f2)(0.f);
#pragma unroll
                    for (int c = 0; c < HQ; ++c) {
                        zlo = qa[c] * k4[c].lo + (qb[c] * q4[c].lo + (k4[c].lo * t4[c].lo + zlo));
                        zhi = qa[c] * k4[c].hi + (qb[c] * q4[c].hi + (k4[c].hi * t4[c].hi + zhi));
                    }
                    const float mn = fmaxf(m, fmaxf(fmaxf(zlo.x, zlo.y), fmaxf(zhi.x, zhi.y)));
                    const f2 alpha = (f2)(__builtin_amdgcn_exp2f(m - mn));
                    m = mn;
                    zlo -= (f2)(m);
                    zhi -= (f2)(m);
                    f2 plo, phi;
                    plo.x = __builtin_amdgcn_exp2f(zlo.x); plo.y = __builtin_amdgcn_exp2f(zlo.y);
                    phi.x = __builtin_amdgcn_exp2f(zhi.x); phi.y = __builtin_amdgcn_exp2f(zhi.y);
                    l2 = l2 * alpha + (plo + phi);
#pragma unroll
                    for (int c = 0; c < GP; ++c) {
                        accv[c] = plo * v4[c].lo + (phi * v4[c].hi + accv[c] * alpha);
                        acce[c] = plo * e4[c].lo + (phi * e4[c].hi + acce[c] * alpha);
                    }
                };
                auto fetch = [&](int j0, f4 (&k4)[HQ], f4 (&q4)[HQ], f4 (&t4)[HQ], f4 (&v4)[GP], f4 (&e4)[GP]) {
#pragma unroll
                    for (int c = 0; c < HQ; ++c) {
                        k4[c] = *reinterpret_cast<const f4*>(kp + c * L + j0);
                        q4[c] = *reinterpret_cast<const f4*>(tabr + (c * 4) * CS + j0);
                        t4[c] = *reinterpret_cast<const f4*>(tabr + ((HQ + c) * 4) * CS + j0);
                    }
#pragma unroll
                    for (int c = 0; c < GP; ++c) {
                        v4[c] = *reinterpret_cast<const f4*>(vp + c * L + j0);
                        e4[c] = *reinterpret_cast<const f4*>(tabr + ((GP + c) * 4) * CS + j0);
                    }
                };
                // Unrolled over the row; sched_barrier keeps the chunks in program order (without it the scheduler
                // hoists every ds_read of the row and the register count explodes).  Latency is hidden by occupancy
                // (>= 4 waves per SIMD), not by software pipelining: both were measured, this needs fewer registers.
                auto sweep = [&](auto&& body) {
#pragma unroll
                    for (int j0 = 0; j0 < L; j0 += 4) {
                        f4 k4[HQ], q4[HQ], t4[HQ], v4[GP], e4[GP];
                        fetch(j0, k4, q4, t4, v4, e4);
                        body(k4, q4, t4, v4, e4);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                if constexpr (EXACT) {
                    m = -INFINITY;
                    sweep(chunk_exact);
                } else {
                    sweep(chunk);
                }
                const float l = l2.x + l2.y;
                if (!EXACT) bad |= !(l > 1e-30f);                // bound too loose for this row: request repair (below)
                const float inv = __builtin_amdgcn_rcpf(l);
                lse = m + __log2f(l);
#pragma unroll
                for (int c = 0; c < GP; ++c) {
                    outv[2 * c] = f_sv * (accv[c].x + accv[c].y) * inv;
                    outv[2 * c + 1] = f_sve * (acce[c].x + acce[c].y) * inv;
                }
            }
            if constexpr (AXIS == 1) {
                // width axis: the compute lanes already run along the contiguous direction -> store from registers
                if (active) {
                    int dn, sq;
                    cur.image_of(g, ls, dn, sq);
                    const int pix = sq * g.W + i;
                    constexpr unsigned ES = kBF ? 2u : 4u;
                    const unsigned off = (unsigned)(dn * g.OC * g.HW + pix) * ES;
                    float* bo = act_base<kBF>(stacked, ((size_t)cur.n0 * g.OC + hg * OCG) * g.HW);      // uniform
#pragma unroll
                    for (int k = 0; k < OCG; ++k) {
                        sta_u<kBF>(bo, off + (unsigned)(k * g.HW) * ES, outv[k]);
                        st_sum[k] += outv[k];
                        st_sq[k] = fmaf(outv[k], outv[k], st_sq[k]);
                    }
                    if (lse_out)
                        stg_u(lse_out + ((size_t)cur.n0 * g.G + hg) * g.HW, (unsigned)(dn * g.G * g.HW + pix) * 4u, lse);
                }
            } else {
                // the L lanes that read sequence ls are the only readers of its region: when they share a wave the
                // in-order LDS pipe makes the overwrite safe without a barrier; L = 128 spans two waves.
                if (L > 64) __syncthreads();
                else MEDT_WAVE_LOCKSTEP();
                if (active) {
#pragma unroll
                    for (int k = 0; k < OCG; ++k) {
                        reg[ls * RS + k * L + i] = outv[k];
                        st_sum[k] += outv[k];
                        st_sq[k] = fmaf(outv[k], outv[k], st_sq[k]);
                    }
                    reg[ls * RS + NCH * L + i] = lse;
                }
            }
        }
        if constexpr (AXIS == 0) {
            __syncthreads();
            store_super<F, AXIS>(reg, 0, stacked, g.OC, hg * OCG, OCG, g, cur, 1);
            if (lse_out) store_super<F, AXIS>(reg, NCH, lse_out, g.G, hg, 1, g, cur);
        }
    }
    // (kept out of the row loop: a branch there makes LLVM sink the whole accumulator chain behind it)
    if (!EXACT && bad) atomicOr(flag, 1u);
    if (out_partials) {
        float v[2 * OCG];
#pragma unroll
        for (int k = 0; k < OCG; ++k) { v[2 * k] = st_sum[k]; v[2 * k + 1] = st_sq[k]; }
        __syncthreads();
        block_sum_d<2 * OCG>(v, red, reinterpret_cast<double*>(out_partials) + ((size_t)blockIdx.x * g.OC + hg * OCG) * 2);
    }
}

// --------------------------------------------------------------------------- //
// Four rows per lane (gp = 2, large problems).  The one-row-per-lane kernel above is LDS-bandwidth bound: every
// lane fetches 7 x 16 B per 4 columns (k, 2 v, 4 table windows).  Here a lane owns rows i0, i0+2, i0+4, i0+6 of a
// sequence, so the k/v fetches are shared by four rows and the table windows of the four rows overlap
// (y = j - i + L - 1 spans y0-6 .. y0+3): 15 x 16 B per 16 (row, column) pairs instead of 28 -- the kernel
// becomes VALU bound.  Rows two apart keep every packed-FMA operand an even-aligned register pair: the pair
// (column c, c+1), c even, of row r' sits at window offsets (6 + c - 2r', +1).  Lanes with odd i0 read table
// copy 1 (U[x]), lanes with even i0 copy 0 (U[x-3]), which makes the window base a multiple of 4 floats for
// both and -- with the copy stride a multiple of 64 floats -- puts the two copies' windows on disjoint banks.
// Round 5 (profiles/r05_attn_fwd_isa_account.txt): the three 16-byte pieces of a chunk's window are pieces t, t+1, t+2 of one
// aligned sequence, and the next chunk's are t+1, t+2, t+3 -- two of the three are carried in registers across the unrolled
// chunks (MEDT_F4R_CARRY): 7 x 16 B per 16 pairs.  EXACT = false redoes a row whose logit bound was too loose itself
// (wave-uniform branch to the online-softmax sweep): one launch, no flag, no repair kernel behind it.
// --------------------------------------------------------------------------- //
#ifndef MEDT_F4R_CARRY
#define MEDT_F4R_CARRY 1
#endif
// true on every lane when the predicate holds on any lane of the wavefront (call it with the whole wave converged)
__device__ __forceinline__ bool wave_any(bool p) {
#ifdef MEDT_LANE_EMU
    int v = p ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
    return v != 0;
#else
    return __builtin_amdgcn_ballot_w64(p) != 0;
#endif
}
template <int L>
struct Fast4 {
    static constexpr int GP = 2, HQ = 1, NCH = 4, OCG = 4, LEN = L;
    static constexpr int LPS = L / 4;                         // lanes per sequence
    static constexpr int S_T = MEDT_THREADS / LPS;            // sequences per sub-tile
    static constexpr int EPT = 4;                             // tile elements per thread and sub-tile
    static constexpr int RS = (NCH + 1) * L + 4;
    static constexpr int NT = 2;
    static constexpr int CS = ((2 * L + 4 + 63) / 64) * 64;
    static constexpr int nta(int axis) { return axis == 1 ? 1 : NT; }
    static constexpr size_t lds_floats(int nt) { return (size_t)S_T * nt * RS + 256 + 8 * CS; }
};

}  // namespace MEDT_FAST_NS
#if !MEDT_FAST_BF16
int fast4_subtile_sequences(int L) { return MEDT_THREADS / (L / 4); }
int fast4_max_subtiles(int axis) { return axis == 1 ? 1 : 2; }
#endif
namespace MEDT_FAST_NS {

// VEC (width axis only): the lanes that move a sequence between global memory and LDS are the lanes that sweep it -- lane a of a
// sequence owns elements 4a .. 4a+3 of each channel row, one 16-byte access per channel and phase instead of four 4-byte ones
// with their address arithmetic -- so a wave only ever touches the LDS rows of its own sequences and the tile loop needs no
// workgroup barrier (the in-order LDS pipe orders a wave's own accesses).  Needs 16-byte aligned tensors (the launcher checks).
#define MEDT_F4R_WAVE_SYNC() do { MEDT_WAVE_LOCKSTEP(); __builtin_amdgcn_wave_barrier(); } while (0)
template <int AXIS, int L, bool EXACT, bool VEC>
__global__ __launch_bounds__(MEDT_THREADS) void attn_fwd4r_kernel(AxialGeom g, const float* __restrict__ qkv_raw,
                                                                  BnStats qs, BnStats ss,
                                                                  const float* __restrict__ relative, GatePtrs gates,
                                                                  float* __restrict__ stacked,
                                                                  float* __restrict__ lse_out,
                                                                  float* __restrict__ out_partials,
                                                                  unsigned* __restrict__ flag) {
    using F = Fast4<L>;
    constexpr int NCH = F::NCH, OCG = F::OCG, RS = F::RS, CS = F::CS, S_T = F::S_T, LPS = F::LPS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* reg = smem;
    float* red = reg + S_T * g.nt4 * RS;
    float* tab = red + 256;
    if (EXACT && flag && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;   // nothing to repair
    const int grp = blockIdx.x / g.oparts, part = blockIdx.x - grp * g.oparts, hg = blockIdx.y;
    const float f_qr = gate(gates.f_qr), f_kr = gate(gates.f_kr), f_sve = gate(gates.f_sve), f_sv = gate(gates.f_sv);
    const float a_qk = ss.scale[grp * g.SC + hg] * MEDT_LOG2E;
    const float a_qr = ss.scale[grp * g.SC + g.G + hg] * f_qr * MEDT_LOG2E;
    const float a_kr = ss.scale[grp * g.SC + 2 * g.G + hg] * f_kr * MEDT_LOG2E;
    {   // tables: row 0 = Rq, 1 = Rk (scaled), 2,3 = Rv; two copies each, column ordered (see stage_tables_cols)
        constexpr int TL = 2 * L - 1;
        for (int e = threadIdx.x; e < 8 * CS; e += MEDT_THREADS) {
            const int rc = e / CS, x = e - rc * CS, row = rc >> 1, copy = rc & 1;
            const int y = x - (copy ? 0 : 3);
            float v = 0.f;
            if (y >= 0 && y < TL) v = row == 1 ? a_kr * relative[TL + y] : relative[row * TL + (TL - 1 - y)];
            tab[e] = v;
        }
    }
    if (threadIdx.x < NCH) {                                  // bn_qkv's affine for this head group: LDS-resident
        red[128 + threadIdx.x] = qs.scale[grp * 2 * g.C + hg * NCH + threadIdx.x];
        red[160 + threadIdx.x] = qs.shift[grp * 2 * g.C + hg * NCH + threadIdx.x];
    }
    __syncthreads();
    float tqmax, tkmax;                                       // max |entry| of Rq and of the scaled Rk (logit bound)
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        float a = 0.f, b = 0.f;
        for (int x = threadIdx.x; x < 2 * L - 1; x += MEDT_THREADS) {
            a = fmaxf(a, fabsf(tab[1 * CS + x]));
            b = fmaxf(b, fabsf(tab[3 * CS + x]));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a = fmaxf(a, __shfl_xor(a, o, 64)); b = fmaxf(b, __shfl_xor(b, o, 64)); }
        if (lane == 0) { red[wave * 2] = a; red[wave * 2 + 1] = b; }
        __syncthreads();
        tqmax = fmaxf(fmaxf(red[0], red[2]), fmaxf(red[4], red[6]));
        tkmax = fmaxf(fmaxf(red[1], red[3]), fmaxf(red[5], red[7]));
#if MEDT_F4R_CARRY
        // wave-uniform: scalar registers (the sweep runs at the edge of the three-waves-per-SIMD register budget)
        tqmax = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tqmax)));
        tkmax = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tkmax)));
#endif
    }
    const float* sc = red + 128;
    const float* sh = red + 160;
    const int lsub = threadIdx.x / LPS, a = threadIdx.x % LPS, b = a & 1, i0 = 4 * (a & ~1) + b;
    const float* tabl = tab + b * CS + (b ? L - 8 : L - 4) - 4 * (a & ~1);    // this lane's window at j0 = 0
    float st_sum[OCG], st_sq[OCG];
#pragma unroll
    for (int k = 0; k < OCG; ++k) { st_sum[k] = 0.f; st_sq[k] = 0.f; }
    const int SSr = S_T * g.nt4;
    const int nsup = (g.spg + SSr - 1) / SSr;
    using Map = SuperMap<F, AXIS>;
    using PF = SuperPrefetch<F, AXIS, NCH>;
    const unsigned step = (unsigned)g.oparts * SSr;
    const int dn_step = step / (unsigned)g.Bo, ds_step = step - dn_step * g.Bo;
    Map cur, nxt;
    {
        const unsigned q0 = (unsigned)part * SSr;
        const int dn = q0 / (unsigned)g.Bo;
        nxt.set(g, grp * g.npg + dn, q0 - dn * g.Bo, min(SSr, g.spg - (int)q0));
    }
    constexpr bool ROWMOVE = VEC && AXIS == 1;                 // (one sub-tile per tile on the width axis: F::nta(1) == 1)
    constexpr unsigned ES = kBF ? 2u : 4u;
    PF pf;
    f4 pv[NCH];                                                // ROWMOVE: the next tile's raw q | k | v0 | v1 pieces of this lane
    auto row_issue = [&](const Map& m) {
        if (lsub < m.nseq) {
            int dn, sq;
            m.image_of(g, lsub, dn, sq);
            const float* base = act_base<kBF>(qkv_raw, ((size_t)m.n0 * 2 * g.C + hg * NCH) * g.HW);      // uniform
            const unsigned off = (unsigned)(dn * (2 * g.C * g.HW) + sq * g.W + 4 * a) * ES;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) pv[ch] = lda4_u<kBF>(base, off + (unsigned)(ch * g.HW) * ES);
        }
    };
    if (part < nsup) {
        if constexpr (ROWMOVE) row_issue(nxt);
        else pf.issue(qkv_raw, g, hg, nxt);
    }
    for (int u = part; u < nsup; u += g.oparts) {
        cur = nxt;
        const int nseq = cur.nseq;
        if constexpr (ROWMOVE) {
            MEDT_F4R_WAVE_SYNC();                              // this wave's stores of the previous tile have read the rows
            if (lsub < nseq) {
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    const float c1 = sc[ch], c0 = sh[ch];
                    f4 t;
                    t.x = fmaf(pv[ch].x, c1, c0); t.y = fmaf(pv[ch].y, c1, c0);
                    t.z = fmaf(pv[ch].z, c1, c0); t.w = fmaf(pv[ch].w, c1, c0);
                    *reinterpret_cast<f4*>(reg + lsub * RS + ch * L + 4 * a) = t;
                }
            }
        } else {
            __syncthreads();                                   // previous super-tile fully stored / tables staged
            pf.commit(reg, g, cur, sc, sh);
        }
        {
            const int un = u + g.oparts;
            int s0 = cur.s0 + ds_step, n0 = cur.n0 + dn_step;
            if (s0 >= g.Bo) { s0 -= g.Bo; ++n0; }
            if (un < nsup) {
                nxt.set(g, n0, s0, min(SSr, g.spg - un * SSr));
                if constexpr (ROWMOVE) row_issue(nxt);
                else pf.issue(qkv_raw, g, hg, nxt);
            }
        }
        if constexpr (ROWMOVE) MEDT_F4R_WAVE_SYNC();
        else __syncthreads();
#pragma unroll 1
        for (int sub = 0; sub < g.nt4; ++sub) {
            const int ls = sub * S_T + lsub;
            const bool active = ls < nseq;
            float* sq = reg + ls * RS;                         // q | k | v0 | v1 rows of this sequence
            float kmx = 0.f;
            if (active) {
#pragma unroll
                for (int r = 0; r < 4; ++r) kmx = fmaxf(kmx, fabsf(sq[L + i0 + 2 * r]));
            }
#pragma unroll
            for (int o = LPS / 2; o > 0; o >>= 1) kmx = fmaxf(kmx, __shfl_xor(kmx, o, 64));
            float qa[4], qb[4], m[4];
            f2 l2[4], av0[4], av1[4], ae0[4], ae1[4];
            const float* kp = sq + L;
            const float* vp = sq + 2 * L;
            // One pass over the L keys for this lane's four query rows.  EX = false: softmax referenced to the per-row upper
            // bound of the logits; EX = true: online softmax (running maximum, one rescale per chunk).  Active lanes only.
            auto sweep_rows = [&](auto exact_tag) {
                constexpr bool EX = decltype(exact_tag)::value;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    m[r] = EX ? -INFINITY : g.bound_shift + fabsf(qa[r]) * kmx + fabsf(qb[r]) * tqmax + kmx * tkmax;
                    l2[r] = av0[r] = av1[r] = ae0[r] = ae1[r] = (f2)(0.f);
                }
#if MEDT_F4R_CARRY
                    // The windows of consecutive chunks overlap in two of their three 16-byte pieces: piece t of chunk c is
                    // piece t - 1 of chunk c + 1.  The loop is fully unrolled, so carrying the two pieces is a renaming, and a
                    // chunk fetches ONE new piece per table (7 ds_read_b128 per chunk instead of 11 + 2 ds_read2st64_b64).
                    // Measured on the roofline shape (profiles/r05_fwd_variants_*.json): 185 -> 169 us with the 4-byte movers.
                    // (-DMEDT_F4R_CARRY=0 builds the round-4 body for A/B runs.)
                    f4 cq[2], ck[2], c0[2], c1[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        cq[t] = *reinterpret_cast<const f4*>(tabl + 0 * CS + 4 * t);
                        ck[t] = *reinterpret_cast<const f4*>(tabl + 2 * CS + 4 * t);
                        c0[t] = *reinterpret_cast<const f4*>(tabl + 4 * CS + 4 * t);
                        c1[t] = *reinterpret_cast<const f4*>(tabl + 6 * CS + 4 * t);
                    }
                    f4 kc = *reinterpret_cast<const f4*>(kp);
#endif
#pragma unroll
                    for (int j0 = 0; j0 < L; j0 += 4) {
#if MEDT_F4R_CARRY
                        const f4 k4 = kc;
#else
                        const f4 k4 = *reinterpret_cast<const f4*>(kp + j0);
#endif
                        const f4 v0 = *reinterpret_cast<const f4*>(vp + j0);
                        const f4 v1 = *reinterpret_cast<const f4*>(vp + L + j0);
                        f2 wq[6], wk[6], w0[6], w1[6];
                        auto row = [&](const int r) {
                            const f2 fqa = (f2)(qa[r]), fqb = (f2)(qb[r]);
                            if constexpr (EX) {
                                f2 zlo = fqa * k4.lo + (fqb * wq[3 - r] + k4.lo * wk[3 - r]);
                                f2 zhi = fqa * k4.hi + (fqb * wq[4 - r] + k4.hi * wk[4 - r]);
                                const float mn = fmaxf(m[r], fmaxf(fmaxf(zlo.x, zlo.y), fmaxf(zhi.x, zhi.y)));
                                const f2 alpha = (f2)(__builtin_amdgcn_exp2f(m[r] - mn));
                                m[r] = mn;
                                zlo -= (f2)(mn);
                                zhi -= (f2)(mn);
                                f2 plo, phi;
                                plo.x = __builtin_amdgcn_exp2f(zlo.x); plo.y = __builtin_amdgcn_exp2f(zlo.y);
                                phi.x = __builtin_amdgcn_exp2f(zhi.x); phi.y = __builtin_amdgcn_exp2f(zhi.y);
                                l2[r] = l2[r] * alpha + (plo + phi);
                                av0[r] = plo * v0.lo + (phi * v0.hi + av0[r] * alpha);
                                av1[r] = plo * v1.lo + (phi * v1.hi + av1[r] * alpha);
                                ae0[r] = plo * w0[3 - r] + (phi * w0[4 - r] + ae0[r] * alpha);
                                ae1[r] = plo * w1[3 - r] + (phi * w1[4 - r] + ae1[r] * alpha);
                            } else {
                                const f2 zi = (f2)(-m[r]);
                                const f2 zlo = fqa * k4.lo + (fqb * wq[3 - r] + (k4.lo * wk[3 - r] + zi));
                                const f2 zhi = fqa * k4.hi + (fqb * wq[4 - r] + (k4.hi * wk[4 - r] + zi));
                                f2 plo, phi;
                                plo.x = __builtin_amdgcn_exp2f(zlo.x); plo.y = __builtin_amdgcn_exp2f(zlo.y);
                                phi.x = __builtin_amdgcn_exp2f(zhi.x); phi.y = __builtin_amdgcn_exp2f(zhi.y);
                                l2[r] += plo + phi;
                                av0[r] = plo * v0.lo + (phi * v0.hi + av0[r]);
                                av1[r] = plo * v1.lo + (phi * v1.hi + av1[r]);
                                ae0[r] = plo * w0[3 - r] + (phi * w0[4 - r] + ae0[r]);
                                ae1[r] = plo * w1[3 - r] + (phi * w1[4 - r] + ae1[r]);
                            }
                        };
#if MEDT_F4R_CARRY
                        wq[0] = cq[0].lo; wq[1] = cq[0].hi; wq[2] = cq[1].lo; wq[3] = cq[1].hi;
                        wk[0] = ck[0].lo; wk[1] = ck[0].hi; wk[2] = ck[1].lo; wk[3] = ck[1].hi;
                        w0[0] = c0[0].lo; w0[1] = c0[0].hi; w0[2] = c0[1].lo; w0[3] = c0[1].hi;
                        w1[0] = c1[0].lo; w1[1] = c1[0].hi; w1[2] = c1[1].lo; w1[3] = c1[1].hi;
                        row(3); row(2);                              // the oldest piece dies here, before the new one is fetched
                        __builtin_amdgcn_sched_barrier(0);
                        {
                            const f4 nq = *reinterpret_cast<const f4*>(tabl + 0 * CS + j0 + 8);
                            const f4 nk = *reinterpret_cast<const f4*>(tabl + 2 * CS + j0 + 8);
                            const f4 n0 = *reinterpret_cast<const f4*>(tabl + 4 * CS + j0 + 8);
                            const f4 n1 = *reinterpret_cast<const f4*>(tabl + 6 * CS + j0 + 8);
                            kc = *reinterpret_cast<const f4*>(kp + (j0 + 4 < L ? j0 + 4 : j0));   // next chunk's keys, half a chunk ahead
                            wq[4] = nq.lo; wk[4] = nk.lo; w0[4] = n0.lo; w1[4] = n1.lo;
                            cq[0] = cq[1]; cq[1] = nq;
                            ck[0] = ck[1]; ck[1] = nk;
                            c0[0] = c0[1]; c0[1] = n0;
                            c1[0] = c1[1]; c1[1] = n1;
                        }
                        row(1); row(0);                              // the row that needs the new piece goes last
#else
#pragma unroll
                        for (int t = 0; t < 3; ++t) {
                            const f4 xq = *reinterpret_cast<const f4*>(tabl + 0 * CS + j0 + 4 * t);
                            const f4 xk = *reinterpret_cast<const f4*>(tabl + 2 * CS + j0 + 4 * t);
                            const f4 x0 = *reinterpret_cast<const f4*>(tabl + 4 * CS + j0 + 4 * t);
                            const f4 x1 = *reinterpret_cast<const f4*>(tabl + 6 * CS + j0 + 4 * t);
                            wq[2 * t] = xq.lo; wq[2 * t + 1] = xq.hi;
                            wk[2 * t] = xk.lo; wk[2 * t + 1] = xk.hi;
                            w0[2 * t] = x0.lo; w0[2 * t + 1] = x0.hi;
                            w1[2 * t] = x1.lo; w1[2 * t + 1] = x1.hi;
                        }
                        row(0); row(1); row(2); row(3);
#endif
                        __builtin_amdgcn_sched_barrier(0);           // keep the unrolled chunks in order (register pressure)
                    }
            };
            bool redo = false;
            if (active) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float q = sq[i0 + 2 * r];
                    qa[r] = q * a_qk;
                    qb[r] = q * a_qr;
                }
                if constexpr (EXACT) {
                    sweep_rows(std::true_type{});
                } else {
                    sweep_rows(std::false_type{});
#pragma unroll
                    for (int r = 0; r < 4; ++r) redo |= !(l2[r].x + l2[r].y > 1e-30f);      // bound too loose for this row
                }
            }
            if constexpr (!EXACT) {
                // A row whose bound was too loose (its sum underflowed) is redone on the spot with the online softmax, by
                // the whole wave (the branch is wave-uniform; the rows are still in LDS, nothing has been written yet).
                // Normally never taken: no flag, no memset, no repair launch behind the kernel.
                if (wave_any(redo)) {
                    if (active) sweep_rows(std::true_type{});
                }
            }
            if (active) {
                // results replace this sequence's q|k|v rows in LDS: its lanes all sit in this wave and have issued
                // every read of the sweep above (in-order LDS pipe)
                MEDT_WAVE_LOCKSTEP();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float l = l2[r].x + l2[r].y;
                    const float inv = __builtin_amdgcn_rcpf(l);
                    const float o[4] = {f_sv * (av0[r].x + av0[r].y) * inv, f_sve * (ae0[r].x + ae0[r].y) * inv,
                                        f_sv * (av1[r].x + av1[r].y) * inv, f_sve * (ae1[r].x + ae1[r].y) * inv};
#pragma unroll
                    for (int k = 0; k < OCG; ++k) {
                        sq[k * L + i0 + 2 * r] = o[k];
                        st_sum[k] += o[k];
                        st_sq[k] = fmaf(o[k], o[k], st_sq[k]);
                    }
                    sq[NCH * L + i0 + 2 * r] = m[r] + __log2f(l);
                }
            }
        }
        if constexpr (ROWMOVE) {
            MEDT_F4R_WAVE_SYNC();
            if (lsub < nseq) {
                int dn, sq;
                cur.image_of(g, lsub, dn, sq);
                const unsigned pix = (unsigned)(sq * g.W + 4 * a);
                float* bo = act_base<kBF>(stacked, ((size_t)cur.n0 * g.OC + hg * OCG) * g.HW);         // uniform
                const unsigned off = ((unsigned)(dn * g.OC * g.HW) + pix) * ES;
#pragma unroll
                for (int k = 0; k < OCG; ++k)
                    sta4_u<kBF>(bo, off + (unsigned)(k * g.HW) * ES, *reinterpret_cast<const f4*>(reg + lsub * RS + k * L + 4 * a));
                if (lse_out)
                    sta4_u<false>(lse_out + ((size_t)cur.n0 * g.G + hg) * g.HW, ((unsigned)(dn * g.G * g.HW) + pix) * 4u,
                                  *reinterpret_cast<const f4*>(reg + lsub * RS + NCH * L + 4 * a));
            }
        } else {
            __syncthreads();
            store_super<F, AXIS>(reg, 0, stacked, g.OC, hg * OCG, OCG, g, cur, 1);
            if (lse_out) store_super<F, AXIS>(reg, NCH, lse_out, g.G, hg, 1, g, cur);
        }
    }
    if (out_partials) {
        float v[2 * OCG];
#pragma unroll
        for (int k = 0; k < OCG; ++k) { v[2 * k] = st_sum[k]; v[2 * k + 1] = st_sq[k]; }
        __syncthreads();
        block_sum_d<2 * OCG>(v, red, reinterpret_cast<double*>(out_partials) + ((size_t)blockIdx.x * g.OC + hg * OCG) * 2);
    }
}

#define MEDT_F3_CASE(KERNEL, GPv, Lv, ...)                                                                          \
    case GPv * 1024 + Lv * 2 + 0:                                                                                   \
        hipLaunchKernelGGL((KERNEL(GPv, 0, Lv)), grid, block, ((Fast3<GPv, Lv>::lds_floats - (size_t)(Fast3<GPv, Lv>::NT - g.nt) * Fast3<GPv, Lv>::S_T * Fast3<GPv, Lv>::RS) * sizeof(float)), s, __VA_ARGS__); \
        break;                                                                                                      \
    case GPv * 1024 + Lv * 2 + 1:                                                                                   \
        hipLaunchKernelGGL((KERNEL(GPv, 1, Lv)), grid, block, ((Fast3<GPv, Lv>::lds_floats - (size_t)(Fast3<GPv, Lv>::NT - g.nt) * Fast3<GPv, Lv>::S_T * Fast3<GPv, Lv>::RS) * sizeof(float)), s, __VA_ARGS__); \
        break;
#define MEDT_F3_L(KERNEL, GPv, ...)                                                                                 \
    MEDT_F3_CASE(KERNEL, GPv, 16, __VA_ARGS__) MEDT_F3_CASE(KERNEL, GPv, 32, __VA_ARGS__)                            \
    MEDT_F3_CASE(KERNEL, GPv, 64, __VA_ARGS__) MEDT_F3_CASE(KERNEL, GPv, 128, __VA_ARGS__)
#define MEDT_FAST3_DISPATCH(KERNEL, ...)                                                                            \
    do {                                                                                                            \
        const dim3 grid(g.groups * g.fparts, g.G), block(MEDT_THREADS);                                             \
        switch (g.gp * 1024 + g.L * 2 + g.axis) {                                                                   \
            MEDT_F3_L(KERNEL, 2, __VA_ARGS__) MEDT_F3_L(KERNEL, 4, __VA_ARGS__)                                      \
            MEDT_F3_L(KERNEL, 8, __VA_ARGS__) MEDT_F3_L(KERNEL, 16, __VA_ARGS__)                                     \
            default: set_error("fast3: no instantiation"); return MEDT_EUNSUPPORTED;                             \
        }                                                                                                           \
        return launch_status("fast3 kernel");                                                                       \
    } while (0)

// --------------------------------------------------------------------------- //
// launchers: return 1 when the geometry is outside the fast path (caller falls back to axial_core.hip)
// --------------------------------------------------------------------------- //
#define MEDT_FAST_DISPATCH(KERNEL, ...)                                                                     \
    do {                                                                                                    \
        if (!g.pos || (g.L & 3)) return 1;                                                                  \
        const size_t lds = fast_lds_bytes(g);                                                               \
        if (lds > 160 * 1024) return 1;                                                                     \
        const dim3 grid(g.groups * g.tpg, g.G), block(MEDT_THREADS);                                        \
        switch (g.gp * 2 + g.axis) {                                                                        \
            case 2 * 2 + 0: hipLaunchKernelGGL((KERNEL<2, 0>), grid, block, lds, s, __VA_ARGS__); break;     \
            case 2 * 2 + 1: hipLaunchKernelGGL((KERNEL<2, 1>), grid, block, lds, s, __VA_ARGS__); break;     \
            case 4 * 2 + 0: hipLaunchKernelGGL((KERNEL<4, 0>), grid, block, lds, s, __VA_ARGS__); break;     \
            case 4 * 2 + 1: hipLaunchKernelGGL((KERNEL<4, 1>), grid, block, lds, s, __VA_ARGS__); break;     \
            case 8 * 2 + 0: hipLaunchKernelGGL((KERNEL<8, 0>), grid, block, lds, s, __VA_ARGS__); break;     \
            case 8 * 2 + 1: hipLaunchKernelGGL((KERNEL<8, 1>), grid, block, lds, s, __VA_ARGS__); break;     \
            case 16 * 2 + 0: hipLaunchKernelGGL((KERNEL<16, 0>), grid, block, lds, s, __VA_ARGS__); break;   \
            case 16 * 2 + 1: hipLaunchKernelGGL((KERNEL<16, 1>), grid, block, lds, s, __VA_ARGS__); break;   \
            default: return 1;                                                                              \
        }                                                                                                   \
        return launch_status(#KERNEL);                                                                      \
    } while (0)

}  // namespace MEDT_FAST_NS
using namespace MEDT_FAST_NS;

template <bool EX, int AX, int Lv>
static void r4_launch(bool vec, const AxialGeom& g, const float* qkv_raw, BnStats qkv, BnStats sim, const float* relative,
                      GatePtrs gates, float* stacked, float* lse, float* out_partials, unsigned* flag, hipStream_t s) {
    const dim3 grid(g.groups * g.oparts, g.G), block(MEDT_THREADS);
    const size_t lds = Fast4<Lv>::lds_floats(g.nt4) * sizeof(float);
    if constexpr (AX == 1) {
        if (vec) {
            hipLaunchKernelGGL((attn_fwd4r_kernel<AX, Lv, EX, true>), grid, block, lds, s, g, qkv_raw, qkv, sim, relative, gates,
                               stacked, lse, out_partials, flag);
            return;
        }
    }
    hipLaunchKernelGGL((attn_fwd4r_kernel<AX, Lv, EX, false>), grid, block, lds, s, g, qkv_raw, qkv, sim, relative, gates, stacked,
                       lse, out_partials, flag);
}

int MEDT_FAST_FN(const AxialGeom& g, const float* qkv_raw, BnStats qkv, BnStats sim, const float* relative,
                        GatePtrs gates, float* stacked, float* lse, float* out_partials, unsigned* flag, hipStream_t s,
                        const FinSrc* simsrc) {
    const FinSrc nosrc = no_fin_src();
    if (simsrc && simsrc->on && !(g.fast3 && !g.rows4 && !(g.bound_path && flag))) {
        set_error("attn_fwd: bn_similarity cannot be finalised in this kernel (axial_attn_fwd_inlines)"); return MEDT_EINVAL;
    }
#define MEDT_K_EXACT(a, b, c) attn_fwd3_kernel<a, b, c, true>
#define MEDT_K_BOUND(a, b, c) attn_fwd3_kernel<a, b, c, false>
    if (g.rows4) {
        // gp = 2, large problem: four rows per lane (VALU bound instead of LDS bound); repairs itself, one launch
        // width axis: 16-byte movers when the tensors allow it (MEDT_F4R_VEC=0: the 4-byte movers, for A/B)
        static const bool vec_on = [] { const char* e = getenv("MEDT_F4R_VEC"); return !(e && atoi(e) == 0); }();
        const bool vec = vec_on && g.axis == 1 && g.nt4 == 1 && (g.W & 3) == 0 && (g.HW & 3) == 0 &&
                         (((uintptr_t)qkv_raw | (uintptr_t)stacked | (uintptr_t)lse) & 15) == 0;
#define MEDT_R4_LAUNCH(EX, AX, Lv)                                                                                   \
    r4_launch<EX, AX, Lv>(vec, g, qkv_raw, qkv, sim, relative, gates, stacked, lse, out_partials, flag, s)
#define MEDT_R4_CASES(EX)                                                                                            \
    switch (g.L * 2 + g.axis) {                                                                                      \
        case 16 * 2 + 0: MEDT_R4_LAUNCH(EX, 0, 16); break;   case 16 * 2 + 1: MEDT_R4_LAUNCH(EX, 1, 16); break;        \
        case 32 * 2 + 0: MEDT_R4_LAUNCH(EX, 0, 32); break;   case 32 * 2 + 1: MEDT_R4_LAUNCH(EX, 1, 32); break;        \
        case 64 * 2 + 0: MEDT_R4_LAUNCH(EX, 0, 64); break;   case 64 * 2 + 1: MEDT_R4_LAUNCH(EX, 1, 64); break;        \
        case 128 * 2 + 0: MEDT_R4_LAUNCH(EX, 0, 128); break; case 128 * 2 + 1: MEDT_R4_LAUNCH(EX, 1, 128); break;     \
        default: set_error("rows4: no instantiation"); return MEDT_EUNSUPPORTED;                                     \
    }
        flag = nullptr;                       // (the four-rows kernel redoes a row whose bound was too loose itself)
        if (g.bound_path) {
            MEDT_R4_CASES(false)
            return launch_status("attn_fwd4r (bound)");
        }
        MEDT_R4_CASES(true)
        return launch_status("attn_fwd4r");
    }
    if (g.fast3 && g.bound_path && flag) {
        // large problems: bound-referenced softmax, then the (normally empty) repair pass
        if (hipMemsetAsync(flag, 0, sizeof(unsigned), s) != hipSuccess) { set_error("attn_fwd: memset failed"); return MEDT_ELAUNCH; }
        int rc = [&]() -> int { MEDT_FAST3_DISPATCH(MEDT_K_BOUND, g, qkv_raw, qkv, sim, relative, gates, stacked, lse, out_partials, flag, nosrc); }();
        if (rc) return rc;
        MEDT_FAST3_DISPATCH(MEDT_K_EXACT, g, qkv_raw, qkv, sim, relative, gates, stacked, lse, out_partials, flag, nosrc);
    }
    if (g.fast3) MEDT_FAST3_DISPATCH(MEDT_K_EXACT, g, qkv_raw, qkv, sim, relative, gates, stacked, lse, out_partials, (unsigned*)nullptr,
                                     (simsrc ? *simsrc : nosrc));
    MEDT_FAST_DISPATCH(attn_fwd4_kernel, g, qkv_raw, qkv, sim, relative, gates, stacked, lse, out_partials);
}

}  // namespace medt
