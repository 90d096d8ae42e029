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
#include "axial_tiles.h"

namespace medt {

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
    tile_load<AXIS>(reg, RS, 0, qkv_raw, 2 * g.C, hg * NCH, NCH, t);
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
    tile_store<AXIS>(reg, RS, 0, stacked, g.OC, hg * OCG, OCG, t);
    if (lse_out) tile_store<AXIS>(reg, RS, NCH, lse_out, g.G, hg, 1, t);
    if (out_partials) {
        float v[2 * OCG];
#pragma unroll
        for (int k = 0; k < OCG; ++k) { v[2 * k] = outv[k]; v[2 * k + 1] = outv[k] * outv[k]; }
        block_sum<2 * OCG>(v, red, out_partials + ((size_t)blockIdx.x * g.OC + hg * OCG) * 2);
    }
}

// --------------------------------------------------------------------------- //
// forward statistics pass: sum / sum of squares of qk, f_qr*qr, f_kr*kr per head
// --------------------------------------------------------------------------- //
template <int GP, int AXIS>
__global__ __launch_bounds__(MEDT_THREADS) void logit_stats4_kernel(AxialGeom g, const float* __restrict__ qkv_raw,
                                                                    BnStats qs, const float* __restrict__ relative,
                                                                    GatePtrs gates, float* __restrict__ partials) {
    constexpr int HQ = GP / 2, NCH = 2 * GP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = g.L, RS = region4(NCH, L), CS = copy_stride(L);
    float* reg = smem;
    float* red = reg + g.S_T * RS;
    float* tab = red + 256;
    const int grp = blockIdx.x / g.tpg, tile = blockIdx.x - grp * g.tpg, hg = blockIdx.y;
    TileCtx t{L, g.Bo, g.W, g.HW, grp * g.spg + tile * g.S_T, min(g.S_T, g.spg - tile * g.S_T)};
    tile_load<AXIS>(reg, RS, 0, qkv_raw, 2 * g.C, hg * NCH, GP, t);           // q and k channels only
    const float f_qr = gate(gates.f_qr), f_kr = gate(gates.f_kr);
    stage_tables4<GP>(tab, relative, L, 1.f);
    __syncthreads();
    const int ls = threadIdx.x / L, i = threadIdx.x - ls * L;
    const bool active = ls < t.nseq;
    const float* sc = qs.scale + grp * 2 * g.C + hg * NCH;
    const float* sh = qs.shift + grp * 2 * g.C + hg * NCH;
    if (active) {
#pragma unroll
        for (int ch = 0; ch < GP; ++ch) {
            const int idx = ls * RS + ch * L + i;
            reg[idx] = fmaf(reg[idx], sc[ch], sh[ch]);
        }
    }
    __syncthreads();
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (active) {
        float q[HQ];
#pragma unroll
        for (int c = 0; c < HQ; ++c) q[c] = reg[ls * RS + c * L + i];
        const float* kp = reg + ls * RS + HQ * L;
        const int r = i & 3;
        const float* tabr = tab + r * CS;
        for (int j0 = 0; j0 < L; j0 += 4) {
            const int xoff = i - j0 + L - 4 - r;
            float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f}, cc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < HQ; ++c) {
                const float4 k4 = *reinterpret_cast<const float4*>(kp + c * L + j0);
                const float4 q4 = *reinterpret_cast<const float4*>(tabr + (c * 4) * CS + xoff);
                const float4 t4 = *reinterpret_cast<const float4*>(tabr + ((HQ + c) * 4) * CS + xoff);
                a[0] = fmaf(q[c], k4.x, a[0]); b[0] = fmaf(q[c], q4.w, b[0]); cc[0] = fmaf(k4.x, t4.w, cc[0]);
                a[1] = fmaf(q[c], k4.y, a[1]); b[1] = fmaf(q[c], q4.z, b[1]); cc[1] = fmaf(k4.y, t4.z, cc[1]);
                a[2] = fmaf(q[c], k4.z, a[2]); b[2] = fmaf(q[c], q4.y, b[2]); cc[2] = fmaf(k4.z, t4.y, cc[2]);
                a[3] = fmaf(q[c], k4.w, a[3]); b[3] = fmaf(q[c], q4.x, b[3]); cc[3] = fmaf(k4.w, t4.x, cc[3]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float tb = f_qr * b[u], tc = f_kr * cc[u];
                acc[0] += a[u];
                acc[1] = fmaf(a[u], a[u], acc[1]);
                acc[2] += tb;
                acc[3] = fmaf(tb, tb, acc[3]);
                acc[4] += tc;
                acc[5] = fmaf(tc, tc, acc[5]);
            }
        }
    }
    float* dst = partials + ((size_t)blockIdx.x * g.SC + hg) * 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float s = wave_sum(acc[k]);
        if (lane == 0) red[wave * 6 + k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        const float s = (red[k] + red[6 + k]) + (red[12 + k] + red[18 + k]);
        dst[(size_t)(k >> 1) * g.G * 2 + (k & 1)] = s;
    }
}

// --------------------------------------------------------------------------- //
// Compile-time-L variants (L in {16,32,64,128}: every layer of the 128- and 256-pixel models).
// On top of the above: (1) the workgroup is persistent over "super-tiles" of SS = NT*S_T adjacent sequences,
// so the relative tables are staged once per workgroup instead of once per S_T sequences and the NCHW rows
// fetched for the height axis are SS*4 bytes long instead of S_T*4; (2) bn_qkv's affine is applied while
// staging; (3) all LDS addresses are base + immediate (RS, CS, L are constants), the j-loop is unrolled.
// --------------------------------------------------------------------------- //
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

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
    static constexpr int HQ = GP / 2, NCH = 2 * GP, OCG = 2 * GP;
    static constexpr int S_T = MEDT_THREADS / L;
    static constexpr int RS = (NCH + 1) * L + 4;
    static constexpr int NT0 = 8192 / (S_T * RS);
    static constexpr int NT = NT0 < 1 ? 1 : (NT0 > 8 ? 8 : NT0);
    static constexpr int SS = S_T * NT;
    static constexpr int CS = 2 * L + (((16 - 2 * L) % 64) + 64) % 64;
    static constexpr size_t lds_floats = (size_t)SS * RS + 256 + (size_t)NCH * 4 * CS;
};

int fast3_max_subtiles(int gp, int L) {     // host mirror of Fast3<GP,L>::NT (0 = no compile-time variant)
    if (L != 16 && L != 32 && L != 64 && L != 128) return 0;
    const int S_T = MEDT_THREADS / L, RS = (2 * gp + 1) * L + 4;
    const int NT = 8192 / (S_T * RS);
    return NT < 1 ? 1 : (NT > 8 ? 8 : NT);
}

// Stage `nch` channels of a super-tile, normalised by bn_qkv, lanes along the contiguous NCHW direction.
template <int GP, int L, int AXIS>
__device__ __forceinline__ void stage_super(float* reg, const float* __restrict__ qkv_raw, const AxialGeom& g, int hg,
                                            int seq0, int nseq, int nch, const float* __restrict__ sc,
                                            const float* __restrict__ sh) {
    using F = Fast3<GP, L>;
    for (int e = threadIdx.x; e < nseq * L; e += MEDT_THREADS) {
        int ls, i;
        if (AXIS == 1) { ls = e / L; i = e % L; }
        else if (nseq == F::SS) { i = e / F::SS; ls = e % F::SS; }
        else { i = e / nseq; ls = e - i * nseq; }
        const int b = seq0 + ls, n = b / g.Bo, s = b - n * g.Bo;
        const float* src = qkv_raw + ((size_t)n * 2 * g.C + hg * F::NCH) * g.HW + (AXIS == 1 ? s * g.W + i : i * g.W + s);
        float* dst = reg + ls * F::RS + i;
        for (int ch = 0; ch < nch; ++ch) dst[ch * L] = fmaf(src[(size_t)ch * g.HW], sc[ch], sh[ch]);
    }
}

template <int GP, int L, int AXIS>
__device__ __forceinline__ void store_super(const float* reg, int lch0, float* __restrict__ dst, int CH, int ch0, int nch,
                                            const AxialGeom& g, int seq0, int nseq) {
    using F = Fast3<GP, L>;
    for (int e = threadIdx.x; e < nseq * L; e += MEDT_THREADS) {
        int ls, i;
        if (AXIS == 1) { ls = e / L; i = e % L; }
        else if (nseq == F::SS) { i = e / F::SS; ls = e % F::SS; }
        else { i = e / nseq; ls = e - i * nseq; }
        const int b = seq0 + ls, n = b / g.Bo, s = b - n * g.Bo;
        float* out = dst + ((size_t)n * CH + ch0) * g.HW + (AXIS == 1 ? s * g.W + i : i * g.W + s);
        const float* src = reg + ls * F::RS + lch0 * L + i;
        for (int ch = 0; ch < nch; ++ch) out[(size_t)ch * g.HW] = src[ch * L];
    }
}

template <int GP, int AXIS, int L>
__global__ __launch_bounds__(MEDT_THREADS) void attn_fwd3_kernel(AxialGeom g, const float* __restrict__ qkv_raw,
                                                                 BnStats qs, BnStats ss,
                                                                 const float* __restrict__ relative, GatePtrs gates,
                                                                 float* __restrict__ stacked,
                                                                 float* __restrict__ lse_out,
                                                                 float* __restrict__ out_partials) {
    using F = Fast3<GP, L>;
    constexpr int HQ = F::HQ, NCH = F::NCH, OCG = F::OCG, RS = F::RS, CS = F::CS, S_T = F::S_T, SS = F::SS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* reg = smem;
    float* red = reg + SS * RS;
    float* tab = red + 256;
    const int grp = blockIdx.x / g.fparts, part = blockIdx.x - grp * g.fparts, hg = blockIdx.y;
    const float f_qr = gate(gates.f_qr), f_kr = gate(gates.f_kr), f_sve = gate(gates.f_sve), f_sv = gate(gates.f_sv);
    const float a_qk = ss.scale[grp * g.SC + hg] * MEDT_LOG2E;
    const float a_qr = ss.scale[grp * g.SC + g.G + hg] * f_qr * MEDT_LOG2E;
    const float a_kr = ss.scale[grp * g.SC + 2 * g.G + hg] * f_kr * MEDT_LOG2E;
    stage_tables_cols<GP>(tab, relative, L, a_kr);
    const float* sc = qs.scale + grp * 2 * g.C + hg * NCH;
    const float* sh = qs.shift + grp * 2 * g.C + hg * NCH;
    const int lsub = threadIdx.x / L, i = threadIdx.x % L;
    const int r = (3 - i) & 3;
    const float* tabr = tab + r * CS + (L - 1 - i - r);       // + j0 per chunk (immediate offset)
    float st_sum[OCG], st_sq[OCG];
#pragma unroll
    for (int k = 0; k < OCG; ++k) { st_sum[k] = 0.f; st_sq[k] = 0.f; }
    const int SSr = S_T * g.nt;                               // runtime super-tile (g.nt <= NT sub-tiles)
    const int nsup = (g.spg + SSr - 1) / SSr;
    for (int u = part; u < nsup; u += g.fparts) {
        const int seq0 = grp * g.spg + u * SSr, nseq = min(SSr, g.spg - u * SSr);
        __syncthreads();                                       // previous super-tile fully stored / tables staged
        stage_super<GP, L, AXIS>(reg, qkv_raw, g, hg, seq0, nseq, NCH, sc, sh);
        __syncthreads();
#pragma unroll 1
        for (int sub = 0; sub < g.nt; ++sub) {
            const int ls = sub * S_T + lsub;
            const bool active = ls < nseq;
            float outv[OCG], lse = 0.f;
#pragma unroll
            for (int k = 0; k < OCG; ++k) outv[k] = 0.f;
            if (active) {
                f2 qa[HQ], qb[HQ];
#pragma unroll
                for (int c = 0; c < HQ; ++c) {
                    const float q = reg[ls * RS + c * L + i];
                    qa[c] = (f2)(q * a_qk);
                    qb[c] = (f2)(q * a_qr);
                }
                const float* kp = reg + ls * RS + HQ * L;
                const float* vp = reg + ls * RS + GP * L;
                float m = -INFINITY;
                f2 l2 = (f2)(0.f), accv[GP], acce[GP];         // two partial sums each (even / odd columns)
#pragma unroll
                for (int c = 0; c < GP; ++c) { accv[c] = (f2)(0.f); acce[c] = (f2)(0.f); }
                // One 4-column chunk: logits -> running max / rescale -> probabilities -> P.V accumulation.
                auto chunk = [&](const f4 (&k4)[HQ], const f4 (&q4)[HQ], const f4 (&t4)[HQ], const f4 (&v4)[GP],
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
                if constexpr (GP == 2) {
                    // software pipeline: the 7 ds_read_b128 of chunk t+1 are issued before the arithmetic of chunk t
                    // (sched_barrier pins the order), so LDS latency hides under ~30 VALU issues instead of stalling
                    f4 ka[HQ], qa4[HQ], ta[HQ], va[GP], ea[GP], kb[HQ], qb4[HQ], tb[HQ], vb[GP], eb[GP];
                    fetch(0, ka, qa4, ta, va, ea);
#pragma unroll
                    for (int j0 = 0; j0 < L; j0 += 8) {
                        fetch(j0 + 4, kb, qb4, tb, vb, eb);
                        __builtin_amdgcn_sched_barrier(0);
                        chunk(ka, qa4, ta, va, ea);
                        if (j0 + 8 < L) fetch(j0 + 8, ka, qa4, ta, va, ea);
                        __builtin_amdgcn_sched_barrier(0);
                        chunk(kb, qb4, tb, vb, eb);
                    }
                } else {
#pragma unroll
                    for (int j0 = 0; j0 < L; j0 += 4) {
                        f4 k4[HQ], q4[HQ], t4[HQ], v4[GP], e4[GP];
                        fetch(j0, k4, q4, t4, v4, e4);
                        chunk(k4, q4, t4, v4, e4);
                    }
                }
                const float l = l2.x + l2.y;
                const float inv = 1.f / l;
                lse = m + __log2f(l);
#pragma unroll
                for (int c = 0; c < GP; ++c) {
                    outv[2 * c] = f_sv * (accv[c].x + accv[c].y) * inv;
                    outv[2 * c + 1] = f_sve * (acce[c].x + acce[c].y) * inv;
                }
            }
            // the L lanes that read sequence ls are the only readers of its region: when they share a wave the
            // in-order LDS pipe makes the overwrite safe without a barrier; L = 128 spans two waves.
            if (L > 64) __syncthreads();
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
        __syncthreads();
        store_super<GP, L, AXIS>(reg, 0, stacked, g.OC, hg * OCG, OCG, g, seq0, nseq);
        if (lse_out) store_super<GP, L, AXIS>(reg, NCH, lse_out, g.G, hg, 1, g, seq0, nseq);
    }
    if (out_partials) {
        float v[2 * OCG];
#pragma unroll
        for (int k = 0; k < OCG; ++k) { v[2 * k] = st_sum[k]; v[2 * k + 1] = st_sq[k]; }
        __syncthreads();
        block_sum<2 * OCG>(v, red, out_partials + ((size_t)blockIdx.x * g.OC + hg * OCG) * 2);
    }
}

template <int GP, int AXIS, int L>
__global__ __launch_bounds__(MEDT_THREADS) void logit_stats3_kernel(AxialGeom g, const float* __restrict__ qkv_raw,
                                                                    BnStats qs, const float* __restrict__ relative,
                                                                    GatePtrs gates, float* __restrict__ partials) {
    using F = Fast3<GP, L>;
    constexpr int HQ = F::HQ, NCH = F::NCH, RS = F::RS, CS = F::CS, S_T = F::S_T, SS = F::SS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* reg = smem;
    float* red = reg + SS * RS;
    float* tab = red + 256;
    const int grp = blockIdx.x / g.fparts, part = blockIdx.x - grp * g.fparts, hg = blockIdx.y;
    const float f_qr = gate(gates.f_qr), f_kr = gate(gates.f_kr);
    stage_tables_cols<GP>(tab, relative, L, 1.f);
    const float* sc = qs.scale + grp * 2 * g.C + hg * NCH;
    const float* sh = qs.shift + grp * 2 * g.C + hg * NCH;
    const int lsub = threadIdx.x / L, i = threadIdx.x % L;
    const int r = (3 - i) & 3;
    const float* tabr = tab + r * CS + (L - 1 - i - r);
    f2 s_qk = (f2)(0.f), q_qk = (f2)(0.f), s_qr = (f2)(0.f), q_qr = (f2)(0.f), s_kr = (f2)(0.f), q_kr = (f2)(0.f);
    const int SSr = S_T * g.nt;                               // runtime super-tile (g.nt <= NT sub-tiles)
    const int nsup = (g.spg + SSr - 1) / SSr;
    for (int u = part; u < nsup; u += g.fparts) {
        const int seq0 = grp * g.spg + u * SSr, nseq = min(SSr, g.spg - u * SSr);
        __syncthreads();
        stage_super<GP, L, AXIS>(reg, qkv_raw, g, hg, seq0, nseq, GP, sc, sh);          // q and k channels only
        __syncthreads();
#pragma unroll 1
        for (int sub = 0; sub < g.nt; ++sub) {
            const int ls = sub * S_T + lsub;
            if (ls < nseq) {
                f2 q[HQ];
#pragma unroll
                for (int c = 0; c < HQ; ++c) q[c] = (f2)(reg[ls * RS + c * L + i]);
                const float* kp = reg + ls * RS + HQ * L;
#pragma unroll
                for (int j0 = 0; j0 < L; j0 += 4) {
                    f2 alo = (f2)(0.f), ahi = (f2)(0.f), blo = (f2)(0.f), bhi = (f2)(0.f), clo = (f2)(0.f), chi = (f2)(0.f);
#pragma unroll
                    for (int c = 0; c < HQ; ++c) {
                        const f4 k4 = *reinterpret_cast<const f4*>(kp + c * L + j0);
                        const f4 q4 = *reinterpret_cast<const f4*>(tabr + (c * 4) * CS + j0);
                        const f4 t4 = *reinterpret_cast<const f4*>(tabr + ((HQ + c) * 4) * CS + j0);
                        alo = q[c] * k4.lo + alo;  ahi = q[c] * k4.hi + ahi;
                        blo = q[c] * q4.lo + blo;  bhi = q[c] * q4.hi + bhi;
                        clo = k4.lo * t4.lo + clo; chi = k4.hi * t4.hi + chi;
                    }
                    blo *= (f2)(f_qr); bhi *= (f2)(f_qr);
                    clo *= (f2)(f_kr); chi *= (f2)(f_kr);
                    s_qk += alo + ahi;  q_qk = alo * alo + (ahi * ahi + q_qk);
                    s_qr += blo + bhi;  q_qr = blo * blo + (bhi * bhi + q_qr);
                    s_kr += clo + chi;  q_kr = clo * clo + (chi * chi + q_kr);
                }
            }
        }
    }
    float acc[6] = {s_qk.x + s_qk.y, q_qk.x + q_qk.y, s_qr.x + s_qr.y, q_qr.x + q_qr.y, s_kr.x + s_kr.y, q_kr.x + q_kr.y};
    __syncthreads();
    float* dst = partials + ((size_t)blockIdx.x * g.SC + hg) * 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float s = wave_sum(acc[k]);
        if (lane == 0) red[wave * 6 + k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        const float s = (red[k] + red[6 + k]) + (red[12 + k] + red[18 + k]);
        dst[(size_t)(k >> 1) * g.G * 2 + (k & 1)] = s;
    }
}

#define MEDT_F3_CASE(KERNEL, GPv, Lv, ...)                                                                          \
    case GPv * 1024 + Lv * 2 + 0:                                                                                   \
        hipLaunchKernelGGL((KERNEL<GPv, 0, Lv>), grid, block, (Fast3<GPv, Lv>::lds_floats * sizeof(float)), s, __VA_ARGS__); \
        break;                                                                                                      \
    case GPv * 1024 + Lv * 2 + 1:                                                                                   \
        hipLaunchKernelGGL((KERNEL<GPv, 1, Lv>), grid, block, (Fast3<GPv, Lv>::lds_floats * sizeof(float)), s, __VA_ARGS__); \
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
            default: set_error(#KERNEL ": no instantiation"); return MEDT_EUNSUPPORTED;                             \
        }                                                                                                           \
        return launch_status(#KERNEL);                                                                              \
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

int axial_logit_stats_fast(const AxialGeom& g, const float* qkv_raw, BnStats qkv, const float* relative,
                           GatePtrs gates, float* partials, hipStream_t s) {
    if (g.fast3) MEDT_FAST3_DISPATCH(logit_stats3_kernel, g, qkv_raw, qkv, relative, gates, partials);
    MEDT_FAST_DISPATCH(logit_stats4_kernel, g, qkv_raw, qkv, relative, gates, partials);
}

int axial_attn_fwd_fast(const AxialGeom& g, const float* qkv_raw, BnStats qkv, BnStats sim, const float* relative,
                        GatePtrs gates, float* stacked, float* lse, float* out_partials, hipStream_t s) {
    if (g.fast3) MEDT_FAST3_DISPATCH(attn_fwd3_kernel, g, qkv_raw, qkv, sim, relative, gates, stacked, lse, out_partials);
    MEDT_FAST_DISPATCH(attn_fwd4_kernel, g, qkv_raw, qkv, sim, relative, gates, stacked, lse, out_partials);
}

}  // namespace medt
