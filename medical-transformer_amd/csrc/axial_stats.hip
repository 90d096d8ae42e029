// axial_stats.hip -- bn_similarity batch statistics WITHOUT forming the L x L logits.
//
// Reference lib/models/axialnet.py:157-167 feeds cat([qk, f_qr*qr, f_kr*kr]) of shape (B*, 3G, L, L) to
// BatchNorm2d(3G); in training mode the softmax therefore waits on sum / sum-of-squares of every logit
// channel over (B*, L, L).  Those six moments are separable in q and k (per head, hq = gp/2 channels):
//
//   sum_ij qk      = sum_c  Sq_c Sk_c                          Sq_c   = sum_i q_ci          (per sequence)
//   sum_ij qk^2    = sum_cc' Gq_cc' Gk_cc'                     Gq_cc' = sum_i q_ci q_c'i    (per sequence)
//   sum_ij qr      = sum_i sum_c   q_ci U_c[i]                 U_c[i]   = sum_{d=i}^{i+L-1} Rq[c,d]
//   sum_ij qr^2    = sum_i sum_cc' q_ci q_c'i T_cc'[i]         T_cc'[i] = sum_{d=i}^{i+L-1} Rq[c,d] Rq[c',d]
//   kr likewise with k, Rk and the key index j (kr[i,j] = sum_c k_cj Rk[c, j-i+L-1], :158).
//
// U and T are sliding-window sums of the relative table: per layer, batch independent (sim_tables_kernel,
// 2*(hq + hq(hq+1)/2)*L floats).  The statistics are then ONE read of the q and k channels (C*e*M bytes,
// O(M*hq^2) flops) instead of the O(M*L*hq) logit recompute.
//
// Work decomposition (wave64): a 256-thread workgroup owns 64 whole sequences of one head; lane = sequence, wave =
// quarter of the positions.  The per-sequence Gram / sum accumulators are lane-private, the position index is
// wave-uniform (tables come in through the scalar cache).  Height layers read NCHW directly (64 consecutive columns =
// 256 contiguous bytes per load); width layers stage [channel][sequence][position] chunks through LDS (coalesced along
// the row, read back transposed with an odd stride).
#include "axial_tiles.h"
#include "sim_tables.h"

namespace medt {

static inline int npairs(int hq) { return hq * (hq + 1) / 2; }

size_t sim_tables_floats(const AxialGeom& g) { return g.pos ? (size_t)2 * (g.hq + npairs(g.hq)) * g.L : 0; }
int sim_stats_parts(const AxialGeom& g) { return cdiv(g.spg, 64); }

// tables[(side*L + i)*NR + r]: r < HQ: U_r[i];  r >= HQ: pair (c <= c') in row-major order, off-diagonal pairs
// carry the factor 2 of the symmetric double sum.  side 0 = q rows of `relative`, 1 = k rows.
int sim_tables_blocks(const AxialGeom& g) { return g.pos ? 2 * (g.hq + npairs(g.hq)) : 0; }

// (BF: the storage type of qkv_raw at compile time -- a runtime flag inside ld_act puts every load into a branch of its own)
template <int HQ, bool POS, int AXIS, bool BF>
__device__ __forceinline__ void sim_stats_body(const AxialGeom& g, const float* __restrict__ qkv_raw,
                                               BnStats qs, const float* __restrict__ tables,
                                               GatePtrs gates, float* __restrict__ partials,
                                               int sparts, int pc_log) {
    constexpr int GP = 2 * HQ, NP = HQ * (HQ + 1) / 2, NR = HQ + NP, NV = 2 * NR, RND = 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = g.L, PC = 1 << pc_log, PCQ = PC >> 2, RS = PC + 1;
    const int grp = blockIdx.x / sparts, tile = blockIdx.x - grp * sparts, hg = blockIdx.y;
    const int seq0 = tile * 64, nseq = min(64, g.spg - seq0);
    const int lane = threadIdx.x & 63, slice = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool active = lane < nseq;
    const int NCH = 4 * HQ;                                   // channels per head: q | k | v
    int* seqoff = (int*)smem;                                 // [64] element offset of (sequence, channel 0 of the head, position 0)
    float* red = smem + 64;                                   // [3][RND][64] cross-slice reduction rounds, then [4][8]
    float* stage = smem + 64;                                 // AXIS == 1: [GP][64][PC+1] (dead before `red` is used)
    const int pstride = AXIS == 1 ? 1 : g.W;                  // element stride between positions of a sequence
    if (threadIdx.x < 64) {
        const int b = grp * g.spg + seq0 + min(lane, nseq - 1);
        const int n = b / g.Bo, sq = b - n * g.Bo;
        seqoff[lane] = (n * 2 * g.C + hg * NCH) * g.HW + (AXIS == 1 ? sq * g.W : sq);
    }
    __syncthreads();
    const int myoff = seqoff[lane];
    float sc[GP], sh[GP];
#pragma unroll
    for (int ch = 0; ch < GP; ++ch) {
        sc[ch] = qs.scale[grp * 2 * g.C + hg * NCH + ch];
        sh[ch] = qs.shift[grp * 2 * g.C + hg * NCH + ch];
    }
    // v[0..HQ) = Sq, [HQ..NR) = Gq pairs, [NR..NR+HQ) = Sk, [NR+HQ..NV) = Gk pairs
    float v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = 0.f;
    float r1q = 0.f, r2q = 0.f, r1k = 0.f, r2k = 0.f;
    for (int chunk0 = 0; chunk0 < L; chunk0 += PC) {
        if (AXIS == 1) {
            if ((L & 3) == 0) {
                // rows are contiguous: 16-byte (8-byte for bf16) loads, four independent ones in flight per thread
                const int pq_log = pc_log - 2, PC4 = PC >> 2;
#pragma unroll 4
                for (int e = threadIdx.x; e < GP * 64 * PC4; e += MEDT_THREADS) {
                    const int p4 = e & (PC4 - 1), ls = (e >> pq_log) & 63, ch = e >> (pq_log + 6);
                    const int i = chunk0 + 4 * p4;
                    if (ls < nseq && i < L) {
                        const size_t src = (size_t)seqoff[ls] + (size_t)ch * g.HW + i;
                        float4 v;
                        if (BF) {
                            const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(qkv_raw) + src);
                            v = make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                                            __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
                        } else {
                            v = *reinterpret_cast<const float4*>(qkv_raw + src);
                        }
                        float* dst = stage + (ch * 64 + ls) * RS + 4 * p4;
                        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
                    }
                }
            } else {
                for (int e = threadIdx.x; e < GP * 64 * PC; e += MEDT_THREADS) {
                    const int p = e & (PC - 1), ls = (e >> pc_log) & 63, ch = e >> (pc_log + 6);
                    const int i = chunk0 + p;
                    if (ls < nseq && i < L)
                        stage[(ch * 64 + ls) * RS + p] = ld_act(qkv_raw, (size_t)seqoff[ls] + (size_t)ch * g.HW + i, BF);
                }
            }
            __syncthreads();
        }
#pragma unroll 4
        for (int t = 0; t < PCQ; ++t) {
            const int p = slice * PCQ + t, i = chunk0 + p;        // wave-uniform
            if (i < L) {
                float x[GP];
#pragma unroll
                for (int ch = 0; ch < GP; ++ch) {
                    float raw;
                    if (AXIS == 1) raw = stage[(ch * 64 + lane) * RS + p];
                    else raw = ld_act(qkv_raw, (size_t)myoff + (size_t)ch * g.HW + (size_t)i * pstride, BF);
                    x[ch] = active ? fmaf(raw, sc[ch], sh[ch]) : 0.f;
                }
                const float* tq = tables + (size_t)i * NR;
                const float* tk = tables + (size_t)(L + i) * NR;
#pragma unroll
                for (int c = 0; c < HQ; ++c) {
                    v[c] += x[c];
                    v[NR + c] += x[HQ + c];
                    if (POS) {
                        r1q = fmaf(x[c], tq[c], r1q);
                        r1k = fmaf(x[HQ + c], tk[c], r1k);
                    }
                }
                int pr = 0;
#pragma unroll
                for (int a = 0; a < HQ; ++a) {
#pragma unroll
                    for (int b = a; b < HQ; ++b, ++pr) {
                        const float pq = x[a] * x[b], pk = x[HQ + a] * x[HQ + b];
                        v[HQ + pr] += pq;
                        v[NR + HQ + pr] += pk;
                        if (POS) {
                            r2q = fmaf(pq, tq[HQ + pr], r2q);
                            r2k = fmaf(pk, tk[HQ + pr], r2k);
                        }
                    }
                }
            }
        }
        if (AXIS == 1) __syncthreads();
    }
    // per-sequence totals: slices 1..3 hand their partial sums to slice 0, RND values per round
#pragma unroll
    for (int r0 = 0; r0 < NV; r0 += RND) {
        if (slice > 0) {
#pragma unroll
            for (int k = 0; k < RND; ++k)
                if (r0 + k < NV) red[((slice - 1) * RND + k) * 64 + lane] = v[r0 + k];
        }
        __syncthreads();
        if (slice == 0) {
#pragma unroll
            for (int k = 0; k < RND; ++k)
                if (r0 + k < NV) v[r0 + k] += (red[k * 64 + lane] + red[(RND + k) * 64 + lane]) + red[(2 * RND + k) * 64 + lane];
        }
        __syncthreads();
    }
    float acc[6];
    acc[0] = 0.f;
    acc[1] = 0.f;
    if (slice == 0) {
#pragma unroll
        for (int c = 0; c < HQ; ++c) acc[0] = fmaf(v[c], v[NR + c], acc[0]);
        int pr = 0;
#pragma unroll
        for (int a = 0; a < HQ; ++a) {
#pragma unroll
            for (int b = a; b < HQ; ++b, ++pr) {
                const float w = (b > a) ? 2.f : 1.f;
                acc[1] = fmaf(w * v[HQ + pr], v[NR + HQ + pr], acc[1]);
            }
        }
    }
    const int gseq = grp * g.spg + seq0 + min(lane, nseq - 1);
    const float f_qr = gate_at(gates.f_qr, gates.stride, gseq), f_kr = gate_at(gates.f_kr, gates.stride, gseq);
    acc[2] = f_qr * r1q;
    acc[3] = f_qr * f_qr * r2q;
    acc[4] = f_kr * r1k;
    acc[5] = f_kr * f_kr * r2k;
    constexpr int NA = POS ? 6 : 2;
    // (double from the cross-lane tree on, like every forward BatchNorm statistic: block_sum_d in medt_common.h)
    double* redd = reinterpret_cast<double*>(red);              // [4][8] doubles (64-float region at an even offset)
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const double s = wave_sum_d((double)acc[k]);
        if (lane == 0) redd[slice * 8 + k] = s;
    }
    __syncthreads();
    if (threadIdx.x < NA) {                                      // partial layout [grp][part][SC][2] doubles, channel x*G + hg
        const int k = threadIdx.x;
        const double s = (redd[k] + redd[8 + k]) + (redd[16 + k] + redd[24 + k]);
        reinterpret_cast<double*>(partials)[((size_t)blockIdx.x * g.SC + hg) * 2 + (size_t)(k >> 1) * g.G * 2 + (k & 1)] = s;
    }
}

template <int HQ, bool POS, int AXIS>
__global__ __launch_bounds__(MEDT_THREADS) void sim_stats_kernel(AxialGeom g, const float* __restrict__ qkv_raw,
                                                                 BnStats qs, const float* __restrict__ tables,
                                                                 GatePtrs gates, float* __restrict__ partials,
                                                                 int sparts, int pc_log) {
    if (g.bf16) sim_stats_body<HQ, POS, AXIS, true>(g, qkv_raw, qs, tables, gates, partials, sparts, pc_log);
    else sim_stats_body<HQ, POS, AXIS, false>(g, qkv_raw, qs, tables, gates, partials, sparts, pc_log);
}

// --------------------------------------------------------------------------- //
// Width layers (the sequence is the contiguous NCHW direction) without the LDS transpose: a ROW of 16 lanes owns one
// sequence, every lane V = L/16 consecutive positions (one 16-byte load per channel at L = 64), a wave moves 4 whole
// sequences per step with fully coalesced loads.  The table-weighted sums (qr, kr) only ever need the grand total, so
// they stay lane-private until the end; the per-sequence sums / Grams the qk moments need are all-reduced inside the
// 16-lane row with four DPP row rotations each (no LDS, no barrier).  Every lane of a row then holds the same
// per-sequence products: they are accumulated on all 16 and the total is scaled by 1/16 (exact).
// --------------------------------------------------------------------------- //
__device__ __forceinline__ float row16_allsum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
    return v;
}

template <int V>
__device__ __forceinline__ void load_run(const float* __restrict__ p, size_t idx, int bf, float (&out)[V]) {
    if (bf) {
        const unsigned short* q = reinterpret_cast<const unsigned short*>(p) + idx;
        if constexpr (V >= 4) {
#pragma unroll
            for (int h = 0; h < V / 4; ++h) {
                const uint2 r = reinterpret_cast<const uint2*>(q)[h];
                out[4 * h] = __uint_as_float(r.x << 16); out[4 * h + 1] = __uint_as_float(r.x & 0xffff0000u);
                out[4 * h + 2] = __uint_as_float(r.y << 16); out[4 * h + 3] = __uint_as_float(r.y & 0xffff0000u);
            }
        } else {
#pragma unroll
            for (int v = 0; v < V; ++v) out[v] = bf16_bits_to_f32(q[v]);
        }
    } else {
        const float* q = p + idx;
        if constexpr (V >= 4) {
#pragma unroll
            for (int h = 0; h < V / 4; ++h) {
                const float4 r = reinterpret_cast<const float4*>(q)[h];
                out[4 * h] = r.x; out[4 * h + 1] = r.y; out[4 * h + 2] = r.z; out[4 * h + 3] = r.w;
            }
        } else if constexpr (V == 2) {
            const float2 r = *reinterpret_cast<const float2*>(q);
            out[0] = r.x; out[1] = r.y;
        } else {
            out[0] = q[0];
        }
    }
}

template <int HQ, bool POS, int V>
__global__ __launch_bounds__(MEDT_THREADS) void sim_stats_rows_kernel(AxialGeom g, const float* __restrict__ qkv_raw,
                                                                      BnStats qs, const float* __restrict__ tables,
                                                                      GatePtrs gates, float* __restrict__ partials,
                                                                      int sparts) {
    constexpr int GP = 2 * HQ, NP = HQ * (HQ + 1) / 2, NR = HQ + NP, NCH = 4 * HQ, L = 16 * V;
    MEDT_STATIC_SHARED double red[MEDT_WAVES * 8];
    const int grp = blockIdx.x / sparts, tile = blockIdx.x - grp * sparts, hg = blockIdx.y;
    const int seq0 = tile * 64, nseq = min(64, g.spg - seq0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane >> 4, p0 = (lane & 15) * V;
    float sc[GP], sh[GP];
#pragma unroll
    for (int ch = 0; ch < GP; ++ch) {
        sc[ch] = qs.scale[grp * 2 * g.C + hg * NCH + ch];
        sh[ch] = qs.shift[grp * 2 * g.C + hg * NCH + ch];
    }
    float tq[POS ? NR : 1][V], tk[POS ? NR : 1][V];         // this lane's positions of the sliding-window tables
    if (POS) {
#pragma unroll
        for (int v = 0; v < V; ++v)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                tq[r][v] = tables[(size_t)(p0 + v) * NR + r];
                tk[r][v] = tables[(size_t)(L + p0 + v) * NR + r];
            }
    }
    float qk1 = 0.f, qk2 = 0.f, g1q = 0.f, g2q = 0.f, g1k = 0.f, g2k = 0.f;
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {
        float r1q = 0.f, r2q = 0.f, r1k = 0.f, r2k = 0.f;
        const int sl = (it * MEDT_WAVES + wave) * 4 + row;                  // sequence of the tile owned by this row
        const bool active = sl < nseq;
        const int b = grp * g.spg + seq0 + (active ? sl : 0);
        const int n = b / g.Bo, sq = b - n * g.Bo;
        const size_t base = ((size_t)n * 2 * g.C + hg * NCH) * g.HW + (size_t)sq * g.W + p0;
        float x[GP][V];
#pragma unroll
        for (int ch = 0; ch < GP; ++ch) load_run<V>(qkv_raw, base + (size_t)ch * g.HW, g.bf16, x[ch]);
        float s[2 * NR];                                                    // Sq | Gq pairs | Sk | Gk pairs of this lane's run
#pragma unroll
        for (int k = 0; k < 2 * NR; ++k) s[k] = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float y[GP];
#pragma unroll
            for (int ch = 0; ch < GP; ++ch) y[ch] = active ? fmaf(x[ch][v], sc[ch], sh[ch]) : 0.f;
#pragma unroll
            for (int c = 0; c < HQ; ++c) {
                s[c] += y[c];
                s[NR + c] += y[HQ + c];
                if (POS) {
                    r1q = fmaf(y[c], tq[c][v], r1q);
                    r1k = fmaf(y[HQ + c], tk[c][v], r1k);
                }
            }
            int pr = 0;
#pragma unroll
            for (int a = 0; a < HQ; ++a)
#pragma unroll
                for (int c = a; c < HQ; ++c, ++pr) {
                    const float pq = y[a] * y[c], pk = y[HQ + a] * y[HQ + c];
                    s[HQ + pr] += pq;
                    s[NR + HQ + pr] += pk;
                    if (POS) {
                        r2q = fmaf(pq, tq[HQ + pr][v], r2q);
                        r2k = fmaf(pk, tk[HQ + pr][v], r2k);
                    }
                }
        }
#pragma unroll
        for (int k = 0; k < 2 * NR; ++k) s[k] = row16_allsum(s[k]);         // per-sequence totals, in all 16 lanes
#pragma unroll
        for (int c = 0; c < HQ; ++c) qk1 = fmaf(s[c], s[NR + c], qk1);
        int pr = 0;
#pragma unroll
        for (int a = 0; a < HQ; ++a)
#pragma unroll
            for (int c = a; c < HQ; ++c, ++pr) qk2 = fmaf((c > a ? 2.f : 1.f) * s[HQ + pr], s[NR + HQ + pr], qk2);
        if (POS) {                                                          // the sequence's gates scale its qr / kr moments
            const float f_qr = gate_at(gates.f_qr, gates.stride, b), f_kr = gate_at(gates.f_kr, gates.stride, b);
            g1q = fmaf(f_qr, r1q, g1q);
            g2q = fmaf(f_qr * f_qr, r2q, g2q);
            g1k = fmaf(f_kr, r1k, g1k);
            g2k = fmaf(f_kr * f_kr, r2k, g2k);
        }
    }
    float acc[6] = {qk1 * 0.0625f, qk2 * 0.0625f, g1q, g2q, g1k, g2k};
    constexpr int NA = POS ? 6 : 2;
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const double t = wave_sum_d((double)acc[k]);
        if (lane == 0) red[wave * 8 + k] = t;
    }
    __syncthreads();
    if (threadIdx.x < NA) {
        const int k = threadIdx.x;
        const double t = (red[k] + red[8 + k]) + (red[16 + k] + red[24 + k]);
        reinterpret_cast<double*>(partials)[((size_t)blockIdx.x * g.SC + hg) * 2 + (size_t)(k >> 1) * g.G * 2 + (k & 1)] = t;
    }
}

static int stats_chunk_log(const AxialGeom& g) {
    int cap = 128 / g.gp;                 // [gp q|k channels][64][PC+1] floats <= ~33 KB
    if (cap < 4) cap = 4;
    int pc = 4, lg = 2;
    while (pc < g.L && pc * 2 <= cap) { pc *= 2; ++lg; }
    return lg;
}

int axial_logit_stats(const AxialGeom& g, const float* qkv_raw, BnStats qkv, const float* relative, GatePtrs gates,
                      const float* tables, float* partials, hipStream_t s) {
    const int sparts = sim_stats_parts(g), lg = stats_chunk_log(g);
    if (g.pos && !tables) { set_error("sim_stats: no tables"); return MEDT_EINVAL; }
    if (g.axis == 1 && g.hq <= 2 && (g.L == 16 || g.L == 32 || g.L == 64 || g.L == 128) && (g.W & 3) == 0) {
        // rows-of-16-lanes kernel: no LDS transpose (width layers with the model's power-of-two lengths, hq <= 2)
        const dim3 gridr(g.groups * sparts, g.G), blockr(MEDT_THREADS);
#define MEDT_SR(HQv, POSv, Vv) \
    hipLaunchKernelGGL((sim_stats_rows_kernel<HQv, POSv, Vv>), gridr, blockr, 0, s, g, qkv_raw, qkv, tables, gates, partials, sparts)
#define MEDT_SR_V(HQv, POSv)                                                                                          \
    do {                                                                                                              \
        if (g.L == 16) MEDT_SR(HQv, POSv, 1); else if (g.L == 32) MEDT_SR(HQv, POSv, 2);                               \
        else if (g.L == 64) MEDT_SR(HQv, POSv, 4); else MEDT_SR(HQv, POSv, 8);                                         \
    } while (0)
        if (g.hq == 1) { if (g.pos) MEDT_SR_V(1, true); else MEDT_SR_V(1, false); }
        else { if (g.pos) MEDT_SR_V(2, true); else MEDT_SR_V(2, false); }
#undef MEDT_SR_V
#undef MEDT_SR
        return launch_status("sim_stats_rows_kernel");
    }
    const size_t stage = g.axis == 1 ? (size_t)g.gp * 64 * ((1 << lg) + 1) : 0, red = 3 * 16 * 64;
    const size_t lds = (64 + (stage > red ? stage : red)) * sizeof(float);
    const dim3 grid(g.groups * sparts, g.G), block(MEDT_THREADS);
#define MEDT_SS(HQv, POSv, AXv)                                                                                       \
    hipLaunchKernelGGL((sim_stats_kernel<HQv, POSv, AXv>), grid, block, lds, s, g, qkv_raw, qkv, tables, gates, partials, \
                       sparts, lg)
    switch (g.hq * 4 + g.pos * 2 + g.axis) {
        case 1 * 4 + 0: MEDT_SS(1, false, 0); break;
        case 1 * 4 + 1: MEDT_SS(1, false, 1); break;
        case 1 * 4 + 2: MEDT_SS(1, true, 0); break;
        case 1 * 4 + 3: MEDT_SS(1, true, 1); break;
        case 2 * 4 + 0: MEDT_SS(2, false, 0); break;
        case 2 * 4 + 1: MEDT_SS(2, false, 1); break;
        case 2 * 4 + 2: MEDT_SS(2, true, 0); break;
        case 2 * 4 + 3: MEDT_SS(2, true, 1); break;
        case 4 * 4 + 0: MEDT_SS(4, false, 0); break;
        case 4 * 4 + 1: MEDT_SS(4, false, 1); break;
        case 4 * 4 + 2: MEDT_SS(4, true, 0); break;
        case 4 * 4 + 3: MEDT_SS(4, true, 1); break;
        case 8 * 4 + 0: MEDT_SS(8, false, 0); break;
        case 8 * 4 + 1: MEDT_SS(8, false, 1); break;
        case 8 * 4 + 2: MEDT_SS(8, true, 0); break;
        case 8 * 4 + 3: MEDT_SS(8, true, 1); break;
        default: set_error("sim_stats: no instantiation for hq=%d", g.hq); return MEDT_EUNSUPPORTED;
    }
#undef MEDT_SS
    return launch_status("sim_stats_kernel");
}

}  // namespace medt
