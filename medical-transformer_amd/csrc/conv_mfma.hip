// conv_mfma.hip -- fp32 matrix-core (MFMA) implicit-GEMM convolutions for the channel-heavy layers:
// MedT's local stem (64 <-> 128 channels, 3x3: 53 % of the model's FLOPs), the 3x3 decoders and the
// wide 1x1 convolutions.  gfx950 has no TF32: v_mfma_f32_16x16x4_f32 is an exact-fp32 FMA chain at the vector
// peak rate, so parity with the VALU path is rounding-order only.
//
//   forward        Y[o, q]  = sum_k W[o, k] * Xcol[k, q]        k = (c,kh,kw),  q = (n,ho,wo)
//   backward-data  (stride 1) = forward of dY with the weights transposed and tap-flipped (flip_weights_kernel)
//   weight grad    dW[o, k] = sum_q dY[o, q] * Xcol[k, q]       (same tiles, reduction over q)
//
// Workgroup = 4 waves = one 64(o) x 64(q) tile; wave w owns rows 16w..16w+15 and four 16x16 accumulators.
// Operands go through LDS in 16-deep K slices: A[64][16] straight from the (contiguous) weight rows,
// B[16][64] gathered from NCHW with the column's (n,ho,wo) decoded once per lane.
// MFMA fragment maps (cdna_hip_programming.md section 3): A: lane l = A[l&15][l>>4]; B: lane l = B[l>>4][l&15];
// D: reg r of lane l = D[(l>>4)*4 + r][l&15].
#include "defer.h"
#include <algorithm>
#include <vector>
#include <stdlib.h>
#include <stdio.h>

namespace medt {

typedef medt_f4 f32x4;

// Measured on MI355X (profiles/, round 1): the MFMA tile kernel beats the VALU direct kernel only for the 3x3
// layers with a deep contraction AND enough 64x64 tiles to occupy the chip -- conv2_p 64->128 (55 vs 76 us),
// conv3_p 128->64 (108 vs 137 us) and their backward twins; on the 2x2/4x4 maps of the deep LoGo layers
// (<= 16 tiles) it is 2-4x slower, and for 1x1 convolutions it ties.  Hence:
bool conv_use_mfma(int Cin, int Cout, int K, int stride, long positions) {
    static const bool off = [] { const char* e = getenv("MEDT_DISABLE_MFMA"); return e && e[0] == '1'; }();
    // A/B switch (scripts/conv_ab.py -> profiles/r02_conv_ab.json): the matrix-core tile kernel wherever it is legal
    static const bool force = false;
    if (force && !off) return (K == 1 || K == 3) && (stride == 1 || stride == 2);
    const long tiles = ((positions + 63) / 64) * ((Cout + 63) / 64);
    if (off || Cout < 32 || Cin * K * K < 256 || (stride != 1 && stride != 2)) return false;
    if (K == 3 && tiles >= 128) return true;
    // few tiles (2x2 / 4x4 maps of the deep LoGo layers, <= 1024 positions): split-K over workgroups + epilogue
    static const long few_pos = 2048L;
    return (K == 3 || K == 1) && tiles < 128 && positions <= few_pos && Cin * K * K >= 512;
}

int conv_mfma_parts_per_group(int N, int groups, int HoWo) { return cdiv((N / groups) * HoWo, 64); }

template <int K>
__global__ __launch_bounds__(MEDT_THREADS) void conv_mfma_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
    float* __restrict__ partials, int Cin, int H, int W, int Cout, int Ho, int Wo, int stride, int pad, int relu,
    int npg, int kchunk, float* __restrict__ ksplit_out) {
    constexpr int KK = K * K;
    MEDT_STATIC_SHARED float As[64][17];
    MEDT_STATIC_SHARED float Bs[16][65];
    const int HoWo = Ho * Wo, per_group = npg * HoWo, ppg = (per_group + 63) / 64, Ktot = Cin * KK;
    const int grp = blockIdx.x / ppg, part = blockIdx.x - grp * ppg, o0 = blockIdx.y * 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // B staging: this lane always fills column j = lane; decode its output position once
    const int qj = part * 64 + lane;
    const bool jok = qj < per_group;
    const int nj = grp * npg + (jok ? qj / HoWo : 0), pj = jok ? qj % HoWo : 0;
    const int hbj = (pj / Wo) * stride - pad, wbj = (pj % Wo) * stride - pad;
    const float* xj = x + (size_t)nj * Cin * H * W;
    // A staging: element (row ar + 16*pass, column ak)
    const int ak = threadIdx.x & 15, ar = threadIdx.x >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4)(0.f);
    // split-K: slice blockIdx.z owns k in [kbeg, kend) and writes its raw partial tile to ksplit_out (the few-tile
    // layers on 2x2 / 4x4 maps would otherwise run on a handful of CUs); conv_splitk_epilogue sums the slices.
    const int kbeg = blockIdx.z * kchunk, kend = min(Ktot, kbeg + kchunk);
    // Software pipeline: the 8 global loads of K-slice s+1 are issued before the 16 MFMAs of slice s and only written
    // to LDS after them, so their latency hides under the matrix work (one workgroup per CU has little else to hide it).
    float ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int o = o0 + ar + 16 * ps, k = k0 + ak;
            ra[ps] = (o < Cout && k < kend) ? w[(size_t)o * Ktot + k] : 0.f;
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int k = k0 + wv + 4 * ps;
            float v = 0.f;
            if (jok && k < kend) {
                const int c = k / KK, t = k - c * KK;
                const int h = hbj + t / K, ww = wbj + t % K;
                if (h >= 0 && h < H && ww >= 0 && ww < W) v = xj[((size_t)c * H + h) * W + ww];
            }
            rb[ps] = v;
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            As[ar + 16 * ps][ak] = ra[ps];
            Bs[wv + 4 * ps][lane] = rb[ps];
        }
        __syncthreads();
        if (k0 + 16 < kend) fetch(k0 + 16);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float a = As[16 * wv + (lane & 15)][ks * 4 + (lane >> 4)];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float b = Bs[ks * 4 + (lane >> 4)][t * 16 + (lane & 15)];
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // epilogue: D[(lane>>4)*4 + r][lane&15] of tile t  ->  o = o0 + 16*wv + (lane>>4)*4 + r,  q = part*64 + t*16 + (lane&15)
    if (ksplit_out) {
        float* dst = ksplit_out + (size_t)blockIdx.z * gridDim.x * 64 * Cout;       // [slice][global position][Cout]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const size_t qg = (size_t)blockIdx.x * 64 + t * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = o0 + 16 * wv + (lane >> 4) * 4 + r;
                if (o < Cout) dst[qg * Cout + o] = acc[t][r];
            }
        }
        return;
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int q = part * 64 + t * 16 + (lane & 15);
        const bool ok = q < per_group;
        const int n = grp * npg + (ok ? q / HoWo : 0), p = ok ? q % HoWo : 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = o0 + 16 * wv + (lane >> 4) * 4 + r;
            if (ok && o < Cout) {
                float v = acc[t][r] + (bias ? bias[o] : 0.f);
                s1[r] += v;
                s2[r] = fmaf(v, v, s2[r]);
                y[((size_t)n * Cout + o) * HoWo + p] = relu ? fmaxf(v, 0.f) : v;
            }
        }
    }
    if (partials) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double a = s1[r], b = s2[r];                       // (double from the cross-lane tree on: block_sum_d, medt_common.h)
#pragma unroll
            for (int m = 8; m > 0; m >>= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }   // over lane&15
            const int o = o0 + 16 * wv + (lane >> 4) * 4 + r;
            if ((lane & 15) == 0 && o < Cout) {
                double* dst = reinterpret_cast<double*>(partials) + ((size_t)blockIdx.x * Cout + o) * 2;
                dst[0] = a;
                dst[1] = b;
            }
        }
    }
}

// --------------------------------------------------------------------------- //
// 3x3 / stride 1 / pad 1 on 16-wide maps: MedT's local stem (conv2_p 64->128, conv3_p 128->64 on the 16x16 maps of the
// 32-px patches -- 53 % of the model's FLOPs), forward, backward-data (flipped weights) and weight gradient.
// The generic tile kernel above makes one global round trip per 16-deep K slice (16 MFMAs per wave) and decodes
// (c,kh,kw) per gathered element; one workgroup per CU cannot hide that latency (round 2: 58 / 86 us for 2.4 GFLOP
// = 27 / 18 % of the fp32 MFMA peak).  Here a workgroup owns 4 full rows of one image (64 positions) and walks the
// input channels in chunks of 16: the chunk's 6 x 18 halo patch and its 64 x 144 weight slab are staged in LDS once
// and all 9 taps run from them -- 144 MFMAs per wave per global round trip, no index arithmetic in the loop, each
// input element fetched once instead of 9 times.  LDS strides are chosen so that every fragment read is conflict-free:
//   weights  As[o][148]: bank = 20 (l&15) + 9 (l>>4) + const  -- 16 multiples of 4 plus residues 0..3 mod 4
//   patch    Ps[c][112]: bank = 48 (l>>4) + (l&15) + const    -- four 16-bank windows
// --------------------------------------------------------------------------- //
constexpr int R16_AST = 148, R16_CST = 112, R16_RST = 18;
constexpr int R16_DST = 68, R16_WCST = 132;

bool conv_rows16_ok(int Cin, int H, int W, int K, int stride, int pad) {
    static const bool off = [] { const char* e = getenv("MEDT_CONV_ROWS16"); return e && e[0] == '0'; }();
    return !off && K == 3 && stride == 1 && pad == 1 && W == 16 && H % 4 == 0 && Cin % 16 == 0;
}

// OT = 64: a wave owns a 16-channel row block and all four position tiles of the workgroup's 4 rows (round 2).
// OT = 32 (round 5): half the output channels per workgroup -- wave w owns row block (w & 1) and the two position tiles 2 (w >> 1),
// + 1 -- so conv3_p (64 output channels: 256 workgroups = ONE per CU, one wave per SIMD, the matrix pipe idle through every staging
// phase) becomes 512 workgroups of 26 KB LDS, conv2_p 1024: several waves per SIMD take turns on the matrix pipe.
template <int OT>
__global__ __launch_bounds__(MEDT_THREADS) void conv3x3_rows16_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
    float* __restrict__ partials, int Cin, int H, int Cout, int relu) {
    constexpr int RB = OT / 16;                                    // 16-channel row blocks per workgroup
    constexpr int NTQ = OT == 64 ? 4 : 2;                          // position tiles (image rows) per wave
    MEDT_STATIC_SHARED float As[OT * R16_AST];
    MEDT_STATIC_SHARED float Ps[16 * R16_CST];
    MEDT_STATIC_SHARED double Rd[2][2][16][2];                     // OT = 32: BatchNorm partials of the upper tile pair, per row block
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int rb = OT == 64 ? wv : (wv & 1), t0 = OT == 64 ? 0 : 2 * (wv >> 1);
    const int tpi = H >> 2;                                        // 4-row tiles per image
    const int n = blockIdx.x / tpi, r0 = (blockIdx.x - n * tpi) * 4, o0 = blockIdx.y * OT;
    const int Ktot = Cin * 9, HW = H * 16;
    if (tid < 192) {                                               // the padding columns (-1 and 16) stay zero
        const int c = tid / 12, rem = tid - c * 12;
        Ps[c * R16_CST + (rem >> 1) * R16_RST + (rem & 1) * 17] = 0.f;
    }
    const int ar = tid >> 4, ae = tid & 15;                        // weight slab: rows ar + 16 j, columns ae + 16 m
    const float* xn = x + (size_t)n * Cin * HW;
    int poff[6], goff[6];                                          // patch element u = tid + 256 i: (c, row, col)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int u = tid + MEDT_THREADS * i, c = u / 96, rem = u - c * 96, row = rem >> 4, col = rem & 15;
        const int gr = r0 - 1 + row;
        poff[i] = c * R16_CST + row * R16_RST + 1 + col;
        goff[i] = (gr >= 0 && gr < H) ? c * HW + gr * 16 + col : -1;       // rows above / below the image stay zero
    }
    float ra[RB * 9], rp[6];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int o = min(o0 + ar + 16 * j, Cout - 1);          // rows past Cout: their accumulators are never stored
            const float* wr = w + (size_t)o * Ktot + c0 * 9 + ae;
#pragma unroll
            for (int m = 0; m < 9; ++m) ra[j * 9 + m] = wr[16 * m];
        }
        const float* xc = xn + (size_t)c0 * HW;
#pragma unroll
        for (int i = 0; i < 6; ++i) {                               // unconditional loads (no branch per element), then select
            const float v = xc[max(goff[i], 0)];
            rp[i] = goff[i] >= 0 ? v : 0.f;
        }
    };
    f32x4 acc[NTQ];
#pragma unroll
    for (int t = 0; t < NTQ; ++t) acc[t] = (f32x4)(0.f);
    const float* arow = As + (16 * rb + (lane & 15)) * R16_AST + (lane >> 4) * 9;
    const float* prow = Ps + (lane >> 4) * R16_CST + (lane & 15) + t0 * R16_RST;
    const int nch = Cin >> 4;
    fetch(0);
    for (int ch = 0; ch < nch; ++ch) {
#pragma unroll
        for (int j = 0; j < RB; ++j)
#pragma unroll
            for (int m = 0; m < 9; ++m) As[(ar + 16 * j) * R16_AST + ae + 16 * m] = ra[j * 9 + m];
#pragma unroll
        for (int i = 0; i < 6; ++i) Ps[poff[i]] = rp[i];
        __syncthreads();
        if (ch + 1 < nch) fetch((ch + 1) * 16);                    // flies during the MFMAs below
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float a[9], b[NTQ + 2][3];
#pragma unroll
            for (int t = 0; t < 9; ++t) a[t] = arow[ks * 36 + t];
#pragma unroll
            for (int r = 0; r < NTQ + 2; ++r)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) b[r][kw] = prow[ks * 4 * R16_CST + r * R16_RST + kw];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int tq = 0; tq < NTQ; ++tq)
                        acc[tq] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kh * 3 + kw], b[tq + kh][kw], acc[tq], 0, 0, 0);
        }
        __syncthreads();
    }
    // D[(lane>>4)*4 + r][lane&15] of tile tq  ->  o = o0 + 16 rb + (lane>>4)*4 + r,  position (r0 + t0 + tq, lane&15)
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = o0 + 16 * rb + (lane >> 4) * 4 + r;
        if (o < Cout) {
            const float bo = bias ? bias[o] : 0.f;
            float* yo = y + ((size_t)n * Cout + o) * HW + (r0 + t0) * 16 + (lane & 15);
#pragma unroll
            for (int tq = 0; tq < NTQ; ++tq) {
                const float v = acc[tq][r] + bo;
                s1[r] += v;
                s2[r] = fmaf(v, v, s2[r]);
                yo[tq * 16] = relu ? fmaxf(v, 0.f) : v;
            }
        }
    }
    if (partials) {
        double pa[4], pb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double a = s1[r], b = s2[r];                       // (double from the cross-lane tree on: block_sum_d, medt_common.h)
#pragma unroll
            for (int m = 8; m > 0; m >>= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }   // over lane&15
            pa[r] = a;
            pb[r] = b;
        }
        if (OT == 32) {                                        // the two waves of a row block each hold half of the 64 positions
            if (wv >= 2 && (lane & 15) == 0)
#pragma unroll
                for (int r = 0; r < 4; ++r) { Rd[rb][0][(lane >> 4) * 4 + r][0] = pa[r]; Rd[rb][0][(lane >> 4) * 4 + r][1] = pb[r]; }
            __syncthreads();
            if (wv < 2)
#pragma unroll
                for (int r = 0; r < 4; ++r) { pa[r] += Rd[rb][0][(lane >> 4) * 4 + r][0]; pb[r] += Rd[rb][0][(lane >> 4) * 4 + r][1]; }
        }
        if (OT == 64 || wv < 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = o0 + 16 * rb + (lane >> 4) * 4 + r;
                if ((lane & 15) == 0 && o < Cout) {
                    double* dst = reinterpret_cast<double*>(partials) + ((size_t)blockIdx.x * Cout + o) * 2;
                    dst[0] = pa[r];
                    dst[1] = pb[r];
                }
            }
        }
    }
}

// --------------------------------------------------------------------------- //
// 7x7 / stride 2 / pad 3 with 3 input channels: the stems (conv1_p 3 -> 64 on the 64 patch-images of MedT's local branch: 308 MFLOP
// that took 29 us on the VALU kernel -- every lane walked 8 x 147 taps with scalar-loaded weights -- on the CRITICAL forward chain).
// Round 5, the LDS-patch scheme of the 3x3 kernel above: a workgroup owns 8 MFMA tiles = 128 output positions of one image (8 output
// rows of a 16-wide map, or 2 rows of a 64-wide one) x 64 output channels; the zero-padded input patch (3 x (2 R + 5) x (W + 6)) and the
// 64 x 147 weight slab are staged in LDS ONCE and all 37 k-steps run from them -- 296 MFMAs per wave, one global round trip.
//   k = (c, kh, kw) = 4 ks + (lane >> 4);  B fragment of tile (row tr, columns 16 tc ..): Ps[c][2 tr + kh][2 (16 tc + (lane & 15)) + kw]
// --------------------------------------------------------------------------- //
constexpr int S7_LDW = 149;                    // weight-slab row stride (147 taps + the zero 148th, odd: conflict-free fragment reads)
constexpr int S7_OT = 32;                      // output channels per workgroup: 2 row blocks x 2 tile halves over the four waves
bool conv_stem7_ok(int Cin, int H, int W, int Cout, int K, int stride, int pad) {
    static const bool off = [] { const char* e = getenv("MEDT_CONV_STEM7"); return e && e[0] == '0'; }();
    if (off || K != 7 || stride != 2 || pad != 3 || Cin != 3 || Cout < 32 || (H & 1) || (W & 1)) return false;
    const int Ho = H / 2, Wo = W / 2;
    if (Wo % 16 || (Ho * Wo) % 128 || Wo > 128) return false;
    const int R = 128 / Wo;                    // output rows per workgroup
    return (size_t)(3 * (2 * R + 5) * (W + 6) + S7_OT * S7_LDW) * sizeof(float) <= 64 * 1024;
}
int conv_stem7_parts_per_group(int N, int groups, int HoWo) { return (N / groups) * HoWo / 128; }

// A workgroup = 128 output positions of one image x 32 output channels: wave w owns the 16-channel row block (w & 1) and the
// four MFMA tiles 4 (w >> 1) .. + 3 (the first version -- 64 channels, a wave = a row block x all eight tiles -- put 128 workgroups
// on 256 CUs and staged twice the weight slab per workgroup).
__global__ __launch_bounds__(MEDT_THREADS) void conv_stem7_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
    float* __restrict__ partials, int H, int W, int Cout, int relu, unsigned inv_pw, unsigned inv_phpw) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    MEDT_STATIC_SHARED double Rd7[2][16][2];                             // BatchNorm partials of the upper tile half, per row block
    const int Ho = H >> 1, Wo = W >> 1, HoWo = Ho * Wo, R = 128 / Wo, TC = Wo >> 4;      // rows per workgroup, tiles per row
    const int PH = 2 * R + 5, PW = W + 6;
    float* As = smem;                          // [S7_OT][S7_LDW]
    float* Ps = smem + S7_OT * S7_LDW;         // [3][PH][PW]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int rb = wv & 1, t0 = 4 * (wv >> 1);
    const int ppi = HoWo / 128;                // workgroups per image
    const int n = blockIdx.x / ppi, ho0 = (blockIdx.x - n * ppi) * R, o0 = blockIdx.y * S7_OT;
    // ---- stage the weight slab (rows past Cout and column 147 are zero) and the zero-padded patch.  Every global load of the stage is
    // issued BEFORE the first LDS store (unconditional clamped addresses + select): a rolled `for (e ...) lds[e] = cond ? g[..] : 0` loop
    // is one global round trip per element (measured: the first version of this kernel took 28 us, 20 of them in these two loops)
    const float* xn = x + (size_t)n * 3 * H * W;
    const int PE = 3 * PH * PW;
    constexpr int WPT = (S7_OT * 147 + MEDT_THREADS - 1) / MEDT_THREADS;   // 19 slab elements per thread
    constexpr int PPT = 12;                                                 // patch elements per thread and batch
    float wreg[WPT], preg[PPT];
    // the slab's rows are one contiguous block of w: element g of the block, no index arithmetic in front of the loads
    const char* wblk = reinterpret_cast<const char*>(w + (size_t)o0 * 147);
    const unsigned wvalid = (unsigned)min(S7_OT, Cout - o0) * 147u;
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const unsigned g = (unsigned)(tid + MEDT_THREADS * i);
        const float v = *reinterpret_cast<const float*>(wblk + (size_t)((g < wvalid ? g : 0u) * 4u));
        wreg[i] = g < wvalid ? v : 0.f;
    }
    auto patch_load = [&](int base) {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int e = base + tid + MEDT_THREADS * i;
            const int ec = e < PE ? e : 0;
            // (e < 2^16: floor(e / d) == mulhi(e, floor(2^32 / d) + 1) -- no runtime integer division per element)
            const int c = (int)__umulhi((unsigned)ec, inv_phpw), rem = ec - c * PH * PW;
            const int r = (int)__umulhi((unsigned)rem, inv_pw), col = rem - r * PW;
            const int gh = 2 * ho0 - 3 + r, gw = col - 3;
            const bool ok = e < PE && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            const float v = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xn) + (size_t)((ok ? (unsigned)((c * H + gh) * W + gw) : 0u) * 4u));
            preg[i] = ok ? v : 0.f;
        }
    };
    patch_load(0);
    MEDT_SCHED_FENCE();
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int g = tid + MEDT_THREADS * i, o = g / 147, k = g - o * 147;
        if (g < S7_OT * 147) As[o * S7_LDW + k] = wreg[i];
    }
    if (tid < S7_OT) As[tid * S7_LDW + 147] = 0.f;                          // the zero 148th column (k-step 36's fourth lane group)
    for (int base = 0; base < PE; base += PPT * MEDT_THREADS) {
        if (base) { patch_load(base); MEDT_SCHED_FENCE(); }
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int e = base + tid + MEDT_THREADS * i;
            if (e < PE) Ps[e] = preg[i];
        }
    }
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4)(0.f);
    const float* arow = As + (16 * rb + (lane & 15)) * S7_LDW + (lane >> 4);
    // tile t of the workgroup: output row ho0 + t / TC, columns 16 (t % TC) ...; this lane's column inside the patch
    int toff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) toff[t] = 2 * ((t0 + t) / TC) * PW + 2 * (16 * ((t0 + t) % TC) + (lane & 15));
    // k-step ks + 1's fragments are read from LDS while step ks multiplies
    auto frags = [&](int ks, float& a, float (&b)[4]) {
        const int k = min(4 * ks + (lane >> 4), 146);             // (k = 147: the slab's zero column multiplies a valid address)
        const int c = k / 49, r49 = k - c * 49, kh = r49 / 7, kw = r49 - kh * 7;
        a = arow[4 * ks];
        const float* pk = Ps + (c * PH + kh) * PW + kw;
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = pk[toff[t]];
    };
    float a0, b0[4], a1, b1[4];
    frags(0, a0, b0);
#pragma unroll 1
    for (int ks = 0; ks < 36; ks += 2) {
        frags(ks + 1, a1, b1);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[t], acc[t], 0, 0, 0);
        frags(ks + 2, a0, b0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1[t], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[t], acc[t], 0, 0, 0);      // k-step 36
#ifndef MEDT_LANE_EMU        // (accumulators are read in another basic block than the last v_mfma: explicit wait states, see conv_wgrad_v4_body32)
    MEDT_SCHED_FENCE();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    MEDT_SCHED_FENCE();
#endif
    // D[(lane>>4)*4 + r][lane&15] of tile t -> o = o0 + 16 rb + (lane>>4)*4 + r, position (ho0 + (t0 + t) / TC, 16 ((t0 + t) % TC) + (lane&15))
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = o0 + 16 * rb + (lane >> 4) * 4 + r;
        if (o < Cout) {
            const float bo = bias ? bias[o] : 0.f;
            float* yo = y + ((size_t)n * Cout + o) * HoWo + ho0 * Wo + (lane & 15);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float v = acc[t][r] + bo;
                s1[r] += v;
                s2[r] = fmaf(v, v, s2[r]);
                yo[((t0 + t) / TC) * Wo + 16 * ((t0 + t) % TC)] = relu ? fmaxf(v, 0.f) : v;
            }
        }
    }
    if (partials) {
        double pa[4], pb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double a = s1[r], b = s2[r];                       // (double from the cross-lane tree on: block_sum_d, medt_common.h)
#pragma unroll
            for (int m = 8; m > 0; m >>= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }   // over lane&15
            pa[r] = a;
            pb[r] = b;
        }
        if (wv >= 2 && (lane & 15) == 0)                       // the two waves of a row block each hold half of the 128 positions
#pragma unroll
            for (int r = 0; r < 4; ++r) { Rd7[rb][(lane >> 4) * 4 + r][0] = pa[r]; Rd7[rb][(lane >> 4) * 4 + r][1] = pb[r]; }
        __syncthreads();
        if (wv < 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = o0 + 16 * rb + (lane >> 4) * 4 + r;
                if ((lane & 15) == 0 && o < Cout) {
                    double* dst = reinterpret_cast<double*>(partials) + ((size_t)blockIdx.x * Cout + o) * 2;
                    dst[0] = pa[r] + Rd7[rb][(lane >> 4) * 4 + r][0];
                    dst[1] = pb[r] + Rd7[rb][(lane >> 4) * 4 + r][1];
                }
            }
        }
    }
}

int conv_stem7_fwd(const float* x, const float* w, const float* bias, float* y, float* partials, int N, int H, int W, int Cout,
                   int relu, hipStream_t s) {
    const int Ho = H / 2, Wo = W / 2, R = 128 / Wo;
    const size_t lds = (size_t)(S7_OT * S7_LDW + 3 * (2 * R + 5) * (W + 6)) * sizeof(float);
    const unsigned PW = (unsigned)(W + 6), PHPW = (unsigned)(2 * R + 5) * PW;
    hipLaunchKernelGGL(conv_stem7_fwd_kernel, dim3(N * (Ho * Wo / 128), cdiv(Cout, S7_OT)), dim3(MEDT_THREADS), lds, s, x, w, bias, y,
                       partials, H, W, Cout, relu, (unsigned)(0x100000000ull / PW) + 1u, (unsigned)(0x100000000ull / PHPW) + 1u);
    return launch_status("conv_stem7_fwd");
}

// --------------------------------------------------------------------------- //
// THIN-CHANNEL 3x3 convolutions on the matrix cores (round 6).  conv_use_mfma refuses Cout < 32 and Cin * 9 < 256 -- the 64 x 64
// implicit-GEMM tile would be mostly padding -- so the global stem (conv2 8 -> 128, conv3 128 -> 8 at 64 x 64), the decoders
// (decoder4 64 -> 32, decoder5 32 -> 16, decoderf 16 -> 16 on 32 ... 128-wide maps), decoder5_p and every backward-data twin of
// those ran the VALU direct kernels at ~10 TFLOP/s (16 - 32 us each for 150 - 300 MFLOP; lib/models/axialnet.py:530-531, 571-588,
// 650-652 of the reference).  Here the LDS-patch scheme of conv3x3_rows16_fwd_kernel with the tile turned around:
//   * MFMA M = 16 OUTPUT CHANNELS (a row block; NRB = 1 or 2 row blocks per workgroup -- 8 channels run a half-empty block),
//     MFMA N = 16 consecutive output COLUMNS of one image row, K-step = 4 input channels of one tap;
//   * a workgroup owns RG * TR rows x 16 * TCW columns of one image (TCW = min(4, W / 16) column tiles over the four waves,
//     RG = 4 / TCW row groups); a wave owns one column tile x TR rows x all NRB row blocks: its B fragments of a K-step are
//     (TR + 2) x 3 shifted reads of ONE patch column strip for 9 TR NRB MFMAs;
//   * per chunk of CC input channels (16, or 8 when Cin is 8) the zero-padded halo patch Ps[CC][RG TR + 2][16 TCW + 2] and the
//     weight slab As[16 NRB][9 CC] are staged in LDS once -- every load of the NEXT chunk is in flight under the MFMAs of this one;
//   * conflict-free fragment reads: patch channel stride CST = 16 (mod 64)  [bank = 16 (l >> 4) + (l & 15) + const],
//     weight row stride 9 CC + 4  [bank = 20 | 12 (l & 15) + 9 (l >> 4) + const: 16 multiples of 4 plus residues 0..3];
//   * epilogue: bias / ReLU / the fan-in addend of a backward-data call (`add`), BatchNorm partial sums in double (one row per
//     workgroup tile, channel-block columns), 64-byte row segments of y.
// Backward-data = the same kernel on dY with the flipped weights (conv2d_bwd_data: wt[c][o][8 - t]).
// --------------------------------------------------------------------------- //
struct ThinPlan { int ok, CC, NRB, TR, KG, TCW, RG, RST, CST, AST, rows_wg, cols_wg, wgs_img, ppg, grid_y; size_t lds; };

ThinPlan conv_thin_plan(int N, int groups, int Cin, int H, int W, int Cout, int K, int stride, int pad) {
    static const bool off = [] { const char* e = getenv("MEDT_CONV_THIN"); return e && e[0] == '0'; }();
    static const int force_tr = 0;      // (debugging: rows per wave)
    ThinPlan p{};
    if (off || K != 3 || stride != 1 || pad != 1 || W < 16 || (W & 15) || (Cin & 7) || Cin < 8 || Cout < 1 || groups < 1 || N % groups) return p;
    if ((long)N * H * W < 4096) return p;                        // (tiny maps: the split-K / small-map kernels)
    p.CC = (Cin & 15) ? 8 : 16;
    p.TCW = W >= 64 ? 4 : (W >= 32 ? 2 : 1);
    if (W % (16 * p.TCW)) return p;
    p.RG = 4 / p.TCW;
    // the largest (TR, NRB) that still gives the chip a workgroup per CU; else the finest split
    const int rbs = cdiv(Cout, 16);
    int best_tr = 0, best_nrb = 0;
    long best_wgs = -1;
    for (int tr = 4; tr >= 1; tr >>= 1) {
        if (H % (p.RG * tr) || (force_tr && tr != force_tr)) continue;
        for (int nrb = 2; nrb >= 1; --nrb) {
            if (nrb > rbs) continue;
            if (p.CC == 8 && nrb == 1 && rbs > 1) continue;     // (instances: CC = 8 only with two row blocks unless Cout <= 16)
            const long wgs = (long)N * (H / (p.RG * tr)) * (W / (16 * p.TCW)) * cdiv(rbs, nrb);
            const bool better = best_wgs < 0 || (best_wgs < 256 && wgs > best_wgs);
            if (better) { best_tr = tr; best_nrb = nrb; best_wgs = wgs; }
        }
    }
    if (best_wgs < 0) return p;
#ifndef MEDT_AB_THIN_TR                 // (A/B build: without this rule)
    // Round 6: a DEEP contraction (>= 4 chunks of 16 channels: conv3 128 -> 8, conv2's backward-data) wants one row per wave + K-groups
    // even where two rows per wave already give a workgroup per CU -- the rule above was tuned on 4 images; at 8 images of 64 x 64
    // (gatedaxialunet bs 8) and 2 of 128 x 128 (MedT-256 bs 2) it picked <16,1,2,1>: 256 four-wave workgroups walking 8 chunks at a
    // global round trip + two barriers each, 55 - 58 us for the 0.6 GFLOP that take 22 us at 4 images with KG = 4
    if (p.CC == 16 && Cin / 16 >= 4 && best_tr > 1 && H % p.RG == 0 && !force_tr) {
        const long w1 = (long)N * (H / p.RG) * (W / (16 * p.TCW)) * cdiv(rbs, best_nrb);
        if (w1 <= 512) { best_tr = 1; best_wgs = w1; }
    }
#endif
    p.TR = best_tr; p.NRB = best_nrb;
    p.rows_wg = p.RG * p.TR; p.cols_wg = 16 * p.TCW;
    p.RST = p.cols_wg + 2;
    const int pr = p.rows_wg + 2;
    p.CST = pr * p.RST;
    p.CST += ((16 - p.CST) % 64 + 64) % 64;                      // = 16 (mod 64)
    p.AST = 9 * p.CC + 4;
    p.wgs_img = (H / p.rows_wg) * (W / p.cols_wg);
    p.ppg = (N / groups) * p.wgs_img;
    p.grid_y = cdiv(rbs, p.NRB);
    // K-groups (see the kernel): deep contractions at one row per wave on a grid that does not fill the SIMDs by itself
    static const int force_kg = 0;
    const int nch = Cin / p.CC;
    p.KG = 1;
    if (p.CC == 16 && p.TR == 1 && best_wgs <= 512) p.KG = (nch % 4 == 0) ? 4 : ((nch % 2 == 0) ? 2 : 1);
    if (force_kg && p.CC == 16 && p.TR == 1 && nch % force_kg == 0 && (force_kg == 1 || force_kg == 2 || force_kg == 4)) p.KG = force_kg;
    p.lds = (size_t)p.KG * ((size_t)p.CC * p.CST + (size_t)16 * p.NRB * p.AST) * sizeof(float) + 4 * 2 * 16 * 2 * sizeof(double);
    p.ok = 1;
    return p;
}

bool conv_thin_ok(int N, int groups, int Cin, int H, int W, int Cout, int K, int stride, int pad) {
    return conv_thin_plan(N, groups, Cin, H, W, Cout, K, stride, pad).ok != 0;
}
int conv_thin_parts_per_group(int N, int groups, int Cin, int H, int W, int Cout, int K, int stride, int pad) {
    return conv_thin_plan(N, groups, Cin, H, W, Cout, K, stride, pad).ppg;
}

struct ThinArgs {
    const float *x, *w, *bias, *add;
    float *y, *partials;
    int Cin, H, W, Cout, relu;
    int TCW, RG, RST, CST, PR, PC, cwg;          // geometry of the plan; cwg = column workgroups per image row band
    unsigned m_pc, m_prpc;                       // ceil(2^32 / PC), ceil(2^32 / (PR PC)): exact quotients for u < 2^16
};

// KG > 1 (deep contractions on few workgroups -- conv3 128 -> 8 and conv2's backward-data at 64 x 64: 256 tiles, 8 chunks): KG groups of
// four waves share the workgroup's output tile and split the CHUNKS (group kg takes chunks kg, kg + KG, ...), each with its own patch /
// weight-slab region, in lockstep behind the same two barriers per round; the groups' totals are combined through LDS in fixed order.
// With one workgroup per CU a single four-wave group had one wave per SIMD and paid a full global round trip + two barriers per chunk
// (measured: 40 us for conv3's 302 MFLOP, 8 rounds); KG = 4 runs two rounds with four waves per SIMD.
template <int CC, int NRB, int TR, int KG>
__global__ __launch_bounds__(MEDT_THREADS * KG) void conv3x3_thin_fwd_kernel(ThinArgs a) {
    constexpr int AST = 9 * CC + 4, KS = CC / 4;
    // patch staging, main columns: a load instruction of a wave = RG "units" (c, patch row) x 16 TCW columns; unit v = pr * CC + c, so
    // the RG units of one load share the patch row and have consecutive channels: scalar base + one chunk-invariant lane offset
    constexpr int NU = CC * (TR + 2) / 4;                // loads per wave and chunk at most (RG = 1: PR = TR + 2 rows, 4 units per round)
    constexpr int NH = (2 * CC * (4 * TR + 2) + MEDT_THREADS - 1) / MEDT_THREADS;     // halo-column elements per thread at most
    constexpr int WROWS = MEDT_THREADS / CC;             // weight-slab rows per pass: thread (ar, ae) loads columns ae + CC m, m < 9
    constexpr int WP = (16 * NRB + WROWS - 1) / WROWS;   // passes
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x & (MEDT_THREADS - 1), lane = tid & 63;                     // tid: inside the K-group
    const int wv = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3), kg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    const int tc = wv % a.TCW, rg = wv / a.TCW;
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;
    const int rows_wg = a.RG * TR, bands = H / rows_wg;
    int b = blockIdx.x;
    const int cw = b % a.cwg; b /= a.cwg;
    const int band = b % bands, n = b / bands;
    const int r0 = band * rows_wg, c0 = cw * 16 * a.TCW, o0 = blockIdx.y * 16 * NRB;
    const int kg_floats = CC * a.CST + 16 * NRB * AST;  // one K-group's staging region
    float* Ps = smem + kg * kg_floats;                  // [CC][CST]
    float* As = Ps + CC * a.CST;                        // [16 NRB][AST]
    double* Rd = reinterpret_cast<double*>(smem + KG * kg_floats);     // [4 waves][NRB][16][2]
    const int Ktot = Cin * 9, PR = a.PR, mainw = 16 * a.TCW;
    const float* xn = a.x + (size_t)n * Cin * HW;
    const int ug = lane / mainw, ucol = lane - ug * mainw;             // unit inside the load's group, main column
    const int lane_g = ug * HW + ucol, lane_p = ug * a.CST + 1 + ucol; // chunk-invariant lane parts of the global / LDS offsets
    const int nrounds = (PR * CC) / (4 * a.RG);                        // loads per wave and chunk (RG | CC, 4 RG | PR CC: PR CC = 16 k)
    // halo columns (patch columns 0 and PC - 1) of every unit: element h = tid + 256 i -> (unit, side)
    int hp[NH], hg[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int h = tid + MEDT_THREADS * i, v = h >> 1, side = h & 1, pr = v / CC, c = v - pr * CC;
        const int gr = r0 - 1 + pr, gc = side ? c0 + mainw : c0 - 1;
        const bool in = h < 2 * PR * CC;
        hp[i] = in ? c * a.CST + pr * a.RST + (side ? mainw + 1 : 0) : -1;
        hg[i] = (in && gr >= 0 && gr < H && gc >= 0 && gc < W) ? c * HW + gr * W + gc : -1;
    }
    const int ar = tid / CC, ae = tid - ar * CC;
    float rp[NU], rh[NH], ra[WP][9];
    auto fetch = [&](int ch0) {
        const float* xc = xn + (size_t)ch0 * HW;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            if (j < nrounds) {                           // (wave-uniform)
                const int v0 = (j * 4 + wv) * a.RG, pr = v0 / CC, cb = v0 - pr * CC, gr = r0 - 1 + pr;
                const float v = xc[cb * HW + min(max(gr, 0), H - 1) * W + c0 + lane_g];
                rp[j] = (gr >= 0 && gr < H) ? v : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < NH; ++i) {                  // unconditional loads (no branch per element), then select
            const float v = xc[max(hg[i], 0)];
            rh[i] = hg[i] >= 0 ? v : 0.f;
        }
#pragma unroll
        for (int q = 0; q < WP; ++q) {
            const int o = min(o0 + ar + WROWS * q, Cout - 1);       // rows past Cout: their accumulators are never stored
            const float* wr = a.w + (size_t)o * Ktot + ch0 * 9 + ae;
#pragma unroll
            for (int m = 0; m < 9; ++m) ra[q][m] = wr[CC * m];
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            if (j < nrounds) {
                const int v0 = (j * 4 + wv) * a.RG, pr = v0 / CC, cb = v0 - pr * CC;
                Ps[cb * a.CST + pr * a.RST + lane_p] = rp[j];
            }
        }
#pragma unroll
        for (int i = 0; i < NH; ++i)
            if (hp[i] >= 0) Ps[hp[i]] = rh[i];
#pragma unroll
        for (int q = 0; q < WP; ++q)
            if (ar + WROWS * q < 16 * NRB) {
#pragma unroll
                for (int m = 0; m < 9; ++m) As[(ar + WROWS * q) * AST + ae + CC * m] = ra[q][m];
            }
    };
    f32x4 acc[NRB][TR];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int t = 0; t < TR; ++t) acc[rb][t] = (f32x4)(0.f);
    const float* arow = As + (lane & 15) * AST + (lane >> 4) * 9;
    const float* prow = Ps + (lane >> 4) * a.CST + (rg * TR) * a.RST + 16 * tc + (lane & 15);
    const int nit = Cin / (CC * KG);                     // rounds: K-group kg takes chunk it * KG + kg
    fetch(kg * CC);
    for (int it = 0; it < nit; ++it) {
        stage();
        __syncthreads();
        if (it + 1 < nit) fetch(((it + 1) * KG + kg) * CC);      // flies during the MFMAs below
        // BLOCKED summation: the chunk's 9 CC products per output are summed in fresh accumulators and added to the running totals
        // once per chunk.  One MFMA chain over a deep contraction (conv3: 1152 terms) measured 2x the rounding error of the VALU
        // kernel it replaces (rms 9.3e-7 vs 4.7e-7 of the output's rms, which splits the contraction over four waves) -- enough to
        // fail the 2-image training fixture's gradient bound in the ill-conditioned batch-statistics network (profiles/r06_thin_precision.txt).
        // With one output tile per wave the k-steps alternate between two chains (an MFMA need not wait for its predecessor).
        constexpr int NA = (TR * NRB == 1) ? 2 : 1;
        f32x4 cacc[NA][NRB][TR];
#pragma unroll
        for (int q = 0; q < NA; ++q)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int t = 0; t < TR; ++t) cacc[q][rb][t] = (f32x4)(0.f);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float av[NRB][9], bv[TR + 2][3];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int t = 0; t < 9; ++t) av[rb][t] = arow[rb * 16 * AST + ks * 36 + t];
#pragma unroll
            for (int r = 0; r < TR + 2; ++r)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) bv[r][kw] = prow[ks * 4 * a.CST + r * a.RST + kw];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int t = 0; t < TR; ++t)
#pragma unroll
                        for (int rb = 0; rb < NRB; ++rb)
                            cacc[(ks * 9 + kh * 3 + kw) % NA][rb][t] =
                                __builtin_amdgcn_mfma_f32_16x16x4f32(av[rb][kh * 3 + kw], bv[t + kh][kw], cacc[(ks * 9 + kh * 3 + kw) % NA][rb][t], 0, 0, 0);
        }
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int t = 0; t < TR; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {            // (element-wise: the lane emulator's vector shim has no operator+=)
                    float c = cacc[0][rb][t][r];
                    if (NA == 2) c += cacc[1][rb][t][r];
                    acc[rb][t][r] += c;
                }
            }
        __syncthreads();
    }
    if (KG > 1) {
        // the K-groups' totals meet in LDS (the staging regions are free behind the loop's last barrier): group 0 adds groups 1 .. KG - 1
        // in that order -- a fixed summation order, bit-reproducible
        float* X = smem + ((kg > 0 ? kg - 1 : 0) * 4 + wv) * (NRB * TR * 4 * 64) + lane;
        if (kg > 0) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int t = 0; t < TR; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[((rb * TR + t) * 4 + r) * 64] = acc[rb][t][r];
        }
        __syncthreads();
        if (kg == 0)
#pragma unroll
        for (int q = 1; q < KG; ++q) {
            const float* Y = smem + ((q - 1) * 4 + wv) * (NRB * TR * 4 * 64) + lane;
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int t = 0; t < TR; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[rb][t][r] += Y[((rb * TR + t) * 4 + r) * 64];
        }
    }
    // D[(lane>>4)*4 + r][lane&15] of (rb, t)  ->  o = o0 + 16 rb + (lane>>4)*4 + r,  position (r0 + rg TR + t, c0 + 16 tc + (lane&15))
    const int col = c0 + 16 * tc + (lane & 15), row0 = r0 + rg * TR;
    if (kg == 0)
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = o0 + 16 * rb + (lane >> 4) * 4 + r;
            if (o < Cout) {
                const float bo = a.bias ? a.bias[o] : 0.f;
                const size_t e0 = ((size_t)n * Cout + o) * HW + (size_t)row0 * W + col;
#pragma unroll
                for (int t = 0; t < TR; ++t) {
                    float v = acc[rb][t][r] + bo;
                    if (a.add) v += a.add[e0 + (size_t)t * W];
                    s1[r] += v;
                    s2[r] = fmaf(v, v, s2[r]);
                    a.y[e0 + (size_t)t * W] = a.relu ? fmaxf(v, 0.f) : v;
                }
            }
        }
        if (a.partials) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double pa = s1[r], pb = s2[r];           // (double from the cross-lane tree on: block_sum_d, medt_common.h)
#pragma unroll
                for (int m = 8; m > 0; m >>= 1) { pa += __shfl_xor(pa, m, 64); pb += __shfl_xor(pb, m, 64); }   // over lane & 15
                if ((lane & 15) == 0) {
                    double* d = Rd + ((wv * NRB + rb) * 16 + (lane >> 4) * 4 + r) * 2;
                    d[0] = pa; d[1] = pb;
                }
            }
        }
    }
    if (a.partials) {
        __syncthreads();                                 // (every K-group arrives; only group 0 has written Rd and goes on)
        if (kg == 0 && tid < 16 * NRB) {                            // the four waves hold the four quarters of the workgroup's positions
            const int o = o0 + tid;
            if (o < Cout) {
                double sa = 0.0, sb = 0.0;
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) { sa += Rd[(w4 * 16 * NRB + tid) * 2]; sb += Rd[(w4 * 16 * NRB + tid) * 2 + 1]; }
                double* dst = reinterpret_cast<double*>(a.partials) + ((size_t)blockIdx.x * Cout + o) * 2;
                dst[0] = sa;
                dst[1] = sb;
            }
        }
    }
}

int conv_thin_fwd(const float* x, const float* w, const float* bias, const float* add, float* y, float* partials, int N, int groups,
                  int Cin, int H, int W, int Cout, int relu, hipStream_t s) {
    const ThinPlan p = conv_thin_plan(N, groups, Cin, H, W, Cout, 3, 1, 1);
    if (!p.ok) { set_error("conv_thin_fwd: shape not supported"); return MEDT_EUNSUPPORTED; }
    if (abl_skip("conv_thin")) return MEDT_OK;
    ThinArgs a;
    a.x = x; a.w = w; a.bias = bias; a.add = add; a.y = y; a.partials = partials;
    a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.relu = relu;
    a.TCW = p.TCW; a.RG = p.RG; a.RST = p.RST; a.CST = p.CST; a.PR = p.rows_wg + 2; a.PC = p.cols_wg + 2; a.cwg = W / p.cols_wg;
    a.m_pc = (unsigned)(0x100000000ull / (unsigned)a.PC) + 1u;
    a.m_prpc = (unsigned)(0x100000000ull / (unsigned)(a.PR * a.PC)) + 1u;
    const dim3 grid((unsigned)(N * p.wgs_img), (unsigned)p.grid_y), block(MEDT_THREADS * p.KG);
#define MEDT_THIN(CCv, NRBv, TRv, KGv) hipLaunchKernelGGL((conv3x3_thin_fwd_kernel<CCv, NRBv, TRv, KGv>), grid, block, p.lds, s, a)
#define MEDT_THIN_TR(CCv, NRBv)                         \
    do {                                                \
        if (p.TR == 4) MEDT_THIN(CCv, NRBv, 4, 1);      \
        else if (p.TR == 2) MEDT_THIN(CCv, NRBv, 2, 1); \
        else MEDT_THIN(CCv, NRBv, 1, 1);                \
    } while (0)
    if (p.KG > 1) {                                      // (CC = 16, one row per wave; > 64 KB of LDS with four K-groups)
        static unsigned char attr[4][64];
        int rc;
        if ((rc = lds_opt_in((const void*)conv3x3_thin_fwd_kernel<16, 1, 1, 4>, attr[0], "conv3x3_thin_fwd")) ||
            (rc = lds_opt_in((const void*)conv3x3_thin_fwd_kernel<16, 2, 1, 4>, attr[1], "conv3x3_thin_fwd")) ||
            (rc = lds_opt_in((const void*)conv3x3_thin_fwd_kernel<16, 1, 1, 2>, attr[2], "conv3x3_thin_fwd")) ||
            (rc = lds_opt_in((const void*)conv3x3_thin_fwd_kernel<16, 2, 1, 2>, attr[3], "conv3x3_thin_fwd"))) return rc;
        if (p.NRB == 2) { if (p.KG == 4) MEDT_THIN(16, 2, 1, 4); else MEDT_THIN(16, 2, 1, 2); }
        else { if (p.KG == 4) MEDT_THIN(16, 1, 1, 4); else MEDT_THIN(16, 1, 1, 2); }
    }
    else if (p.CC == 16 && p.NRB == 2) MEDT_THIN_TR(16, 2);
    else if (p.CC == 16) MEDT_THIN_TR(16, 1);
    else if (p.NRB == 2) MEDT_THIN_TR(8, 2);
    else MEDT_THIN_TR(8, 1);
#undef MEDT_THIN_TR
#undef MEDT_THIN
    return launch_status("conv3x3_thin_fwd");
}

// Weight gradient of the same layers: dW[o][c][t] = sum_q dY[o][q] * X[c][q + t].  A workgroup owns 64 output channels x
// one 16-channel chunk x all 9 taps (9 accumulator tiles per wave) over a chunk of 64-position tiles: the dY tile and the
// halo patch are staged once per position tile and the 9 taps read shifted windows of the patch -- 144 MFMAs per wave per
// round trip.  Both operands have the channel on the fragment's row index and the position on k:
//   dY  Ds[o][68]:  bank = 4 (l&15) + (l>>4) + const;   patch Ps[c][132]: bank = 4 (l&15) + (l>>4) + const.
// grid (Cout/64, Cin/16, splits); slab bz of `scratch` receives the partial sum over its position tiles.
struct R16WJob {                         // one weight-gradient problem of the kernel below (kernel arguments / job table)
    const float *dy, *raw, *coef, *x;
    float* scratch;
    int Cin, H, Cout, tiles_per_split, ptiles, npg, gx, gy, gz;
};

// OT = 64: a wave owns a 16-channel row block and all nine taps (round 2).  OT = 32 (round 5): wave w owns row block (w & 1) and the
// taps 0..4 / 5..8 ((w >> 1); the unused tenth slot multiplies a zero operand -- no branch around an MFMA, see conv_wgrad_v4_body32):
// half the accumulators, twice the workgroups -- measured slower for the weight-gradient pair (see conv_wgrad_rows16_grouped), kept
// behind MEDT_R16W_OT=32; the explicit tap-offset table of this round took the full kernel from 189 + 40 to 123 + 40 registers.
template <int OT>
__device__ __forceinline__ void conv3x3_rows16_wgrad_body(const R16WJob& jb, int bx, int by, int bz, float* Ds, float* Ps) {
    constexpr int NR = OT / 4;                                   // dY rows per thread and tile
    constexpr int NA = OT == 64 ? 9 : 5;                         // accumulator tiles (taps) per wave
    const float* __restrict__ dy = jb.dy;
    const float* __restrict__ raw = jb.raw;
    const float* __restrict__ coef = jb.coef;
    const float* __restrict__ x = jb.x;
    float* __restrict__ scratch = jb.scratch;
    const int Cin = jb.Cin, H = jb.H, Cout = jb.Cout, tiles_per_split = jb.tiles_per_split, ptiles = jb.ptiles, npg = jb.npg;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rb = OT == 64 ? wv : (wv & 1), tlo = OT == 64 ? 0 : 5 * (wv >> 1);
    const int o0 = bx * OT, c0 = by * 16;
    const int Ktot = Cin * 9, HW = H * 16, tpi = H >> 2;
    const int pt_begin = bz * tiles_per_split;
    const int pt_end = min(ptiles, pt_begin + tiles_per_split);
    if (tid < 192) {
        const int c = tid / 12, rem = tid - c * 12;
        Ps[c * R16_WCST + (rem >> 1) * R16_RST + (rem & 1) * 17] = 0.f;
    }
    int poff[6], pc[6], prw[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int u = tid + MEDT_THREADS * i, c = u / 96, rem = u - c * 96, row = rem >> 4, col = rem & 15;
        poff[i] = c * R16_WCST + row * R16_RST + 1 + col;
        pc[i] = (c0 + c) * HW + col;
        prw[i] = row - 1;
    }
    float rd[NR], rp[6];
    auto fetch = [&](int pt) {
        const int n = pt / tpi, r0 = (pt - n * tpi) * 4;
        const size_t base = (size_t)n * Cout * HW + r0 * 16 + lane;
        const float* cf = coef ? coef + (size_t)(n / npg) * Cout * 3 : nullptr;
#pragma unroll
        for (int j = 0; j < NR; ++j)                                // rows past Cout: their accumulators are never stored
            rd[j] = dy[base + (size_t)min(o0 + wv + 4 * j, Cout - 1) * HW];
        if (cf) {
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int o = min(o0 + wv + 4 * j, Cout - 1);
                rd[j] = fmaf(cf[o * 3], rd[j], fmaf(cf[o * 3 + 1], raw[base + (size_t)o * HW], cf[o * 3 + 2]));
            }
        }
        const float* xn = x + (size_t)n * Cin * HW;
#pragma unroll
        for (int i = 0; i < 6; ++i) {                               // unconditional loads (no branch per element), then select
            const int gr = r0 + prw[i];
            const float v = xn[pc[i] + min(max(gr, 0), H - 1) * 16];
            rp[i] = (unsigned)gr < (unsigned)H ? v : 0.f;
        }
    };
    f32x4 acc[NA];
#pragma unroll
    for (int t = 0; t < NA; ++t) acc[t] = (f32x4)(0.f);
    // this wave's taps: LDS offset of tap slot tt inside the patch window, and whether the slot exists (the tenth does not)
    int tapoff[NA];
    float tapon[NA];
#pragma unroll
    for (int tt = 0; tt < NA; ++tt) {
        const int t = min(tlo + tt, 8);
        tapoff[tt] = (t / 3) * R16_RST + t % 3;
        tapon[tt] = tlo + tt <= 8 ? 1.f : 0.f;
    }
    const float* drow = Ds + (16 * rb + (lane & 15)) * R16_DST + (lane >> 4);
    const float* prow = Ps + (lane & 15) * R16_WCST + (lane >> 4);
    if (pt_begin < pt_end) fetch(pt_begin);
    for (int pt = pt_begin; pt < pt_end; ++pt) {
#pragma unroll
        for (int j = 0; j < NR; ++j) Ds[(wv + 4 * j) * R16_DST + lane] = rd[j];
#pragma unroll
        for (int i = 0; i < 6; ++i) Ps[poff[i]] = rp[i];
        __syncthreads();
        if (pt + 1 < pt_end) fetch(pt + 1);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {                          // k = position (row ks>>2, column 4 (ks&3) + (lane>>4))
            const float a = drow[ks * 4];
            const float* pk = prow + (ks >> 2) * R16_RST + (ks & 3) * 4;
#pragma unroll
            for (int tt = 0; tt < NA; ++tt)
                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(OT == 64 ? a : a * tapon[tt], pk[tapoff[tt]], acc[tt], 0, 0, 0);
        }
        __syncthreads();
    }
    float* out = scratch + (size_t)bz * Cout * Ktot;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = o0 + 16 * rb + (lane >> 4) * 4 + r;
        if (o < Cout) {
            float* dst = out + (size_t)o * Ktot + (c0 + (lane & 15)) * 9;
#pragma unroll
            for (int tt = 0; tt < NA; ++tt)
                if (tlo + tt <= 8) dst[tlo + tt] = acc[tt][r];
        }
    }
}

// Several of these problems in ONE launch (the recorded weight gradients of conv2_p and conv3_p at the end of the local
// branch's backward): each alone puts one workgroup of one wave per SIMD on every CU and spends two thirds of its time
// waiting for the next tile; side by side they fill each other's gaps (46 + 46 us back to back before).
using R16WBatch = JobBatch<R16WJob, 4>;
template <int OT>
__global__ __launch_bounds__(MEDT_THREADS) void conv3x3_rows16_wgrad_kernel(R16WBatch b) {
    MEDT_STATIC_SHARED float Ds[OT * R16_DST];
    MEDT_STATIC_SHARED float Ps[16 * R16_WCST];
    const int j = find_job(b, blockIdx.x);
    const R16WJob& jb = b.job[j];
    const int r = blockIdx.x - b.start[j], bx = r % jb.gx, by = (r / jb.gx) % jb.gy, bz = r / (jb.gx * jb.gy);
    conv3x3_rows16_wgrad_body<OT>(jb, bx, by, bz, Ds, Ps);
}

// y[n,o,p] = bias[o] + sum over K slices; optional ReLU and BatchNorm partials ([group][256-position part][Cout][2]).
// grid (groups*ppg256, Cout): lanes over the positions of one group, one channel per blockIdx.y.
__global__ __launch_bounds__(MEDT_THREADS) void conv_splitk_epilogue_kernel(
    const float* __restrict__ slices, int nslices, size_t slice_stride, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ partials, int Cout, int HoWo, int npg, int ppg64, int relu) {
    MEDT_STATIC_SHARED float red[MEDT_WAVES * 2 * 2];
    const int per_group = npg * HoWo, ppg = (per_group + MEDT_THREADS - 1) / MEDT_THREADS;
    const int grp = blockIdx.x / ppg, part = blockIdx.x - grp * ppg, o = blockIdx.y;
    const int q = part * MEDT_THREADS + threadIdx.x;
    float v[2] = {0.f, 0.f};
    if (q < per_group) {
        const size_t qg = (size_t)grp * ppg64 * 64 + q;          // position index used by the tile kernel
        float a = bias ? bias[o] : 0.f;
        const float* sp = slices + qg * Cout + o;
        int s = 0;
        for (; s + 8 <= nslices; s += 8) {                 // 8 slices' loads in flight, summed in slice order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = sp[(size_t)(s + u) * slice_stride];
            MEDT_SCHED_FENCE();
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
            MEDT_SCHED_FENCE();
        }
        for (; s < nslices; ++s) a += sp[(size_t)s * slice_stride];
        const int n = grp * npg + q / HoWo, p = q % HoWo;
        y[((size_t)n * Cout + o) * HoWo + p] = relu ? fmaxf(a, 0.f) : a;
        v[0] = a;
        v[1] = a * a;
    }
    if (partials) block_sum_d<2>(v, red, reinterpret_cast<double*>(partials) + ((size_t)blockIdx.x * Cout + o) * 2);
}

// number of K slices for a problem with `tiles` output tiles (1 = no split)
int conv_mfma_ksplit(int Ktot, long tiles, bool rows16) {
#ifdef MEDT_AB_KSPLIT128                // (A/B build: no split from 128 tiles on)
    if (tiles >= 128) return 1;
#else
    // (round 6: 128 ... 255 tiles -- decoder4 of gatedaxialunet at bs 8: 128 workgroups walking 36 k-steps, 44 us on half the CUs -- split in two;
    //  NOT on the 16-wide maps, where the unsplit problem takes the LDS-patch kernel: conv3_p of MedT-256 at bs 2 is such a case)
    if (tiles >= 256 || (rows16 && tiles >= 128)) return 1;
    if (tiles >= 128) return Ktot >= 256 ? 2 : 1;
#endif
    int ks = (int)(256 / tiles);
    const int maxks = Ktot / 64 > 0 ? Ktot / 64 : 1;          // at least 64 k (4 LDS steps) per slice
    if (ks > maxks) ks = maxks;
    return ks < 1 ? 1 : ks;
}

size_t conv_mfma_scratch_floats(int N, int groups, int HoWo, int Cin, int Cout, int K, bool rows16) {
    const long qt = (long)groups * conv_mfma_parts_per_group(N, groups, HoWo);
    const int ks = conv_mfma_ksplit(Cin * K * K, qt * cdiv(Cout, 64), rows16);
    return ks > 1 ? (size_t)ks * qt * 64 * Cout : 0;
}

int conv_mfma_fwd(const float* x, const float* w, const float* bias, float* y, float* partials, float* scratch, int N,
                  int Cin, int H, int W, int Cout, int K, int stride, int pad, int relu, int groups, hipStream_t s) {
    const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
    const int ppg64 = conv_mfma_parts_per_group(N, groups, Ho * Wo), Ktot = Cin * K * K;
    const long qt = (long)groups * ppg64;
    int ks = scratch ? conv_mfma_ksplit(Ktot, qt * cdiv(Cout, 64), conv_rows16_ok(Cin, H, W, K, stride, pad)) : 1;
    int kchunk = Ktot;
    if (ks > 1) {
        kchunk = cdiv(cdiv(Ktot, ks), 16) * 16;
        ks = cdiv(Ktot, kchunk);
    }
    const dim3 grid((unsigned)qt, cdiv(Cout, 64), ks), block(MEDT_THREADS);
    if (ks == 1 && (N / groups) * Ho * Wo % 64 == 0 && conv_rows16_ok(Cin, H, W, K, stride, pad)) {
        if (abl_skip("rows16")) return MEDT_OK;
        // fewer than three 64-channel workgroups per CU: half-width workgroups instead (MEDT_R16_OT=64 / 32 forces one)
        static const int force_ot = [] { const char* e = getenv("MEDT_R16_OT"); return e ? atoi(e) : 0; }();
        const bool half = force_ot == 32 || (force_ot != 64 && qt * cdiv(Cout, 64) < 768);
        if (half)
            hipLaunchKernelGGL(conv3x3_rows16_fwd_kernel<32>, dim3((unsigned)qt, cdiv(Cout, 32)), block, 0, s, x, w, bias, y,
                               partials, Cin, H, Cout, relu);
        else
            hipLaunchKernelGGL(conv3x3_rows16_fwd_kernel<64>, dim3((unsigned)qt, cdiv(Cout, 64)), block, 0, s, x, w, bias, y,
                               partials, Cin, H, Cout, relu);
        return launch_status("conv3x3_rows16_fwd");
    }
    float* kout = ks > 1 ? scratch : nullptr;
    if (K == 1)
        hipLaunchKernelGGL(conv_mfma_fwd_kernel<1>, grid, block, 0, s, x, w, bias, y, partials, Cin, H, W, Cout, Ho, Wo,
                           stride, pad, relu, N / groups, kchunk, kout);
    else
        hipLaunchKernelGGL(conv_mfma_fwd_kernel<3>, grid, block, 0, s, x, w, bias, y, partials, Cin, H, W, Cout, Ho, Wo,
                           stride, pad, relu, N / groups, kchunk, kout);
    int rc = launch_status("conv_mfma_fwd");
    if (rc || ks == 1) return rc;
    const int npg = N / groups, ppg = cdiv(npg * Ho * Wo, MEDT_THREADS);
    hipLaunchKernelGGL(conv_splitk_epilogue_kernel, dim3(groups * ppg, Cout), block, 0, s, scratch, ks,
                       (size_t)qt * 64 * Cout, bias, y, partials, Cout, Ho * Wo, npg, ppg64, relu);
    return launch_status("conv_splitk_epilogue");
}

// wt[c][o][K*K-1-t] = w[o][c][t]: backward-data of a stride-1 convolution is the forward convolution of dY with wt
// and padding K-1-pad.
__global__ __launch_bounds__(MEDT_THREADS) void flip_weights_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                                    int Cout, int Cin, int KK) {
    const int idx = blockIdx.x * MEDT_THREADS + threadIdx.x;
    if (idx >= Cout * Cin * KK) return;
    const int t = idx % KK, c = (idx / KK) % Cin, o = idx / (KK * Cin);
    wt[((size_t)c * Cout + o) * KK + (KK - 1 - t)] = w[idx];
}

int conv_flip_weights(const float* w, float* wt, int Cout, int Cin, int K, hipStream_t s) {
    const int total = Cout * Cin * K * K;
    hipLaunchKernelGGL(flip_weights_kernel, dim3(cdiv(total, MEDT_THREADS)), dim3(MEDT_THREADS), 0, s, w, wt, Cout, Cin, K * K);
    return launch_status("flip_weights");
}

// The recorded flips of a flush in ONE launch (round 6: a flush of the forward pass issued 4 - 6 of them back to back, ~5 us + a boundary each --
// on the layer chain itself in the single-stream networks)
using FBatchW = JobBatch<FlipJob, 96>;
__global__ __launch_bounds__(MEDT_THREADS) void flip_weights_grouped_kernel(FBatchW b) {
    const int j = find_job(b, blockIdx.x);
    const FlipJob& f = b.job[j];
    const int KK = f.K * f.K;
    const int idx = (blockIdx.x - b.start[j]) * MEDT_THREADS + threadIdx.x;
    if (idx >= f.Cout * f.Cin * KK) return;
    const int t = idx % KK, c = (idx / KK) % f.Cin, o = idx / (KK * f.Cin);
    f.wt[((size_t)c * f.Cout + o) * KK + (KK - 1 - t)] = f.w[idx];
}

int conv_flip_weights_grouped(const FlipJob* jobs, int n, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += 96) {
        FBatchW b;
        b.n = n - i0 < 96 ? n - i0 : 96;
        int blocks = 0;
        for (int i = 0; i < b.n; ++i) {
            b.job[i] = jobs[i0 + i];
            b.start[i] = blocks;
            blocks += cdiv(jobs[i0 + i].Cout * jobs[i0 + i].Cin * jobs[i0 + i].K * jobs[i0 + i].K, MEDT_THREADS);
        }
        b.start[b.n] = blocks;
        hipLaunchKernelGGL(flip_weights_grouped_kernel, dim3(blocks), dim3(MEDT_THREADS), 0, s, b);
        int rc = launch_status("flip_weights_grouped");
        if (rc) return rc;
    }
    return MEDT_OK;
}

// wt_ready: the flipped weights already sit in wt_scratch (round 5: the TRAINING forward pass leaves them behind the layer's saved
// statistics -- recorded for the forward's grouped flush, off the backward chain; medt_api.hip)
int conv_mfma_bwd_data_s1(const float* dy, const float* w, float* wt_scratch, float* ksplit_scratch, float* dx, int N,
                          int Cin, int H, int W, int Cout, int K, int pad, hipStream_t s, bool wt_ready) {
    if (!wt_ready) {
        int rc = conv_flip_weights(w, wt_scratch, Cout, Cin, K, s);
        if (rc) return rc;
    }
    // dy (N,Cout,Ho,Wo) with Ho = H + 2*pad - K + 1  ->  dx (N,Cin,H,W): forward conv, pad' = K-1-pad
    const int Ho = H + 2 * pad - K + 1, Wo = W + 2 * pad - K + 1;
    return conv_mfma_fwd(dy, wt_scratch, nullptr, dx, nullptr, ksplit_scratch, N, Cout, Ho, Wo, Cin, K, 1, K - 1 - pad, 0,
                         1, s);
}

// --------------------------------------------------------------------------- //
// weight gradient on the matrix cores (same tiling as conv_wgrad_kernel in conv.hip)
// --------------------------------------------------------------------------- //
template <int K>
__device__ __forceinline__ void conv_wgrad_mfma_body(
    const float* __restrict__ dy, const float* __restrict__ raw, const float* __restrict__ coef,
    const float* __restrict__ x, float* __restrict__ scratch, int N, int Cin, int H, int W, int Cout, int Ho, int Wo,
    int stride, int pad, int QS, int npg, int bx, int by, int bz, float (*A)[65], float (*B)[65]) {
    constexpr int KK = K * K;
    const int Ktot = Cin * KK, HoWo = Ho * Wo;
    const int o0 = bx * 64, k0 = by * 64;
    const long NP = (long)N * HoWo;
    const long q_begin = (long)bz * QS;
    const long q_end = q_begin + QS < NP ? q_begin + QS : NP;
    // the wave index as an SGPR: rows (o, k) are then wave-uniform, so their channel / tap decomposition, the bounds
    // tests on them and the BatchNorm coefficient loads run on the scalar unit instead of once per lane
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane, r0 = wv;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4)(0.f);
    // Software pipeline as in the forward kernel: the 32 global loads of the next 64 positions fly during the 64 MFMAs
    float ra[16], rb[16];
    auto fetch = [&](long q0) {
        const long q = q0 + j;
        const bool qok = q < q_end;
        const int n = qok ? (int)(q / HoWo) : 0, p = qok ? (int)(q - (long)n * HoWo) : 0;
        const int ho = p / Wo, wo = p - ho * Wo;
        const int hb = ho * stride - pad, wb = wo * stride - pad;
        const float* dyp = dy + (size_t)n * Cout * HoWo + p;
        const float* rawp = raw ? raw + (size_t)n * Cout * HoWo + p : nullptr;
        const float* xp = x + (size_t)n * Cin * H * W + (long)hb * W + wb;
        // BatchNorm group of the 64 positions: one group in all but the step that straddles a group boundary
        const long qlast = q0 + 63 < q_end ? q0 + 63 : q_end - 1;
        const int g_first = (int)(q0 / HoWo) / npg, g_last = (int)(qlast / HoWo) / npg;
        const bool one_group = g_first == g_last;
        const float* cfu = coef ? coef + (size_t)g_first * Cout * 3 : nullptr;
        const float* cfl = coef ? coef + (size_t)(n / npg) * Cout * 3 : nullptr;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = r0 + 4 * i;
            float a = 0.f, b = 0.f;
            const int o = o0 + r, k = k0 + r;
            if (o < Cout) {
                if (qok) a = dyp[(size_t)o * HoWo];
                if (coef) {
                    const float rw = qok ? rawp[(size_t)o * HoWo] : 0.f;
                    if (one_group) a = fmaf(cfu[o * 3], a, fmaf(cfu[o * 3 + 1], rw, cfu[o * 3 + 2]));
                    else a = fmaf(cfl[o * 3], a, fmaf(cfl[o * 3 + 1], rw, cfl[o * 3 + 2]));
                    if (!qok) a = 0.f;
                }
            }
            if (k < Ktot) {
                const int c = k / KK, t = k - c * KK;
                const int dh = t / K, dw = t % K;
                const int h = hb + dh, w = wb + dw;
                if (qok && h >= 0 && h < H && w >= 0 && w < W) b = xp[((size_t)c * H + dh) * W + dw];
            }
            ra[i] = a;
            rb[i] = b;
        }
    };
    const bool wave_rows = o0 + 16 * wv < Cout;
    const int tcols = Ktot - k0 >= 64 ? 4 : (Ktot - k0 + 15) / 16;
    if (q_begin < q_end) fetch(q_begin);
    for (long q0 = q_begin; q0 < q_end; q0 += 64) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            A[r0 + 4 * i][j] = ra[i];
            B[r0 + 4 * i][j] = rb[i];
        }
        __syncthreads();
        if (q0 + 64 < q_end) fetch(q0 + 64);
        // most recorded layers are narrower than the tile (Cout 16-32, Ktot 16-64): a wave whose 16 output rows lie
        // beyond Cout has nothing to multiply, and 16-column blocks beyond Ktot are skipped (both wave-uniform)
        if (wave_rows) {
#pragma unroll 4
            for (int ks = 0; ks < 16; ++ks) {
                const float a = A[16 * wv + (lane & 15)][ks * 4 + (lane >> 4)];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t < tcols) {
                        const float b = B[t * 16 + (lane & 15)][ks * 4 + (lane >> 4)];
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
    float* out = scratch + (size_t)bz * Cout * Ktot;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = o0 + 16 * wv + (lane >> 4) * 4 + r, k = k0 + t * 16 + (lane & 15);
            if (o < Cout && k < Ktot) out[(size_t)o * Ktot + k] = acc[t][r];
        }
}

// The same with a 32(o) x 64(k) tile for the recorded (grouped) launches: most recorded layers have 16-32 output
// channels, so the 64-row tile left two or three of the four waves without a row block and spent half of its LDS on zero
// rows.  Here wave w owns row block (w & 1) and the two column blocks 2*(w >> 1), +1; the A stage is 32 x 65 floats
// (25 KB of LDS per workgroup instead of 33: six resident workgroups per CU instead of four -- the kernel waits on
// global loads for half of its wave cycles, profiles/r03_step_pmc.json, so residency is what it needs).
template <int K>
__device__ __forceinline__ void conv_wgrad_mfma_body32(
    const float* __restrict__ dy, const float* __restrict__ raw, const float* __restrict__ coef,
    const float* __restrict__ x, float* __restrict__ scratch, int N, int Cin, int H, int W, int Cout, int Ho, int Wo,
    int stride, int pad, int QS, int npg, int bx, int by, int bz, float (*A)[65], float (*B)[65]) {
    constexpr int KK = K * K;
    const int Ktot = Cin * KK, HoWo = Ho * Wo;
    const int o0 = bx * 32, k0 = by * 64;
    const long NP = (long)N * HoWo;
    const long q_begin = (long)bz * QS;
    const long q_end = q_begin + QS < NP ? q_begin + QS : NP;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (SGPR: see above)
    const int j = lane, r0 = wv;
    f32x4 acc[2];
    acc[0] = (f32x4)(0.f);
    acc[1] = (f32x4)(0.f);
    float ra[8], rb[16];
    auto fetch = [&](long q0) {
        const long q = q0 + j;
        const bool qok = q < q_end;
        const int n = qok ? (int)(q / HoWo) : 0, p = qok ? (int)(q - (long)n * HoWo) : 0;
        const int ho = p / Wo, wo = p - ho * Wo;
        const int hb = ho * stride - pad, wb = wo * stride - pad;
        const float* dyp = dy + (size_t)n * Cout * HoWo + p;
        const float* rawp = raw ? raw + (size_t)n * Cout * HoWo + p : nullptr;
        const float* xp = x + (size_t)n * Cin * H * W + (long)hb * W + wb;
        const long qlast = q0 + 63 < q_end ? q0 + 63 : q_end - 1;
        const int g_first = (int)(q0 / HoWo) / npg, g_last = (int)(qlast / HoWo) / npg;
        const bool one_group = g_first == g_last;
        const float* cfu = coef ? coef + (size_t)g_first * Cout * 3 : nullptr;
        const float* cfl = coef ? coef + (size_t)(n / npg) * Cout * 3 : nullptr;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int o = o0 + r0 + 4 * i;
            float a = 0.f;
            if (o < Cout) {
                if (qok) a = dyp[(size_t)o * HoWo];
                if (coef) {
                    const float rw = qok ? rawp[(size_t)o * HoWo] : 0.f;
                    if (one_group) a = fmaf(cfu[o * 3], a, fmaf(cfu[o * 3 + 1], rw, cfu[o * 3 + 2]));
                    else a = fmaf(cfl[o * 3], a, fmaf(cfl[o * 3 + 1], rw, cfl[o * 3 + 2]));
                    if (!qok) a = 0.f;
                }
            }
            ra[i] = a;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k = k0 + r0 + 4 * i;
            float b = 0.f;
            if (k < Ktot) {
                const int c = k / KK, t = k - c * KK;
                const int dh = t / K, dw = t % K;
                const int h = hb + dh, w = wb + dw;
                if (qok && h >= 0 && h < H && w >= 0 && w < W) b = xp[((size_t)c * H + dh) * W + dw];
            }
            rb[i] = b;
        }
    };
    const int rblk = wv & 1, t0 = 2 * (wv >> 1);
    const bool wave_rows = o0 + 16 * rblk < Cout;
    const int tcols = Ktot - k0 >= 64 ? 4 : (Ktot - k0 + 15) / 16;
    if (q_begin < q_end) fetch(q_begin);
    for (long q0 = q_begin; q0 < q_end; q0 += 64) {
#pragma unroll
        for (int i = 0; i < 8; ++i) A[r0 + 4 * i][j] = ra[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) B[r0 + 4 * i][j] = rb[i];
        __syncthreads();
        if (q0 + 64 < q_end) fetch(q0 + 64);
        if (wave_rows && t0 < tcols) {
#pragma unroll 4
            for (int ks = 0; ks < 16; ++ks) {
                const float a = A[16 * rblk + (lane & 15)][ks * 4 + (lane >> 4)];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    if (t0 + tt < tcols) {
                        const float b = B[(t0 + tt) * 16 + (lane & 15)][ks * 4 + (lane >> 4)];
                        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[tt], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
    float* out = scratch + (size_t)bz * Cout * Ktot;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = o0 + 16 * rblk + (lane >> 4) * 4 + r, k = k0 + (t0 + tt) * 16 + (lane & 15);
            if (o < Cout && k < Ktot) out[(size_t)o * Ktot + k] = acc[tt][r];
        }
}

// ------------------------------------------------------------------------------------------------------------------------ //
// Round 5: the recorded weight gradients of the 1x1 stride-1 layers (every qkv_transform, conv_down, conv_up: most of a flush)
// with 16-BYTE loads along the position axis.  dW[o][c] = sum_q dY[o][q] X[c][q] is a plain A B^T with both operands contiguous
// in q.  The scalar-load body above handles 64 positions per global round trip and spends ~1500 SALU instructions per wave on
// per-row address arithmetic, bounds branches and coefficient loads (SQ counters, profiles/r04_step_pmc.json: SALU > VALU, 1.25
// workgroups per CU in flight: the CU's one scalar unit is the bottleneck of a launch of ~4600 short workgroups).  Here:
//   * a step is 128 positions: thread t moves float4 number (t & 31) of rows (t >> 5) + 8 i -- 4 + 4 (dY, raw) + 8 (X) loads of
//     16 bytes per thread and step, unconditional (clamped row / position, zeroed by select), saddr + 32-bit voffset addressing;
//   * the MFMA fragments are read as ds_read_b128: lane group g of MFMA j in a 16-position block takes position 4 g + j for BOTH
//     operands (the contraction does not care about the order), so one 16-byte LDS read feeds four MFMAs;
//   * chunks are 256 ... 1024 positions (2 ... 8 steps), 4x fewer workgroups and 4x fewer partial slabs for the row reduction;
//   * K = 3 (stride 1, pad 1: the decoders, conv2 / conv3 of the global stem -- the other half of a flush's workgroups): row k =
//     (c, dh, dw) of the tile reads the ALIGNED float4 of input row h + dh - 1 plus ONE neighbour element (left for dw = 0, right
//     for dw = 2) and shifts in registers -- no unaligned or out-of-tensor address, image borders zeroed by select.
// ------------------------------------------------------------------------------------------------------------------------ //
constexpr int V4_PS = 128, V4_LD = V4_PS + 4;
template <int K>
__device__ __forceinline__ void conv_wgrad_v4_body32(const float* __restrict__ dy, const float* __restrict__ raw,
                                                     const float* __restrict__ coef, const float* __restrict__ x,
                                                     float* __restrict__ scratch, int N, int Cin, int HW, int Wd, int Cout, int QS,
                                                     int npg, int bx, int by, int bz, float* smem) {
    static_assert(K == 1 || K == 3, "1x1 or 3x3 (stride 1, pad K / 2)");
    constexpr int KK = K * K;
    const int Ktot = Cin * KK;
    float* A = smem;                              // [32][V4_LD]
    float* B = smem + 32 * V4_LD;                 // [64][V4_LD]
    const int o0 = bx * 32, c0 = by * 64;         // (c0: first k = (c, tap) row of the tile)
    const unsigned NP = (unsigned)N * HW;
    const unsigned q_begin = (unsigned)bz * QS;
    const unsigned q_end = q_begin + QS < NP ? q_begin + QS : NP;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = tid >> 5, col4 = tid & 31;
    // element offsets of this thread's rows inside one image's (Cout | Cin, HW) block; rows past the end are clamped and zeroed
    unsigned aoff[4], boff[8], acf[4];
    int btap[8];                                  // K = 3: the tap t = 3 dh + dw of row i (row shift (dh - 1) W, column shift dw - 1)
    bool aok[4], bok[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int o = o0 + lrow + 8 * i;
        aok[i] = o < Cout;
        const int oc = aok[i] ? o : Cout - 1;
        aoff[i] = (unsigned)oc * HW;
        acf[i] = (unsigned)oc * 3;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = c0 + lrow + 8 * i;
        bok[i] = k < Ktot;
        const int kc = bok[i] ? k : Ktot - 1;
        const int c = kc / KK, t = kc - c * KK;
        boff[i] = (unsigned)c * HW;
        btap[i] = K == 3 ? t : 4;
    }
    f32x4 acc[2];
    acc[0] = (f32x4)(0.f);
    acc[1] = (f32x4)(0.f);
    f32x4 ra[4], rb[8];
    const char* dyb = reinterpret_cast<const char*>(dy);
    const char* rwb = reinterpret_cast<const char*>(raw);
    const char* xb = reinterpret_cast<const char*>(x);
    auto fetch = [&](unsigned q0) {
        const unsigned q = q0 + 4u * col4;
        const bool qok = q < q_end;                                  // (HW % 4 == 0: the four positions share the image)
        const unsigned qc = qok ? q : q_begin;
        const unsigned n = qc / (unsigned)HW, p = qc - n * HW;
        const unsigned abase = n * (unsigned)Cout * HW + p, bbase = n * (unsigned)Cin * HW + p;
        f32x4 rr[4];
        float cf[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(dyb + (size_t)((abase + aoff[i]) * 4u));
        if (coef) {
            const float* cg = coef + (size_t)(n / (unsigned)npg) * Cout * 3;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rr[i] = *reinterpret_cast<const f32x4*>(rwb + (size_t)((abase + aoff[i]) * 4u));
                cf[i][0] = cg[acf[i]]; cf[i][1] = cg[acf[i] + 1]; cf[i][2] = cg[acf[i] + 2];
            }
        }
        float nb[8];                                                 // K = 3: the neighbour element of the shifted taps
        bool hok[8], nok[8];
        if (K == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) rb[i] = *reinterpret_cast<const f32x4*>(xb + (size_t)((bbase + boff[i]) * 4u));
        } else {
            const int h = (int)(p / (unsigned)Wd), w0 = (int)p - h * Wd, Hd = HW / Wd;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int dh1 = btap[i] / 3 - 1, dw = btap[i] - 3 * (dh1 + 1);
                hok[i] = (unsigned)(h + dh1) < (unsigned)Hd;         // input row of the tap
                const unsigned e0 = bbase + boff[i] + (hok[i] ? dh1 * Wd : 0);      // aligned float4 of that row (this row if outside)
                rb[i] = *reinterpret_cast<const f32x4*>(xb + (size_t)(e0 * 4u));
                // left neighbour x[w0 - 1] for dw = 0, right neighbour x[w0 + 4] for dw = 2 (inside the row, else zero)
                const int wn = dw == 0 ? w0 - 1 : w0 + 4;
                nok[i] = hok[i] && dw != 1 && (unsigned)wn < (unsigned)Wd;
                nb[i] = *reinterpret_cast<const float*>(xb + (size_t)((e0 + (nok[i] ? wn - w0 : 0)) * 4u));
            }
        }
        MEDT_SCHED_FENCE();                                          // every load of the step is in flight before the first use
        if (K == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float nv = nok[i] ? nb[i] : 0.f;
                const f32x4 v = hok[i] ? rb[i] : (f32x4)(0.f);
                const int dw = btap[i] % 3;
                if (dw == 0) rb[i] = f32x4{nv, v[0], v[1], v[2]};
                else if (dw == 2) rb[i] = f32x4{v[1], v[2], v[3], nv};
                else rb[i] = v;
            }
        }
        if (coef) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[i][e] = fmaf(cf[i][0], ra[i][e], fmaf(cf[i][1], rr[i][e], cf[i][2]));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (!(qok && aok[i])) ra[i] = (f32x4)(0.f);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (!(qok && bok[i])) rb[i] = (f32x4)(0.f);
    };
    const int rblk = wv & 1, t0 = 2 * (wv >> 1);
    const float* afrag = A + (16 * rblk + (lane & 15)) * V4_LD + 4 * (lane >> 4);
    const float* bfrag = B + (t0 * 16 + (lane & 15)) * V4_LD + 4 * (lane >> 4);
    if (q_begin < q_end) fetch(q_begin);
    for (unsigned q0 = q_begin; q0 < q_end; q0 += V4_PS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(A + (lrow + 8 * i) * V4_LD + 4 * col4) = ra[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(B + (lrow + 8 * i) * V4_LD + 4 * col4) = rb[i];
        __syncthreads();
        if (q0 + V4_PS < q_end) fetch(q0 + V4_PS);                   // flies during the MFMAs below
        // UNCONDITIONAL, straight-line MFMA section: rows past Cout / Cin are zero rows of the tiles, so the waves and column blocks
        // that have nothing to add multiply zeros (the matrix pipe is < 5 % busy in this kernel).  With `if (t0 + 1 < tcols)` around
        // the second tile's MFMAs hipcc (ROCm 7.2) carried acc[0] around the loop through a v_accvgpr_mov copy placed in the loop's
        // back-edge block, two wait states behind the last v_mfma that writes it -- a gfx950 hazard it only accounts for inside a
        // basic block: the copy read a stale fourth register and every output row 4 m + 3 of single-block tiles came out wrong ON
        // THE MI355X (the CPU lane emulator knows nothing of ISA hazards and passed; tests/test_ops_gpu.py::
        // test_recorded_weight_gradient caught it, scripts/r5_dbg_v4.py + profiles/r05_v4_hazard.txt pinned it down).
#pragma unroll
        for (int blk = 0; blk < V4_PS / 16; ++blk) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(afrag + 16 * blk);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bfrag + 16 * blk);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(bfrag + 16 * V4_LD + 16 * blk);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj], b0[jj], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj], b1[jj], acc[1], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // (the accumulators are read in another basic block than the one that issues the last v_mfma: keep the matrix pipe's write-back
    //  latency between them explicitly -- see the hazard note above)
#ifndef MEDT_LANE_EMU
    MEDT_SCHED_FENCE();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    MEDT_SCHED_FENCE();
#endif
    float* out = scratch + (size_t)bz * Cout * Ktot;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = o0 + 16 * rblk + (lane >> 4) * 4 + r, k = c0 + (t0 + tt) * 16 + (lane & 15);
            if (o < Cout && k < Ktot) out[(size_t)o * Ktot + k] = acc[tt][r];
        }
}

template <int K>
__global__ __launch_bounds__(MEDT_THREADS) void conv_wgrad_mfma_kernel(
    const float* __restrict__ dy, const float* __restrict__ raw, const float* __restrict__ coef,
    const float* __restrict__ x, float* __restrict__ scratch, int N, int Cin, int H, int W, int Cout, int Ho, int Wo,
    int stride, int pad, int QS, int npg) {
    MEDT_STATIC_SHARED float A[64][65];
    MEDT_STATIC_SHARED float B[64][65];
    conv_wgrad_mfma_body<K>(dy, raw, coef, x, scratch, N, Cin, H, W, Cout, Ho, Wo, stride, pad, QS, npg, blockIdx.x,
                            blockIdx.y, blockIdx.z, A, B);
}

// Up to four of these problems in one launch (the recorded weight gradients of the local decoders' deep 3x3 layers: three
// launches of 144-288 workgroups back to back at the end of the backward pass before)
struct MWJob {
    const float *dy, *raw, *coef, *x;
    float* scratch;
    int N, Cin, H, W, Cout, Ho, Wo, stride, pad, QS, npg, gx, gy, gz, K;
};
using MWBatch = JobBatch<MWJob, 4>;
__global__ __launch_bounds__(MEDT_THREADS) void conv_wgrad_mfma_batch_kernel(MWBatch b) {
    MEDT_STATIC_SHARED float A[64][65];
    MEDT_STATIC_SHARED float B[64][65];
    const int j = find_job(b, blockIdx.x);
    const MWJob& m = b.job[j];
    const int r = blockIdx.x - b.start[j], bx = r % m.gx, by = (r / m.gx) % m.gy, bz = r / (m.gx * m.gy);
    if (m.K == 1)
        conv_wgrad_mfma_body<1>(m.dy, m.raw, m.coef, m.x, m.scratch, m.N, m.Cin, m.H, m.W, m.Cout, m.Ho, m.Wo, m.stride, m.pad,
                                m.QS, m.npg, bx, by, bz, A, B);
    else
        conv_wgrad_mfma_body<3>(m.dy, m.raw, m.coef, m.x, m.scratch, m.N, m.Cin, m.H, m.W, m.Cout, m.Ho, m.Wo, m.stride, m.pad,
                                m.QS, m.npg, bx, by, bz, A, B);
}

int conv_wgrad_mfma_batch(const MJob* const* jobs, int n, hipStream_t s) {
    if (abl_skip(jobs[0]->N >= 16 ? "wgrad_mfma_l" : "wgrad_mfma_g")) return MEDT_OK;
    for (int i0 = 0; i0 < n; i0 += 4) {
        MWBatch b;
        b.n = n - i0 < 4 ? n - i0 : 4;
        int blocks = 0;
        for (int i = 0; i < b.n; ++i) {
            const MJob& m = *jobs[i0 + i];
            MWJob& j = b.job[i];
            j.dy = m.dy; j.raw = m.raw; j.coef = m.coef; j.x = m.x; j.scratch = m.scratch;
            j.N = m.N; j.Cin = m.Cin; j.H = m.H; j.W = m.W; j.Cout = m.Cout; j.Ho = m.Ho; j.Wo = m.Wo; j.stride = m.stride;
            j.pad = m.pad; j.QS = m.QS; j.npg = m.npg; j.K = m.K;
            j.gx = cdiv(m.Cout, 64); j.gy = cdiv(m.Cin * m.K * m.K, 64); j.gz = m.splits;
            b.start[i] = blocks;
            blocks += j.gx * j.gy * j.gz;
        }
        b.start[b.n] = blocks;
        hipLaunchKernelGGL(conv_wgrad_mfma_batch_kernel, dim3(blocks), dim3(MEDT_THREADS), 0, s, b);
        int rc = launch_status("conv_wgrad_mfma_batch");
        if (rc) return rc;
    }
    return MEDT_OK;
}

// The recorded weight gradients of many layers -- all kernel sizes -- in ONE launch (defer.h): 64 x 64 tiles on the
// matrix cores for every layer.  Per FMA the MFMA tile reads 16x fewer LDS bytes than the 4x4 register tile
// (profiles/r02_wgrad_ab.json).  A flush holds 20-40 layers and 1-3 thousand workgroups; issued per kernel size the
// launches ran back to back, most of them with fewer workgroups than the chip has slots (the 7x7 stem: 96), so their
// time was the serial latency of one workgroup's position chunk three times over.  One launch overlaps them, and the
// jobs are ordered longest chunk first so the long workgroups start first.
struct WJobP {                       // WJob packed for the kernel-argument block (< 4 KB): 44 jobs per launch
    const float *dy, *raw, *coef, *x;
    float* scratch;
    int N, Cin, H, W, Cout, Ho, Wo, QS, npg, gz;
    unsigned char stride, pad, K, v4;           // v4: conv_wgrad_v4_body32<K> (1x1 / 3x3, stride 1, 16-byte aligned rows)
};
using WBatch = JobBatch<WJobP, 42>;
static_assert(sizeof(WBatch) <= 4000, "job table must fit the kernel-argument block");
template <int TO>
__global__ __launch_bounds__(MEDT_THREADS, 3) void conv_wgrad_mfma_grouped_kernel(WBatch b) {      // (3 workgroups per CU: the LDS limit)
    // one LDS block for both tile movers: A[TO][65] | B[64][65] of the scalar-load bodies, A[32][132] | B[64][132] of the 16-byte one
    MEDT_STATIC_SHARED __attribute__((aligned(16))) float smem[(TO == 32 ? 96 * V4_LD : (TO + 64) * 65)];
    float (*A)[65] = reinterpret_cast<float (*)[65]>(smem);
    float (*B)[65] = reinterpret_cast<float (*)[65]>(smem + TO * 65);
    const int j = find_job(b, blockIdx.x);
    const WJobP& w = b.job[j];
    const int local = blockIdx.x - b.start[j];
    const int gx = (w.Cout + TO - 1) / TO, gy = (w.Cin * w.K * w.K + 63) / 64;
    const int bx = local % gx, t = local / gx, by = t % gy, bz = t / gy;
    if constexpr (TO == 32) {
        if (w.v4) {
            if (w.K == 1) conv_wgrad_v4_body32<1>(w.dy, w.raw, w.coef, w.x, w.scratch, w.N, w.Cin, w.H * w.W, w.W, w.Cout, w.QS, w.npg, bx, by, bz, smem);
            else conv_wgrad_v4_body32<3>(w.dy, w.raw, w.coef, w.x, w.scratch, w.N, w.Cin, w.H * w.W, w.W, w.Cout, w.QS, w.npg, bx, by, bz, smem);
            return;
        }
    }
#define MEDT_WG_BODY(KV)                                                                                                 \
    do {                                                                                                                 \
        if constexpr (TO == 64)                                                                                          \
            conv_wgrad_mfma_body<KV>(w.dy, w.raw, w.coef, w.x, w.scratch, w.N, w.Cin, w.H, w.W, w.Cout, w.Ho, w.Wo,      \
                                     w.stride, w.pad, w.QS, w.npg, bx, by, bz, A, B);                                    \
        else                                                                                                             \
            conv_wgrad_mfma_body32<KV>(w.dy, w.raw, w.coef, w.x, w.scratch, w.N, w.Cin, w.H, w.W, w.Cout, w.Ho, w.Wo,    \
                                       w.stride, w.pad, w.QS, w.npg, bx, by, bz, A, B);                                  \
    } while (0)
    if (w.K == 1) MEDT_WG_BODY(1);
    else if (w.K == 3) MEDT_WG_BODY(3);
    else MEDT_WG_BODY(7);
#undef MEDT_WG_BODY
}

// The 16-byte body's preconditions (the chunk policy of conv.hip asks before it sizes the job's chunks)
bool conv_wgrad_v4_ok(const float* dy, const float* raw, const float* x, int N, int Cin, int H, int W, int Cout, int Ho, int Wo,
                      int K, int stride, int pad) {
    if (stride != 1 || Ho != H || Wo != W) return false;
    if (!((K == 1 && pad == 0 && (H * W) % 4 == 0) || (K == 3 && pad == 1 && W % 4 == 0))) return false;
    if ((((uintptr_t)dy | (uintptr_t)raw | (uintptr_t)x) & 15) != 0) return false;
    const size_t HW = (size_t)H * W;
    return (size_t)N * Cout * HW * 4 < 0xffffffffull && (size_t)N * Cin * HW * 4 < 0xffffffffull;      // 32-bit byte offsets
}

int conv_wgrad_grouped(const WJob* jobs, int n, hipStream_t s) {
    constexpr int to = 32;        // o-tile (round 5: the 64-row tile and the VALU body lost their A/B and are gone)
    // longest workgroups first: steps of 64 positions per chunk, weighted by the taps a step gathers
    std::vector<int> order(n);
    for (int j = 0; j < n; ++j) order[j] = j;
    auto cost = [&](int j) { return (long)((jobs[j].QS + 63) / 64) * (jobs[j].K == 1 ? 2 : 3); };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost(a) > cost(b); });
    WBatch b;
    b.n = 0;
    int blocks = 0;
    auto launch = [&]() -> int {
        b.start[b.n] = blocks;
#ifdef MEDT_ABLATE                      // (timing-experiment build only: the job table of a launch, MEDT_WG_DEBUG=1 -> profiles/r06_wgrad_jobs.txt)
        if (getenv("MEDT_WG_DEBUG")) {
            fprintf(stderr, "wgrad grouped: %d jobs, %d blocks\n", b.n, blocks);
            for (int j = 0; j < b.n; ++j) {
                const WJobP& w = b.job[j];
                fprintf(stderr, "  wjob K%d v4 %d Cout %3d Cin %3d HoWo %3dx%3d N %2d QS %4d blocks %4d (o-tiles %d k-tiles %d chunks %d)\n", (int)w.K, (int)w.v4,
                        w.Cout, w.Cin, w.Ho, w.Wo, w.N, w.QS, b.start[j + 1] - b.start[j], (w.Cout + 31) / 32, (w.Cin * w.K * w.K + 63) / 64, w.gz);
            }
        }
#endif
        hipLaunchKernelGGL(conv_wgrad_mfma_grouped_kernel<32>, dim3(blocks), dim3(MEDT_THREADS), 0, s, b);
        b.n = 0;
        blocks = 0;
        return launch_status("conv_wgrad_mfma_grouped");
    };
    for (int i = 0; i < n; ++i) {
        const WJob& w = jobs[order[i]];
        if (w.K != 1 && w.K != 3 && w.K != 7) { set_error("conv2d: kernel size %d unsupported (1, 3, 7)", w.K); return MEDT_EUNSUPPORTED; }
        b.job[b.n] = WJobP{w.dy, w.raw, w.coef, w.x, w.scratch, w.N, w.Cin, w.H, w.W, w.Cout, w.Ho, w.Wo, w.QS, w.npg, w.gz,
                           (unsigned char)w.stride, (unsigned char)w.pad, (unsigned char)w.K, (unsigned char)(to == 32 && w.v4)};
        b.start[b.n] = blocks;
        blocks += cdiv(w.Cout, to) * w.gy * w.gz;
        if (++b.n == 42) { int rc = launch(); if (rc) return rc; }
    }
    if (b.n) { int rc = launch(); if (rc) return rc; }
    return MEDT_OK;
}

// (Round 5, measured and removed: the three kinds of recorded MFMA weight gradients of a flush -- grouped tiles, LDS-patch problems,
//  few-tile problems -- as ONE launch with a mixed job table.  The LDS-patch body needs 213 + 40 registers, so the merged grid holds 2
//  workgroups per CU: with the grouped tiles inside the step was 32 us slower (2.201 vs 2.168 ms), with only the two dedicated kinds
//  merged it was neutral (2.078 vs 2.082 ms: the merged kernel took 71 us = 53 + 18, the LDS-patch workgroups fill every slot first).
//  profiles/r05_step_ab.json.)
bool conv_wgrad_rows16_ok(int Cin, int H, int W, int Ho, int Wo, int K, int stride, int pad, int QS) {
    return QS % 64 == 0 && Ho == H && Wo == W && conv_rows16_ok(Cin, H, W, K, stride, pad);
}

// the LDS-patch weight-gradient kernel for up to four recorded problems at once
int conv_wgrad_rows16_grouped(const MJob* const* jobs, int n, hipStream_t s) {
    if (abl_skip(jobs[0]->N >= 16 ? "wgrad_mfma_l" : "wgrad_mfma_g")) return MEDT_OK;
    // output channels per workgroup.  Measured (profiles/r05_step_ab.json): the half-width instance is 12 us SLOWER here (56.8 us per
    // launch: twice the workgroups restage the patch, and the pair is matrix-pipe-bound already) -- MEDT_R16W_OT=32 selects it
    static const int ot = 64;
    for (int i0 = 0; i0 < n; i0 += 4) {
        R16WBatch b;
        b.n = n - i0 < 4 ? n - i0 : 4;
        int blocks = 0;
        for (int i = 0; i < b.n; ++i) {
            const MJob& m = *jobs[i0 + i];
            R16WJob& j = b.job[i];
            j.dy = m.dy; j.raw = m.raw; j.coef = m.coef; j.x = m.x; j.scratch = m.scratch;
            j.Cin = m.Cin; j.H = m.H; j.Cout = m.Cout; j.tiles_per_split = m.QS / 64; j.ptiles = m.N * m.H / 4; j.npg = m.npg;
            j.gx = cdiv(m.Cout, ot); j.gy = m.Cin / 16; j.gz = m.splits;
            b.start[i] = blocks;
            blocks += j.gx * j.gy * j.gz;
        }
        b.start[b.n] = blocks;
        if (ot == 32) hipLaunchKernelGGL(conv3x3_rows16_wgrad_kernel<32>, dim3(blocks), dim3(MEDT_THREADS), 0, s, b);
        else hipLaunchKernelGGL(conv3x3_rows16_wgrad_kernel<64>, dim3(blocks), dim3(MEDT_THREADS), 0, s, b);
        int rc = launch_status("conv3x3_rows16_wgrad");
        if (rc) return rc;
    }
    return MEDT_OK;
}

int conv_wgrad_mfma(const float* dy, const float* raw, const float* coef, const float* x, float* scratch, int N, int Cin,
                    int H, int W, int Cout, int Ho, int Wo, int K, int stride, int pad, int QS, int splits, int npg,
                    hipStream_t s) {
    const dim3 grid(cdiv(Cout, 64), cdiv(Cin * K * K, 64), splits), block(MEDT_THREADS);
    if (abl_skip(N >= 16 ? "wgrad_mfma_l" : "wgrad_mfma_g")) return MEDT_OK;
    if (conv_wgrad_rows16_ok(Cin, H, W, Ho, Wo, K, stride, pad, QS)) {
        const MJob m{dy, raw, coef, x, scratch, N, Cin, H, W, Cout, Ho, Wo, K, stride, pad, QS, splits, npg};
        const MJob* one = &m;
        return conv_wgrad_rows16_grouped(&one, 1, s);
    }
    if (K == 1)
        hipLaunchKernelGGL(conv_wgrad_mfma_kernel<1>, grid, block, 0, s, dy, raw, coef, x, scratch, N, Cin, H, W, Cout, Ho,
                           Wo, stride, pad, QS, npg);
    else
        hipLaunchKernelGGL(conv_wgrad_mfma_kernel<3>, grid, block, 0, s, dy, raw, coef, x, scratch, N, Cin, H, W, Cout, Ho,
                           Wo, stride, pad, QS, npg);
    return launch_status("conv_wgrad_mfma");
}

}  // namespace medt
