// axial_small.hip -- whole AxialAttention_wopos layers in one workgroup per (BatchNorm group, head).
//
// MedT's local branch runs 16 position-free attention layers on 16x16 ... 4x4 maps of 4-image patch groups
// (lib/models/axialnet.py:222-253, 346-391, 661-700).  Every BatchNorm of such a layer is per channel, the channels
// of bn_qkv / bn_output partition by head, bn_similarity has one channel per head, and the batch statistics are
// per patch group: the work of one (group, head) pair -- qkv_transform rows, bn_qkv statistics, logit statistics,
// softmax, P.V, bn_output statistics, AvgPool -- touches a few thousand floats and depends on nothing outside the
// pair.  One 256-thread workgroup does all of it out of LDS; what used to be seven dependent launches per layer
// (conv1x1, 3 x bn_finalize, logit statistics, attention, output pass: ~5 us each, latency bound) becomes this kernel
// plus one merged finalisation (running statistics need the groups in order, bn_finalize3 in pointwise.hip).
// The saved tensors (qkv_raw, stacked, lse, statistics) are exactly those of the layer-by-layer path, so the
// backward entry point does not care which forward ran.
#include "medt_common.h"
#include "medt_kernels.h"
#include <type_traits>

namespace medt {

struct SmallFwdArgs {
    const float* x;                     // (N, C, H, W)
    const float* w;                     // (2C, C)
    medt_bn_ptrs bq, bs, bo;            // bn_qkv (2C), bn_similarity (G), bn_output (C)
    float *qkv_raw, *stacked, *lse, *y;
    float *part_q, *part_s, *part_o;    // [groups][CH][2] sum / sum of squares for the finalisation pass
    int N, C, H, W, G, gp, L, npg, stride, training, out_relu;
    float eps;
};

// scale/shift of one channel, bit-identical to bn_finalize_kernel (pointwise.hip) given the same float sums
__device__ __forceinline__ void small_scale_shift(float s, float ss, double count, const medt_bn_ptrs& bn, int ch, float eps,
                                                  int training, float& scale, float& shift) {
    const float g = bn.weight[ch], b = bn.bias[ch];
    if (training) {
        const double mean = (double)s / count;
        double var = (double)ss / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        scale = (float)(g * rstd);
        shift = (float)(b - mean * g * rstd);
    } else {
        const float mean = bn.running_mean[ch];
        const float rstd = (float)(1.0 / sqrt((double)bn.running_var[ch] + (double)eps));
        scale = g * rstd;
        shift = b - mean * g * rstd;
    }
}

template <int AXIS, int L, int GP>
__global__ __launch_bounds__(MEDT_THREADS) void wopos_small_fwd_kernel(SmallFwdArgs a) {
    constexpr int HQ = GP / 2, NCH = 2 * GP, RMAX = 4;          // up to 4 rows (positions) per thread: P <= 1024
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int grp = blockIdx.x, hg = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = a.C, W = a.W, HW = a.H * a.W, P = a.npg * HW;
    const int n0 = grp * a.npg;
    float* Q = smem;                    // [NCH][P]  q | k | v rows of this head, raw then normalised
    float* S = Q + NCH * P;             // [GP][P]   sv
    float* sc = S + GP * P;             // [32] scale
    float* sh = sc + 32;                // [32] shift
    float* red = sh + 32;               // [64] reduction scratch
    float* Wl = red + 64;               // [C][NCH] qkv_transform rows of this head, transposed

    // 1. qkv_transform rows hg*2gp .. +2gp                                               (axialnet.py:228)
    //    A work item is a position and a chunk of NOC <= 16 output channels: x is loaded once per item (16 loads in
    //    flight: one workgroup per CU has nothing else to hide the L2 latency) and the 256 threads stay busy when
    //    the group has fewer than 256 positions.
    for (int e = tid; e < NCH * C; e += MEDT_THREADS) {
        const int oc = e / C, c = e - oc * C;
        Wl[c * NCH + oc] = a.w[(size_t)(hg * NCH + oc) * C + c];
    }
    __syncthreads();
    {
        constexpr int NOC_MAX = NCH < 16 ? NCH : 16;
        int chunks = P >= MEDT_THREADS ? 1 : MEDT_THREADS / P;
        if (chunks > NCH / 4) chunks = NCH / 4;
        if (chunks < NCH / NOC_MAX) chunks = NCH / NOC_MAX;
        const int noc = NCH / chunks;                           // 4, 8 or 16
        auto project = [&](auto cb_tag) {
            constexpr int CB = decltype(cb_tag)::value;         // x values in flight per thread
            for (int item = tid; item < P * chunks; item += MEDT_THREADS) {
                const int chunk = item / P, q = item - chunk * P, ni = q / HW, p = q - ni * HW, oc0 = chunk * noc;
                const float* xp = a.x + ((size_t)(n0 + ni) * C) * HW + p;
                float acc[NOC_MAX];
#pragma unroll
                for (int o = 0; o < NOC_MAX; ++o) acc[o] = 0.f;
                for (int c0 = 0; c0 < C; c0 += CB) {
                    float xv[CB];
#pragma unroll
                    for (int k = 0; k < CB; ++k) xv[k] = xp[(size_t)(c0 + k) * HW];
#pragma unroll
                    for (int k = 0; k < CB; ++k) {
                        const float* wr = Wl + (c0 + k) * NCH + oc0;
#pragma unroll
                        for (int o = 0; o < NOC_MAX; o += 4)
                            if (o < noc) {
                                const float4 w4 = *reinterpret_cast<const float4*>(wr + o);
                                acc[o] = fmaf(w4.x, xv[k], acc[o]);
                                acc[o + 1] = fmaf(w4.y, xv[k], acc[o + 1]);
                                acc[o + 2] = fmaf(w4.z, xv[k], acc[o + 2]);
                                acc[o + 3] = fmaf(w4.w, xv[k], acc[o + 3]);
                            }
                    }
                }
#pragma unroll
                for (int o = 0; o < NOC_MAX; ++o)
                    if (o < noc) {
                        Q[(oc0 + o) * P + q] = acc[o];
                        a.qkv_raw[((size_t)(n0 + ni) * 2 * C + hg * NCH + oc0 + o) * HW + p] = acc[o];
                    }
            }
        };
        if ((C & 31) == 0) project(std::integral_constant<int, 32>{});
        else project(std::integral_constant<int, 16>{});
    }
    __syncthreads();
    // 2. bn_qkv: batch statistics over the group's positions (one wave per channel)      (:228)
    for (int oc = wave; oc < NCH; oc += MEDT_WAVES) {
        float s = 0.f, ss = 0.f;
        for (int q = lane; q < P; q += 64) {
            const float v = Q[oc * P + q];
            s += v;
            ss = fmaf(v, v, ss);
        }
        s = wave_sum(s);
        ss = wave_sum(ss);
        if (lane == 0) {
            const int ch = hg * NCH + oc;
            small_scale_shift(s, ss, (double)P, a.bq, ch, a.eps, a.training, sc[oc], sh[oc]);
            if (a.training) {
                a.part_q[((size_t)grp * 2 * C + ch) * 2] = s;
                a.part_q[((size_t)grp * 2 * C + ch) * 2 + 1] = ss;
            }
        }
    }
    __syncthreads();
    for (int item = tid; item < NCH * P; item += MEDT_THREADS) {
        const int oc = item / P;
        Q[item] = fmaf(Q[item], sc[oc], sh[oc]);
    }
    __syncthreads();
    // 3. logits of this thread's rows (position q is query i of its sequence; key j sits at base + j*sj), kept in
    //    registers across the bn_similarity reduction                                     (:232-236)
    constexpr int SJ1 = 1;
    const int sj = AXIS == 1 ? SJ1 : W;
    float z[RMAX][L];
    int rbase[RMAX];
    float v[2] = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        const int q = tid + r * MEDT_THREADS;
        if (q < P) {
            const int i = AXIS == 1 ? q % W : (q % HW) / W;
            rbase[r] = q - i * sj;
            float qv[HQ];
#pragma unroll
            for (int c = 0; c < HQ; ++c) qv[c] = Q[c * P + q];
#pragma unroll
            for (int j = 0; j < L; ++j) {
                float qk = 0.f;
#pragma unroll
                for (int c = 0; c < HQ; ++c) qk = fmaf(qv[c], Q[(HQ + c) * P + rbase[r] + j * sj], qk);
                z[r][j] = qk;
                v[0] += qk;
                v[1] = fmaf(qk, qk, v[1]);
            }
        }
    }
    block_sum<2>(v, red, red + 32);
    if (tid == 0) {
        float scale, shift;
        small_scale_shift(red[32], red[33], (double)P * L, a.bs, hg, a.eps, a.training, scale, shift);
        red[40] = scale;
        if (a.training) {
            a.part_s[((size_t)grp * a.G + hg) * 2] = red[32];
            a.part_s[((size_t)grp * a.G + hg) * 2 + 1] = red[33];
        }
    }
    __syncthreads();
    const float a_qk = red[40] * MEDT_LOG2E;     // the shift is constant along a softmax row
    // 4. softmax + P.V per row; sv stays in LDS for the output statistics                 (:237-241)
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        const int q = tid + r * MEDT_THREADS;
        if (q < P) {
            const int ni = q / HW, p = q - ni * HW;
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < L; ++j) { z[r][j] *= a_qk; m = fmaxf(m, z[r][j]); }
            float l = 0.f, acc[GP];
#pragma unroll
            for (int c = 0; c < GP; ++c) acc[c] = 0.f;
#pragma unroll
            for (int j = 0; j < L; ++j) {
                const float pj = __builtin_amdgcn_exp2f(z[r][j] - m);
                l += pj;
#pragma unroll
                for (int c = 0; c < GP; ++c) acc[c] = fmaf(pj, Q[(GP + c) * P + rbase[r] + j * sj], acc[c]);
            }
            const float inv = 1.f / l;
#pragma unroll
            for (int c = 0; c < GP; ++c) {
                const float o = acc[c] * inv;
                S[c * P + q] = o;
                a.stacked[((size_t)(n0 + ni) * C + hg * GP + c) * HW + p] = o;
            }
            a.lse[((size_t)(n0 + ni) * a.G + hg) * HW + p] = m + __log2f(l);
        }
    }
    __syncthreads();
    // 5. bn_output statistics                                                             (:242)
    for (int c = wave; c < GP; c += MEDT_WAVES) {
        float s = 0.f, ss = 0.f;
        for (int q = lane; q < P; q += 64) {
            const float x = S[c * P + q];
            s += x;
            ss = fmaf(x, x, ss);
        }
        s = wave_sum(s);
        ss = wave_sum(ss);
        if (lane == 0) {
            const int ch = hg * GP + c;
            small_scale_shift(s, ss, (double)P, a.bo, ch, a.eps, a.training, sc[c], sh[c]);
            if (a.training) {
                a.part_o[((size_t)grp * C + ch) * 2] = s;
                a.part_o[((size_t)grp * C + ch) * 2 + 1] = ss;
            }
        }
    }
    __syncthreads();
    // 6. bn_output apply + AvgPool2d(stride) [+ the block's ReLU]                          (:242-253, :381-383)
    const int st = a.stride, Ho = a.H / st, Wo = a.W / st, HoWo = Ho * Wo;
    const float pool = 1.f / (float)(st * st);
    for (int item = tid; item < GP * a.npg * HoWo; item += MEDT_THREADS) {
        const int c = item / (a.npg * HoWo), r = item - c * a.npg * HoWo, ni = r / HoWo, po = r - ni * HoWo;
        const int ho = po / Wo, wo = po - ho * Wo;
        const float* src = S + c * P + ni * HW;
        float acc = 0.f;
        for (int dh = 0; dh < st; ++dh)
            for (int dw = 0; dw < st; ++dw) acc += fmaf(sc[c], src[(ho * st + dh) * W + wo * st + dw], sh[c]);
        acc *= pool;
        a.y[((size_t)(n0 + ni) * C + hg * GP + c) * HoWo + po] = a.out_relu ? fmaxf(acc, 0.f) : acc;
    }
}

static size_t small_lds_bytes(int gp, int P, int C) { return ((size_t)3 * gp * P + 32 + 32 + 64 + 2 * gp * C) * sizeof(float); }

static bool small_enabled() {
    static const bool on = [] { const char* e = getenv("MEDT_DISABLE_SMALL"); return !(e && e[0] == '1'); }();
    return on;
}

// The fused path applies to position-free layers whose (group, head) slice fits one workgroup.
bool wopos_small_ok(const AxialGeom& g, const medt_axial_desc& d) {
    if (!small_enabled() || g.pos) return false;
    const int P = g.npg * g.HW;
    if (P > 1024 || (g.C & 15)) return false;
    if (g.L != 4 && g.L != 8 && g.L != 16) return false;
    if (g.gp != 2 && g.gp != 4 && g.gp != 8 && g.gp != 16) return false;
    if (d.stride < 1 || d.H % d.stride || d.W % d.stride) return false;
    return small_lds_bytes(g.gp, P, g.C) <= 64 * 1024;
}

int wopos_small_fwd(const AxialGeom& g, const medt_axial_desc& d, const medt_axial_params& p, const float* x, float* y,
                    float* qkv_raw, float* stacked, float* lse, float* part_q, float* part_s, float* part_o,
                    hipStream_t s) {
    SmallFwdArgs a;
    a.x = x; a.w = p.w_qkv;
    a.bq = p.bn_qkv; a.bs = p.bn_similarity; a.bo = p.bn_output;
    a.qkv_raw = qkv_raw; a.stacked = stacked; a.lse = lse; a.y = y;
    a.part_q = part_q; a.part_s = part_s; a.part_o = part_o;
    a.N = g.N; a.C = g.C; a.H = g.H; a.W = g.W; a.G = g.G; a.gp = g.gp; a.L = g.L; a.npg = g.npg;
    a.stride = d.stride; a.training = d.training ? 1 : 0; a.out_relu = d.out_relu; a.eps = d.eps;
    const dim3 grid(g.groups, g.G), block(MEDT_THREADS);
    const size_t lds = small_lds_bytes(g.gp, g.npg * g.HW, g.C);
#define MEDT_SMALL(AX, Lv, GPv) hipLaunchKernelGGL((wopos_small_fwd_kernel<AX, Lv, GPv>), grid, block, lds, s, a)
#define MEDT_SMALL_GP(AX, Lv)                                                                       \
    switch (g.gp) {                                                                                 \
        case 2: MEDT_SMALL(AX, Lv, 2); break;                                                       \
        case 4: MEDT_SMALL(AX, Lv, 4); break;                                                       \
        case 8: MEDT_SMALL(AX, Lv, 8); break;                                                       \
        default: MEDT_SMALL(AX, Lv, 16); break;                                                     \
    }
#define MEDT_SMALL_L(AX)                                                                            \
    switch (g.L) {                                                                                  \
        case 4: MEDT_SMALL_GP(AX, 4) break;                                                         \
        case 8: MEDT_SMALL_GP(AX, 8) break;                                                         \
        default: MEDT_SMALL_GP(AX, 16) break;                                                       \
    }
    if (g.axis == 1) { MEDT_SMALL_L(1) } else { MEDT_SMALL_L(0) }
#undef MEDT_SMALL_L
#undef MEDT_SMALL_GP
#undef MEDT_SMALL
    return launch_status("wopos_small_fwd");
}

}  // namespace medt
