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
#include "defer.h"
#include <type_traits>

namespace medt {

// -DMEDT_STAMPS (scripts/phase_stamps.sh builds a second library with it): thread 0 of workgroup (0, 0) of the fused
// forward kernel records the 100 MHz wall clock at its phase boundaries; medt_debug_stamps() copies them out.
#ifdef MEDT_STAMPS
__device__ unsigned long long g_stamps[16];
#define MEDT_STAMP(i)                                                                          \
    do {                                                                                       \
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_stamps[i] = wall_clock64(); \
    } while (0)
#else
#define MEDT_STAMP(i) do { } while (0)
#endif

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
__device__ __forceinline__ void small_scale_shift(double s, double ss, double count, const medt_bn_ptrs& bn, int ch, float eps,
                                                  int training, float& scale, float& shift) {
    const float g = bn.weight[ch], b = bn.bias[ch];
    if (training) {
        const double mean = s / count;
        double var = ss / count - mean * mean;
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

// The same from per-channel parameters that were prefetched into LDS at kernel start (prm[4]: weight, bias, running mean,
// running variance): the lane that finalises a BatchNorm does not start a dependent global round trip in mid-kernel.
__device__ __forceinline__ void small_scale_shift_p(double s, double ss, double count, const float* prm, float eps, int training,
                                                    float& scale, float& shift) {
    const float g = prm[0], b = prm[1];
    if (training) {
        const double mean = s / count;
        double var = ss / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        scale = (float)(g * rstd);
        shift = (float)(b - mean * g * rstd);
    } else {
        const float mean = prm[2];
        const float rstd = (float)(1.0 / sqrt((double)prm[3] + (double)eps));
        scale = g * rstd;
        shift = b - mean * g * rstd;
    }
}

// block_sum for workgroups of NW waves (256 or 512 threads)
template <int K>
__device__ __forceinline__ void small_block_sum(float (&v)[K], float* red, float* dst, int NW) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float s = wave_sum(v[k]);
        if (lane == 0) red[wave * K + k] = s;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        float s = 0.f;
        for (int w = 0; w < NW; ++w) s += red[w * K + k];
        dst[k] = s;
    }
    __syncthreads();
}

// input-channel split of the projection phase: P * KS <= threads, slices of a multiple of 8 channels
static __host__ __device__ inline int small_ksplit(int P, int T, int C) {
    int ks = 1;
    while (ks * 2 <= 16 && ks * 2 * P <= T && C % (ks * 2 * 8) == 0) ks *= 2;
    return ks;
}

template <int AXIS, int L, int GP>
__global__ __launch_bounds__(512) void wopos_small_fwd_kernel(SmallFwdArgs a) {
    constexpr int HQ = GP / 2, NCH = 2 * GP, RMAX = 2;          // up to 2 rows (positions) per thread: P <= 2 * blockDim.x
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int grp = blockIdx.x, hg = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = a.C, W = a.W, HW = a.H * a.W, P = a.npg * HW;
    const int n0 = grp * a.npg, T = blockDim.x, NW = T >> 6;
    float* Q = smem;                    // [NCH][P]  q | k | v rows of this head, raw then normalised
    float* S = Q + NCH * P;             // [GP][P]   sv
    float* sc = S + GP * P;             // [32] scale
    float* sh = sc + 32;                // [32] shift
    float* red = sh + 32;               // [64] reduction scratch
    float* Wl = red + 64;               // [C][NCH] qkv_transform rows of this head, transposed
    float* prm = Wl + C * NCH;          // [NCH + 1 + GP][4] BatchNorm parameters of this head's channels (bn_qkv | sim | out)
    MEDT_STAMP(0);
    if (tid < NCH + 1 + GP) {
        const medt_bn_ptrs& bn = tid < NCH ? a.bq : (tid == NCH ? a.bs : a.bo);
        const int ch = tid < NCH ? hg * NCH + tid : (tid == NCH ? hg : hg * GP + (tid - NCH - 1));
        prm[tid * 4] = bn.weight[ch];
        prm[tid * 4 + 1] = bn.bias[ch];
        if (!a.training) {
            prm[tid * 4 + 2] = bn.running_mean[ch];
            prm[tid * 4 + 3] = bn.running_var[ch];
        }
    }

    // 1. qkv_transform rows hg*2gp .. +2gp                                               (axialnet.py:228)
    //    The whole kernel is a chain of dependent global round trips, so this phase is built to need one: the weight rows
    //    and the first 32 input channels of the thread's position are requested together, and when the group has fewer
    //    positions than the workgroup has threads the input channels are split KS ways over the threads (partial sums
    //    combined through LDS in a fixed order), so one batch of loads per thread covers its slice.
    {
        const int KS = small_ksplit(P, T, C), cpk = C / KS;
        float* Zp = prm + 4 * (NCH + 1 + GP);                   // [KS][NCH][P] partial sums (KS > 1)
        constexpr int WB = 8, XB = GP >= 16 ? 16 : 32;          // (32 accumulators per position when gp = 16)
        const int nW = NCH * C;
        float wr[WB], xv[XB];
        auto load_w = [&](int base) {
#pragma unroll
            for (int k = 0; k < WB; ++k) {
                const int e = min(base + tid + k * T, nW - 1), oc = e / C;
                wr[k] = a.w[(size_t)(hg * NCH + oc) * C + (e - oc * C)];
            }
        };
        auto store_w = [&](int base) {
#pragma unroll
            for (int k = 0; k < WB; ++k) {
                const int e = base + tid + k * T;
                if (e < nW) { const int oc = e / C; Wl[(e - oc * C) * NCH + oc] = wr[k]; }
            }
        };
        const int nitems = KS > 1 ? 1 : (P + T - 1) / T;         // KS > 1: P * KS <= T, one (position, slice) per thread
        auto locate = [&](int r, bool& act, int& ks, int& q, const float*& xp) {
            const int item = tid + r * T;
            act = item < P * KS;
            ks = act ? item / P : 0;
            q = act ? item - ks * P : 0;
            const int ni = q / HW, p = q - ni * HW;
            xp = a.x + ((size_t)(n0 + ni) * C + (size_t)ks * cpk) * HW + p;
        };
        auto load_x = [&](const float* xp, int c0) {
#pragma unroll
            for (int k8 = 0; k8 < XB; k8 += 8)
                if (c0 + k8 < cpk) {                             // (uniform: slices are multiples of 8 channels)
#pragma unroll
                    for (int k = k8; k < k8 + 8; ++k) xv[k] = xp[(size_t)(c0 + k) * HW];
                }
        };
        bool act; int ks, q; const float* xp;
        locate(0, act, ks, q, xp);
        load_w(0);
        load_x(xp, 0);                                          // both batches in flight together
        MEDT_SCHED_FENCE();
        store_w(0);
        for (int base = WB * T; base < nW; base += WB * T) {
            load_w(base);
            MEDT_SCHED_FENCE();
            store_w(base);
        }
        __syncthreads();
        MEDT_STAMP(1);                                          // weights in LDS, first input batch in registers
        for (int r = 0; r < nitems; ++r) {
            if (r) locate(r, act, ks, q, xp);
            float acc[NCH];
#pragma unroll
            for (int o = 0; o < NCH; ++o) acc[o] = 0.f;
            for (int c0 = 0; c0 < cpk; c0 += XB) {
                if (r || c0) {
                    load_x(xp, c0);
                    MEDT_SCHED_FENCE();
                }
#pragma unroll
                for (int k8 = 0; k8 < XB; k8 += 8)
                    if (c0 + k8 < cpk) {                         // slices are multiples of 8 channels
#pragma unroll
                        for (int k = k8; k < k8 + 8; ++k) {
                            const float* wrow = Wl + (ks * cpk + c0 + k) * NCH;
#pragma unroll
                            for (int o = 0; o < NCH; o += 4) {
                                const float4 w4 = *reinterpret_cast<const float4*>(wrow + o);
                                acc[o] = fmaf(w4.x, xv[k], acc[o]);
                                acc[o + 1] = fmaf(w4.y, xv[k], acc[o + 1]);
                                acc[o + 2] = fmaf(w4.z, xv[k], acc[o + 2]);
                                acc[o + 3] = fmaf(w4.w, xv[k], acc[o + 3]);
                            }
                        }
                    }
            }
            if (act) {
                if (KS == 1) {
                    const int ni = q / HW, p = q - ni * HW;
#pragma unroll
                    for (int o = 0; o < NCH; ++o) {
                        Q[o * P + q] = acc[o];
                        a.qkv_raw[((size_t)(n0 + ni) * 2 * C + hg * NCH + o) * HW + p] = acc[o];
                    }
                } else {
#pragma unroll
                    for (int o = 0; o < NCH; ++o) Zp[(ks * NCH + o) * P + q] = acc[o];
                }
            }
        }
        if (KS > 1) {
            __syncthreads();
            for (int item = tid; item < NCH * P; item += T) {
                const int oc = item / P, qq = item - oc * P, ni = qq / HW, p = qq - ni * HW;
                float v = Zp[item];
                for (int k = 1; k < KS; ++k) v += Zp[k * NCH * P + item];
                Q[item] = v;
                a.qkv_raw[((size_t)(n0 + ni) * 2 * C + hg * NCH + oc) * HW + p] = v;
            }
        }
    }
    __syncthreads();
    MEDT_STAMP(2);                                              // projection done, q | k | v rows in LDS
    // 2. bn_qkv: batch statistics over the group's positions (one wave per channel)      (:228)
    for (int oc = wave; oc < NCH; oc += NW) {
        float s = 0.f, ss = 0.f;
        for (int q = lane; q < P; q += 64) s += Q[oc * P + q];
        s = wave_sum(s);
        const float m = s * (1.f / (float)P);           // second pass: squares about the mean (centered_to_raw, medt_common.h)
        for (int q = lane; q < P; q += 64) {
            const float dv = Q[oc * P + q] - m;
            ss = fmaf(dv, dv, ss);
        }
        ss = wave_sum(ss);
        if (lane == 0) {
            red[oc * 2] = s;
            red[oc * 2 + 1] = ss;
        }
    }
    __syncthreads();
    MEDT_STAMP(3);                                              // bn_qkv sums
    if (tid < NCH) {                                   // the double-precision finalisations side by side, not one per wave turn
        const int ch = hg * NCH + tid;
        double s, ss;
        centered_to_raw(red[tid * 2], red[tid * 2 + 1], red[tid * 2] * (1.f / (float)P), (double)P, s, ss);
        small_scale_shift_p(s, ss, (double)P, prm + tid * 4, a.eps, a.training, sc[tid], sh[tid]);
        if (a.training) {                              // doubles, like every forward BatchNorm partial
            double* dp = reinterpret_cast<double*>(a.part_q) + ((size_t)grp * 2 * C + ch) * 2;
            dp[0] = s;
            dp[1] = ss;
        }
    }
    __syncthreads();
    MEDT_STAMP(4);                                              // bn_qkv finalised (double)
    for (int q = tid; q < P; q += T) {
#pragma unroll
        for (int oc = 0; oc < NCH; ++oc) Q[oc * P + q] = fmaf(Q[oc * P + q], sc[oc], sh[oc]);
    }
    __syncthreads();
    MEDT_STAMP(5);                                              // normalised
    // 3. logits of this thread's rows (position q is query i of its sequence; key j sits at base + j*sj), kept in
    //    registers across the bn_similarity reduction                                     (:232-236)
    constexpr int SJ1 = 1;
    const int sj = AXIS == 1 ? SJ1 : W;
    float z[RMAX][L];
    int rbase[RMAX];
    float v[2] = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        const int q = tid + r * T;
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
    small_block_sum<2>(v, red, red + 32, NW);
    MEDT_STAMP(6);                                              // logits + their sums
    if (tid == 0) {
        float scale, shift;
        small_scale_shift_p((double)red[32], (double)red[33], (double)P * L, prm + NCH * 4, a.eps, a.training, scale, shift);
        red[40] = scale;
        if (a.training) {
            double* dp = reinterpret_cast<double*>(a.part_s) + ((size_t)grp * a.G + hg) * 2;
            dp[0] = (double)red[32];
            dp[1] = (double)red[33];
        }
    }
    __syncthreads();
    MEDT_STAMP(7);                                              // bn_similarity finalised
    const float a_qk = red[40] * MEDT_LOG2E;     // the shift is constant along a softmax row
    // 4. softmax + P.V per row; sv stays in LDS for the output statistics                 (:237-241)
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        const int q = tid + r * T;
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
    MEDT_STAMP(8);                                              // softmax + P.V
    // 5. bn_output statistics                                                             (:242)
    for (int c = wave; c < GP; c += NW) {
        float s = 0.f, ss = 0.f;
        for (int q = lane; q < P; q += 64) s += S[c * P + q];
        s = wave_sum(s);
        const float m = s * (1.f / (float)P);
        for (int q = lane; q < P; q += 64) {
            const float dv = S[c * P + q] - m;
            ss = fmaf(dv, dv, ss);
        }
        ss = wave_sum(ss);
        if (lane == 0) {
            red[c * 2] = s;
            red[c * 2 + 1] = ss;
        }
    }
    __syncthreads();
    MEDT_STAMP(9);                                              // bn_output sums
    if (tid < GP) {
        const int ch = hg * GP + tid;
        double s, ss;
        centered_to_raw(red[tid * 2], red[tid * 2 + 1], red[tid * 2] * (1.f / (float)P), (double)P, s, ss);
        small_scale_shift_p(s, ss, (double)P, prm + (NCH + 1 + tid) * 4, a.eps, a.training, sc[tid], sh[tid]);
        if (a.training) {
            double* dp = reinterpret_cast<double*>(a.part_o) + ((size_t)grp * C + ch) * 2;
            dp[0] = s;
            dp[1] = ss;
        }
    }
    __syncthreads();
    MEDT_STAMP(10);                                             // bn_output finalised
    // 6. bn_output apply + AvgPool2d(stride) [+ the block's ReLU]                          (:242-253, :381-383)
    const int st = a.stride, Ho = a.H / st, Wo = a.W / st, HoWo = Ho * Wo;
    if (st == 1) {
        // no pooling: one position per thread turn, its GP channels in a compile-time loop (the generic loop below spends
        // its time on three runtime divisions per element: 2.6 of the kernel's 18 us, scripts/phase_stamps.py)
        for (int q = tid; q < P; q += T) {
            const int ni = q / HW, p = q - ni * HW;
            float* dst = a.y + ((size_t)(n0 + ni) * C + hg * GP) * HW + p;
#pragma unroll
            for (int c = 0; c < GP; ++c) {
                const float v = fmaf(sc[c], S[c * P + q], sh[c]);
                dst[(size_t)c * HW] = a.out_relu ? fmaxf(v, 0.f) : v;
            }
        }
        MEDT_STAMP(11);
        return;
    }
    const float pool = 1.f / (float)(st * st);
    for (int item = tid; item < GP * a.npg * HoWo; item += T) {
        const int c = item / (a.npg * HoWo), r = item - c * a.npg * HoWo, ni = r / HoWo, po = r - ni * HoWo;
        const int ho = po / Wo, wo = po - ho * Wo;
        const float* src = S + c * P + ni * HW;
        float acc = 0.f;
        for (int dh = 0; dh < st; ++dh)
            for (int dw = 0; dw < st; ++dw) acc += fmaf(sc[c], src[(ho * st + dh) * W + wo * st + dw], sh[c]);
        acc *= pool;
        a.y[((size_t)(n0 + ni) * C + hg * GP + c) * HoWo + po] = a.out_relu ? fmaxf(acc, 0.f) : acc;
    }
    MEDT_STAMP(11);
}

static size_t small_lds_bytes(int gp, int P, int C) {
    const int ks = small_ksplit(P, P > 512 ? 512 : 256, C);
    return ((size_t)3 * gp * P + 32 + 32 + 64 + 2 * gp * C + 4 * (3 * gp + 1) + (ks > 1 ? (size_t)ks * 2 * gp * P : 0)) *
           sizeof(float);
}

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
    if (abl_skip("wopos_fwd")) return MEDT_OK;
    SmallFwdArgs a;
    a.x = x; a.w = p.w_qkv;
    a.bq = p.bn_qkv; a.bs = p.bn_similarity; a.bo = p.bn_output;
    a.qkv_raw = qkv_raw; a.stacked = stacked; a.lse = lse; a.y = y;
    a.part_q = part_q; a.part_s = part_s; a.part_o = part_o;
    a.N = g.N; a.C = g.C; a.H = g.H; a.W = g.W; a.G = g.G; a.gp = g.gp; a.L = g.L; a.npg = g.npg;
    a.stride = d.stride; a.training = d.training ? 1 : 0; a.out_relu = d.out_relu; a.eps = d.eps;
    const dim3 grid(g.groups, g.G), block(g.npg * g.HW > 512 ? 512 : 256);
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

// --------------------------------------------------------------------------- //
// Backward of the same layers, core part: from dy to the gradient at the output of qkv_transform (before the
// bn_qkv backward affine, which the conv kernels apply on load), again one workgroup per (BN group, head).
// Replaces relu_mask, axial_out_bwd_stats, attn_bwd_stats, attn_bwd and the in-between finalisations' critical-path
// role: the BatchNorm backward coefficients of bn_output and bn_similarity are group-local and computed here; the
// per-group partial sums still go out so the finalisation kernels produce the parameter gradients (sum over the
// groups) and bn_qkv's coefficients exactly as on the layer-by-layer path.
// --------------------------------------------------------------------------- //
struct SmallBwdArgs {
    const float *qkv_raw, *stacked, *lse, *dy, *y;      // y: forward output, only read when out_relu
    BnStats sq, ss, so;                                 // saved statistics of bn_qkv, bn_similarity, bn_output
    const float *w_out, *w_sim;                         // bn_output.weight (C), bn_similarity.weight (G)
    float *dqkv;                                        // (N, 2C, H, W)
    float *part_ob, *part_sb, *part_qb;                 // [groups][C][2], [groups][G][4], [groups][2C][2]
    const float* w_qkv_bn;                              // bn_qkv.weight (2C)
    float* coef_qkv;                                    // [groups][2C][3]: bn_qkv backward as dx = c0*d + c1*x + c2
    double row_count;
    int N, C, H, W, G, npg, stride, training, out_relu;
};

template <int AXIS, int L, int GP>
__global__ __launch_bounds__(512) void wopos_small_bwd_kernel(SmallBwdArgs a) {
    constexpr int HQ = GP / 2, NCH = 2 * GP, R = L == 16 ? 2 : 1;     // rows (positions) per thread (512 threads when L = 16)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int grp = blockIdx.x, hg = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = a.C, W = a.W, HW = a.H * a.W, P = a.npg * HW, G = a.G;
    const int n0 = grp * a.npg, T = blockDim.x, NW = T >> 6;
    float* Q = smem;                    // [NCH][P] normalised q | k | v; later the gradients of the same rows
    float* D = Q + NCH * P;             // [GP][P]  d(loss)/d(sv)
    float* lse = D + GP * P;            // [P]
    float* dlt = lse + P;               // [P]      Delta_i = sum_c dsv[c,i] sv[c,i]
    float* red = dlt + P;               // [576]    reduction scratch (8 waves x 32 values + 64 results)
    float* cf = red + 576;              // [3*GP] bn_output coefficients, then [8] bn_similarity (e, u, w)
    float* prm = cf + 80;               // saved statistics / weights the later phases need, prefetched now:
    float* p_out = prm;                 //   [GP][3]  bn_output mean, rstd, weight
    float* p_sim = p_out + 3 * GP;      //   [4]      bn_similarity scale, mean, rstd, weight
    float* p_qkv = p_sim + 4;           //   [NCH][3] bn_qkv mean, rstd, weight
    const int st = a.stride, Ho = a.H / st, Wo = a.W / st;
    const double cnt = (double)P;
    if (tid < GP) {
        const int ch = hg * GP + tid;
        p_out[tid * 3] = a.so.mean[grp * C + ch];
        p_out[tid * 3 + 1] = a.so.rstd[grp * C + ch];
        p_out[tid * 3 + 2] = a.w_out[ch];
    } else if (tid == GP) {
        p_sim[0] = a.ss.scale[grp * G + hg];
        p_sim[1] = a.ss.mean[grp * G + hg];
        p_sim[2] = a.ss.rstd[grp * G + hg];
        p_sim[3] = a.w_sim[hg];
    } else if (tid >= 64 && tid < 64 + NCH) {
        const int ch = hg * NCH + (tid - 64);
        p_qkv[(tid - 64) * 3] = a.sq.mean[grp * 2 * C + ch];
        p_qkv[(tid - 64) * 3 + 1] = a.sq.rstd[grp * 2 * C + ch];
        p_qkv[(tid - 64) * 3 + 2] = a.w_qkv_bn[ch];
    }

    // Every global value the first phases need is requested in one go (a branch or a wait per element would turn this
    // into a dozen dependent round trips): the row log-sum-exps, the first batch of q|k|v rows with their bn_qkv affine,
    // and dy / y / stacked of this thread's positions.  Threads past the last position read position P-1 (ignored).
    constexpr int QB = 8;
    const int nQ = NCH * P;
    float lsev[R], rawv[QB], scv[QB], shv[QB];
    auto load_q = [&](int base) {
#pragma unroll
        for (int k = 0; k < QB; ++k) {
            const int item = min(base + tid + k * T, nQ - 1);
            const int oc = item / P, q = item - oc * P, ni = q / HW, p = q - ni * HW, ch = hg * NCH + oc;
            rawv[k] = a.qkv_raw[((size_t)(n0 + ni) * 2 * C + ch) * HW + p];
            scv[k] = a.sq.scale[grp * 2 * C + ch];
            shv[k] = a.sq.shift[grp * 2 * C + ch];
        }
    };
    auto store_q = [&](int base) {                          // normalised q|k|v of this head
#pragma unroll
        for (int k = 0; k < QB; ++k) {
            const int item = base + tid + k * T;
            if (item < nQ) Q[item] = fmaf(rawv[k], scv[k], shv[k]);
        }
    };
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int q = min(tid + r * T, P - 1), ni = q / HW, p = q - ni * HW;
        lsev[r] = a.lse[((size_t)(n0 + ni) * G + hg) * HW + p];
    }
    load_q(0);
    // 1. AvgPool2d + ReLU-mask + bn_output backward                                        (axialnet.py:242-253)
    float gy[R][GP], sv[R][GP], yv[R][GP];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int q = min(tid + r * T, P - 1), ni = q / HW, p = q - ni * HW, h = p / W, w = p - h * W;
        const size_t po = (size_t)(h / st) * Wo + (w / st);
#pragma unroll
        for (int c = 0; c < GP; ++c) {
            const int ch = hg * GP + c;
            const size_t yo = ((size_t)(n0 + ni) * C + ch) * Ho * Wo + po;
            gy[r][c] = a.dy[yo];
            if (a.out_relu) yv[r][c] = a.y[yo];
            sv[r][c] = a.stacked[((size_t)(n0 + ni) * C + ch) * HW + p];
        }
    }
    MEDT_SCHED_FENCE();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int q = tid + r * T;
        if (q < P) lse[q] = lsev[r];
    }
    store_q(0);
    for (int base = QB * T; base < nQ; base += QB * T) {
        load_q(base);
        MEDT_SCHED_FENCE();
        store_q(base);
    }
    __syncthreads();                                        // (also publishes the prefetched statistics p_out / p_sim / p_qkv)
    {
        float v[2 * GP];
#pragma unroll
        for (int k = 0; k < 2 * GP; ++k) v[k] = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool valid = tid + r * T < P;
#pragma unroll
            for (int c = 0; c < GP; ++c) {
                float d = gy[r][c];
                if (a.out_relu && !(yv[r][c] > 0.f)) d = 0.f;
                if (!valid) d = 0.f;
                gy[r][c] = d;
                v[2 * c] += d;
                v[2 * c + 1] += d * ((sv[r][c] - p_out[3 * c]) * p_out[3 * c + 1]);
            }
        }
        small_block_sum<2 * GP>(v, red, red + 512, NW);
        if (tid < GP) {
            const int ch = hg * GP + tid;
            const float dscale = 1.f / (float)(st * st);
            const float r1 = red[512 + 2 * tid], r2 = red[512 + 2 * tid + 1];
            a.part_ob[((size_t)grp * C + ch) * 2] = r1;
            a.part_ob[((size_t)grp * C + ch) * 2 + 1] = r2;
            // same arithmetic as bn_bwd_finalize_kernel (pointwise.hip)
            const double s1 = (double)r1 * dscale, s2 = (double)r2 * dscale;
            const double mean = p_out[3 * tid], rstd = p_out[3 * tid + 1], A = (double)p_out[3 * tid + 2] * rstd;
            cf[3 * tid] = (float)(A * dscale);
            if (a.training) {
                const double m1 = s1 / cnt, m2 = s2 / cnt;
                cf[3 * tid + 1] = (float)(-A * rstd * m2);
                cf[3 * tid + 2] = (float)(A * (rstd * mean * m2 - m1));
            } else {
                cf[3 * tid + 1] = 0.f;
                cf[3 * tid + 2] = 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int q = tid + r * T;
            if (q < P) {
                float dl = 0.f;
#pragma unroll
                for (int c = 0; c < GP; ++c) {
                    const float d = fmaf(cf[3 * c], gy[r][c], fmaf(cf[3 * c + 1], sv[r][c], cf[3 * c + 2]));
                    D[c * P + q] = d;
                    dl = fmaf(d, sv[r][c], dl);
                }
                dlt[q] = dl;
            }
        }
    }
    __syncthreads();
    const int sj = AXIS == 1 ? 1 : W;
    const float a_qk = p_sim[0] * MEDT_LOG2E;
    // everything one (i, j) pair contributes; qi/kj: positions of query i and key j
    auto pair = [&](int qi, int kj, float& S, float& Pij, float& dZ) {
        S = 0.f;
#pragma unroll
        for (int c = 0; c < HQ; ++c) S = fmaf(Q[c * P + qi], Q[(HQ + c) * P + kj], S);
        Pij = __builtin_amdgcn_exp2f(fmaf(S, a_qk, -lse[qi]));
        float dP = 0.f;
#pragma unroll
        for (int c = 0; c < GP; ++c) dP = fmaf(D[c * P + qi], Q[(GP + c) * P + kj], dP);
        dZ = Pij * (dP - dlt[qi]);
    };
    // 2. bn_similarity backward statistics: sum dZ, sum dZ * S over the group              (:236)
    {
        float v[2] = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int q = tid + r * T;
            if (q < P) {
                const int i = AXIS == 1 ? q % W : (q % HW) / W, base = q - i * sj;
#pragma unroll
                for (int j = 0; j < L; ++j) {
                    float S, Pij, dZ;
                    pair(q, base + j * sj, S, Pij, dZ);
                    v[0] += dZ;
                    v[1] = fmaf(dZ, S, v[1]);
                }
            }
        }
        small_block_sum<2>(v, red, red + 512, NW);
        if (tid == 0) {
            const float a0f = red[512], axf = red[513];
            float* ps = a.part_sb + ((size_t)grp * G + hg) * 4;
            ps[0] = a0f; ps[1] = axf; ps[2] = 0.f; ps[3] = 0.f;
            // same arithmetic as sim_bwd_finalize_kernel / sim_coef (axial_core.hip)
            const double a0 = a0f, ax = axf, count = cnt * L;
            const double mean = p_sim[1], rstd = p_sim[2];
            const double sxh = rstd * (ax - mean * a0), e = (double)p_sim[3] * rstd;
            cf[64] = (float)e;
            if (a.training) {
                const double m1 = a0 / count, m2 = sxh / count, u = -e * rstd * m2;
                cf[65] = (float)u;
                cf[66] = (float)(-e * m1 - u * mean);
            } else {
                cf[65] = 0.f;
                cf[66] = 0.f;
            }
        }
        __syncthreads();
    }
    const float ce = cf[64], cu = cf[65], cw = cf[66];
    // 3. dq (this thread's positions as queries) and dk, dv (as keys)                       (:232-241)
    float gq[R][NCH];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) gq[r][k] = 0.f;
        const int q = tid + r * T;
        if (q < P) {
            const int i = AXIS == 1 ? q % W : (q % HW) / W, base = q - i * sj;
#pragma unroll
            for (int j = 0; j < L; ++j) {
                const int o = base + j * sj;
                float S, Pij, dZ;
                pair(q, o, S, Pij, dZ);                       // o as key of query q
                const float dS = fmaf(ce, dZ, fmaf(cu, S, cw));
#pragma unroll
                for (int c = 0; c < HQ; ++c) gq[r][c] = fmaf(dS, Q[(HQ + c) * P + o], gq[r][c]);
                pair(o, q, S, Pij, dZ);                       // q as key of query o
                const float dS2 = fmaf(ce, dZ, fmaf(cu, S, cw));
#pragma unroll
                for (int c = 0; c < HQ; ++c) gq[r][HQ + c] = fmaf(dS2, Q[c * P + o], gq[r][HQ + c]);
#pragma unroll
                for (int c = 0; c < GP; ++c) gq[r][GP + c] = fmaf(Pij, D[c * P + o], gq[r][GP + c]);
            }
        }
    }
    __syncthreads();                                           // Q is free: it now receives the gradients
    // 4. gradient at the bn_qkv output -> global, and its bn_qkv backward statistics        (:228)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int q = tid + r * T;
        if (q < P) {
            const int ni = q / HW, p = q - ni * HW;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                Q[k * P + q] = gq[r][k];
                a.dqkv[((size_t)(n0 + ni) * 2 * C + hg * NCH + k) * HW + p] = gq[r][k];
            }
        }
    }
    __syncthreads();
    {
        // each wave reduces channels wave, wave + NW, ...: the saved pre-BatchNorm activations of all of them are requested
        // in one batch (4 positions per lane at a time) instead of one global round trip per channel
        constexpr int NWc = R == 2 ? 8 : 4, CPW = (NCH + NWc - 1) / NWc, QB = 4;
        float s1[CPW], s2[CPW];
#pragma unroll
        for (int j = 0; j < CPW; ++j) s1[j] = s2[j] = 0.f;
        for (int q0 = lane; q0 < P; q0 += 64 * QB) {
            float rw[CPW][QB];
#pragma unroll
            for (int j = 0; j < CPW; ++j) {
                const int ch = hg * NCH + min(wave + j * NWc, NCH - 1);
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    const int q = min(q0 + 64 * u, P - 1), ni = q / HW, p = q - ni * HW;
                    rw[j][u] = a.qkv_raw[((size_t)(n0 + ni) * 2 * C + ch) * HW + p];
                }
            }
            MEDT_SCHED_FENCE();
#pragma unroll
            for (int j = 0; j < CPW; ++j) {
                const int oc = wave + j * NWc;
                if (oc < NCH) {
                    const float mean = p_qkv[3 * oc], rstd = p_qkv[3 * oc + 1];
#pragma unroll
                    for (int u = 0; u < QB; ++u) {
                        const int q = q0 + 64 * u;
                        if (q < P) {
                            const float g = Q[oc * P + q];
                            s1[j] += g;
                            s2[j] = fmaf(g, (rw[j][u] - mean) * rstd, s2[j]);
                        }
                    }
                }
            }
            MEDT_SCHED_FENCE();
        }
#pragma unroll
        for (int j = 0; j < CPW; ++j) {
            const int oc = wave + j * NWc;
            if (oc < NCH) {
                const float t1 = wave_sum(s1[j]), t2 = wave_sum(s2[j]);
                if (lane == 0) {
                    red[oc * 2] = t1;
                    red[oc * 2 + 1] = t2;
                }
            }
        }
    }
    __syncthreads();
    if (tid < NCH) {                                   // the double-precision parts side by side, not one per wave turn
        const int ch = hg * NCH + tid;
        const float s1 = red[tid * 2], s2 = red[tid * 2 + 1], mean = p_qkv[3 * tid], rstd = p_qkv[3 * tid + 1];
        a.part_qb[((size_t)grp * 2 * C + ch) * 2] = s1;
        a.part_qb[((size_t)grp * 2 * C + ch) * 2 + 1] = s2;
        // bn_qkv's backward coefficients of this (group, channel) only need this workgroup's own sums: written here
        // so the 1x1 dgrad behind does not wait for the finalisation launch (which is left with parameter gradients)
        const double A = (double)p_qkv[3 * tid + 2] * (double)rstd;
        float* cfo = a.coef_qkv + ((size_t)grp * 2 * C + ch) * 3;
        cfo[0] = (float)A;
        if (a.training) {
            const double m1 = (double)s1 / a.row_count, m2 = (double)s2 / a.row_count;
            cfo[1] = (float)(-A * (double)rstd * m2);
            cfo[2] = (float)(A * ((double)rstd * (double)mean * m2 - m1));
        } else {
            cfo[1] = 0.f;
            cfo[2] = 0.f;
        }
    }
}

static size_t small_bwd_lds_bytes(int gp, int P) { return ((size_t)3 * gp * P + 2 * P + 576 + 80 + 9 * gp + 4) * sizeof(float); }

bool wopos_small_bwd_ok(const AxialGeom& g, const medt_axial_desc& d) {
    if (!wopos_small_ok(g, d)) return false;
    const int P = g.npg * g.HW;
    if (P > (g.L == 16 ? 1024 : 256)) return false;          // R rows per thread x 512 / 256 threads
    return small_bwd_lds_bytes(g.gp, P) <= 64 * 1024;
}

int wopos_small_bwd(const AxialGeom& g, const medt_axial_desc& d, const medt_axial_params& p, const float* y,
                    const float* dy, const float* qkv_raw, const float* stacked, const float* lse, BnStats sq, BnStats ss,
                    BnStats so, float* dqkv, float* part_ob, float* part_sb, float* part_qb, float* coef_qkv,
                    hipStream_t s) {
    if (abl_skip("wopos_bwd")) return MEDT_OK;
    SmallBwdArgs a;
    a.w_qkv_bn = p.bn_qkv.weight; a.coef_qkv = coef_qkv; a.row_count = g.row_count;
    a.qkv_raw = qkv_raw; a.stacked = stacked; a.lse = lse; a.dy = dy; a.y = y;
    a.sq = sq; a.ss = ss; a.so = so;
    a.w_out = p.bn_output.weight; a.w_sim = p.bn_similarity.weight;
    a.dqkv = dqkv; a.part_ob = part_ob; a.part_sb = part_sb; a.part_qb = part_qb;
    a.N = g.N; a.C = g.C; a.H = g.H; a.W = g.W; a.G = g.G; a.npg = g.npg;
    a.stride = d.stride; a.training = d.training ? 1 : 0; a.out_relu = d.out_relu;
    const dim3 grid(g.groups, g.G), block(g.L == 16 ? 512 : 256);
    const size_t lds = small_bwd_lds_bytes(g.gp, g.npg * g.HW);
#define MEDT_SMALL(AX, Lv, GPv) hipLaunchKernelGGL((wopos_small_bwd_kernel<AX, Lv, GPv>), grid, block, lds, s, a)
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
    return launch_status("wopos_small_bwd");
}

// Parameter gradients of the three BatchNorms (sum over the BN groups of the per-group partials written by the kernel
// above) and bn_qkv's backward coefficients for the conv kernels, in one launch: block -> (BN, channel), lane -> group.
// Same formulas as bn_bwd_finalize_kernel (pointwise.hip) and sim_bwd_finalize_kernel (axial_core.hip).
__device__ __forceinline__ double small_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ void wopos_small_bwd_finalize_body(const SmallFinArgs& a, int ch) {
    const int lane = threadIdx.x, C = a.C, G = a.G;
    double dg = 0.0, db = 0.0;
    if (ch < C) {                                               // bn_output
        for (int grp = lane; grp < a.groups; grp += 64) {
            const float* q = a.part_ob + ((size_t)grp * C + ch) * 2;
            db += (double)q[0] * a.dscale_out;
            dg += (double)q[1] * a.dscale_out;
        }
        dg = small_wave_sum_d(dg);
        db = small_wave_sum_d(db);
        if (lane == 0) { a.d_out_w[ch] = (float)dg; a.d_out_b[ch] = (float)db; }
        return;
    }
    ch -= C;
    if (ch < G) {                                               // bn_similarity
        for (int grp = lane; grp < a.groups; grp += 64) {
            const float* q = a.part_sb + ((size_t)grp * G + ch) * 4;
            const double a0 = q[0], ax = q[1];
            const double mean = a.ss.mean[grp * G + ch], rstd = a.ss.rstd[grp * G + ch];
            dg += rstd * (ax - mean * a0);
            db += a0;
        }
        dg = small_wave_sum_d(dg);
        db = small_wave_sum_d(db);
        if (lane == 0) { a.d_sim_w[ch] = (float)dg; a.d_sim_b[ch] = (float)db; }
        return;
    }
    ch -= G;                                                    // bn_qkv (its coefficients come from the backward kernel)
    const int CH = 2 * C;
    for (int grp = lane; grp < a.groups; grp += 64) {
        const float* q = a.part_qb + ((size_t)grp * CH + ch) * 2;
        dg += (double)q[1];
        db += (double)q[0];
    }
    dg = small_wave_sum_d(dg);
    db = small_wave_sum_d(db);
    if (lane == 0) { a.d_qkv_w[ch] = (float)dg; a.d_qkv_b[ch] = (float)db; }
}

__global__ __launch_bounds__(64) void wopos_small_bwd_finalize_kernel(SmallFinArgs a) {
    wopos_small_bwd_finalize_body(a, blockIdx.x);
}

using SfBatch = JobBatch<SmallFinArgs, 16>;
__global__ __launch_bounds__(64) void wopos_small_bwd_finalize_grouped_kernel(SfBatch b) {
    const int j = find_job(b, blockIdx.x);
    wopos_small_bwd_finalize_body(b.job[j], blockIdx.x - b.start[j]);
}

int wopos_small_bwd_finalize_grouped(const SmallFinArgs* jobs, int n, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += 16) {
        SfBatch b;
        b.n = n - i0 < 16 ? n - i0 : 16;
        int blocks = 0;
        for (int i = 0; i < b.n; ++i) { b.job[i] = jobs[i0 + i]; b.start[i] = blocks; blocks += 3 * jobs[i0 + i].C + jobs[i0 + i].G; }
        b.start[b.n] = blocks;
        hipLaunchKernelGGL(wopos_small_bwd_finalize_grouped_kernel, dim3(blocks), dim3(64), 0, s, b);
        int rc = launch_status("wopos_small_bwd_finalize_grouped");
        if (rc) return rc;
    }
    return MEDT_OK;
}

// BatchNorm parameter gradients of the layer (sums over the groups); recorded into `q` when given
int wopos_small_bwd_finalize(const AxialGeom& g, const medt_axial_desc& d, const medt_axial_params& p,
                             const float* part_ob, const float* part_sb, const float* part_qb, BnStats sq, BnStats ss,
                             BnStats so, const medt_axial_grads& gr, hipStream_t s, Queue* q) {
    SmallFinArgs a;
    a.part_ob = part_ob; a.part_sb = part_sb; a.part_qb = part_qb;
    a.so = so; a.ss = ss; a.sq = sq;
    a.d_out_w = gr.bn_out_weight; a.d_out_b = gr.bn_out_bias;
    a.d_sim_w = gr.bn_sim_weight; a.d_sim_b = gr.bn_sim_bias;
    a.d_qkv_w = gr.bn_qkv_weight; a.d_qkv_b = gr.bn_qkv_bias;
    a.C = g.C; a.G = g.G; a.groups = g.groups; a.training = d.training ? 1 : 0;
    a.row_count = g.row_count; a.sim_count = g.sim_count;
    a.dscale_out = 1.f / (float)(d.stride * d.stride);
    if (q) { q->sfin.push_back(a); return MEDT_OK; }
    hipLaunchKernelGGL(wopos_small_bwd_finalize_kernel, dim3(g.C + g.G + 2 * g.C), dim3(64), 0, s, a);
    return launch_status("wopos_small_bwd_finalize");
}

}  // namespace medt

#ifdef MEDT_STAMPS
extern "C" int medt_debug_stamps(unsigned long long* out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(medt::g_stamps), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
