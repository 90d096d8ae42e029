// conv_small.hip -- 1x1 convolution + train-mode BatchNorm (+ residual, ReLU) blocks whose BatchNorm group is small
// enough for one workgroup (the conv_down / conv_up / downsample blocks of MedT's local branch: 4-image patch groups on
// 16x16 ... 2x2 maps, lib/models/axialnet.py:346-391, 450-454).
//
// Forward: one workgroup per (BN group, chunk of output channels) computes the convolution for every position of the
// group, so the batch statistics of its channels are exact block-level sums and the normalise + residual + ReLU pass
// runs in the same kernel: conv -> bn_finalize -> bn_apply_act becomes this kernel + bn_finalize (saved statistics and
// the group-ordered running-stat updates).  Backward: one wave per (BN group, channel) does the ReLU mask, the two
// BatchNorm-backward sums and dz = c0*g + c1*z + c2: bn_act_bwd_stats -> bn_bwd_finalize -> bn_bwd_apply becomes this
// kernel + bn_bwd_finalize (parameter gradients: sums over the groups).  Same saved tensors either way.
#include "medt_common.h"
#include "medt_kernels.h"
#include <stdlib.h>
#include <type_traits>

namespace medt {

struct SmallConvArgs {
    const float *x, *w, *res;
    medt_bn_ptrs bn;
    float *z, *y, *partials;            // partials: [groups][Cout][2]
    int N, Cin, H, W, Cout, stride, npg, relu, training;
    float eps;
};

// identical arithmetic to bn_finalize_kernel (pointwise.hip) given the same float sums
__device__ __forceinline__ void conv_small_scale_shift(double s, double ss, double count, const medt_bn_ptrs& bn, int ch,
                                                       float eps, int training, float& scale, float& shift) {
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

// The same from parameters prefetched into LDS at kernel start (prm[4]: weight, bias, running mean, running variance)
__device__ __forceinline__ void conv_small_scale_shift_p(double s, double ss, double count, const float* prm, float eps,
                                                         int training, float& scale, float& shift) {
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

constexpr int SC_NOC_MAX = 16;          // output channels per workgroup (8 when the group's 16 x P tile does not fit LDS)

// Everything in this kernel is a chain of dependent global round trips (one workgroup, a few thousand floats), so the
// structure minimises them: the weight slab and the input values of the first 32 channels are requested together, and
// when the group has fewer positions than the workgroup has threads the input channels are split KS ways over the
// threads (partial sums combined through LDS in a fixed order) so that every thread's loads fit one batch.
static __host__ __device__ inline int conv_small_threads(int P) { return P <= 256 ? 256 : (P <= 512 ? 512 : 1024); }
static __host__ __device__ inline int conv_small_ksplit(int P, int Cin) {
    int ks = 1;
    const int T = conv_small_threads(P);
    while (ks * 2 <= 16 && ks * 2 * P <= T && Cin % (ks * 2 * 8) == 0) ks *= 2;      // slices of a multiple of 8 channels
    return ks;
}

template <int T, int SC_NOC>                        // T = 256 ... 1024 threads (one position per thread when the group has
__global__ __launch_bounds__(T) void conv1x1_bn_small_fwd_kernel(SmallConvArgs a) {      // 1024), SC_NOC output channels
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int grp = blockIdx.x, oc0 = blockIdx.y * SC_NOC, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = T >> 6;
    const int Cin = a.Cin, W = a.W, HW = a.H * a.W, st = a.stride, Ho = a.H / st, Wo = a.W / st, HoWo = Ho * Wo;
    const int P = a.npg * HoWo, n0 = grp * a.npg;
    const int KS = conv_small_ksplit(P, Cin), cpk = Cin / KS;
    float* Z = smem;                    // [SC_NOC][P]
    float* Wl = Z + SC_NOC * P;         // [Cin][SC_NOC]
    float* sc = Wl + Cin * SC_NOC;      // [SC_NOC]
    float* sh = sc + SC_NOC;            // [SC_NOC]
    float* prm = sh + SC_NOC;           // [SC_NOC][4] BatchNorm parameters, prefetched (no dependent round trip in mid-kernel)
    float* sums = prm + 4 * SC_NOC;     // [SC_NOC][2]
    float* Zp = sums + 2 * SC_NOC;      // [KS][SC_NOC][P] partial sums of the channel slices (KS > 1)
    if (tid < SC_NOC) {
        const int ch = oc0 + tid;
        prm[tid * 4] = a.bn.weight[ch];
        prm[tid * 4 + 1] = a.bn.bias[ch];
        if (!a.training) {
            prm[tid * 4 + 2] = a.bn.running_mean[ch];
            prm[tid * 4 + 3] = a.bn.running_var[ch];
        }
    }
    // this thread's work item: position q, channel slice ks
    const bool act = tid < P * KS;
    const int ks = act ? tid / P : 0, q = act ? tid - ks * P : 0, ni = q / HoWo, po = q - ni * HoWo;
    const int ho = po / Wo, wo = po - ho * Wo;
    const float* xp = a.x + ((size_t)(n0 + ni) * Cin + (size_t)ks * cpk) * HW + (ho * st) * W + wo * st;
    constexpr int WB = 8, XB = T == 1024 ? 16 : 32;   // (1024 threads: 128 VGPRs per lane)
    const int nW = SC_NOC * Cin;
    float wr[WB], xv[XB];
    auto load_w = [&](int base) {
#pragma unroll
        for (int k = 0; k < WB; ++k) {
            const int e = min(base + tid + k * T, nW - 1), oc = e / Cin;
            wr[k] = a.w[(size_t)(oc0 + oc) * Cin + (e - oc * Cin)];
        }
    };
    auto store_w = [&](int base) {
#pragma unroll
        for (int k = 0; k < WB; ++k) {
            const int e = base + tid + k * T;
            if (e < nW) { const int oc = e / Cin; Wl[(e - oc * Cin) * SC_NOC + oc] = wr[k]; }
        }
    };
    auto load_x = [&](int c0) {
        if constexpr (T == 1024) {                   // (128 VGPRs per lane: the branch-free form allocates fewer)
#pragma unroll
            for (int k = 0; k < XB; ++k) xv[k] = xp[(size_t)min(c0 + k, cpk - 1) * HW];
        } else {
#pragma unroll
            for (int k8 = 0; k8 < XB; k8 += 8)
                if (c0 + k8 < cpk) {                 // (uniform: slices are multiples of 8 channels)
#pragma unroll
                    for (int k = k8; k < k8 + 8; ++k) xv[k] = xp[(size_t)(c0 + k) * HW];
                }
        }
    };
    load_w(0);
    load_x(0);                                        // both batches in flight together
    MEDT_SCHED_FENCE();
    store_w(0);
    for (int base = WB * T; base < nW; base += WB * T) {
        load_w(base);
        MEDT_SCHED_FENCE();
        store_w(base);
    }
    __syncthreads();
    float acc[SC_NOC];
#pragma unroll
    for (int o = 0; o < SC_NOC; ++o) acc[o] = 0.f;
    for (int c0 = 0; c0 < cpk; c0 += XB) {
        if (c0) {
            load_x(c0);
            MEDT_SCHED_FENCE();
        }
#pragma unroll
        for (int k8 = 0; k8 < XB; k8 += 8)
            if (c0 + k8 < cpk) {                     // slices are multiples of 8 channels
#pragma unroll
                for (int k = k8; k < k8 + 8; ++k) {
                    const float* wrow = Wl + (ks * cpk + c0 + k) * SC_NOC;
#pragma unroll
                    for (int o = 0; o < SC_NOC; o += 4) {
                        const float4 w4 = *reinterpret_cast<const float4*>(wrow + o);
                        acc[o] = fmaf(w4.x, xv[k], acc[o]);
                        acc[o + 1] = fmaf(w4.y, xv[k], acc[o + 1]);
                        acc[o + 2] = fmaf(w4.z, xv[k], acc[o + 2]);
                        acc[o + 3] = fmaf(w4.w, xv[k], acc[o + 3]);
                    }
                }
            }
    }
    if (KS == 1) {
        if (act) {
#pragma unroll
            for (int o = 0; o < SC_NOC; ++o) {
                Z[o * P + q] = acc[o];
                a.z[((size_t)(n0 + ni) * a.Cout + oc0 + o) * HoWo + po] = acc[o];
            }
        }
    } else {
        if (act) {
#pragma unroll
            for (int o = 0; o < SC_NOC; ++o) Zp[(ks * SC_NOC + o) * P + q] = acc[o];
        }
        __syncthreads();
        for (int item = tid; item < SC_NOC * P; item += T) {
            const int oc = item / P, qq = item - oc * P, n2 = qq / HoWo, p2 = qq - n2 * HoWo;
            float v = Zp[item];
            for (int k = 1; k < KS; ++k) v += Zp[k * SC_NOC * P + item];
            Z[item] = v;
            a.z[((size_t)(n0 + n2) * a.Cout + oc0 + oc) * HoWo + p2] = v;
        }
    }
    // the residual of this thread's outputs is fetched now and lands while the statistics are reduced
    constexpr int RI = 16;                                          // SC_NOC * P / T <= 16 for every launch shape
    float rv[RI];
    if (a.res) {
#pragma unroll
        for (int k = 0; k < RI; ++k) {
            const int item = min(tid + k * T, SC_NOC * P - 1);
            const int oc = item / P, qq = item - oc * P, n2 = qq / HoWo, p2 = qq - n2 * HoWo;
            rv[k] = a.res[((size_t)(n0 + n2) * a.Cout + oc0 + oc) * HoWo + p2];
        }
    }
    __syncthreads();
    for (int oc = wave; oc < SC_NOC; oc += NW) {
        float s = 0.f, ss = 0.f;
        for (int qq = lane; qq < P; qq += 64) s += Z[oc * P + qq];
        s = wave_sum(s);
        const float m = s * (1.f / (float)P);           // second pass: squares about the mean (centered_to_raw, medt_common.h)
        for (int qq = lane; qq < P; qq += 64) {
            const float dv = Z[oc * P + qq] - m;
            ss = fmaf(dv, dv, ss);
        }
        ss = wave_sum(ss);
        if (lane == 0) {
            sums[oc * 2] = s;
            sums[oc * 2 + 1] = ss;
        }
    }
    __syncthreads();
    if (tid < SC_NOC) {                               // the double-precision finalisation of the 16 channels side by side
        const int ch = oc0 + tid;
        double s, ss;
        centered_to_raw(sums[tid * 2], sums[tid * 2 + 1], sums[tid * 2] * (1.f / (float)P), (double)P, s, ss);
        conv_small_scale_shift_p(s, ss, (double)P, prm + tid * 4, a.eps, a.training, sc[tid], sh[tid]);
        if (a.training) {                             // [groups][Cout][2] doubles, like every forward BatchNorm partial
            double* dp = reinterpret_cast<double*>(a.partials) + ((size_t)grp * a.Cout + ch) * 2;
            dp[0] = s;
            dp[1] = ss;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RI; ++k) {
        const int item = tid + k * T;
        if (item < SC_NOC * P) {
            const int oc = item / P, qq = item - oc * P, n2 = qq / HoWo, p2 = qq - n2 * HoWo;
            const size_t idx = ((size_t)(n0 + n2) * a.Cout + oc0 + oc) * HoWo + p2;
            float v = fmaf(Z[item], sc[oc], sh[oc]);
            if (a.res) v += rv[k];
            if (a.relu) v = fmaxf(v, 0.f);
            a.y[idx] = v;
        }
    }
}

static size_t conv_small_lds(int P, int Cin, int noc) {
    const int ks = conv_small_ksplit(P, Cin);
    return ((size_t)noc * P + (size_t)Cin * noc + 8 * noc + (ks > 1 ? (size_t)ks * noc * P : 0)) * sizeof(float);
}

// 16, or 0 (does not fit).  The 8-channel tile (MEDT_SMALL_NOC8=1) would admit the 1024-position groups of layer1_p, but
// measured there the fused forward ties with conv + finalize + apply (14.5 vs 15.6 us) and the one-wave-per-channel
// backward loses (20 vs 15 us), so those blocks stay on the layer-by-layer kernels.
static int conv_small_noc(int P, int Cin) {
    static const int min_noc = 16;
    for (int noc = SC_NOC_MAX; noc >= min_noc; noc >>= 1)
        if (conv_small_lds(P, Cin, noc) <= 64 * 1024) return noc;
    return 0;
}

static bool conv_small_enabled() {
    static const bool on = [] { const char* e = getenv("MEDT_DISABLE_SMALL"); return !(e && e[0] == '1'); }();
    return on;
}

bool conv_small_ok(const medt_conv_desc& d) {
    if (!conv_small_enabled() || !d.has_bn || d.K != 1 || d.pad != 0 || d.has_bias) return false;
    if (d.stride < 1 || d.H % d.stride || d.W % d.stride) return false;
    if ((d.Cin & 15) || (d.Cout % SC_NOC_MAX)) return false;
    const int P = (d.N / d.bn_groups) * (d.H / d.stride) * (d.W / d.stride);
    if (!(P <= 1024 && conv_small_noc(P, d.Cin) != 0)) return false;
#ifndef MEDT_AB_SMALL_ANYGRID           // (A/B build: without this rule)
    // (round 6: the kernels are one workgroup per (BatchNorm group, 16 output channels) -- built for the 16 patch groups of MedT's local
    //  branch.  With ONE group (layer4 of the unets: 4 x 4 maps, 256 -> 256 channels) that is 16 workgroups walking the whole contraction,
    //  23 - 28 us against ~12 for convolution + bn_fin_apply on the whole chip.  Only the one-group case: the patch groups of the local branch
    //  keep these kernels at every batch size)
    if (d.bn_groups == 1 && d.Cout / conv_small_noc(P, d.Cin) < 32) return false;
#endif
    return true;
}

int conv_small_fwd(const medt_conv_desc& d, const float* x, const float* w, const medt_bn_ptrs& bn, const float* res,
                   float* z, float* y, float* partials, hipStream_t s) {
    if (abl_skip("conv_small_fwd")) return MEDT_OK;
    SmallConvArgs a;
    a.x = x; a.w = w; a.res = d.has_res ? res : nullptr; a.bn = bn; a.z = z; a.y = y; a.partials = partials;
    a.N = d.N; a.Cin = d.Cin; a.H = d.H; a.W = d.W; a.Cout = d.Cout; a.stride = d.stride; a.npg = d.N / d.bn_groups;
    a.relu = d.relu; a.training = d.training ? 1 : 0; a.eps = d.eps;
    const int P = a.npg * (d.H / d.stride) * (d.W / d.stride);
    const int threads = conv_small_threads(P);
    const int noc = conv_small_noc(P, d.Cin);
    const dim3 grid(d.bn_groups, d.Cout / noc);
    const size_t lds = conv_small_lds(P, d.Cin, noc);
#define MEDT_SMALL_CONV(TT)                                                                              \
    do {                                                                                                 \
        if (noc == 16) hipLaunchKernelGGL((conv1x1_bn_small_fwd_kernel<TT, 16>), grid, dim3(TT), lds, s, a); \
        else hipLaunchKernelGGL((conv1x1_bn_small_fwd_kernel<TT, 8>), grid, dim3(TT), lds, s, a);         \
    } while (0)
    if (threads == 256) MEDT_SMALL_CONV(256);
    else if (threads == 512) MEDT_SMALL_CONV(512);
    else MEDT_SMALL_CONV(1024);
#undef MEDT_SMALL_CONV
    return launch_status("conv1x1_bn_small_fwd");
}

// --------------------------------------------------------------------------- //
// backward of BatchNorm (+ ReLU mask) for the same blocks: one wave per (group, channel)
// --------------------------------------------------------------------------- //
struct SmallBnBwdArgs {
    const float *dy, *y, *z, *weight;
    BnStats st;
    float *g, *dz, *partials;           // g: masked incoming gradient (= d(res)); partials [groups][C][2]
    int C, HW, npg, relu, training;
};

__global__ __launch_bounds__(MEDT_THREADS) void bn_act_bwd_small_kernel(SmallBnBwdArgs a) {
    const int grp = blockIdx.x, c = blockIdx.y * MEDT_WAVES + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= a.C) return;
    const int HW = a.HW, P = a.npg * HW, gc = grp * a.C + c;
    const float mean = a.st.mean[gc], rstd = a.st.rstd[gc];
    float s1 = 0.f, s2 = 0.f;
    for (int q = lane; q < P; q += 64) {
        const int ni = q / HW, p = q - ni * HW;
        const size_t idx = ((size_t)(grp * a.npg + ni) * a.C + c) * HW + p;
        float d = a.dy[idx];
        if (a.relu && !(a.y[idx] > 0.f)) d = 0.f;
        if (a.g) a.g[idx] = d;
        s1 += d;
        s2 = fmaf(d, (a.z[idx] - mean) * rstd, s2);
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
        a.partials[(size_t)gc * 2] = s1;
        a.partials[(size_t)gc * 2 + 1] = s2;
    }
    // same arithmetic as bn_bwd_coef (pointwise.hip), dscale = 1
    const double A = (double)a.weight[c] * (double)rstd;
    float c0 = (float)A, c1 = 0.f, c2 = 0.f;
    if (a.training) {
        const double m1 = (double)s1 / (double)P, m2 = (double)s2 / (double)P;
        c1 = (float)(-A * (double)rstd * m2);
        c2 = (float)(A * ((double)rstd * (double)mean * m2 - m1));
    }
    for (int q = lane; q < P; q += 64) {
        const int ni = q / HW, p = q - ni * HW;
        const size_t idx = ((size_t)(grp * a.npg + ni) * a.C + c) * HW + p;
        float d = a.dy[idx];
        if (a.relu && !(a.y[idx] > 0.f)) d = 0.f;
        a.dz[idx] = fmaf(c0, d, fmaf(c1, a.z[idx], c2));
    }
}

// The same for any convolution whose BatchNorm population fits one workgroup's registers: one WORKGROUP per (group,
// channel), 16 values per thread read once (float4), the two sums reduced over the block, dz written from registers.
// Replaces bn_act_bwd_stats -> bn_bwd_finalize -> bn_bwd_apply (three dependent launches of ~5 us each, whatever the
// tensor size) on the layer chain; the parameter gradients are produced off the chain by the recorded finalisation.
// (NK float4s per thread: 4 -> populations up to 4 * 4 * TT; round 6: NK = 8 for the 32768-value populations of gatedaxialunet bs 8's and
//  MedT-256 bs 2's first layers, which took the three-launch path -- 7 x (8.4 + 4.6 + 5.9 us + two boundaries) on the layer chain)
template <int TT, int NK = 4>
__global__ __launch_bounds__(TT) void bn_act_bwd_chan_kernel(SmallBnBwdArgs a) {
    MEDT_STATIC_SHARED float red[2 * (TT / 64)];
    const int grp = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
    const int HW = a.HW, P = a.npg * HW, gc = grp * a.C + c;
    const float mean = a.st.mean[gc], rstd = a.st.rstd[gc];
    float4 dv[NK], zv[NK];
    size_t at[NK];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int q = 4 * (tid + k * TT);
        dv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        zv[k] = dv[k];
        at[k] = 0;
        if (q < P) {
            const int ni = q / HW, p = q - ni * HW;                 // HW % 4 == 0: the four values share a plane
            const size_t idx = ((size_t)(grp * a.npg + ni) * a.C + c) * HW + p;
            at[k] = idx;
            float4 d = *reinterpret_cast<const float4*>(a.dy + idx);
            if (a.relu) {
                const float4 yv = *reinterpret_cast<const float4*>(a.y + idx);
                if (!(yv.x > 0.f)) d.x = 0.f;
                if (!(yv.y > 0.f)) d.y = 0.f;
                if (!(yv.z > 0.f)) d.z = 0.f;
                if (!(yv.w > 0.f)) d.w = 0.f;
            }
            if (a.g) *reinterpret_cast<float4*>(a.g + idx) = d;
            const float4 zz = *reinterpret_cast<const float4*>(a.z + idx);
            dv[k] = d;
            zv[k] = zz;
            s1 += (d.x + d.y) + (d.z + d.w);
            s2 = fmaf(d.x, (zz.x - mean) * rstd, s2);
            s2 = fmaf(d.y, (zz.y - mean) * rstd, s2);
            s2 = fmaf(d.z, (zz.z - mean) * rstd, s2);
            s2 = fmaf(d.w, (zz.w - mean) * rstd, s2);
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if ((tid & 63) == 0) { red[(tid >> 6) * 2] = s1; red[(tid >> 6) * 2 + 1] = s2; }
    __syncthreads();
    s1 = 0.f;
    s2 = 0.f;
#pragma unroll
    for (int w = 0; w < TT / 64; ++w) { s1 += red[w * 2]; s2 += red[w * 2 + 1]; }       // fixed order, every thread
    if (tid == 0) {
        a.partials[(size_t)gc * 2] = s1;
        a.partials[(size_t)gc * 2 + 1] = s2;
    }
    // same arithmetic as bn_bwd_coef (pointwise.hip), dscale = 1
    const double A = (double)a.weight[c] * (double)rstd;
    float c0 = (float)A, c1 = 0.f, c2 = 0.f;
    if (a.training) {
        const double m1 = (double)s1 / (double)P, m2 = (double)s2 / (double)P;
        c1 = (float)(-A * (double)rstd * m2);
        c2 = (float)(A * ((double)rstd * (double)mean * m2 - m1));
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int q = 4 * (tid + k * TT);
        if (q < P) {
            float4 o;
            o.x = fmaf(c0, dv[k].x, fmaf(c1, zv[k].x, c2));
            o.y = fmaf(c0, dv[k].y, fmaf(c1, zv[k].y, c2));
            o.z = fmaf(c0, dv[k].z, fmaf(c1, zv[k].z, c2));
            o.w = fmaf(c0, dv[k].w, fmaf(c1, zv[k].w, c2));
            *reinterpret_cast<float4*>(a.dz + at[k]) = o;
        }
    }
}

// one workgroup of 256 (population <= 4096) or 1024 threads (<= 16384; <= 32768 with eight float4s per thread) per (group, channel);
// 0: not applicable
int bn_chan_threads(const medt_conv_desc& d, int HoWo) {
#ifdef MEDT_AB_BN_CHAN_16K              // (A/B build: round 5's limit)
    static const int pmax = 16384;
#else
    static const int pmax = 32768;
#endif
    if (!d.has_bn || (HoWo & 3)) return 0;
    const long P = (long)(d.N / d.bn_groups) * HoWo;
    if (P > pmax) return 0;
    return P <= 4096 ? 256 : 1024;
}

int bn_act_bwd_chan(const medt_conv_desc& d, const float* dy, const float* y, const float* z, BnStats st,
                    const float* weight, float* g, float* dz, float* partials, int HoWo, hipStream_t s) {
    SmallBnBwdArgs a;
    a.dy = dy; a.y = y; a.z = z; a.weight = weight; a.st = st; a.g = g; a.dz = dz; a.partials = partials;
    a.C = d.Cout; a.HW = HoWo; a.npg = d.N / d.bn_groups; a.relu = d.relu; a.training = d.training ? 1 : 0;
    const dim3 grid(d.bn_groups, d.Cout);
    if (bn_chan_threads(d, HoWo) == 256) hipLaunchKernelGGL(bn_act_bwd_chan_kernel<256>, grid, dim3(256), 0, s, a);
    else if ((long)a.npg * HoWo <= 16384) hipLaunchKernelGGL(bn_act_bwd_chan_kernel<1024>, grid, dim3(1024), 0, s, a);
    else hipLaunchKernelGGL((bn_act_bwd_chan_kernel<1024, 8>), grid, dim3(1024), 0, s, a);
    return launch_status("bn_act_bwd_chan");
}

int bn_act_bwd_small(const medt_conv_desc& d, const float* dy, const float* y, const float* z, BnStats st,
                     const float* weight, float* g, float* dz, float* partials, int HoWo, hipStream_t s) {
    SmallBnBwdArgs a;
    a.dy = dy; a.y = y; a.z = z; a.weight = weight; a.st = st; a.g = g; a.dz = dz; a.partials = partials;
    a.C = d.Cout; a.HW = HoWo; a.npg = d.N / d.bn_groups; a.relu = d.relu; a.training = d.training ? 1 : 0;
    hipLaunchKernelGGL(bn_act_bwd_small_kernel, dim3(d.bn_groups, cdiv(d.Cout, MEDT_WAVES)), dim3(MEDT_THREADS), 0, s, a);
    return launch_status("bn_act_bwd_small");
}

// --------------------------------------------------------------------------- //
// BatchNorm (+ ReLU mask) backward AND the 1x1 backward-data behind it in ONE launch (round 4).
// On the local branch's backward chain every conv_down / conv_up block was two dependent launches: bn_act_bwd_small
// (one wave per (group, channel): mask, the two sums, dz) and a 1x1 dgrad (5 + 5..14 us for a few hundred KFLOP).  The
// BatchNorm population of these blocks is one patch group, so a workgroup can redo the whole group's statistics itself:
// workgroup (group, tile of CT input channels) reads dy | y | z of ALL Cout channels of its group once (<= 32 values per
// thread, float4), reduces the two sums per channel over the TPC lanes that share a channel, forms dz = c0*g + c1*z + c2 in
// registers, parks it in LDS as [Cout][P] and contracts it with its weight slice: dx[ci, q] = sum_o w[o, ci] dz[o, q]
// (+ the fan-in deposit dx_add).  The statistics are recomputed Cin / CT (= 8) times per group -- a few hundred KB of L2
// reads -- instead of travelling through a second launch; tile 0 also writes g (= d(res)), dz (the recorded weight
// gradient reads it) and the per-group partial sums (BatchNorm parameter gradients, recorded finalisation).
// Same coefficient arithmetic as bn_act_bwd_small_kernel / bn_bwd_coef.
// --------------------------------------------------------------------------- //
struct BnDgradArgs {
    const float *dy, *y, *z, *gamma, *w, *dx_add;
    BnStats st;
    float *g, *dz, *partials, *dx;
    int Cout, Cin, HW, npg, relu, training, E, CT;      // E values per thread (multiple of 4), CT input channels per workgroup
};

template <int T, int CPT>                               // CPT input channels per thread = CT / (T / P)
__global__ __launch_bounds__(T) void bn_dgrad1x1_small_kernel(BnDgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int E4MAX = T == 1024 ? 4 : 8;
    const int grp = blockIdx.x, ci0 = blockIdx.y * a.CT, tid = threadIdx.x;
    const int Cout = a.Cout, HW = a.HW, P = a.npg * HW, n0 = grp * a.npg, CT = a.CT;
    const int E4 = a.E >> 2, TPC = P / a.E;             // threads per channel (power of two, <= 64)
    float* Dz = smem;                                   // [Cout][P]
    float* Wl = Dz + (size_t)Cout * P;                  // [Cout][CT]
    const int c = tid / TPC, tq = tid - c * TPC;
    const int gc = grp * Cout + c;
    // every global value of the statistics phase in one batch: dy, y, z of this thread's E positions of channel c, the
    // channel's saved statistics, and the weight slice (consumed last)
    float4 dv[E4MAX], zv[E4MAX], yv[E4MAX];
    size_t at[E4MAX];
#pragma unroll
    for (int j = 0; j < E4MAX; ++j) {
        const int q = 4 * (tq + min(j, E4 - 1) * TPC), ni = q / HW, p = q - ni * HW;       // HW % 4 == 0: one plane per float4
        at[j] = ((size_t)(n0 + ni) * Cout + c) * HW + p;
        dv[j] = *reinterpret_cast<const float4*>(a.dy + at[j]);
        zv[j] = *reinterpret_cast<const float4*>(a.z + at[j]);
        if (a.relu) yv[j] = *reinterpret_cast<const float4*>(a.y + at[j]);
    }
    const float mean = a.st.mean[gc], rstd = a.st.rstd[gc], gam = a.gamma[c];
    constexpr int WB = 8;
    const int nW = Cout * CT;
    float wr[WB];
#pragma unroll
    for (int k = 0; k < WB; ++k) {
        const int e = min(tid + k * T, nW - 1), oc = e / CT;
        wr[k] = a.w[(size_t)oc * a.Cin + ci0 + (e - oc * CT)];
    }
    MEDT_SCHED_FENCE();
#pragma unroll
    for (int k = 0; k < WB; ++k) {
        const int e = tid + k * T;
        if (e < nW) Wl[e] = wr[k];
    }
    for (int base = WB * T; base < nW; base += WB * T) {
#pragma unroll
        for (int k = 0; k < WB; ++k) {
            const int e = min(base + tid + k * T, nW - 1), oc = e / CT;
            wr[k] = a.w[(size_t)oc * a.Cin + ci0 + (e - oc * CT)];
        }
        MEDT_SCHED_FENCE();
#pragma unroll
        for (int k = 0; k < WB; ++k) {
            const int e = base + tid + k * T;
            if (e < nW) Wl[e] = wr[k];
        }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < E4MAX; ++j) {
        if (j < E4) {
            float4 d = dv[j];
            if (a.relu) {
                if (!(yv[j].x > 0.f)) d.x = 0.f;
                if (!(yv[j].y > 0.f)) d.y = 0.f;
                if (!(yv[j].z > 0.f)) d.z = 0.f;
                if (!(yv[j].w > 0.f)) d.w = 0.f;
            }
            dv[j] = d;
            s1 += (d.x + d.y) + (d.z + d.w);
            s2 = fmaf(d.x, (zv[j].x - mean) * rstd, s2);
            s2 = fmaf(d.y, (zv[j].y - mean) * rstd, s2);
            s2 = fmaf(d.z, (zv[j].z - mean) * rstd, s2);
            s2 = fmaf(d.w, (zv[j].w - mean) * rstd, s2);
        }
    }
    for (int o = TPC >> 1; o > 0; o >>= 1) {            // the TPC lanes of a channel are an aligned run inside one wave
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    // same arithmetic as bn_bwd_coef (medt_common.h), dscale = 1
    const double A = (double)gam * (double)rstd;
    float c0 = (float)A, c1 = 0.f, c2 = 0.f;
    if (a.training) {
        const double m1 = (double)s1 / (double)P, m2 = (double)s2 / (double)P;
        c1 = (float)(-A * (double)rstd * m2);
        c2 = (float)(A * ((double)rstd * (double)mean * m2 - m1));
    }
    const bool writer = blockIdx.y == 0;
    if (writer && tq == 0) {
        a.partials[(size_t)gc * 2] = s1;
        a.partials[(size_t)gc * 2 + 1] = s2;
    }
#pragma unroll
    for (int j = 0; j < E4MAX; ++j) {
        if (j < E4) {
            float4 o;
            o.x = fmaf(c0, dv[j].x, fmaf(c1, zv[j].x, c2));
            o.y = fmaf(c0, dv[j].y, fmaf(c1, zv[j].y, c2));
            o.z = fmaf(c0, dv[j].z, fmaf(c1, zv[j].z, c2));
            o.w = fmaf(c0, dv[j].w, fmaf(c1, zv[j].w, c2));
            *reinterpret_cast<float4*>(Dz + (size_t)c * P + 4 * (tq + j * TPC)) = o;
            if (writer) {
                *reinterpret_cast<float4*>(a.dz + at[j]) = o;
                if (a.g) *reinterpret_cast<float4*>(a.g + at[j]) = dv[j];
            }
        }
    }
    // the contraction: thread (q, r) owns CPT input channels of position q
    const int q = tid % P, r = tid / P;                 // P <= T, T / P groups of CPT channels each (= CT)
    const int ni = q / HW, p = q - ni * HW;
    const int cl = r * CPT;                             // first local input channel of this thread
    const bool act = cl < CT;
    float addv[CPT];
    if (a.dx_add && act) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) addv[i] = a.dx_add[((size_t)(n0 + ni) * a.Cin + ci0 + cl + i) * HW + p];
    }
    __syncthreads();
    float acc[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) acc[i] = 0.f;
    if (act) {
        const float* dzq = Dz + q;
        const float* wq = Wl + cl;
#pragma unroll 8
        for (int o = 0; o < Cout; ++o) {
            const float v = dzq[(size_t)o * P];
            if constexpr (CPT == 4) {
                const float4 w4 = *reinterpret_cast<const float4*>(wq + o * CT);
                acc[0] = fmaf(w4.x, v, acc[0]);
                acc[1] = fmaf(w4.y, v, acc[1]);
                acc[2] = fmaf(w4.z, v, acc[2]);
                acc[3] = fmaf(w4.w, v, acc[3]);
            } else if constexpr (CPT == 2) {
                const float2 w2 = *reinterpret_cast<const float2*>(wq + o * CT);
                acc[0] = fmaf(w2.x, v, acc[0]);
                acc[1] = fmaf(w2.y, v, acc[1]);
            } else {
                acc[0] = fmaf(wq[o * CT], v, acc[0]);
            }
        }
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            float v = acc[i];
            if (a.dx_add) v += addv[i];
            a.dx[((size_t)(n0 + ni) * a.Cin + ci0 + cl + i) * HW + p] = v;
        }
    }
}

struct BnDgradPlan { int T, E, CT, CPT; size_t lds; };

static bool bn_dgrad_fused_enabled() {
    static const bool on = true;
    return on;
}

// Applies to the conv_small blocks (1x1, no bias, BatchNorm group <= 1024 positions) with stride 1 whose group tile
// Cout x P fits 32 values per thread and the workgroup's LDS.
static bool bn_dgrad1x1_plan(const medt_conv_desc& d, BnDgradPlan* pl) {
    if (!bn_dgrad_fused_enabled() || !conv_small_ok(d) || d.stride != 1 || d.K != 1) return false;
    const int HW = d.H * d.W, P = (d.N / d.bn_groups) * HW;
    if ((HW & 3) || (P & (P - 1))) return false;
    const long tile = (long)d.Cout * P;
    int T = 256;
    while (T < 1024 && tile / T > 32) T <<= 1;
    const long E = tile / T;
    if (tile % T || E > (T == 1024 ? 16 : 32) || E < 4 || (E & (E - 1)) || P % E || P / E > 64 || P > T) return false;
    const int R = T / P;                                // thread groups over the positions
    int CT = d.Cin / 8 > R ? d.Cin / 8 : R;             // eight tiles per group when the contraction allows it
    if (CT > 4 * R) CT = 4 * R;
    if (d.Cin % CT || CT % R) return false;
    const int CPT = CT / R;
    if (CPT != 1 && CPT != 2 && CPT != 4) return false;
    pl->T = T; pl->E = (int)E; pl->CT = CT; pl->CPT = CPT;
    pl->lds = ((size_t)tile + (size_t)d.Cout * CT) * sizeof(float);
    static const size_t lds_cap = (size_t)150 * 1024;
    return pl->lds <= lds_cap;
}

bool bn_dgrad1x1_small_ok(const medt_conv_desc& d) {
    BnDgradPlan pl;
    return bn_dgrad1x1_plan(d, &pl);
}

int bn_dgrad1x1_small(const medt_conv_desc& d, const float* dy, const float* y, const float* z, BnStats st,
                      const float* gamma, const float* w, const float* dx_add, float* g, float* dz, float* partials,
                      float* dx, hipStream_t s) {
    BnDgradPlan pl;
    if (!bn_dgrad1x1_plan(d, &pl)) { set_error("bn_dgrad1x1_small: shape not supported"); return MEDT_EUNSUPPORTED; }
    if (abl_skip("bn_dgrad")) return MEDT_OK;
    BnDgradArgs a;
    a.dy = dy; a.y = y; a.z = z; a.gamma = gamma; a.w = w; a.dx_add = dx_add; a.st = st;
    a.g = g; a.dz = dz; a.partials = partials; a.dx = dx;
    a.Cout = d.Cout; a.Cin = d.Cin; a.HW = d.H * d.W; a.npg = d.N / d.bn_groups; a.relu = d.relu;
    a.training = d.training ? 1 : 0; a.E = pl.E; a.CT = pl.CT;
    const dim3 grid(d.bn_groups, d.Cin / pl.CT);
#define MEDT_BND(TT, CP)                                                                                              \
    do {                                                                                                              \
        static unsigned char attr[64];                                                                                \
        if (pl.lds > 64 * 1024)              /* more than 64 KB of dynamic LDS needs the opt-in (gfx950: 160 KB per CU) */ \
            if (int rca = lds_opt_in((const void*)bn_dgrad1x1_small_kernel<TT, CP>, attr, "bn_dgrad1x1_small")) return rca; \
        hipLaunchKernelGGL((bn_dgrad1x1_small_kernel<TT, CP>), grid, dim3(TT), pl.lds, s, a);                         \
    } while (0)
#define MEDT_BND_T(TT)                                                                                                \
    switch (pl.CPT) {                                                                                                 \
        case 1: MEDT_BND(TT, 1); break;                                                                               \
        case 2: MEDT_BND(TT, 2); break;                                                                               \
        default: MEDT_BND(TT, 4); break;                                                                              \
    }
    if (pl.T == 256) { MEDT_BND_T(256) } else if (pl.T == 512) { MEDT_BND_T(512) } else { MEDT_BND_T(1024) }
#undef MEDT_BND_T
#undef MEDT_BND
    return launch_status("bn_dgrad1x1_small");
}


}  // namespace medt
