// conv.hip -- direct NCHW fp32 convolutions for the encoder/decoder around the attention layers
// (reference lib/models/axialnet.py:416-419, 434-439, 557-562, 581-588: 7x7 s2, 3x3 s1/s2, 1x1 s1/s2).
//
// The reference dispatches these to cuDNN/MIOpen; at this model's sizes (8..256 channels on 2x2..128x128
// maps) MIOpen falls back to naive / badly-shaped solvers (rocprof: 0.5-1.1 ms per call), so they are
// written directly: lanes run along output pixels (coalesced NCHW rows), every lane keeps OT output
// channels in registers, the K*K taps of one input channel are loaded once per lane and the weights
// arrive through the scalar path (wave-uniform addresses, K*K contiguous floats per (o,c) pair).
#include "defer.h"
#include <type_traits>

namespace medt {

// --------------------------------------------------------------------------- //
// forward:  y[n,o,ho,wo] = bias[o] + sum_{c,kh,kw} w[o,c,kh,kw] * x[n,c,ho*s-p+kh,wo*s-p+kw]
// Lanes run over the flattened (image, output pixel) index *within one BatchNorm group*, so 2x2 / 4x4 maps
// of the deep LoGo layers still fill the wave.  Optional per-channel [sum, sum^2] partials are laid out
// [group][part][Cout][2] with part = 256-position chunk inside the group (BatchNorm statistics).
// --------------------------------------------------------------------------- //
template <int K, int OT>
__global__ __launch_bounds__(MEDT_THREADS) void conv2d_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
    float* __restrict__ partials, int Cin, int H, int W, int Cout, int Ho, int Wo, int stride, int pad, int relu,
    int npg, int y_bf16) {
    constexpr int KK = K * K;
    MEDT_STATIC_SHARED float red[MEDT_WAVES * OT * 2 * 2];
    const int HoWo = Ho * Wo, per_group = npg * HoWo, ppg = (per_group + MEDT_THREADS - 1) / MEDT_THREADS;
    const int grp = blockIdx.x / ppg, part = blockIdx.x - grp * ppg, o0 = blockIdx.y * OT;
    const int q = part * MEDT_THREADS + threadIdx.x;
    const bool ok = q < per_group;
    const int ni = ok ? q / HoWo : 0, p = ok ? q - ni * HoWo : 0;
    const int n = grp * npg + ni;
    const int ho = p / Wo, wo = p - ho * Wo;
    const int h0 = ho * stride - pad, w0 = wo * stride - pad;
    float acc[OT];
#pragma unroll
    for (int o = 0; o < OT; ++o) acc[o] = bias ? bias[o0 + o] : 0.f;
    const float* xn = x + (size_t)n * Cin * H * W;
    // tap offsets inside one channel plane, decoded once (-1: padding or a lane past the end).  The loads in the channel
    // loop are unconditional (clamped offset, then a select): a branch per tap would keep one load in flight at a time.
    int off[KK];
#pragma unroll
    for (int kh = 0; kh < K; ++kh)
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
            const int h = h0 + kh, ww = w0 + kw;
            off[kh * K + kw] = (ok && (unsigned)h < (unsigned)H && (unsigned)ww < (unsigned)W) ? h * W + ww : -1;
        }
    const int HWi = H * W;
    // U channels' taps are loaded as one batch (U*K*K loads in flight), then consumed: MEDT_SCHED_FENCE keeps the
    // scheduler from interleaving load / use pairs, which would make the loop one global round trip per tap.
    constexpr int U = K == 1 ? 8 : (K == 3 ? 4 : 1);
    auto taps = [&](auto u_tag, int c) {
        constexpr int UU = decltype(u_tag)::value;
        float xr[UU][KK];
#pragma unroll
        for (int u = 0; u < UU; ++u)
#pragma unroll
            for (int t = 0; t < KK; ++t) xr[u][t] = xn[(size_t)(c + u) * HWi + max(off[t], 0)];
        MEDT_SCHED_FENCE();
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            float xv[KK];
#pragma unroll
            for (int t = 0; t < KK; ++t) xv[t] = off[t] >= 0 ? xr[u][t] : 0.f;
#pragma unroll
            for (int o = 0; o < OT; ++o) {
                const float* wp = w + ((size_t)(o0 + o) * Cin + c + u) * KK;
#pragma unroll
                for (int t = 0; t < KK; ++t) acc[o] = fmaf(wp[t], xv[t], acc[o]);
            }
        }
        MEDT_SCHED_FENCE();
    };
    int c = 0;
    for (; c + U <= Cin; c += U) taps(std::integral_constant<int, U>{}, c);
    for (; c < Cin; ++c) taps(std::integral_constant<int, 1>{}, c);
    if (ok) {
        const size_t yo = ((size_t)n * Cout + o0) * HoWo + p;
#pragma unroll
        for (int o = 0; o < OT; ++o) st_act(y, yo + (size_t)o * HoWo, relu ? fmaxf(acc[o], 0.f) : acc[o], y_bf16);
    }
    if (partials) {
        float v[2 * OT];
#pragma unroll
        for (int o = 0; o < OT; ++o) {
            const float a = ok ? acc[o] : 0.f;
            v[2 * o] = a;
            v[2 * o + 1] = a * a;
        }
        block_sum_d<2 * OT>(v, red, reinterpret_cast<double*>(partials) + ((size_t)blockIdx.x * Cout + o0) * 2);
    }
}

int conv2d_parts_per_group(int N, int groups, int HoWo) { return cdiv((N / groups) * HoWo, MEDT_THREADS); }

// Output-channel tile per lane: 16 when there are enough workgroups to fill the chip, else smaller tiles
// (more workgroups, less register reuse) -- the deep LoGo layers have as few as 64 output positions.
static int pick_tile(int C, int max_tile, long position_blocks) {
    // (>= 512 workgroups; measured round 4 on the MedT step: 1024 -> 2.32 ms, 256 -> 2.18 ms, 512 -> 2.19 ms)
    int t = max_tile;
    while (t > 1 && (C % t != 0 || position_blocks * (C / t) < 512)) t >>= 1;
    while (C % t != 0) t >>= 1;
    return t;
}

template <int K>
static int conv2d_fwd_k(const float* x, const float* w, const float* bias, float* y, float* partials, int N, int Cin,
                        int H, int W, int Cout, int Ho, int Wo, int stride, int pad, int relu, int groups,
                        hipStream_t s, int y_bf16) {
    const int npg = N / groups;
    const unsigned gx = (unsigned)(groups * conv2d_parts_per_group(N, groups, Ho * Wo));
#define MEDT_LAUNCH_FWD(OT)                                                                                        \
    hipLaunchKernelGGL((conv2d_fwd_kernel<K, OT>), dim3(gx, Cout / OT), dim3(MEDT_THREADS), 0, s, x, w, bias, y, partials, \
                       Cin, H, W, Cout, Ho, Wo, stride, pad, relu, npg, y_bf16)
    switch (pick_tile(Cout, K == 7 ? 8 : 16, gx)) {
        case 16: if constexpr (K != 7) { MEDT_LAUNCH_FWD(16); } break;
        case 8: MEDT_LAUNCH_FWD(8); break;
        case 4: MEDT_LAUNCH_FWD(4); break;
        case 2: MEDT_LAUNCH_FWD(2); break;
        default: MEDT_LAUNCH_FWD(1); break;
    }
#undef MEDT_LAUNCH_FWD
    return launch_status("conv2d_fwd");
}

static int conv_thin_mode() {          // MEDT_CONV_THIN: 0 off, 1 forward only, 2 backward-data only, default both
    static const int m = [] { const char* e = getenv("MEDT_CONV_THIN"); return e ? atoi(e) : 3; }();
    return m;
}
int conv_parts_per_group(int N, int groups, int HoWo, int Cin, int Cout, int K, int stride, int H, int W, int pad) {
    // (the same order as conv2d_fwd's dispatch; H = 0: a caller that only runs 1x1 layers)
    if (H > 0 && (conv_thin_mode() & 1) && !conv_use_mfma(Cin, Cout, K, stride, (long)N * HoWo) && conv_thin_ok(N, groups, Cin, H, W, Cout, K, stride, pad))
        return conv_thin_parts_per_group(N, groups, Cin, H, W, Cout, K, stride, pad);
    if (conv_fwd_ws_ok(Cin, Cout, K, stride, (long)N * HoWo)) return cdiv((N / groups) * HoWo, 64);
    if (!conv_use_mfma(Cin, Cout, K, stride, (long)N * HoWo)) return conv2d_parts_per_group(N, groups, HoWo);
    if (conv_mfma_scratch_floats(N, groups, HoWo, Cin, Cout, K, H > 0 && conv_rows16_ok(Cin, H, W, K, stride, pad)) > 0) return conv2d_parts_per_group(N, groups, HoWo);
    return conv_mfma_parts_per_group(N, groups, HoWo);
}

size_t conv2d_fwd_scratch_floats(int N, int groups, int Cin, int H, int W, int Cout, int K, int stride, int pad) {
    const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
    if (!conv_use_mfma(Cin, Cout, K, stride, (long)N * Ho * Wo)) return 0;
    return conv_mfma_scratch_floats(N, groups, Ho * Wo, Cin, Cout, K, conv_rows16_ok(Cin, H, W, K, stride, pad));
}

size_t conv2d_bwd_data_scratch_floats(int N, int Cin, int H, int W, int Cout, int K, int stride, int pad) {
    if (stride != 1 || K - 1 - pad < 0 || !conv_use_mfma(Cout, Cin, K, 1, (long)N * H * W)) return 0;
    return conv_mfma_scratch_floats(N, 1, H * W, Cout, Cin, K, conv_rows16_ok(Cout, H, W, K, 1, K - 1 - pad));
}

// Deep contraction, narrow output (conv3 128 -> 8 at 64 x 64, the decoders' 64 -> 32 and 128 -> 64 on small maps): in
// the kernel above every lane walks all Cin * K * K taps serially -- Cin / 4 dependent global round trips (43 us for
// conv3's 151 MFLOP).  Here a workgroup owns 64 output positions, its four waves split the channel contraction and
// combine through LDS in fixed order, and the weight slice of the workgroup's OT output channels sits in LDS.
// BatchNorm partials: one [sum, sum^2] pair of doubles per (64-position part, channel).
template <int K, int OT>
__global__ __launch_bounds__(MEDT_THREADS) void conv2d_fwd_ws_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
    float* __restrict__ partials, int Cin, int H, int W, int Cout, int Ho, int Wo, int stride, int pad, int relu,
    int npg, int y_bf16) {
    constexpr int KK = K * K;
    MEDT_STATIC_SHARED float red[3][OT][64];
    extern __shared__ __attribute__((aligned(16))) float wl[];       // [OT][Cin][KK]: rows o0 .. o0+OT-1 of w, as stored
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int HoWo = Ho * Wo, per_group = npg * HoWo, ppg = (per_group + 63) / 64;
    const int grp = blockIdx.x / ppg, part = blockIdx.x - grp * ppg, o0 = blockIdx.y * OT;
    {
        const int nw = OT * Cin * KK;
        const float* src = w + (size_t)o0 * Cin * KK;
        for (int e0 = threadIdx.x; e0 < nw; e0 += 8 * MEDT_THREADS) {            // 8 loads in flight per lane
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[min(e0 + u * MEDT_THREADS, nw - 1)];
            MEDT_SCHED_FENCE();
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + u * MEDT_THREADS < nw) wl[e0 + u * MEDT_THREADS] = v[u];
        }
    }
    const int q = part * 64 + lane;
    const bool ok = q < per_group;
    const int ni = ok ? q / HoWo : 0, p = ok ? q - ni * HoWo : 0;
    const int n = grp * npg + ni;
    const int ho = p / Wo, wo = p - ho * Wo;
    const int h0 = ho * stride - pad, w0 = wo * stride - pad;
    int off[KK];
#pragma unroll
    for (int kh = 0; kh < K; ++kh)
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
            const int h = h0 + kh, ww = w0 + kw;
            off[kh * K + kw] = (ok && (unsigned)h < (unsigned)H && (unsigned)ww < (unsigned)W) ? h * W + ww : -1;
        }
    float acc[OT];
#pragma unroll
    for (int o = 0; o < OT; ++o) acc[o] = 0.f;
    const float* xn = x + (size_t)n * Cin * H * W;
    const int HWi = H * W;
    const int cb = (Cin * wv) / 4, ce = (Cin * (wv + 1)) / 4;
    __syncthreads();
    constexpr int U = K == 1 ? 16 : 4;                             // input channels per batch of loads
    auto taps = [&](auto u_tag, int c) {
        constexpr int UU = decltype(u_tag)::value;
        float xr[UU][KK];
#pragma unroll
        for (int u = 0; u < UU; ++u)
#pragma unroll
            for (int t = 0; t < KK; ++t) xr[u][t] = xn[(size_t)(c + u) * HWi + max(off[t], 0)];
        MEDT_SCHED_FENCE();
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            float xv[KK];
#pragma unroll
            for (int t = 0; t < KK; ++t) xv[t] = off[t] >= 0 ? xr[u][t] : 0.f;
#pragma unroll
            for (int o = 0; o < OT; ++o) {
                const float* wp = wl + (o * Cin + c + u) * KK;
#pragma unroll
                for (int t = 0; t < KK; ++t) acc[o] = fmaf(wp[t], xv[t], acc[o]);
            }
        }
        MEDT_SCHED_FENCE();
    };
    int c = cb;
    for (; c + U <= ce; c += U) taps(std::integral_constant<int, U>{}, c);
    for (; c < ce; ++c) taps(std::integral_constant<int, 1>{}, c);
    if (wv > 0) {
#pragma unroll
        for (int o = 0; o < OT; ++o) red[wv - 1][o][lane] = acc[o];
    }
    __syncthreads();
    if (wv == 0) {
        const size_t yo = ((size_t)n * Cout + o0) * HoWo + p;
        double* pd = partials ? reinterpret_cast<double*>(partials) + ((size_t)blockIdx.x * Cout + o0) * 2 : nullptr;
#pragma unroll
        for (int o = 0; o < OT; ++o) {
            float v = acc[o] + (red[0][o][lane] + red[1][o][lane]) + red[2][o][lane];
            if (bias) v += bias[o0 + o];
            if (ok) st_act(y, yo + (size_t)o * HoWo, relu ? fmaxf(v, 0.f) : v, y_bf16);
            if (pd) {
                const float a = ok ? v : 0.f;
                const double s1 = wave_sum_d((double)a), s2 = wave_sum_d((double)(a * a));
                if (lane == 0) { pd[2 * o] = s1; pd[2 * o + 1] = s2; }
            }
        }
    }
}

static int conv_fwd_ws_tile(int Cin, int Cout, int K, long positions) {       // 0: not this kernel
    int ot = pick_tile(Cout, 8, (positions + 63) / 64);
    while (ot > 1 && (size_t)ot * Cin * K * K * sizeof(float) > 48 * 1024) ot >>= 1;
    return (size_t)ot * Cin * K * K * sizeof(float) <= 48 * 1024 ? ot : 0;
}

bool conv_fwd_ws_ok(int Cin, int Cout, int K, int stride, long positions) {
    static const bool on = true;
    return on && K == 3 && Cin * K * K >= 512 && positions <= 16384 && !conv_use_mfma(Cin, Cout, K, stride, positions) &&
           conv_fwd_ws_tile(Cin, Cout, K, positions) != 0;
}

static int conv2d_fwd_ws(const float* x, const float* w, const float* bias, float* y, float* partials, int N, int Cin, int H,
                         int W, int Cout, int Ho, int Wo, int stride, int pad, int relu, int groups, hipStream_t s,
                         int y_bf16) {
    const int npg = N / groups, ot = conv_fwd_ws_tile(Cin, Cout, 3, (long)N * Ho * Wo);
    const unsigned gx = (unsigned)(groups * cdiv(npg * Ho * Wo, 64));
    const size_t lds = (size_t)ot * Cin * 9 * sizeof(float);
#define MEDT_LAUNCH_FWS(OT)                                                                                            \
    hipLaunchKernelGGL((conv2d_fwd_ws_kernel<3, OT>), dim3(gx, Cout / OT), dim3(MEDT_THREADS), lds, s, x, w, bias, y, partials, \
                       Cin, H, W, Cout, Ho, Wo, stride, pad, relu, npg, y_bf16)
    switch (ot) {
        case 8: MEDT_LAUNCH_FWS(8); break;
        case 4: MEDT_LAUNCH_FWS(4); break;
        case 2: MEDT_LAUNCH_FWS(2); break;
        default: MEDT_LAUNCH_FWS(1); break;
    }
#undef MEDT_LAUNCH_FWS
    return launch_status("conv2d_fwd_ws");
}

int conv2d_fwd(const float* x, const float* w, const float* bias, float* y, float* partials, float* scratch, int N,
               int Cin, int H, int W, int Cout, int K, int stride, int pad, int relu, int groups, hipStream_t s,
               int y_bf16) {
    const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
    if (abl_skip(N >= 16 ? (K == 3 ? "conv3_fwd_l" : (K == 1 ? "conv1_fwd_l" : "conv7_fwd_l")) : (K == 3 ? "conv3_fwd_g" : (K == 1 ? "conv1_fwd_g" : "conv7_fwd_g")))) return MEDT_OK;
    if (!y_bf16 && conv_stem7_ok(Cin, H, W, Cout, K, stride, pad))        // (medt_api.hip sizes the partial sums for it: conv_geom)
        return conv_stem7_fwd(x, w, bias, y, partials, N, H, W, Cout, relu, s);
    if (!y_bf16 && conv_use_mfma(Cin, Cout, K, stride, (long)N * Ho * Wo) &&
        (scratch || conv_mfma_scratch_floats(N, groups, Ho * Wo, Cin, Cout, K, conv_rows16_ok(Cin, H, W, K, stride, pad)) == 0))
        return conv_mfma_fwd(x, w, bias, y, partials, scratch, N, Cin, H, W, Cout, K, stride, pad, relu, groups, s);
    // thin-channel 3x3 layers (Cout < 32 or Cin * 9 < 256: refused above) on the LDS-patch MFMA kernel of round 6
    if (!y_bf16 && (conv_thin_mode() & 1) && !conv_use_mfma(Cin, Cout, K, stride, (long)N * Ho * Wo) && conv_thin_ok(N, groups, Cin, H, W, Cout, K, stride, pad))
        return conv_thin_fwd(x, w, bias, nullptr, y, partials, N, groups, Cin, H, W, Cout, relu, s);
    if (conv_fwd_ws_ok(Cin, Cout, K, stride, (long)N * Ho * Wo))
        return conv2d_fwd_ws(x, w, bias, y, partials, N, Cin, H, W, Cout, Ho, Wo, stride, pad, relu, groups, s, y_bf16);
    switch (K) {
        case 1: return conv2d_fwd_k<1>(x, w, bias, y, partials, N, Cin, H, W, Cout, Ho, Wo, stride, pad, relu, groups, s, y_bf16);
        case 3: return conv2d_fwd_k<3>(x, w, bias, y, partials, N, Cin, H, W, Cout, Ho, Wo, stride, pad, relu, groups, s, y_bf16);
        case 7: return conv2d_fwd_k<7>(x, w, bias, y, partials, N, Cin, H, W, Cout, Ho, Wo, stride, pad, relu, groups, s, y_bf16);
    }
    set_error("conv2d: kernel size %d unsupported (1, 3, 7)", K);
    return MEDT_EUNSUPPORTED;
}

// --------------------------------------------------------------------------- //
// backward-data:  dx[n,c,h,w] = sum_{o,kh,kw} w[o,c,kh,kw] * dy[n,o,(h+p-kh)/s,(w+p-kw)/s]
// lanes over the flattened (image, input pixel) index
// --------------------------------------------------------------------------- //
template <int K, int CT>
__global__ __launch_bounds__(MEDT_THREADS) void conv2d_bwd_data_kernel(
    const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int N, int Cin, int H, int W,
    int Cout, int Ho, int Wo, int stride, int pad, const float* __restrict__ add) {
    constexpr int KK = K * K;
    const int HW = H * W;
    const long q = (long)blockIdx.x * MEDT_THREADS + threadIdx.x;
    const int c0 = blockIdx.y * CT;
    const bool ok = q < (long)N * HW;
    const int n = ok ? (int)(q / HW) : 0, p = ok ? (int)(q - (long)n * HW) : 0;
    const int h = p / W, ww = p - h * W;
    int off[KK];
#pragma unroll
    for (int kh = 0; kh < K; ++kh)
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
            const int hh = h + pad - kh, wq = ww + pad - kw;
            int o = -1;
            if (ok && hh >= 0 && wq >= 0 && hh % stride == 0 && wq % stride == 0) {
                const int ho = hh / stride, wo = wq / stride;
                if (ho < Ho && wo < Wo) o = ho * Wo + wo;
            }
            off[kh * K + kw] = o;
        }
    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = 0.f;
    const float* dyn = dy + (size_t)n * Cout * Ho * Wo;
    constexpr int U = K == 1 ? 8 : (K == 3 ? 4 : 1);               // output channels per batch of loads (see conv2d_fwd_kernel)
    const int HoWo = Ho * Wo;
    auto taps = [&](auto u_tag, int o) {
        constexpr int UU = decltype(u_tag)::value;
        float dr[UU][KK];
#pragma unroll
        for (int u = 0; u < UU; ++u)
#pragma unroll
            for (int t = 0; t < KK; ++t) dr[u][t] = dyn[(size_t)(o + u) * HoWo + max(off[t], 0)];
        MEDT_SCHED_FENCE();
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            float dv[KK];
#pragma unroll
            for (int t = 0; t < KK; ++t) dv[t] = off[t] >= 0 ? dr[u][t] : 0.f;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float* wp = w + ((size_t)(o + u) * Cin + c0 + c) * KK;
#pragma unroll
                for (int t = 0; t < KK; ++t) acc[c] = fmaf(wp[t], dv[t], acc[c]);
            }
        }
        MEDT_SCHED_FENCE();
    };
    int o = 0;
    for (; o + U <= Cout; o += U) taps(std::integral_constant<int, U>{}, o);
    for (; o < Cout; ++o) taps(std::integral_constant<int, 1>{}, o);
    if (ok) {
        const size_t o0 = ((size_t)n * Cin + c0) * HW + p;
#pragma unroll
        for (int c = 0; c < CT; ++c) dx[o0 + (size_t)c * HW] = add ? acc[c] + add[o0 + (size_t)c * HW] : acc[c];
    }
}

// Few positions, many channels (the 4x4 / 2x2 maps of the deep LoGo layers, decoder1_p): the lanes of the kernel above
// would each walk all Cout output channels serially -- a chain of Cout/4 dependent global round trips (round 2: 78 us
// for decoder1_p's 67 MFLOP).  Here a workgroup owns 64 positions, its four waves split the o-contraction and combine
// through LDS (fixed order), and the loop keeps 8-16 channels' loads in flight.
// TM: the taps (bit t = 3 kh + kw) that ANY input pixel of the geometry uses -- compile-time, so dead taps cost nothing: a stride-2 layer on
// 2 x 2 maps (decoder1_p) meets its single output pixel through the four taps (1..2, 1..2) only (round 6; as a run-time mask behind
// uniform branches the same idea was SLOWER, 49.6 vs 28.2 us: the unrolled tap loop lost its shape)
template <int K, int CT, int TM = (1 << (K * K)) - 1>
__global__ __launch_bounds__(MEDT_THREADS) void conv2d_bwd_data_ws_kernel(
    const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int N, int Cin, int H, int W,
    int Cout, int Ho, int Wo, int stride, int pad, const float* __restrict__ add, int stage_dy) {
    constexpr int KK = K * K;
    MEDT_STATIC_SHARED float red[3][CT][64];
    extern __shared__ __attribute__((aligned(16))) float wl[];       // [Cout][CT][KK]: this workgroup's weight slice
    // stage_dy: the output gradient of the images this workgroup touches fits in LDS too ([image][Cout][Ho*Wo], after the
    // weights): the contraction loop then has no global loads at all (decoder1_p: 2x2 -> 1x1 maps, 16 images per workgroup)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int HW = H * W;
    const long q = (long)blockIdx.x * 64 + lane;
    const int c0 = blockIdx.y * CT;
    if constexpr (TM == (1 << KK) - 1) {
        const int nw = Cout * CT * KK;
        for (int e0 = threadIdx.x; e0 < nw; e0 += 8 * MEDT_THREADS) {            // 8 loads in flight per lane
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = min(e0 + u * MEDT_THREADS, nw - 1), o = e / (CT * KK), r = e - o * (CT * KK);
                v[u] = w[((size_t)o * Cin + c0) * KK + r];
            }
            MEDT_SCHED_FENCE();
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + u * MEDT_THREADS < nw) wl[e0 + u * MEDT_THREADS] = v[u];
        }
    } else {
        // only the live taps of the slice are staged (same LDS layout, the dead slots are never read)
        constexpr int NL = __builtin_popcount((unsigned)TM);
        const int nw = Cout * CT * NL;
        for (int e0 = threadIdx.x; e0 < nw; e0 += 8 * MEDT_THREADS) {
            float v[8];
            int dst[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = min(e0 + u * MEDT_THREADS, nw - 1), oc = e / NL, tl = e - oc * NL;
                int t = 0, seen = 0;                                              // the tl-th set bit of TM
#pragma unroll
                for (int b = 0; b < KK; ++b)
                    if ((TM >> b) & 1) { if (seen == tl) t = b; ++seen; }
                const int o = oc / CT, c = oc - o * CT;
                dst[u] = oc * KK + t;
                v[u] = w[((size_t)o * Cin + c0 + c) * KK + t];
            }
            MEDT_SCHED_FENCE();
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + u * MEDT_THREADS < nw) wl[dst[u]] = v[u];
        }
    }
    float* dl = wl + Cout * CT * KK;
    const int HoWo_ = Ho * Wo;
    const int nb0 = (int)(((long)blockIdx.x * 64) / HW);
    const int img_stride = Cout * HoWo_ + 1;                       // +1: lanes of different images hit different banks
    if (stage_dy) {
        const int nb1 = min(N - 1, (int)(((long)blockIdx.x * 64 + 63) / HW));
        const int per = Cout * HoWo_, cnt = (nb1 - nb0 + 1) * per;
        const float* src = dy + (size_t)nb0 * per;
        for (int e0 = threadIdx.x; e0 < cnt; e0 += 8 * MEDT_THREADS) {      // 8 loads in flight per lane
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[min(e0 + u * MEDT_THREADS, cnt - 1)];
            MEDT_SCHED_FENCE();
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * MEDT_THREADS;
                if (e < cnt) { const int im = e / per; dl[im * img_stride + (e - im * per)] = v[u]; }
            }
        }
    }
    const bool ok = q < (long)N * HW;
    const int n = ok ? (int)(q / HW) : 0, p = ok ? (int)(q - (long)n * HW) : 0;
    const int h = p / W, ww = p - h * W;
    int off[KK];
#pragma unroll
    for (int kh = 0; kh < K; ++kh)
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
            const int hh = h + pad - kh, wq = ww + pad - kw;
            int o = -1;
            if (ok && hh >= 0 && wq >= 0 && hh % stride == 0 && wq % stride == 0) {
                const int ho = hh / stride, wo = wq / stride;
                if (ho < Ho && wo < Wo) o = ho * Wo + wo;
            }
            off[kh * K + kw] = o;
        }
    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = 0.f;
    const float* dyn = dy + (size_t)n * Cout * Ho * Wo;
    const int ob = (Cout * wv) / 4, oe = (Cout * (wv + 1)) / 4;
    __syncthreads();
    constexpr int U = K == 1 ? 16 : 4;                             // output channels per batch of loads
    const int HoWo = Ho * Wo;
    const float* dln = dl + (n - nb0) * img_stride;            // (stage_dy) this lane's image in LDS
    auto taps = [&](auto u_tag, int o) {
        constexpr int UU = decltype(u_tag)::value;
        float dr[UU][KK];
        if (stage_dy) {
#pragma unroll
            for (int u = 0; u < UU; ++u)
#pragma unroll
                for (int t = 0; t < KK; ++t) dr[u][t] = ((TM >> t) & 1) ? dln[(o + u) * HoWo + max(off[t], 0)] : 0.f;
        } else {
#pragma unroll
            for (int u = 0; u < UU; ++u)
#pragma unroll
                for (int t = 0; t < KK; ++t) dr[u][t] = ((TM >> t) & 1) ? dyn[(size_t)(o + u) * HoWo + max(off[t], 0)] : 0.f;
            MEDT_SCHED_FENCE();
        }
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            float dv[KK];
#pragma unroll
            for (int t = 0; t < KK; ++t) dv[t] = off[t] >= 0 ? dr[u][t] : 0.f;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float* wp = wl + ((o + u) * CT + c) * KK;
#pragma unroll
                for (int t = 0; t < KK; ++t)
                    if ((TM >> t) & 1) acc[c] = fmaf(wp[t], dv[t], acc[c]);
            }
        }
        MEDT_SCHED_FENCE();
    };
    int o = ob;
    for (; o + U <= oe; o += U) taps(std::integral_constant<int, U>{}, o);
    for (; o < oe; ++o) taps(std::integral_constant<int, 1>{}, o);
    if (wv > 0) {
#pragma unroll
        for (int c = 0; c < CT; ++c) red[wv - 1][c][lane] = acc[c];
    }
    __syncthreads();
    if (wv == 0 && ok) {
        const size_t o0 = ((size_t)n * Cin + c0) * HW + p;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const float v = acc[c] + (red[0][c][lane] + red[1][c][lane]) + red[2][c][lane];
            dx[o0 + (size_t)c * HW] = add ? v + add[o0 + (size_t)c * HW] : v;
        }
    }
}

static bool conv_bwd_data_ws_enabled() {
    static const bool on = true;
    return on;
}

template <int K>
static int conv2d_bwd_data_k(const float* dy, const float* w, float* dx, int N, int Cin, int H, int W, int Cout, int Ho,
                             int Wo, int stride, int pad, hipStream_t s, const float* add) {
    if constexpr (K != 7) {
        // (3 x 3 with a deep contraction up to 16384 positions: conv2's 8 <- 128 at 64 x 64 was 58 us in the kernel below)
        static const long ws_pos3 = 16384L;
        const long npos = (long)N * H * W;
        if ((npos <= 4096 || (K == 3 && Cout * K * K >= 512 && npos <= ws_pos3)) && Cout >= 64 && conv_bwd_data_ws_enabled()) {
            const unsigned g64 = (unsigned)(((long)N * H * W + 63) / 64);
            // images one 64-position workgroup can touch, and their output gradient in floats
            const int imgs = H * W >= 64 ? 2 : 64 / (H * W) + 1;
            const size_t dy_floats = (size_t)imgs * (Cout * Ho * Wo + 1);
            // the taps any input pixel uses (see the kernel's TM)
            int tmask = 0;
            if (K == 3)
                for (int h = 0; h < H && h < 8; ++h)
                    for (int ww = 0; ww < W && ww < 8; ++ww)
                        for (int t = 0; t < 9; ++t) {
                            const int hh = h + pad - t / 3, wq = ww + pad - t % 3;
                            if (hh >= 0 && wq >= 0 && hh % stride == 0 && wq % stride == 0 && hh / stride < Ho && wq / stride < Wo) tmask |= 1 << t;
                        }
#ifdef MEDT_AB_DGRAD_ALLTAPS             // (A/B build: every tap)
            tmask = 0x1ff;
#endif
            const bool four = K == 3 && H <= 8 && W <= 8 && tmask == 0x1b0;
#define MEDT_LAUNCH_WS(CT)                                                                                           \
    do {                                                                                                             \
        if constexpr (K == 3) {                                                                                      \
            if (four) {                                                                                              \
                hipLaunchKernelGGL((conv2d_bwd_data_ws_kernel<K, CT, 0x1b0>), dim3(g64, Cin / CT), dim3(MEDT_THREADS), \
                                   ((size_t)Cout * CT * K * K + (stage ? dy_floats : 0)) * sizeof(float), s, dy, w, dx, N, Cin, H, W, \
                                   Cout, Ho, Wo, stride, pad, add, stage);                                          \
                break;                                                                                               \
            }                                                                                                        \
        }                                                                                                            \
        hipLaunchKernelGGL((conv2d_bwd_data_ws_kernel<K, CT>), dim3(g64, Cin / CT), dim3(MEDT_THREADS),               \
                           ((size_t)Cout * CT * K * K + (stage ? dy_floats : 0)) * sizeof(float), s, dy, w, dx, N, Cin, H, W, \
                           Cout, Ho, Wo, stride, pad, add, stage);                                                  \
    } while (0)
            int ct = pick_tile(Cin, K == 1 ? 8 : 4, g64);
            while (ct > 1 && (size_t)Cout * ct * K * K * sizeof(float) > 48 * 1024) ct >>= 1;
            if ((size_t)Cout * ct * K * K * sizeof(float) > 48 * 1024) ct = 0;          // does not fit: the kernel below
            const int stage = (K == 3 && ct && ((size_t)Cout * ct * K * K + dy_floats) * sizeof(float) <= 60 * 1024) ? 1 : 0;
            switch (ct) {
                case 0: break;
                case 8: if constexpr (K == 1) { MEDT_LAUNCH_WS(8); } break;
                case 4: MEDT_LAUNCH_WS(4); break;
                case 2: MEDT_LAUNCH_WS(2); break;
                default: MEDT_LAUNCH_WS(1); break;
            }
#undef MEDT_LAUNCH_WS
            if (ct) return launch_status("conv2d_bwd_data_ws");
        }
    }
    const unsigned gx = (unsigned)(((long)N * H * W + MEDT_THREADS - 1) / MEDT_THREADS);
#define MEDT_LAUNCH_BWD(CT)                                                                                       \
    hipLaunchKernelGGL((conv2d_bwd_data_kernel<K, CT>), dim3(gx, Cin / CT), dim3(MEDT_THREADS), 0, s, dy, w, dx, N, Cin, \
                       H, W, Cout, Ho, Wo, stride, pad, add)
    switch (pick_tile(Cin, K == 7 ? 4 : 16, gx)) {
        case 16: if constexpr (K != 7) { MEDT_LAUNCH_BWD(16); } break;
        case 8: if constexpr (K != 7) { MEDT_LAUNCH_BWD(8); } break;
        case 4: MEDT_LAUNCH_BWD(4); break;
        case 2: MEDT_LAUNCH_BWD(2); break;
        default: MEDT_LAUNCH_BWD(1); break;
    }
#undef MEDT_LAUNCH_BWD
    return launch_status("conv2d_bwd_data");
}

// the backward-data of this layer runs as a forward MFMA convolution with flipped weights (conv_mfma_bwd_data_s1)
// backward-data as a forward convolution of dY with the flipped weights: the MFMA tile kernels, or (round 6) the thin-channel kernel
static bool conv2d_bwd_data_thin(int N, int Cin, int H, int W, int Cout, int K, int stride, int pad) {
    return (conv_thin_mode() & 2) && stride == 1 && K == 3 && pad == 1 && !conv_use_mfma(Cout, Cin, K, 1, (long)N * H * W) &&
           conv_thin_ok(N, 1, Cout, H, W, Cin, K, 1, 1);
}
bool conv2d_bwd_data_flips(int N, int Cin, int H, int W, int Cout, int K, int stride, int pad) {
    return stride == 1 && K - 1 - pad >= 0 &&
           (conv_use_mfma(Cout, Cin, K, 1, (long)N * H * W) || conv2d_bwd_data_thin(N, Cin, H, W, Cout, K, stride, pad));
}

int conv2d_bwd_data(const float* dy, const float* w, float* dx, float* wt_scratch, float* ksplit_scratch, int N, int Cin,
                    int H, int W, int Cout, int K, int stride, int pad, hipStream_t s, const float* add, bool wt_ready) {
    const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
    if (abl_skip(N >= 16 ? (K == 3 ? "conv3_dgrad_l" : "conv1_dgrad_l") : (K == 3 ? "conv3_dgrad_g" : "conv1_dgrad_g"))) return MEDT_OK;
    // as a forward convolution of dY: "Cout" = Cin, contraction over Cout*K*K
    if (wt_scratch && conv2d_bwd_data_thin(N, Cin, H, W, Cout, K, stride, pad)) {      // (its epilogue takes the fan-in addend)
        if (!wt_ready) {
            int rc = conv_flip_weights(w, wt_scratch, Cout, Cin, K, s);
            if (rc) return rc;
        }
        return conv_thin_fwd(dy, wt_scratch, nullptr, add, dx, nullptr, N, 1, Cout, H, W, Cin, 0, s);
    }
    // (the `add` epilogue: thin kernel above and the VALU path only)
    if (!add && wt_scratch && conv_use_mfma(Cout, Cin, K, 1, (long)N * H * W) && conv2d_bwd_data_flips(N, Cin, H, W, Cout, K, stride, pad) &&
        (ksplit_scratch || conv_mfma_scratch_floats(N, 1, H * W, Cout, Cin, K, conv_rows16_ok(Cout, H, W, K, 1, K - 1 - pad)) == 0))
        return conv_mfma_bwd_data_s1(dy, w, wt_scratch, ksplit_scratch, dx, N, Cin, H, W, Cout, K, pad, s, wt_ready);
    switch (K) {
        case 1: return conv2d_bwd_data_k<1>(dy, w, dx, N, Cin, H, W, Cout, Ho, Wo, stride, pad, s, add);
        case 3: return conv2d_bwd_data_k<3>(dy, w, dx, N, Cin, H, W, Cout, Ho, Wo, stride, pad, s, add);
        case 7: return conv2d_bwd_data_k<7>(dy, w, dx, N, Cin, H, W, Cout, Ho, Wo, stride, pad, s, add);
    }
    set_error("conv2d: kernel size %d unsupported (1, 3, 7)", K);
    return MEDT_EUNSUPPORTED;
}

// --------------------------------------------------------------------------- //
// backward-weight as an implicit GEMM:  dW[o, k] = sum_q dY[o, q] * Xcol[k, q]
//   k = (c, kh, kw) flattened (the weight tensor's own layout), q = (n, ho, wo) flattened.
// One workgroup = one 64(o) x 64(k) tile x one chunk of QS positions; 64-position LDS steps, 4x4 register
// tile per lane.  Flattening the taps into k keeps the tile full for the 3- and 8-channel stems (k = 147, 72)
// and QS shrinks until there are enough workgroups to fill the chip; chunk partials go through a deterministic
// reduction (reduce_rows), no float atomics.  The optional (raw, coef) pair applies a BatchNorm-backward affine
// to dY on load (qkv_transform's gradient, whose dY is still in normalised space).
// --------------------------------------------------------------------------- //
template <int K, int TO, int TC>          // TO x TC output tile (16 | 32 | 64 each): small layers do not pay for 64 x 64
__device__ __forceinline__ void conv_wgrad_body(
    const float* __restrict__ dy, const float* __restrict__ raw, const float* __restrict__ coef,
    const float* __restrict__ x, float* __restrict__ scratch, int N, int Cin, int H, int W, int Cout, int Ho, int Wo,
    int stride, int pad, int QS, int npg, int bx, int by, int bz) {
    constexpr int KK = K * K, RO = TO / 16, RC = TC / 16;
    // row stride 68: 16-byte aligned rows for ds_read_b128 along the position index, and 68 mod 64 = 4 puts the 16
    // B rows a wave reads at once on disjoint 4-bank groups
    MEDT_STATIC_SHARED __attribute__((aligned(16))) float A[TO][68];
    MEDT_STATIC_SHARED __attribute__((aligned(16))) float B[TC][68];
    const int Ktot = Cin * KK, HoWo = Ho * Wo;
    const int o0 = bx * TO, k0 = by * TC;
    const long NP = (long)N * HoWo;
    const long q_begin = (long)bz * QS;
    const long q_end = q_begin + QS < NP ? q_begin + QS : NP;
    const int to = threadIdx.x >> 4, tc = threadIdx.x & 15;
    const int j = threadIdx.x & 63, r0 = threadIdx.x >> 6;          // staging: fixed position j, rows r0, r0+4, ...
    float acc[RO][RC];
#pragma unroll
    for (int a = 0; a < RO; ++a)
#pragma unroll
        for (int b = 0; b < RC; ++b) acc[a][b] = 0.f;
    // Software pipeline: the global loads of the next 64 positions are issued before the FMA block of the current ones
    // and written to LDS after it (most layers run one workgroup per CU: nothing else hides the load latency).
    constexpr int NA = TO / 4, NB = TC / 4;
    float ra[NA], rb[NB];
    auto fetch = [&](long q0) {
        const long q = q0 + j;
        const bool qok = q < q_end;
        const int n = qok ? (int)(q / HoWo) : 0, p = qok ? (int)(q - (long)n * HoWo) : 0;
        const int ho = p / Wo, wo = p - ho * Wo;
        const int hb = ho * stride - pad, wb = wo * stride - pad;
        const float* dyp = dy + (size_t)n * Cout * HoWo + p;
        const float* rawp = raw ? raw + (size_t)n * Cout * HoWo + p : nullptr;
        const float* cf = coef ? coef + (size_t)(n / npg) * Cout * 3 : nullptr;
        const float* xp = x + (size_t)n * Cin * H * W;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            float a = 0.f;
            const int o = o0 + r0 + 4 * i;
            if (qok && o < Cout) {
                a = dyp[(size_t)o * HoWo];
                if (cf) a = fmaf(cf[o * 3], a, fmaf(cf[o * 3 + 1], rawp[(size_t)o * HoWo], cf[o * 3 + 2]));
            }
            ra[i] = a;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            float b = 0.f;
            const int k = k0 + r0 + 4 * i;
            if (qok && k < Ktot) {
                const int c = k / KK, t = k - c * KK;
                const int kh = t / K, kw = t - kh * K;
                const int h = hb + kh, w = wb + kw;
                if (h >= 0 && h < H && w >= 0 && w < W) b = xp[((size_t)c * H + h) * W + w];
            }
            rb[i] = b;
        }
    };
    if (q_begin < q_end) fetch(q_begin);
    for (long q0 = q_begin; q0 < q_end; q0 += 64) {
#pragma unroll
        for (int i = 0; i < NA; ++i) A[r0 + 4 * i][j] = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) B[r0 + 4 * i][j] = rb[i];
        __syncthreads();
        if (q0 + 64 < q_end) fetch(q0 + 64);
#pragma unroll 2
        for (int jj = 0; jj < 64; jj += 4) {                      // four positions per 16-byte LDS read
            float4 av[RO], bv[RC];
#pragma unroll
            for (int a = 0; a < RO; ++a) av[a] = *reinterpret_cast<const float4*>(&A[to + 16 * a][jj]);
#pragma unroll
            for (int b = 0; b < RC; ++b) bv[b] = *reinterpret_cast<const float4*>(&B[tc + 16 * b][jj]);
#pragma unroll
            for (int a = 0; a < RO; ++a)
#pragma unroll
                for (int b = 0; b < RC; ++b) {
                    float t = acc[a][b];
                    t = fmaf(av[a].x, bv[b].x, t);
                    t = fmaf(av[a].y, bv[b].y, t);
                    t = fmaf(av[a].z, bv[b].z, t);
                    t = fmaf(av[a].w, bv[b].w, t);
                    acc[a][b] = t;
                }
        }
        __syncthreads();
    }
    float* out = scratch + (size_t)bz * Cout * Ktot;
#pragma unroll
    for (int a = 0; a < RO; ++a)
#pragma unroll
        for (int b = 0; b < RC; ++b) {
            const int o = o0 + to + 16 * a, k = k0 + tc + 16 * b;
            if (o < Cout && k < Ktot) out[(size_t)o * Ktot + k] = acc[a][b];
        }
}

template <int K, int TO, int TC>
__global__ __launch_bounds__(MEDT_THREADS) void conv_wgrad_kernel(
    const float* __restrict__ dy, const float* __restrict__ raw, const float* __restrict__ coef,
    const float* __restrict__ x, float* __restrict__ scratch, int N, int Cin, int H, int W, int Cout, int Ho, int Wo,
    int stride, int pad, int QS, int npg) {
    conv_wgrad_body<K, TO, TC>(dy, raw, coef, x, scratch, N, Cin, H, W, Cout, Ho, Wo, stride, pad, QS, npg, blockIdx.x,
                               blockIdx.y, blockIdx.z);
}

static inline int wgrad_tile(int n) { return n <= 16 ? 16 : (n <= 32 ? 32 : 64); }

// positions per chunk: as small as 64 while the grid is below ~512 workgroups, at most 512
static int wgrad_chunk(int Cout, int Ktot, long NP) {
    const long tiles = (long)cdiv(Cout, wgrad_tile(Cout)) * cdiv(Ktot, wgrad_tile(Ktot));
    int QS = 512;
    while (QS > 64 && tiles * ((NP + QS - 1) / QS) < 512) QS >>= 1;
    while ((NP + QS - 1) / QS > 64) QS <<= 1;            // at most 64 partial slabs for the deterministic reduction
    return QS;
}

// recorded (grouped) launches: ~64 position chunks of at most 256 positions per job, at most 64 partial slabs.  A flush is
// a few thousand workgroups over all its layers, so the launch's time is the serial latency of ONE workgroup's chunk
// (~4 us per 64 positions): short chunks, many workgroups (measured on the MedT step: 32 chunks / 512 positions 2.54 ms,
// 64 / 256: 2.49 ms, 128 slabs of 128: 2.53 ms -- the slab reduction starts to cost what the shorter chunks save)
static int wgrad_chunk_grouped(long NP) {
    static const int target = 64;
    static const int qmax = 256;
    int QS = 64;
    while (QS < qmax && (NP + QS - 1) / QS > target) QS <<= 1;
    static const int smax = 64;
    // (>= 65536 positions -- the 128 x 128 maps of decoderf / adjust: 16-step chunks were the long pole of the global branch's flush;
    //  their weight matrices are tiny, so 256 slabs cost the row reduction nothing: 2.080 -> 2.067 ms/step, profiles/r05_step_ab.json)
    static const int smax_big = 256;
    while ((NP + QS - 1) / QS > (NP >= 65536 ? smax_big : smax)) QS <<= 1;
    return QS;
}

// chunks of the 16-byte body: 256 positions (two 128-position steps) while that gives at most MEDT_WG4_CHUNKS (16) chunks, up to
// MEDT_WG4_QMAX (1024) positions; never fewer positions per chunk than the scalar policy (so never more slabs)
static int wgrad_chunk_v4(long NP) {
    static const int target = 16;
    static const int qmax = 1024;
    static const int smax_big = 256;
    int QS = 256;
    while (QS < qmax && (NP + QS - 1) / QS > (NP >= 65536 ? smax_big : target)) QS <<= 1;
    while ((NP + QS - 1) / QS > (NP >= 65536 ? smax_big : 64)) QS <<= 1;
    const int QG = wgrad_chunk_grouped(NP);
    return QS > QG ? QS : QG;
}

int conv2d_bwd_weight_splits(int N, int Cin, int Cout, int K, int Ho, int Wo) {      // slabs of scratch: either mode
    const long NP = (long)N * Ho * Wo;
    const int QS = wgrad_chunk(Cout, Cin * K * K, NP), QG = wgrad_chunk_grouped(NP);
    const int a = (int)((NP + QS - 1) / QS), b = (int)((NP + QG - 1) / QG);
    return a > b ? a : b;
}

int conv2d_bwd_weight(const float* dy, const float* raw, const float* coef, const float* x, float* dw, float* scratch,
                      int N, int Cin, int H, int W, int Cout, int K, int stride, int pad, int groups, hipStream_t s,
                      Queue* q) {
    const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
    const long NP = (long)N * Ho * Wo;
    const int Ktot = Cin * K * K;
    const bool mfma = conv_use_mfma(Cin, Cout, K, stride, NP) && Cout >= 64;
    // recorded 1x1 stride-1 problems: the 16-byte position-axis body (conv_mfma.hip) with chunks of 256 ... 1024 positions --
    // 4x fewer workgroups and partial slabs than the scalar-load body's 64 ... 256 (never more slabs than the workspace was
    // sized for: conv2d_bwd_weight_splits takes the scalar policy's count, which is the larger)
    const bool v4 = q && !mfma && conv_wgrad_v4_ok(dy, raw, x, N, Cin, H, W, Cout, Ho, Wo, K, stride, pad);
    const int QS = v4 ? wgrad_chunk_v4(NP) : ((q && !mfma) ? wgrad_chunk_grouped(NP) : wgrad_chunk(Cout, Ktot, NP));
    const int splits = (int)((NP + QS - 1) / QS);
    const dim3 grid(cdiv(Cout, 64), cdiv(Ktot, 64), splits), block(MEDT_THREADS);
    if (splits == 1) scratch = dw;                         // a single slab is the result: no reduction pass
    if (mfma) {
        static const bool defer_m = true;
        if (q && defer_m) {       // recorded like the grouped jobs: nothing on the layer chain reads a weight gradient
            q->mwgrad.push_back(MJob{dy, raw, coef, x, scratch, N, Cin, H, W, Cout, Ho, Wo, K, stride, pad, QS, splits, N / groups});
            if (splits > 1) q->reduce.push_back(RJob{scratch, dw, splits, Cout * Ktot});
            return MEDT_OK;
        }
        int rc = conv_wgrad_mfma(dy, raw, coef, x, scratch, N, Cin, H, W, Cout, Ho, Wo, K, stride, pad, QS, splits,
                                 N / groups, s);
        if (rc) return rc;
        if (splits == 1) return MEDT_OK;
        if (q) { q->reduce.push_back(RJob{scratch, dw, splits, Cout * Ktot}); return MEDT_OK; }
        return reduce_rows(scratch, splits, Cout * Ktot, dw, s);
    }
    if (q) {                                               // deferred: grouped with the other layers' at the flush
        if (K != 1 && K != 3 && K != 7) { set_error("conv2d: kernel size %d unsupported (1, 3, 7)", K); return MEDT_EUNSUPPORTED; }
        q->wgrad.push_back(WJob{dy, raw, coef, x, scratch, N, Cin, H, W, Cout, Ho, Wo, stride, pad, QS, N / groups,
                                cdiv(Cout, 64), cdiv(Ktot, 64), splits, K, v4 ? 1 : 0});
        if (splits > 1) q->reduce.push_back(RJob{scratch, dw, splits, Cout * Ktot});
        return MEDT_OK;
    }
    {
        const int TO = wgrad_tile(Cout), TC = wgrad_tile(Ktot);
        const dim3 g2(cdiv(Cout, TO), cdiv(Ktot, TC), splits);
#define MEDT_WG(KV, TOV, TCV)                                                                                        \
    hipLaunchKernelGGL((conv_wgrad_kernel<KV, TOV, TCV>), g2, block, 0, s, dy, raw, coef, x, scratch, N, Cin, H, W, Cout, Ho, \
                       Wo, stride, pad, QS, N / groups)
#define MEDT_WG_TC(KV, TOV)                                                                                          \
    do { if (TC == 16) MEDT_WG(KV, TOV, 16); else if (TC == 32) MEDT_WG(KV, TOV, 32); else MEDT_WG(KV, TOV, 64); } while (0)
#define MEDT_WG_TO(KV)                                                                                               \
    do { if (TO == 16) MEDT_WG_TC(KV, 16); else if (TO == 32) MEDT_WG_TC(KV, 32); else MEDT_WG_TC(KV, 64); } while (0)
        switch (K) {
            case 1: MEDT_WG_TO(1); break;
            case 3: MEDT_WG_TO(3); break;
            case 7: MEDT_WG_TO(7); break;
            default: set_error("conv2d: kernel size %d unsupported (1, 3, 7)", K); return MEDT_EUNSUPPORTED;
        }
#undef MEDT_WG_TO
#undef MEDT_WG_TC
#undef MEDT_WG
    }
    int rc = launch_status("conv_wgrad");
    if (rc) return rc;
    return splits == 1 ? MEDT_OK : reduce_rows(scratch, splits, Cout * Ktot, dw, s);
}

// per-channel sum over (n, pixels):  out[c] = sum x[n,c,:]   (bias gradients); deterministic.  One workgroup per (channel, split);
// round 6: the number of splits follows the size -- a channel of <= 4096 values is ONE workgroup that writes the gradient itself
// (no partial rows, no reduction job).  Sixteen splits everywhere made the local branch's five decoders 7936 workgroups of 4 - 1024
// values each: 48 us of the step's final flush for 2 MB.
#define CS_SPLITS 16
int channel_sum_splits_for(int N, int HW) {
    const long total = (long)N * HW;
    long sp = (total + 4095) / 4096;
    return sp < 1 ? 1 : (sp > CS_SPLITS ? CS_SPLITS : (int)sp);
}
__device__ __forceinline__ void channel_sum_body(const float* __restrict__ x, float* __restrict__ part, int N, int C,
                                                 int HW, int c, int sp, int splits) {
    MEDT_STATIC_SHARED float red[MEDT_WAVES];
    float v[1] = {0.f};
    const long total = (long)N * HW;
    const long per = (total + splits - 1) / splits, beg = sp * per, end = beg + per < total ? beg + per : total;
    if (total < (1l << 31)) {
        // 32-bit index arithmetic (a 64-bit division per element was most of this kernel's time), four loads in flight per lane
        const unsigned e32 = (unsigned)end, hw = (unsigned)HW;
        float a4[4] = {0.f, 0.f, 0.f, 0.f};
        for (unsigned q = (unsigned)beg + threadIdx.x; q < e32; q += 4 * MEDT_THREADS) {
            float t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned qq = min(q + u * MEDT_THREADS, e32 - 1), n = qq / hw, p = qq - n * hw;
                t[u] = x[((size_t)n * C + c) * HW + p];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (q + u * MEDT_THREADS < e32) a4[u] += t[u];
        }
        v[0] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    } else {
        for (long q = beg + threadIdx.x; q < end; q += MEDT_THREADS) {
            const int n = (int)(q / HW), p = (int)(q - (long)n * HW);
            v[0] += x[((size_t)n * C + c) * HW + p];
        }
    }
    block_sum<1>(v, red, part + (size_t)sp * C + c);
}

__global__ __launch_bounds__(MEDT_THREADS) void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                                   int N, int C, int HW) {
    channel_sum_body(x, part, N, C, HW, blockIdx.x, blockIdx.y, gridDim.y);
}

using CBatch = JobBatch<CJob, 96>;
__global__ __launch_bounds__(MEDT_THREADS) void channel_sum_grouped_kernel(CBatch b);

int channel_sum(const float* x, float* out, float* scratch, int N, int C, int HW, hipStream_t s) {
    const int splits = channel_sum_splits_for(N, HW);
    hipLaunchKernelGGL(channel_sum_kernel, dim3(C, splits), dim3(MEDT_THREADS), 0, s, x, splits == 1 ? out : scratch, N, C, HW);
    int rc = launch_status("channel_sum");
    if (rc || splits == 1) return rc;
    return reduce_rows(scratch, splits, C, out, s);
}

__global__ __launch_bounds__(MEDT_THREADS) void channel_sum_grouped_kernel(CBatch b) {
    const int j = find_job(b, blockIdx.x);
    const CJob& c = b.job[j];
    const int local = blockIdx.x - b.start[j];
    channel_sum_body(c.x, c.part, c.N, c.C, c.HW, local % c.C, local / c.C, c.splits);
}


int channel_sum_grouped(const CJob* jobs, int n, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += 96) {
        CBatch b;
        b.n = n - i0 < 96 ? n - i0 : 96;
        int blocks = 0;
        for (int i = 0; i < b.n; ++i) { b.job[i] = jobs[i0 + i]; b.start[i] = blocks; blocks += jobs[i0 + i].C * jobs[i0 + i].splits; }
        b.start[b.n] = blocks;
        hipLaunchKernelGGL(channel_sum_grouped_kernel, dim3(blocks), dim3(MEDT_THREADS), 0, s, b);
        int rc = launch_status("channel_sum_grouped");
        if (rc) return rc;
    }
    return MEDT_OK;
}

}  // namespace medt
