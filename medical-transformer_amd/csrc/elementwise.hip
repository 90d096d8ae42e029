// elementwise.hip -- the streaming kernels between the convolutions and the attention layers:
//   BatchNorm2d apply (+residual, +ReLU) and its backward   (axialnet.py:285-300, 475-483)
//   bilinear x2 upsample + ReLU + skip-add                   (axialnet.py:493-501, 650-652, 690-698)
//   LoGo patch gather / merge                                (axialnet.py:658-702)
//   cross-entropy loss                                       (metrics.py:17-20)
//   Adam with coupled L2 weight decay over a flat buffer     (train.py:111-112,161)
// All HBM-bound: one element per lane, NCHW rows coalesced, per-channel constants via scalar loads.
#include "medt_kernels.h"
#include <stdint.h>
#include <stdlib.h>

namespace medt {

static inline unsigned grid1d(size_t total) { return (unsigned)((total + MEDT_THREADS - 1) / MEDT_THREADS); }

// y = [relu]( z*scale[g,c] + shift[g,c] [+ res] )         one lane per element of (N,C,HW)
__global__ __launch_bounds__(MEDT_THREADS) void bn_apply_act_kernel(const float* __restrict__ z, BnStats st,
                                                                    const float* __restrict__ res,
                                                                    float* __restrict__ y, int C, int HW, int npg,
                                                                    int relu, size_t total) {
    const size_t idx = (size_t)blockIdx.x * MEDT_THREADS + threadIdx.x;
    if (idx >= total) return;
    const size_t nc = idx / HW;
    const int c = (int)(nc % C), n = (int)(nc / C);
    const int gc = (n / npg) * C + c;
    float v = fmaf(z[idx], st.scale[gc], st.shift[gc]);
    if (res) v += res[idx];
    if (relu) v = fmaxf(v, 0.f);
    y[idx] = v;
}

int bn_apply_act(const float* z, BnStats st, const float* res, float* y, int N, int C, int HW, int groups, int relu,
                 hipStream_t s) {
    const size_t total = (size_t)N * C * HW;
    hipLaunchKernelGGL(bn_apply_act_kernel, dim3(grid1d(total)), dim3(MEDT_THREADS), 0, s, z, st, res, y, C, HW,
                       N / groups, relu, total);
    return launch_status("bn_apply_act");
}

// bn_finalize + bn_apply_act in one launch: every workgroup re-derives the statistics of ITS (group, channel) from the
// convolution's partial sums (ppg pairs of doubles: one round trip) and then normalises its slice of that population;
// part 0 saves mean / rstd / scale / shift for backward.  What is left of the finalisation -- the running-statistics
// recurrence over the groups -- is off the layer chain (recorded, or issued right behind when no queue is bound).
struct BnFinApplyArgs {
    const float *z, *partials, *weight, *bias, *running_mean, *running_var, *res;
    float* y;
    BnStats st;
    int C, HW, npg, ppg, parts, relu, training, vec4;     // vec4: HW % 4 == 0 and z / res / y 16-byte aligned
    double count;
    float eps;
};
constexpr int BFA_PER_THREAD = 16;
__global__ __launch_bounds__(MEDT_THREADS) void bn_fin_apply_kernel(BnFinApplyArgs a) {
    MEDT_STATIC_SHARED double redd[2 * MEDT_WAVES];
    const int grp = blockIdx.x / a.parts, part = blockIdx.x - grp * a.parts, c = blockIdx.y, tid = threadIdx.x;
    const int gc = grp * a.C + c;
    const float gam = a.weight[c], bet = a.bias[c];
    float mean_f, rstd_f, scale, shift;
    if (a.training) {
        const double* pd = reinterpret_cast<const double*>(a.partials);
        double s = 0.0, ss = 0.0;
        for (int p = tid; p < a.ppg; p += MEDT_THREADS) {
            const double* q = pd + ((size_t)(grp * a.ppg + p) * a.C + c) * 2;
            s += q[0];
            ss += q[1];
        }
        s = wave_sum_d(s);
        ss = wave_sum_d(ss);
        if ((tid & 63) == 0) { redd[(tid >> 6) * 2] = s; redd[(tid >> 6) * 2 + 1] = ss; }
        __syncthreads();
        s = 0.0;
        ss = 0.0;
#pragma unroll
        for (int w = 0; w < MEDT_WAVES; ++w) { s += redd[w * 2]; ss += redd[w * 2 + 1]; }
        const double mean = s / a.count;
        double var = ss / a.count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)a.eps);
        mean_f = (float)mean;
        rstd_f = (float)rstd;
        scale = (float)(gam * rstd);
        shift = (float)(bet - mean * gam * rstd);
    } else {
        mean_f = a.running_mean[c];
        rstd_f = (float)(1.0 / sqrt((double)a.running_var[c] + (double)a.eps));
        scale = gam * rstd_f;
        shift = bet - mean_f * gam * rstd_f;
    }
    if (part == 0 && tid == 0) {
        a.st.mean[gc] = mean_f;
        a.st.rstd[gc] = rstd_f;
        a.st.scale[gc] = scale;
        a.st.shift[gc] = shift;
    }
    const int HW = a.HW, P = a.npg * HW;
    const int q0 = part * (MEDT_THREADS * BFA_PER_THREAD);
    const int q1 = q0 + MEDT_THREADS * BFA_PER_THREAD < P ? q0 + MEDT_THREADS * BFA_PER_THREAD : P;
    if (a.vec4) {
        for (int q = q0 + 4 * tid; q < q1; q += 4 * MEDT_THREADS) {
            const int ni = q / HW, p = q - ni * HW;
            const size_t idx = ((size_t)(grp * a.npg + ni) * a.C + c) * HW + p;
            const float4 zz = *reinterpret_cast<const float4*>(a.z + idx);
            float4 v;
            v.x = fmaf(zz.x, scale, shift);
            v.y = fmaf(zz.y, scale, shift);
            v.z = fmaf(zz.z, scale, shift);
            v.w = fmaf(zz.w, scale, shift);
            if (a.res) {
                const float4 r = *reinterpret_cast<const float4*>(a.res + idx);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(a.y + idx) = v;
        }
    } else {
        for (int q = q0 + tid; q < q1; q += MEDT_THREADS) {
            const int ni = q / HW, p = q - ni * HW;
            const size_t idx = ((size_t)(grp * a.npg + ni) * a.C + c) * HW + p;
            float v = fmaf(a.z[idx], scale, shift);
            if (a.res) v += a.res[idx];
            if (a.relu) v = fmaxf(v, 0.f);
            a.y[idx] = v;
        }
    }
}

int bn_fin_apply(const float* z, const float* partials, int ppg, double count, const medt_bn_ptrs& bn, float eps, int training,
                 BnStats st, const float* res, float* y, int N, int C, int HW, int groups, int relu, hipStream_t s) {
    BnFinApplyArgs a;
    a.z = z; a.partials = partials; a.weight = bn.weight; a.bias = bn.bias; a.running_mean = bn.running_mean;
    a.running_var = bn.running_var; a.res = res; a.y = y; a.st = st; a.C = C; a.HW = HW; a.npg = N / groups; a.ppg = ppg;
    a.parts = cdiv(a.npg * HW, MEDT_THREADS * BFA_PER_THREAD); a.relu = relu; a.training = training; a.count = count;
    a.vec4 = ((HW & 3) == 0 && (((uintptr_t)z | (uintptr_t)y | (uintptr_t)res) & 15) == 0) ? 1 : 0;
    a.eps = eps;
    hipLaunchKernelGGL(bn_fin_apply_kernel, dim3(groups * a.parts, C), dim3(MEDT_THREADS), 0, s, a);
    return launch_status("bn_fin_apply");
}

// g = dy * (y > 0 if relu) ; partials[group][part][C][2] = [sum g, sum g*zhat]     grid (groups*ppg, C),
// lanes over the flattened (image, pixel) positions of one group and one channel
__global__ __launch_bounds__(MEDT_THREADS) void bn_act_bwd_stats_kernel(const float* __restrict__ dy,
                                                                        const float* __restrict__ y,
                                                                        const float* __restrict__ z, BnStats st,
                                                                        float* __restrict__ g,
                                                                        float* __restrict__ partials, int C, int HW,
                                                                        int npg, int relu) {
    MEDT_STATIC_SHARED float red[MEDT_WAVES * 2];
    const int per_group = npg * HW, ppg = (per_group + MEDT_THREADS - 1) / MEDT_THREADS;
    const int grp = blockIdx.x / ppg, part = blockIdx.x - grp * ppg, c = blockIdx.y;
    const int q = part * MEDT_THREADS + threadIdx.x;
    float v[2] = {0.f, 0.f};
    if (q < per_group) {
        const int ni = q / HW, p = q - ni * HW;
        const int gc = grp * C + c;
        const size_t idx = ((size_t)(grp * npg + ni) * C + c) * HW + p;
        float d = dy[idx];
        if (relu && !(y[idx] > 0.f)) d = 0.f;
        if (g) g[idx] = d;
        v[0] = d;
        v[1] = d * ((z[idx] - st.mean[gc]) * st.rstd[gc]);
    }
    block_sum<2>(v, red, partials + ((size_t)blockIdx.x * C + c) * 2);
}

int bn_act_bwd_stats(const float* dy, const float* y, const float* z, BnStats st, float* g, float* partials, int N, int C,
                     int HW, int groups, int relu, hipStream_t s) {
    const int ppg = cdiv((N / groups) * HW, MEDT_THREADS);
    hipLaunchKernelGGL(bn_act_bwd_stats_kernel, dim3(groups * ppg, C), dim3(MEDT_THREADS), 0, s, dy, y, z, st, g,
                       partials, C, HW, N / groups, relu);
    return launch_status("bn_act_bwd_stats");
}

// dz = c0*g + c1*z + c2   (coef [group][C][3])
__global__ __launch_bounds__(MEDT_THREADS) void bn_bwd_apply_kernel(const float* __restrict__ g,
                                                                    const float* __restrict__ z,
                                                                    const float* __restrict__ coef,
                                                                    float* __restrict__ dz, int C, int HW, int npg,
                                                                    size_t total) {
    const size_t idx = (size_t)blockIdx.x * MEDT_THREADS + threadIdx.x;
    if (idx >= total) return;
    const size_t nc = idx / HW;
    const int c = (int)(nc % C), n = (int)(nc / C);
    const float* cf = coef + ((size_t)(n / npg) * C + c) * 3;
    dz[idx] = fmaf(cf[0], g[idx], fmaf(cf[1], z[idx], cf[2]));
}

int bn_bwd_apply(const float* g, const float* z, const float* coef, float* dz, int N, int C, int HW, int groups,
                 hipStream_t s) {
    const size_t total = (size_t)N * C * HW;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid1d(total)), dim3(MEDT_THREADS), 0, s, g, z, coef, dz, C, HW,
                       N / groups, total);
    return launch_status("bn_bwd_apply");
}

// out = a * (y > 0)            (ReLU backward by output sign)
__global__ __launch_bounds__(MEDT_THREADS) void relu_mask_kernel(const float* __restrict__ a, const float* __restrict__ y,
                                                                 float* __restrict__ out, size_t total) {
    const size_t i = (size_t)blockIdx.x * MEDT_THREADS + threadIdx.x;
    if (i < total) out[i] = y[i] > 0.f ? a[i] : 0.f;
}

int relu_mask(const float* a, const float* y, float* out, size_t total, hipStream_t s) {
    hipLaunchKernelGGL(relu_mask_kernel, dim3(grid1d(total)), dim3(MEDT_THREADS), 0, s, a, y, out, total);
    return launch_status("relu_mask");
}

// y = relu(x)  /  y = relu(x) in place of a previous ReLU-free tensor
__global__ __launch_bounds__(MEDT_THREADS) void relu_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            size_t total) {
    const size_t i = (size_t)blockIdx.x * MEDT_THREADS + threadIdx.x;
    if (i < total) y[i] = fmaxf(x[i], 0.f);
}

int relu_fwd(const float* x, float* y, size_t total, hipStream_t s) {
    hipLaunchKernelGGL(relu_kernel, dim3(grid1d(total)), dim3(MEDT_THREADS), 0, s, x, y, total);
    return launch_status("relu");
}

// --------------------------------------------------------------------------- //
// bilinear x2 (align_corners=False) + ReLU + skip-add
// --------------------------------------------------------------------------- //
struct Lerp { int i0, i1; float l; };
__device__ __forceinline__ Lerp src_index(int o, int n_in) {
    float s = 0.5f * ((float)o + 0.5f) - 0.5f;         // aten area_pixel_compute_source_index, scale = 1/2
    if (s < 0.f) s = 0.f;
    Lerp r;
    r.i0 = (int)s;
    r.i1 = r.i0 + (r.i0 < n_in - 1 ? 1 : 0);
    r.l = s - (float)r.i0;
    return r;
}

// one interpolated value from its four sources: explicit FMAs, so that the forward value and the backward's ReLU mask
// (recomputed from registers) are the same arithmetic whatever the compiler contracts
__device__ __forceinline__ float up_value(float v00, float v01, float v10, float v11, float lh, float lw) {
    const float h0 = 1.f - lh, w0 = 1.f - lw;
    const float t0 = fmaf(lw, v01, w0 * v00), t1 = fmaf(lw, v11, w0 * v10);
    return fmaf(lh, t1, h0 * t0);
}

__device__ __forceinline__ float up_sample(const float* __restrict__ xp, int H, int W, int ho, int wo) {
    const Lerp a = src_index(ho, H), b = src_index(wo, W);
    return up_value(xp[a.i0 * W + b.i0], xp[a.i0 * W + b.i1], xp[a.i1 * W + b.i0], xp[a.i1 * W + b.i1], a.l, b.l);
}

__global__ __launch_bounds__(MEDT_THREADS) void up2x_relu_add_kernel(const float* __restrict__ x,
                                                                     const float* __restrict__ skip,
                                                                     float* __restrict__ y, int H, int W, size_t total) {
    const size_t idx = (size_t)blockIdx.x * MEDT_THREADS + threadIdx.x;
    if (idx >= total) return;
    const int Wo = 2 * W, Ho = 2 * H;
    const int wo = (int)(idx % Wo), ho = (int)((idx / Wo) % Ho);
    const size_t nc = idx / ((size_t)Wo * Ho);
    float v = fmaxf(up_sample(x + nc * H * W, H, W, ho, wo), 0.f);
    if (skip) v += skip[idx];
    y[idx] = v;
}

// Four consecutive output columns per lane (W even): they interpolate between input columns 2t-1 .. 2t+2 of two rows, so
// eight loads serve four outputs and skip / y move as float4.  Same Lerp values and the same up_value arithmetic as the
// scalar kernel (the border clamps coincide with src_index's).
__global__ __launch_bounds__(MEDT_THREADS) void up2x_relu_add4_kernel(const float* __restrict__ x,
                                                                      const float* __restrict__ skip,
                                                                      float* __restrict__ y, int H, int W, size_t total4) {
    const size_t idx = (size_t)blockIdx.x * MEDT_THREADS + threadIdx.x;
    if (idx >= total4) return;
    const int Wq = W / 2, Ho = 2 * H;
    const int t = (int)(idx % Wq), ho = (int)((idx / Wq) % Ho);
    const size_t nc = idx / ((size_t)Wq * Ho);
    const float* xp = x + nc * H * W;
    const Lerp a = src_index(ho, H);
    const int cc[4] = {max(2 * t - 1, 0), 2 * t, 2 * t + 1, min(2 * t + 2, W - 1)};
    float r0[4], r1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r0[k] = xp[a.i0 * W + cc[k]];
        r1[k] = xp[a.i1 * W + cc[k]];
    }
    constexpr int C0[4] = {0, 1, 1, 2};
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const Lerp b = src_index(4 * t + k, W);
        v[k] = fmaxf(up_value(r0[C0[k]], r0[C0[k] + 1], r1[C0[k]], r1[C0[k] + 1], a.l, b.l), 0.f);
    }
    const size_t o = (nc * Ho + ho) * (size_t)(2 * W) + 4 * t;
    if (skip) {
        const float4 sk = *reinterpret_cast<const float4*>(skip + o);
        v[0] += sk.x; v[1] += sk.y; v[2] += sk.z; v[3] += sk.w;
    }
    *reinterpret_cast<float4*>(y + o) = make_float4(v[0], v[1], v[2], v[3]);
}

int up2x_relu_add_fwd(const float* x, const float* skip, float* y, int NC, int H, int W, hipStream_t s) {
    const size_t total = (size_t)NC * 4 * H * W;
    static const bool vec = true;
    if (vec && (W & 1) == 0 && ((uintptr_t)y & 15) == 0 && (!skip || ((uintptr_t)skip & 15) == 0)) {
        hipLaunchKernelGGL(up2x_relu_add4_kernel, dim3(grid1d(total / 4)), dim3(MEDT_THREADS), 0, s, x, skip, y, H, W, total / 4);
        return launch_status("up2x_relu_add4");
    }
    hipLaunchKernelGGL(up2x_relu_add_kernel, dim3(grid1d(total)), dim3(MEDT_THREADS), 0, s, x, skip, y, H, W, total);
    return launch_status("up2x_relu_add");
}

// dx[h,w] = sum over the (<=16) output pixels whose interpolation touches (h,w) of weight * dy * [up(x) > 0].
// The 16 output pixels (rows 2h-1 .. 2h+2, columns 2w-1 .. 2w+2) interpolate between the 3 x 3 neighbourhood of (h,w):
// rows (h-1,h) for the first two, (h,h+1) for the last two (the border clamps coincide with src_index's), so the
// neighbourhood is loaded once and the 16 mask values come from registers (80 -> 25 loads per lane).
__global__ __launch_bounds__(MEDT_THREADS) void up2x_relu_bwd_kernel(const float* __restrict__ x,
                                                                     const float* __restrict__ dy,
                                                                     float* __restrict__ dx, int H, int W, size_t total) {
    const size_t idx = (size_t)blockIdx.x * MEDT_THREADS + threadIdx.x;
    if (idx >= total) return;
    const int w = (int)(idx % W), h = (int)((idx / W) % H);
    const size_t nc = idx / ((size_t)W * H);
    const float* xp = x + nc * H * W;
    const float* dp = dy + nc * 4 * H * W;
    const int Ho = 2 * H, Wo = 2 * W;
    const int rr[3] = {max(h - 1, 0), h, min(h + 1, H - 1)}, cc[3] = {max(w - 1, 0), w, min(w + 1, W - 1)};
    float xs[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) xs[a][b] = xp[rr[a] * W + cc[b]];
    float lw[4], ww[4];
    bool wok[4];
#pragma unroll
    for (int kw = 0; kw < 4; ++kw) {
        const int wo = 2 * w - 1 + kw;
        const Lerp b = src_index(wo < 0 ? 0 : wo, W);
        lw[kw] = b.l;
        ww[kw] = (b.i0 == w ? 1.f - b.l : 0.f) + (b.i1 == w ? b.l : 0.f);
        wok[kw] = wo >= 0 && wo < Wo && ww[kw] != 0.f;
    }
    float acc = 0.f;
#pragma unroll
    for (int kh = 0; kh < 4; ++kh) {
        const int ho = 2 * h - 1 + kh;
        if (ho < 0 || ho >= Ho) continue;
        const Lerp a = src_index(ho, H);
        const float wh = (a.i0 == h ? 1.f - a.l : 0.f) + (a.i1 == h ? a.l : 0.f);
        if (wh == 0.f) continue;
        constexpr int R0[4] = {0, 0, 1, 1};
        const int r0 = R0[kh];
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
            if (!wok[kw]) continue;
            const int c0 = R0[kw];
            const float up = up_value(xs[r0][c0], xs[r0][c0 + 1], xs[r0 + 1][c0], xs[r0 + 1][c0 + 1], a.l, lw[kw]);
            if (up > 0.f) acc = fmaf(wh * ww[kw], dp[(size_t)ho * Wo + 2 * w - 1 + kw], acc);
        }
    }
    dx[idx] = acc;
}

int up2x_relu_bwd(const float* x, const float* dy, float* dx, int NC, int H, int W, hipStream_t s) {
    const size_t total = (size_t)NC * H * W;
    hipLaunchKernelGGL(up2x_relu_bwd_kernel, dim3(grid1d(total)), dim3(MEDT_THREADS), 0, s, x, dy, dx, H, W, total);
    return launch_status("up2x_relu_bwd");
}

// --------------------------------------------------------------------------- //
// LoGo patches: gather (N,C,S,S) -> (G*G*N, C, P, P) patch-major; merge y = x + x_loc
// --------------------------------------------------------------------------- //
__global__ __launch_bounds__(MEDT_THREADS) void patch_gather_kernel(const float* __restrict__ x, float* __restrict__ xp,
                                                                    int N, int C, int S, int P, int G, size_t total) {
    const size_t idx = (size_t)blockIdx.x * MEDT_THREADS + threadIdx.x;
    if (idx >= total) return;
    const int w = (int)(idx % P), h = (int)((idx / P) % P);
    const int c = (int)((idx / ((size_t)P * P)) % C);
    const int b = (int)(idx / ((size_t)P * P * C));         // b = patch*N + n
    const int patch = b / N, n = b - patch * N;
    const int pi = patch / G, pj = patch - pi * G;
    xp[idx] = x[(((size_t)n * C + c) * S + pi * P + h) * S + pj * P + w];
}

int patch_gather(const float* x, float* xp, int N, int C, int S, int P, int G, hipStream_t s) {
    const size_t total = (size_t)G * G * N * C * P * P;
    hipLaunchKernelGGL(patch_gather_kernel, dim3(grid1d(total)), dim3(MEDT_THREADS), 0, s, x, xp, N, C, S, P, G, total);
    return launch_status("patch_gather");
}

// y = x + x_loc, x_loc = x with the G x G grid of P-px patches (top-left G*P square) overwritten by yp   (:658,:700-702)
__global__ __launch_bounds__(MEDT_THREADS) void logo_merge_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ yp, float* __restrict__ y,
                                                                  int N, int C, int S, int P, int G, size_t total) {
    const size_t idx = (size_t)blockIdx.x * MEDT_THREADS + threadIdx.x;
    if (idx >= total) return;
    const int w = (int)(idx % S), h = (int)((idx / S) % S);
    const int c = (int)((idx / ((size_t)S * S)) % C);
    const int n = (int)(idx / ((size_t)S * S * C));
    const float xv = x[idx];
    float loc = xv;
    if (h < G * P && w < G * P) {
        const int patch = (h / P) * G + (w / P);
        loc = yp[((((size_t)patch * N + n) * C + c) * P + (h % P)) * P + (w % P)];
    }
    y[idx] = xv + loc;
}

int logo_merge_fwd(const float* x, const float* yp, float* y, int N, int C, int S, int P, int G, hipStream_t s) {
    const size_t total = (size_t)N * C * S * S;
    hipLaunchKernelGGL(logo_merge_kernel, dim3(grid1d(total)), dim3(MEDT_THREADS), 0, s, x, yp, y, N, C, S, P, G, total);
    return launch_status("logo_merge");
}

// dx = dy * (inside patches ? 1 : 2);  dyp = gather(dy)
__global__ __launch_bounds__(MEDT_THREADS) void logo_merge_bwd_kernel(const float* __restrict__ dy,
                                                                      float* __restrict__ dx, float* __restrict__ dyp,
                                                                      int N, int C, int S, int P, int G, size_t total) {
    const size_t idx = (size_t)blockIdx.x * MEDT_THREADS + threadIdx.x;
    if (idx >= total) return;
    const int w = (int)(idx % S), h = (int)((idx / S) % S);
    const int c = (int)((idx / ((size_t)S * S)) % C);
    const int n = (int)(idx / ((size_t)S * S * C));
    const float d = dy[idx];
    if (h < G * P && w < G * P) {
        const int patch = (h / P) * G + (w / P);
        dyp[((((size_t)patch * N + n) * C + c) * P + (h % P)) * P + (w % P)] = d;
        dx[idx] = d;
    } else {
        dx[idx] = 2.f * d;
    }
}

int logo_merge_bwd(const float* dy, float* dx, float* dyp, int N, int C, int S, int P, int G, hipStream_t s) {
    const size_t total = (size_t)N * C * S * S;
    hipLaunchKernelGGL(logo_merge_bwd_kernel, dim3(grid1d(total)), dim3(MEDT_THREADS), 0, s, dy, dx, dyp, N, C, S, P, G,
                       total);
    return launch_status("logo_merge_bwd");
}

// --------------------------------------------------------------------------- //
// cross entropy (mean over non-ignored pixels), K classes on dim 1
// --------------------------------------------------------------------------- //
__global__ __launch_bounds__(MEDT_THREADS) void ce_fwd_kernel(const float* __restrict__ logits,
                                                              const int64_t* __restrict__ target,
                                                              float* __restrict__ partials, int K, int HW, int ignore,
                                                              size_t total) {
    MEDT_STATIC_SHARED float red[MEDT_WAVES * 3];
    const size_t idx = (size_t)blockIdx.x * MEDT_THREADS + threadIdx.x;     // over N*HW
    float v[3] = {0.f, 0.f, 0.f};
    if (idx < total) {
        const size_t n = idx / HW, p = idx - n * HW;
        const int64_t t = target[idx];
        if (t != ignore && (t < 0 || t >= K)) v[2] = 1.f;      // F.cross_entropy raises on these: counted, reported
        if (t != ignore && t >= 0 && t < K) {
            const float* lp = logits + n * K * HW + p;
            float m = lp[0];
            for (int k = 1; k < K; ++k) m = fmaxf(m, lp[(size_t)k * HW]);
            float sum = 0.f;
            for (int k = 0; k < K; ++k) sum += __expf(lp[(size_t)k * HW] - m);
            v[0] = m + __logf(sum) - lp[(size_t)t * HW];
            v[1] = 1.f;
        }
    }
    block_sum<3>(v, red, partials + (size_t)blockIdx.x * 3);
}

// loss_out[0] = sum/count, loss_out[1] = count, loss_out[2] = number of out-of-range targets
__global__ __launch_bounds__(64) void ce_finalize_kernel(const float* __restrict__ partials, int nparts,
                                                         float* __restrict__ loss_out) {
    double s = 0.0, c = 0.0, b = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 64) {
        s += (double)partials[3 * p];
        c += (double)partials[3 * p + 1];
        b += (double)partials[3 * p + 2];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); c += __shfl_xor(c, o, 64); b += __shfl_xor(b, o, 64); }
    if (threadIdx.x == 0) {
        loss_out[0] = (float)(s / c);
        loss_out[1] = (float)c;
        loss_out[2] = (float)b;
    }
}

int ce_parts(size_t npix) { return (int)grid1d(npix); }

int ce_fwd(const float* logits, const int64_t* target, float* partials, float* loss_out, int N, int K, int HW, int ignore,
           hipStream_t s) {
    const size_t total = (size_t)N * HW;
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(grid1d(total)), dim3(MEDT_THREADS), 0, s, logits, target, partials, K, HW,
                       ignore, total);
    int rc = launch_status("ce_fwd");
    if (rc) return rc;
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, s, partials, (int)grid1d(total), loss_out);
    return launch_status("ce_finalize");
}

// dlogits[n,k,p] = (softmax_k - [k == t]) * dloss / count
__global__ __launch_bounds__(MEDT_THREADS) void ce_bwd_kernel(const float* __restrict__ logits,
                                                              const int64_t* __restrict__ target,
                                                              const float* __restrict__ loss_out,
                                                              const float* __restrict__ dloss,
                                                              float* __restrict__ dlogits, int K, int HW, int ignore,
                                                              size_t total) {
    const size_t idx = (size_t)blockIdx.x * MEDT_THREADS + threadIdx.x;
    if (idx >= total) return;
    const size_t n = idx / HW, p = idx - n * HW;
    const int64_t t = target[idx];
    const float* lp = logits + n * K * HW + p;
    float* dp = dlogits + n * K * HW + p;
    if (t == ignore || t < 0 || t >= K) {
        for (int k = 0; k < K; ++k) dp[(size_t)k * HW] = 0.f;
        return;
    }
    const float scale = (dloss ? dloss[0] : 1.f) / loss_out[1];
    float m = lp[0];
    for (int k = 1; k < K; ++k) m = fmaxf(m, lp[(size_t)k * HW]);
    float sum = 0.f;
    for (int k = 0; k < K; ++k) sum += __expf(lp[(size_t)k * HW] - m);
    const float inv = 1.f / sum;
    for (int k = 0; k < K; ++k) {
        const float pk = __expf(lp[(size_t)k * HW] - m) * inv;
        dp[(size_t)k * HW] = (pk - (k == t ? 1.f : 0.f)) * scale;
    }
}

int ce_bwd(const float* logits, const int64_t* target, const float* loss_out, const float* dloss, float* dlogits, int N,
           int K, int HW, int ignore, hipStream_t s) {
    const size_t total = (size_t)N * HW;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(grid1d(total)), dim3(MEDT_THREADS), 0, s, logits, target, loss_out, dloss,
                       dlogits, K, HW, ignore, total);
    return launch_status("ce_bwd");
}

// --------------------------------------------------------------------------- //
// AxialAttention_gated_sig (reference lib/models/model_codes.py:279-280, 292-293): the four gates enter through a sigmoid
// --------------------------------------------------------------------------- //
__global__ void gate_sigmoid_fwd_kernel(const float* f_qr, const float* f_kr, const float* f_sve, const float* f_sv,
                                        float* __restrict__ eff) {
    const float* src[4] = {f_qr, f_kr, f_sve, f_sv};
    const int k = threadIdx.x;
    if (k < 4) eff[k] = 1.f / (1.f + expf(-*src[k]));
}
__global__ void gate_sigmoid_bwd_kernel(const float* __restrict__ d_eff, const float* __restrict__ eff,
                                        float* __restrict__ dgate) {
    const int k = threadIdx.x;
    if (k < 4) dgate[k] = d_eff[k] * eff[k] * (1.f - eff[k]);
}
int gate_sigmoid_fwd(const float* f_qr, const float* f_kr, const float* f_sve, const float* f_sv, float* eff, hipStream_t s) {
    hipLaunchKernelGGL(gate_sigmoid_fwd_kernel, dim3(1), dim3(64), 0, s, f_qr, f_kr, f_sve, f_sv, eff);
    return launch_status("gate_sigmoid_fwd");
}
int gate_sigmoid_bwd(const float* d_eff, const float* eff, float* dgate, hipStream_t s) {
    hipLaunchKernelGGL(gate_sigmoid_bwd_kernel, dim3(1), dim3(64), 0, s, d_eff, eff, dgate);
    return launch_status("gate_sigmoid_bwd");
}

// --------------------------------------------------------------------------- //
// AxialAttention_gated_data (reference lib/models/model_codes.py:316-443): four gates PER SEQUENCE from a two-layer MLP
// on the sequence-averaged input (:371-380):  xn = mean_L x;  h = relu(W1 xn + b1);  o = relu(W2 h + b2);  s = sigmoid(o);
// gate columns (qr, kr, sv, sve) = s[:, 0..3] (:376-379, 406-407, 420-421).  One wave per sequence.
// --------------------------------------------------------------------------- //
__global__ __launch_bounds__(64) void gate_mlp_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                          const float* __restrict__ b1, const float* __restrict__ w2,
                                                          const float* __restrict__ b2, float* __restrict__ xn,
                                                          float* __restrict__ h, float* __restrict__ o,
                                                          float* __restrict__ gates, int C, int H, int W, int axis) {
    extern __shared__ float sm[];                      // xn[C] | h[C]
    const int b = blockIdx.x, Bo = axis ? H : W, L = axis ? W : H, n = b / Bo, sq = b - n * Bo, HW = H * W;
    const int pstride = axis ? 1 : W;
    for (int c = threadIdx.x; c < C; c += 64) {
        const float* p = x + ((size_t)n * C + c) * HW + (axis ? sq * W : sq);
        float a = 0.f;
        for (int i = 0; i < L; ++i) a += p[(size_t)i * pstride];
        a *= 1.f / (float)L;
        sm[c] = a;
        xn[(size_t)b * C + c] = a;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 64) {
        float a = b1[c];
        for (int k = 0; k < C; ++k) a = fmaf(w1[(size_t)c * C + k], sm[k], a);
        a = fmaxf(a, 0.f);
        sm[C + c] = a;
        h[(size_t)b * C + c] = a;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        float a = b2[threadIdx.x];
        for (int k = 0; k < C; ++k) a = fmaf(w2[threadIdx.x * C + k], sm[C + k], a);
        a = fmaxf(a, 0.f);
        o[b * 4 + threadIdx.x] = a;
        gates[b * 4 + threadIdx.x] = 1.f / (1.f + expf(-a));
    }
}

int gate_mlp_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* xn, float* h,
                 float* o, float* gates, int N, int C, int H, int W, int axis, hipStream_t s) {
    const int nseq = N * (axis ? H : W);
    hipLaunchKernelGGL(gate_mlp_fwd_kernel, dim3(nseq), dim3(64), 2 * C * sizeof(float), s, x, w1, b1, w2, b2, xn, h, o, gates,
                       C, H, W, axis);
    return launch_status("gate_mlp_fwd");
}

// per sequence: d_o = dgates * s(1-s) * [o > 0];  dh = (W2^T d_o) * [h > 0];  dxn = W1^T dh
__global__ __launch_bounds__(64) void gate_mlp_bwd_seq_kernel(const float* __restrict__ dgates, const float* __restrict__ gates,
                                                              const float* __restrict__ o, const float* __restrict__ h,
                                                              const float* __restrict__ w1, const float* __restrict__ w2,
                                                              float* __restrict__ d_o, float* __restrict__ dh,
                                                              float* __restrict__ dxn, int C) {
    extern __shared__ float sm[];                      // d_o[4] | dh[C]
    const int b = blockIdx.x;
    if (threadIdx.x < 4) {
        const float sg = gates[b * 4 + threadIdx.x];
        const float v = o[b * 4 + threadIdx.x] > 0.f ? dgates[b * 4 + threadIdx.x] * sg * (1.f - sg) : 0.f;
        sm[threadIdx.x] = v;
        d_o[b * 4 + threadIdx.x] = v;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 64) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) a = fmaf(w2[k * C + c], sm[k], a);
        a = h[(size_t)b * C + c] > 0.f ? a : 0.f;
        sm[4 + c] = a;
        dh[(size_t)b * C + c] = a;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 64) {
        float a = 0.f;
        for (int k = 0; k < C; ++k) a = fmaf(w1[(size_t)k * C + c], sm[4 + k], a);
        dxn[(size_t)b * C + c] = a;
    }
}

// parameter gradients (sums over the sequences): block r < C: row r of dW1 and db1[r]; block C + k: row k of dW2, db2[k]
__global__ __launch_bounds__(64) void gate_mlp_bwd_param_kernel(const float* __restrict__ d_o, const float* __restrict__ dh,
                                                                const float* __restrict__ h, const float* __restrict__ xn,
                                                                float* __restrict__ dw1, float* __restrict__ db1,
                                                                float* __restrict__ dw2, float* __restrict__ db2, int C,
                                                                int nseq) {
    const int r = blockIdx.x;
    const bool first = r < C;
    const int row = first ? r : r - C, stride = first ? C : 4;
    const float* lhs = first ? dh : d_o;                 // [b][row]
    const float* rhs = first ? xn : h;                   // [b][c]
    for (int c = threadIdx.x; c < C; c += 64) {
        float a = 0.f;
        for (int b = 0; b < nseq; ++b) a = fmaf(lhs[(size_t)b * stride + row], rhs[(size_t)b * C + c], a);
        (first ? dw1 : dw2)[(size_t)row * C + c] = a;
    }
    if (threadIdx.x == 0) {
        float a = 0.f;
        for (int b = 0; b < nseq; ++b) a += lhs[(size_t)b * stride + row];
        (first ? db1 : db2)[row] = a;
    }
}

// dx[n, c, pos] = dxn[b(n, pos), c] / L   (backward of the sequence mean)
__global__ __launch_bounds__(MEDT_THREADS) void gate_mlp_bwd_dx_kernel(const float* __restrict__ dxn, float* __restrict__ dx,
                                                                       int C, int H, int W, int axis, size_t total) {
    const size_t idx = (size_t)blockIdx.x * MEDT_THREADS + threadIdx.x;
    if (idx >= total) return;
    const int w = (int)(idx % W), hh = (int)((idx / W) % H), c = (int)((idx / ((size_t)W * H)) % C);
    const int n = (int)(idx / ((size_t)W * H * C));
    const int Bo = axis ? H : W, L = axis ? W : H, sq = axis ? hh : w;
    dx[idx] = dxn[((size_t)n * Bo + sq) * C + c] * (1.f / (float)L);
}

int gate_mlp_bwd(const float* dgates, const float* gates, const float* o, const float* h, const float* xn, const float* w1,
                 const float* w2, float* d_o, float* dh, float* dxn, float* dw1, float* db1, float* dw2, float* db2,
                 float* dx, int N, int C, int H, int W, int axis, hipStream_t s) {
    const int nseq = N * (axis ? H : W);
    hipLaunchKernelGGL(gate_mlp_bwd_seq_kernel, dim3(nseq), dim3(64), (4 + C) * sizeof(float), s, dgates, gates, o, h, w1, w2,
                       d_o, dh, dxn, C);
    int rc = launch_status("gate_mlp_bwd_seq");
    if (rc) return rc;
    hipLaunchKernelGGL(gate_mlp_bwd_param_kernel, dim3(C + 4), dim3(64), 0, s, d_o, dh, h, xn, dw1, db1, dw2, db2, C, nseq);
    if ((rc = launch_status("gate_mlp_bwd_param"))) return rc;
    const size_t total = (size_t)N * C * H * W;
    hipLaunchKernelGGL(gate_mlp_bwd_dx_kernel, dim3(grid1d(total)), dim3(MEDT_THREADS), 0, s, dxn, dx, C, H, W, axis, total);
    return launch_status("gate_mlp_bwd_dx");
}

// per-sequence gate gradients: sum over the heads, reorder (f_qr, f_kr, f_sve, f_sv) -> gate-tensor columns (qr, kr, sv, sve)
__global__ __launch_bounds__(MEDT_THREADS) void gate_seq_reduce_kernel(const float* __restrict__ partials,
                                                                       float* __restrict__ dgates, int nseq, int G) {
    const int idx = blockIdx.x * MEDT_THREADS + threadIdx.x;
    if (idx >= nseq * 4) return;
    const int b = idx >> 2, col = idx & 3, k = col == 2 ? 3 : (col == 3 ? 2 : col);
    float a = 0.f;
    for (int g = 0; g < G; ++g) a += partials[((size_t)b * G + g) * 4 + k];
    dgates[idx] = a;
}

int gate_seq_reduce(const float* partials, float* dgates, int nseq, int G, hipStream_t s) {
    hipLaunchKernelGGL(gate_seq_reduce_kernel, dim3(grid1d((size_t)nseq * 4)), dim3(MEDT_THREADS), 0, s, partials, dgates, nseq, G);
    return launch_status("gate_seq_reduce");
}

// --------------------------------------------------------------------------- //
// Adam (torch.optim.Adam semantics, coupled L2 weight decay).  The step counter lives on the device so a
// captured hipGraph replays correctly: adam_tick advances it and derives the bias corrections.
//   state[0] = step, state[1] = 1 - b1^step, state[2] = 1 - b2^step
// --------------------------------------------------------------------------- //
__global__ void adam_tick_kernel(float* __restrict__ state, float b1, float b2) {
    const float t = state[0] + 1.f;
    state[0] = t;
    state[1] = 1.f - powf(b1, t);
    state[2] = 1.f - powf(b2, t);
}

__global__ __launch_bounds__(MEDT_THREADS) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                                 float* __restrict__ m, float* __restrict__ v,
                                                                 const float* __restrict__ state, size_t n, float lr,
                                                                 float b1, float b2, float eps, float wd,
                                                                 float gscale) {
    const size_t i = (size_t)blockIdx.x * MEDT_THREADS + threadIdx.x;
    if (i >= n) return;
    const float bc1 = state[1], bc2 = state[2];
    const float pi = p[i];
    const float gi = fmaf(wd, pi, g[i] * gscale);
    const float mi = fmaf(b1, m[i], (1.f - b1) * gi);
    const float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
}

int adam_step(float* p, const float* g, float* m, float* v, float* state, size_t n, float lr, float b1, float b2,
              float eps, float wd, float gscale, hipStream_t s) {
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, s, state, b1, b2);
    int rc = launch_status("adam_tick");
    if (rc) return rc;
    hipLaunchKernelGGL(adam_step_kernel, dim3(grid1d(n)), dim3(MEDT_THREADS), 0, s, p, g, m, v, state, n, lr, b1, b2, eps,
                       wd, gscale);
    return launch_status("adam_step");
}

// --------------------------------------------------------------------------- //
// Per-image confusion counts of a 2-class segmentation: prediction = logits[:,1] >= threshold (what test.py writes
// as PNG, test.py:131-137), ground truth = target > 0.  counts[n] = {tp, fp, fn, tn} (integer atomics: exact and
// order independent).  Replaces the per-pixel loops of performancemetrics_*.m.
// --------------------------------------------------------------------------- //
__global__ __launch_bounds__(MEDT_THREADS) void seg_counts_kernel(const float* __restrict__ logits,
                                                                  const int64_t* __restrict__ target,
                                                                  int* __restrict__ counts, int K, int HW,
                                                                  float threshold) {
    const int n = blockIdx.y;
    const float* fg = logits + ((size_t)n * K + 1) * HW;
    const int64_t* t = target + (size_t)n * HW;
    int tp = 0, fp = 0, fn = 0, tn = 0;
    for (int p = blockIdx.x * MEDT_THREADS + threadIdx.x; p < HW; p += gridDim.x * MEDT_THREADS) {
        const bool pred = fg[p] >= threshold, gt = t[p] > 0;
        tp += pred && gt;
        fp += pred && !gt;
        fn += !pred && gt;
        tn += !pred && !gt;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        tp += __shfl_xor(tp, o, 64);
        fp += __shfl_xor(fp, o, 64);
        fn += __shfl_xor(fn, o, 64);
        tn += __shfl_xor(tn, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(counts + n * 4 + 0, tp);
        atomicAdd(counts + n * 4 + 1, fp);
        atomicAdd(counts + n * 4 + 2, fn);
        atomicAdd(counts + n * 4 + 3, tn);
    }
}

int seg_counts(const float* logits, const int64_t* target, int* counts, int N, int K, int HW, float threshold,
               hipStream_t s) {
    if (hipMemsetAsync(counts, 0, (size_t)N * 4 * sizeof(int), s) != hipSuccess) {
        set_error("seg_counts: memset failed");
        return MEDT_ELAUNCH;
    }
    const int parts = min(cdiv(HW, MEDT_THREADS), 64);
    hipLaunchKernelGGL(seg_counts_kernel, dim3(parts, N), dim3(MEDT_THREADS), 0, s, logits, target, counts, K, HW, threshold);
    return launch_status("seg_counts");
}

}  // namespace medt
