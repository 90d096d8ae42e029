// pointwise.hip -- channel-mixing and per-channel kernels around the attention core:
//   1x1 convolution (qkv_transform, reference lib/models/utils.py:4-5 + axialnet.py:151) fwd / bwd,
//   BatchNorm statistics finalisation (nn.BatchNorm1d/2d semantics, axialnet.py:116-118),
//   bn_output apply + pair-sum + AvgPool (axialnet.py:179-187) and its backward statistics.
// HBM-bound streaming kernels: lanes run along the contiguous pixel dimension, weights and
// per-channel constants come through the scalar path (wave-uniform addresses).
#include "defer.h"
#include "fin_inline.h"
#include "sim_tables.h"
#include <type_traits>

namespace medt {

// --------------------------------------------------------------------------- //
// 1x1 convolution forward (+ per-channel sum / sum-of-squares partials)
// --------------------------------------------------------------------------- //
template <int OT>
__global__ __launch_bounds__(MEDT_THREADS) void conv1x1_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, float* __restrict__ partials,
    int Cin, int Cout, int HW) {
    MEDT_STATIC_SHARED float red[MEDT_WAVES * OT * 2 * 2];
    const int  p  = blockIdx.x * MEDT_THREADS + threadIdx.x;
    const int  n  = blockIdx.y;
    const int  o0 = blockIdx.z * OT;
    const bool ok = p < HW;
    const float* xp = x + (size_t)n * Cin * HW + (ok ? p : 0);
    float acc[OT];
#pragma unroll
    for (int o = 0; o < OT; ++o) acc[o] = 0.f;
    {
        int c = 0;                                      // lanes past HW read pixel 0 (never stored): no branch around the loads
        for (; c + 8 <= Cin; c += 8) {
            float xr[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) xr[u] = xp[(size_t)(c + u) * HW];
            MEDT_SCHED_FENCE();
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int o = 0; o < OT; ++o) acc[o] = fmaf(w[(o0 + o) * Cin + c + u], xr[u], acc[o]);
            MEDT_SCHED_FENCE();
        }
        for (; c < Cin; ++c) {
            const float xv = xp[(size_t)c * HW];
#pragma unroll
            for (int o = 0; o < OT; ++o) acc[o] = fmaf(w[(o0 + o) * Cin + c], xv, acc[o]);
        }
    }
    if (ok) {
        float* yp = y + ((size_t)n * Cout + o0) * HW + p;
#pragma unroll
        for (int o = 0; o < OT; ++o) yp[(size_t)o * HW] = acc[o];
    }
    if (partials) {
        float v[2 * OT];
#pragma unroll
        for (int o = 0; o < OT; ++o) { v[2 * o] = acc[o]; v[2 * o + 1] = acc[o] * acc[o]; }
        block_sum_d<2 * OT>(v, red, reinterpret_cast<double*>(partials) + ((size_t)(n * gridDim.x + blockIdx.x) * Cout + o0) * 2);
    }
}

int conv1x1_ptiles(int HW) { return cdiv(HW, MEDT_THREADS); }

int conv1x1_fwd(const float* x, const float* w, float* y, float* partials, int N, int Cin, int Cout, int HW,
                hipStream_t s) {
    const int pt = conv1x1_ptiles(HW);
    if (Cout % 16 == 0) {
        hipLaunchKernelGGL(conv1x1_fwd_kernel<16>, dim3(pt, N, Cout / 16), dim3(MEDT_THREADS), 0, s, x, w, y, partials,
                           Cin, Cout, HW);
    } else if (Cout % 8 == 0) {
        hipLaunchKernelGGL(conv1x1_fwd_kernel<8>, dim3(pt, N, Cout / 8), dim3(MEDT_THREADS), 0, s, x, w, y, partials,
                           Cin, Cout, HW);
    } else if (Cout % 2 == 0) {
        hipLaunchKernelGGL(conv1x1_fwd_kernel<2>, dim3(pt, N, Cout / 2), dim3(MEDT_THREADS), 0, s, x, w, y, partials,
                           Cin, Cout, HW);
    } else {
        hipLaunchKernelGGL(conv1x1_fwd_kernel<1>, dim3(pt, N, Cout), dim3(MEDT_THREADS), 0, s, x, w, y, partials, Cin,
                           Cout, HW);
    }
    return launch_status("conv1x1_fwd");
}

// --------------------------------------------------------------------------- //
// 1x1 convolution backward-data, with the BatchNorm-backward affine applied on load
// --------------------------------------------------------------------------- //
template <int CT>
__global__ __launch_bounds__(MEDT_THREADS) void conv1x1_bwd_data_kernel(
    const float* __restrict__ dy, const float* __restrict__ raw, const float* __restrict__ coef,
    const float* __restrict__ w, float* __restrict__ dx, int N, int Cin, int Cout, int HW, int npg) {
    const long q = (long)blockIdx.x * MEDT_THREADS + threadIdx.x;     // flattened (image, pixel)
    const int  c0 = blockIdx.y * CT;
    const bool ok = q < (long)N * HW;
    const int  n = ok ? (int)(q / HW) : 0, p = ok ? (int)(q - (long)n * HW) : 0;
    const size_t base = (size_t)n * Cout * HW + p;
    const float* cf = coef ? coef + (size_t)(n / npg) * Cout * 3 : nullptr;
    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = 0.f;
    // Batches of 8 output channels: all their loads in flight, then the FMAs (MEDT_SCHED_FENCE keeps load / use pairs from
    // being interleaved into one global round trip per channel).  Lanes past the end read position 0 (never stored).
    auto batch = [&](auto u_tag, auto cf_tag, int o) {
        constexpr int UU = decltype(u_tag)::value;
        constexpr bool CF = decltype(cf_tag)::value;
        float dv[UU], rv[UU], k0[UU], k1[UU], k2[UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            dv[u] = dy[base + (size_t)(o + u) * HW];
            if constexpr (CF) {
                rv[u] = raw[base + (size_t)(o + u) * HW];
                k0[u] = cf[(o + u) * 3 + 0];
                k1[u] = cf[(o + u) * 3 + 1];
                k2[u] = cf[(o + u) * 3 + 2];
            }
        }
        MEDT_SCHED_FENCE();
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            float v = dv[u];
            if constexpr (CF) v = fmaf(k0[u], v, fmaf(k1[u], rv[u], k2[u]));
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = fmaf(w[(o + u) * Cin + c0 + c], v, acc[c]);
        }
        MEDT_SCHED_FENCE();
    };
    {
        int o = 0;
        if (cf) {
            for (; o + 8 <= Cout; o += 8) batch(std::integral_constant<int, 8>{}, std::true_type{}, o);
            for (; o < Cout; ++o) batch(std::integral_constant<int, 1>{}, std::true_type{}, o);
        } else {
            for (; o + 8 <= Cout; o += 8) batch(std::integral_constant<int, 8>{}, std::false_type{}, o);
            for (; o < Cout; ++o) batch(std::integral_constant<int, 1>{}, std::false_type{}, o);
        }
    }
    if (ok) {
        float* dp = dx + ((size_t)n * Cin + c0) * HW + p;
#pragma unroll
        for (int c = 0; c < CT; ++c) dp[(size_t)c * HW] = acc[c];
    }
}

// Small-problem variant (deep LoGo layers: <= ~4k positions, 128-256 output channels): the o-contraction is
// split over the workgroup's four waves (64 positions per workgroup) and combined through LDS, so the serial,
// latency-bound chain per lane is 4x shorter.
template <int CT>
__global__ __launch_bounds__(MEDT_THREADS) void conv1x1_bwd_data_ws_kernel(
    const float* __restrict__ dy, const float* __restrict__ raw, const float* __restrict__ coef,
    const float* __restrict__ w, float* __restrict__ dx, int N, int Cin, int Cout, int HW, int npg) {
    MEDT_STATIC_SHARED float red[3][CT][64];
    extern __shared__ __attribute__((aligned(16))) float wl[];       // [Cout][CT]: this workgroup's weight slice
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long q = (long)blockIdx.x * 64 + lane;
    const int  c0 = blockIdx.y * CT;
    for (int e0 = threadIdx.x; e0 < Cout * CT; e0 += 8 * MEDT_THREADS) {          // 8 loads in flight per lane
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = min(e0 + u * MEDT_THREADS, Cout * CT - 1);
            v[u] = w[(e / CT) * Cin + c0 + (e % CT)];
        }
        MEDT_SCHED_FENCE();
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e0 + u * MEDT_THREADS < Cout * CT) wl[e0 + u * MEDT_THREADS] = v[u];
    }
    const bool ok = q < (long)N * HW;
    const int  n = ok ? (int)(q / HW) : 0, p = ok ? (int)(q - (long)n * HW) : 0;
    const size_t base = (size_t)n * Cout * HW + p;
    const float* cf = coef ? coef + (size_t)(n / npg) * Cout * 3 : nullptr;
    const int ob = (Cout * wv) / 4, oe = (Cout * (wv + 1)) / 4;
    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = 0.f;
    // the weights come from LDS, so the only global loads in the loop are dy / raw / coef: 16 channels' worth in flight
    // per round trip (the loop is a chain of round trips, nothing else: <= 4k positions)
    __syncthreads();
    // (batches of 16 output channels, loads first: see conv1x1_bwd_data_kernel)
    auto batch = [&](auto u_tag, auto cf_tag, int o) {
        constexpr int UU = decltype(u_tag)::value;
        constexpr bool CF = decltype(cf_tag)::value;
        float dv[UU], rv[UU], k0[UU], k1[UU], k2[UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            dv[u] = dy[base + (size_t)(o + u) * HW];
            if constexpr (CF) {
                rv[u] = raw[base + (size_t)(o + u) * HW];
                k0[u] = cf[(o + u) * 3 + 0];
                k1[u] = cf[(o + u) * 3 + 1];
                k2[u] = cf[(o + u) * 3 + 2];
            }
        }
        MEDT_SCHED_FENCE();
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            float v = dv[u];
            if constexpr (CF) v = fmaf(k0[u], v, fmaf(k1[u], rv[u], k2[u]));
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = fmaf(wl[(o + u) * CT + c], v, acc[c]);
        }
        MEDT_SCHED_FENCE();
    };
    {
        int o = ob;
        if (cf) {
            for (; o + 16 <= oe; o += 16) batch(std::integral_constant<int, 16>{}, std::true_type{}, o);
            for (; o < oe; ++o) batch(std::integral_constant<int, 1>{}, std::true_type{}, o);
        } else {
            for (; o + 16 <= oe; o += 16) batch(std::integral_constant<int, 16>{}, std::false_type{}, o);
            for (; o < oe; ++o) batch(std::integral_constant<int, 1>{}, std::false_type{}, o);
        }
    }
    if (wv > 0) {
#pragma unroll
        for (int c = 0; c < CT; ++c) red[wv - 1][c][lane] = acc[c];
    }
    __syncthreads();
    if (wv == 0 && ok) {
        float* dp = dx + ((size_t)n * Cin + c0) * HW + p;
#pragma unroll
        for (int c = 0; c < CT; ++c) dp[(size_t)c * HW] = acc[c] + (red[0][c][lane] + red[1][c][lane]) + red[2][c][lane];
    }
}

int conv1x1_bwd_data_impl(const float* dy, const float* raw, const float* coef, const float* w, float* dx, int N, int Cin,
                          int Cout, int HW, int groups, hipStream_t s);
int conv1x1_bwd_data(const float* dy, const float* raw, const float* coef, const float* w, float* dx, int N, int Cin,
                     int Cout, int HW, int groups, hipStream_t s) {
    if (abl_skip(groups > 1 ? "qkv_dgrad_l" : "qkv_dgrad_g")) return MEDT_OK;
    return conv1x1_bwd_data_impl(dy, raw, coef, w, dx, N, Cin, Cout, HW, groups, s);
}
int conv1x1_bwd_data_impl(const float* dy, const float* raw, const float* coef, const float* w, float* dx, int N, int Cin,
                     int Cout, int HW, int groups, hipStream_t s) {
    const int npg = N / groups;
    const unsigned gx = (unsigned)(((long)N * HW + MEDT_THREADS - 1) / MEDT_THREADS);
    if ((long)N * HW <= 4096 && Cout >= 64 && Cout <= 768 && Cin % 8 == 0) {   // small, deep: wave-split contraction
        const unsigned g64 = (unsigned)(((long)N * HW + 63) / 64);
        if (Cin % 16 == 0 && g64 * (Cin / 16) >= 64)
            hipLaunchKernelGGL(conv1x1_bwd_data_ws_kernel<16>, dim3(g64, Cin / 16), dim3(MEDT_THREADS),
                               (size_t)Cout * 16 * sizeof(float), s, dy, raw, coef, w, dx, N, Cin, Cout, HW, npg);
        else
            hipLaunchKernelGGL(conv1x1_bwd_data_ws_kernel<8>, dim3(g64, Cin / 8), dim3(MEDT_THREADS),
                               (size_t)Cout * 8 * sizeof(float), s, dy, raw, coef, w, dx, N, Cin, Cout, HW, npg);
        return launch_status("conv1x1_bwd_data_ws");
    }
    int CT = 16;                                   // fewer channels per lane when the grid would not fill the chip
    while (CT > 1 && (Cin % CT != 0 || (long)gx * (Cin / CT) < 512)) CT >>= 1;
    while (Cin % CT != 0) CT >>= 1;
#define MEDT_LAUNCH_1X1(T)                                                                                       \
    hipLaunchKernelGGL(conv1x1_bwd_data_kernel<T>, dim3(gx, Cin / T), dim3(MEDT_THREADS), 0, s, dy, raw, coef, w, dx, N, \
                       Cin, Cout, HW, npg)
    switch (CT) {
        case 16: MEDT_LAUNCH_1X1(16); break;
        case 8: MEDT_LAUNCH_1X1(8); break;
        case 4: MEDT_LAUNCH_1X1(4); break;
        case 2: MEDT_LAUNCH_1X1(2); break;
        default: MEDT_LAUNCH_1X1(1); break;
    }
#undef MEDT_LAUNCH_1X1
    return launch_status("conv1x1_bwd_data");
}

// --------------------------------------------------------------------------- //
// out[k] = sum_p in[p][k]      (deterministic: fixed order, no atomics)
// --------------------------------------------------------------------------- //
__device__ __forceinline__ void reduce_rows_body(const float* __restrict__ in, int P, int K, float* __restrict__ out,
                                                 int block) {
    MEDT_STATIC_SHARED float red[4][64];
    const int k = block * 64 + (threadIdx.x & 63);
    const int slice = threadIdx.x >> 6;
    float s = 0.f;
    if (k < K) {
        // 16 independent chains, their loads issued together: the jobs with ~1000 rows (the relative-table partial rows of the
        // single-sweep attention backward, one per sweep workgroup) were 65 dependent round trips per lane with four
        float c[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) c[u] = 0.f;
        int p = slice;
        for (; p + 60 < P; p += 64) {
            float t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = in[(size_t)(p + 4 * u) * K + k];
#pragma unroll
            for (int u = 0; u < 16; ++u) c[u] += t[u];
        }
        if (p < P) {                                                // the last, partial batch: clamped loads, then a select
            float t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = in[(size_t)min(p + 4 * u, P - 1) * K + k];
#pragma unroll
            for (int u = 0; u < 16; ++u) c[u] += p + 4 * u < P ? t[u] : 0.f;
        }
#pragma unroll
        for (int w = 8; w > 0; w >>= 1)
#pragma unroll
            for (int u = 0; u < w; ++u) c[u] += c[u + w];
        s = c[0];
    }
    red[slice][threadIdx.x & 63] = s;
    __syncthreads();
    if (slice == 0 && k < K) out[k] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(MEDT_THREADS) void reduce_rows_kernel(const float* __restrict__ in, int P, int K,
                                                                   float* __restrict__ out) {
    reduce_rows_body(in, P, K, out, blockIdx.x);
}

// many reductions, one launch (defer.h)
using RBatch = JobBatch<RJob, 144>;
__global__ __launch_bounds__(MEDT_THREADS) void reduce_rows_grouped_kernel(RBatch b) {
    const int j = find_job(b, blockIdx.x);
    reduce_rows_body(b.job[j].src, b.job[j].P, b.job[j].K, b.job[j].dst, blockIdx.x - b.start[j]);
}

int reduce_rows_grouped(const RJob* jobs, int n, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += 144) {
        RBatch b;
        b.n = n - i0 < 144 ? n - i0 : 144;
        int blocks = 0;
        for (int i = 0; i < b.n; ++i) {
            b.job[i] = jobs[i0 + i];
            b.start[i] = blocks;
            blocks += cdiv(jobs[i0 + i].K, 64);
        }
        b.start[b.n] = blocks;
        hipLaunchKernelGGL(reduce_rows_grouped_kernel, dim3(blocks), dim3(MEDT_THREADS), 0, s, b);
        int rc = launch_status("reduce_rows_grouped");
        if (rc) return rc;
    }
    return MEDT_OK;
}

int reduce_rows(const float* in, int P, int K, float* out, hipStream_t s) {
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(cdiv(K, 64)), dim3(MEDT_THREADS), 0, s, in, P, K, out);
    return launch_status("reduce_rows");
}

// --------------------------------------------------------------------------- //
// BatchNorm statistics finalisation.  One wave per channel; double accumulation.
// --------------------------------------------------------------------------- //

// Per-group sums of NV interleaved partial values.  With `groups` a power of two <= 64 the 64 lanes split into
// groups x (64/groups) slots, every lane streams its slot's parts and a butterfly over the slot bits leaves the
// group total in every lane of that group (lane % groups == g) -- no serial dependent loads over the groups.
template <int NV, class T = float>
__device__ __forceinline__ bool group_sums(const T* __restrict__ partials, int ppg, int groups, int CH, int ch,
                                           int stride, const int (&which)[NV], double (&out)[NV]) {
    const int lane = threadIdx.x;
    if (groups > 64 || (groups & (groups - 1))) return false;
    const int slots = 64 / groups, g = lane % groups, slot = lane / groups;
#pragma unroll
    for (int k = 0; k < NV; ++k) out[k] = 0.0;
    for (int p = slot; p < ppg; p += slots) {
        const T* q = partials + ((size_t)(g * ppg + p) * CH + ch) * stride;
#pragma unroll
        for (int k = 0; k < NV; ++k) out[k] += (double)q[which[k]];
    }
    for (int o = groups; o < 64; o <<= 1)
#pragma unroll
        for (int k = 0; k < NV; ++k) out[k] += __shfl_xor(out[k], o, 64);
    return true;
}

// partials: [group][part][CH][2] DOUBLES (sum x, sum x^2) -- see block_sum_d in medt_common.h
__device__ __forceinline__ void bn_finalize_body(int ch, const float* __restrict__ partials_f, int ppg, int groups, int CH,
                                                 double count, const float* __restrict__ weight,
                                                 const float* __restrict__ bias, float* running_mean,
                                                 float* running_var, int64_t* nbt, float momentum, float eps,
                                                 int training, BnStats out) {
    const int lane = threadIdx.x;
    const double* partials = reinterpret_cast<const double*>(partials_f);
    const float g = weight[ch], b = bias[ch];
    if (!training) {
        const float  mean = running_mean[ch];
        const float  rstd = (float)(1.0 / sqrt((double)running_var[ch] + (double)eps));
        for (int grp = lane; grp < groups; grp += 64) {
            out.mean[grp * CH + ch]  = mean;
            out.rstd[grp * CH + ch]  = rstd;
            out.scale[grp * CH + ch] = g * rstd;
            out.shift[grp * CH + ch] = b - mean * g * rstd;
        }
        return;
    }
    // training == 2: the statistics were saved by the kernel that applied them (bn_fin_apply); only the running
    // statistics and the batch counter are left to do
    const bool save = training != 2;
    double rm = running_mean ? (double)running_mean[ch] : 0.0, rv = running_var ? (double)running_var[ch] : 0.0;
    double sums[2];
    const int which[2] = {0, 1};
    if (group_sums<2, double>(partials, ppg, groups, CH, ch, 2, which, sums)) {
        const double mean = sums[0] / count;
        double var = sums[1] / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        if (lane < groups && save) {
            out.mean[lane * CH + ch]  = (float)mean;
            out.rstd[lane * CH + ch]  = (float)rstd;
            out.scale[lane * CH + ch] = (float)(g * rstd);
            out.shift[lane * CH + ch] = (float)(b - mean * g * rstd);
        }
        for (int grp = 0; grp < groups; ++grp) {            // running-stat recurrence in group (= patch) order
            const double m = __shfl(mean, grp, 64), v = __shfl(var, grp, 64);
            rm = (1.0 - momentum) * rm + momentum * m;
            rv = (1.0 - momentum) * rv + momentum * v * (count / (count - 1.0));
        }
    } else {
        for (int grp = 0; grp < groups; ++grp) {
            double s = 0.0, ss = 0.0;
            for (int p = lane; p < ppg; p += 64) {
                const double* q = partials + ((size_t)(grp * ppg + p) * CH + ch) * 2;
                s += q[0];
                ss += q[1];
            }
            s = wave_sum_d(s);
            ss = wave_sum_d(ss);
            const double mean = s / count;
            double var = ss / count - mean * mean;
            if (var < 0.0) var = 0.0;
            const double rstd = 1.0 / sqrt(var + (double)eps);
            if (lane == 0 && save) {
                out.mean[grp * CH + ch]  = (float)mean;
                out.rstd[grp * CH + ch]  = (float)rstd;
                out.scale[grp * CH + ch] = (float)(g * rstd);
                out.shift[grp * CH + ch] = (float)(b - mean * g * rstd);
            }
            rm = (1.0 - momentum) * rm + momentum * mean;
            rv = (1.0 - momentum) * rv + momentum * var * (count / (count - 1.0));
        }
    }
    if (lane == 0) {
        if (running_mean) running_mean[ch] = (float)rm;
        if (running_var) running_var[ch] = (float)rv;
        if (ch == 0 && nbt) *nbt += groups;
    }
}

__global__ __launch_bounds__(64) void bn_finalize_kernel(const float* __restrict__ partials, int ppg, int groups, int CH,
                                                         double count, const float* __restrict__ weight,
                                                         const float* __restrict__ bias, float* running_mean,
                                                         float* running_var, int64_t* nbt, float momentum, float eps,
                                                         int training, BnStats out, TablesJob tj) {
    if ((int)blockIdx.x >= CH) {         // appended blocks: sliding-window tables of the layer's relative table
        MEDT_STATIC_SHARED float lds[512];
        sim_tables_block(blockIdx.x - CH, tj.relative, tj.tables, tj.HQ, tj.L, lds);
        return;
    }
    bn_finalize_body(blockIdx.x, partials, ppg, groups, CH, count, weight, bias, running_mean, running_var, nbt, momentum,
                     eps, training, out);
}

// Three BatchNorms of one layer in one launch (the fused small-layer forward, axial_small.hip): block -> (BN, channel).
__global__ __launch_bounds__(64) void bn_finalize3_kernel(BnFin a, BnFin b, BnFin c, int groups, float momentum, float eps,
                                                          int training) {
    int ch = blockIdx.x;
    const BnFin* f = &a;
    if (ch >= a.CH) {
        ch -= a.CH;
        f = &b;
        if (ch >= b.CH) { ch -= b.CH; f = &c; }
    }
    bn_finalize_body(ch, f->partials, f->ppg, groups, f->CH, f->count, f->weight, f->bias, f->running_mean, f->running_var,
                     f->nbt, momentum, eps, training, f->out);
}

BnFin make_fin(const float* partials, int ppg, int CH, double count, const medt_bn_ptrs& bn, BnStats out) {
    BnFin f;
    f.partials = partials; f.ppg = ppg; f.CH = CH; f.count = count;
    f.weight = bn.weight; f.bias = bn.bias; f.running_mean = bn.running_mean; f.running_var = bn.running_var;
    f.nbt = bn.num_batches_tracked; f.out = out;
    return f;
}

int bn_finalize3(const float* p0, int CH0, double n0, const medt_bn_ptrs& bn0, BnStats o0,
                 const float* p1, int CH1, double n1, const medt_bn_ptrs& bn1, BnStats o1,
                 const float* p2, int CH2, double n2, const medt_bn_ptrs& bn2, BnStats o2,
                 int ppg, int groups, float momentum, float eps, int training, hipStream_t s) {
    hipLaunchKernelGGL(bn_finalize3_kernel, dim3(CH0 + CH1 + CH2), dim3(64), 0, s, make_fin(p0, ppg, CH0, n0, bn0, o0),
                       make_fin(p1, ppg, CH1, n1, bn1, o1), make_fin(p2, ppg, CH2, n2, bn2, o2), groups, momentum, eps,
                       training);
    return launch_status("bn_finalize3");
}

int bn_finalize(const float* partials, int ppg, int groups, int CH, double count, const medt_bn_ptrs& bn,
                float momentum, float eps, int training, BnStats out, hipStream_t s, const TablesJob* tables) {
    const TablesJob tj = tables ? *tables : TablesJob{nullptr, nullptr, 0, 0, 0};
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(CH + tj.blocks), dim3(64), 0, s, partials, ppg, groups, CH, count, bn.weight,
                       bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, momentum, eps, training, out, tj);
    return launch_status("bn_finalize");
}

// (bn_bwd_coef: medt_common.h)
__device__ __forceinline__ void bn_bwd_finalize_body(int ch, const float* __restrict__ partials, int ppg, int groups,
                                                     int CH, double count, float dscale, BnStats st,
                                                     const float* __restrict__ weight, int training,
                                                     float* __restrict__ coef, float* __restrict__ dweight,
                                                     float* __restrict__ dbias) {
    const int lane = threadIdx.x;
    double dg = 0.0, db = 0.0;
    double sums[2];
    const int which[2] = {0, 1};
    if (group_sums<2>(partials, ppg, groups, CH, ch, 2, which, sums)) {
        const double s1 = sums[0] * dscale, s2 = sums[1] * dscale;
        if (lane < groups)
            bn_bwd_coef(s1, s2, count, dscale, st.mean[lane * CH + ch], st.rstd[lane * CH + ch], weight[ch], training,
                        coef + ((size_t)lane * CH + ch) * 3);
        dg = s2;
        db = s1;
        for (int o = groups >> 1; o > 0; o >>= 1) {          // sum over the groups (lanes 0..groups-1 differ)
            dg += __shfl_xor(dg, o, 64);
            db += __shfl_xor(db, o, 64);
        }
    } else {
        for (int grp = 0; grp < groups; ++grp) {
            double s1 = 0.0, s2 = 0.0;
            for (int p = lane; p < ppg; p += 64) {
                const float* q = partials + ((size_t)(grp * ppg + p) * CH + ch) * 2;
                s1 += (double)q[0];
                s2 += (double)q[1];
            }
            s1 = wave_sum_d(s1) * dscale;
            s2 = wave_sum_d(s2) * dscale;
            dg += s2;
            db += s1;
            if (lane == 0)
                bn_bwd_coef(s1, s2, count, dscale, st.mean[grp * CH + ch], st.rstd[grp * CH + ch], weight[ch], training,
                            coef + ((size_t)grp * CH + ch) * 3);
        }
    }
    if (lane == 0) {
        if (dweight) dweight[ch] = (float)dg;
        if (dbias) dbias[ch] = (float)db;
    }
}

__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const float* __restrict__ partials, int ppg, int groups,
                                                             int CH, double count, float dscale, BnStats st,
                                                             const float* __restrict__ weight, int training,
                                                             float* __restrict__ coef, float* __restrict__ dweight,
                                                             float* __restrict__ dbias) {
    bn_bwd_finalize_body(blockIdx.x, partials, ppg, groups, CH, count, dscale, st, weight, training, coef, dweight, dbias);
}

using BfBatch = JobBatch<BfinJob, 32>;
__global__ __launch_bounds__(64) void bn_bwd_finalize_grouped_kernel(BfBatch b) {
    const int j = find_job(b, blockIdx.x);
    const BfinJob& f = b.job[j];
    bn_bwd_finalize_body(blockIdx.x - b.start[j], f.partials, f.ppg, f.groups, f.CH, f.count, f.dscale, f.st, f.weight,
                         f.training, f.coef, f.dweight, f.dbias);
}

int bn_bwd_finalize_grouped(const BfinJob* jobs, int n, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += 32) {
        BfBatch b;
        b.n = n - i0 < 32 ? n - i0 : 32;
        int blocks = 0;
        for (int i = 0; i < b.n; ++i) { b.job[i] = jobs[i0 + i]; b.start[i] = blocks; blocks += jobs[i0 + i].CH; }
        b.start[b.n] = blocks;
        hipLaunchKernelGGL(bn_bwd_finalize_grouped_kernel, dim3(blocks), dim3(64), 0, s, b);
        int rc = launch_status("bn_bwd_finalize_grouped");
        if (rc) return rc;
    }
    return MEDT_OK;
}

using FBatch = JobBatch<FinJob, 32>;
__global__ __launch_bounds__(64) void bn_finalize_grouped_kernel(FBatch b) {
    const int j = find_job(b, blockIdx.x);
    const FinJob& q = b.job[j];
    bn_finalize_body(blockIdx.x - b.start[j], q.f.partials, q.f.ppg, q.groups, q.f.CH, q.f.count, q.f.weight, q.f.bias,
                     q.f.running_mean, q.f.running_var, q.f.nbt, q.momentum, q.eps, q.training, q.f.out);
}

int bn_finalize_grouped(const FinJob* jobs, int n, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += 32) {
        FBatch b;
        b.n = n - i0 < 32 ? n - i0 : 32;
        int blocks = 0;
        for (int i = 0; i < b.n; ++i) { b.job[i] = jobs[i0 + i]; b.start[i] = blocks; blocks += jobs[i0 + i].f.CH; }
        b.start[b.n] = blocks;
        hipLaunchKernelGGL(bn_finalize_grouped_kernel, dim3(blocks), dim3(64), 0, s, b);
        int rc = launch_status("bn_finalize_grouped");
        if (rc) return rc;
    }
    return MEDT_OK;
}

int bn_bwd_finalize(const float* partials, int ppg, int groups, int CH, double count, float dscale, BnStats st,
                    const float* weight, int training, float* coef, float* dweight, float* dbias, hipStream_t s) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(CH), dim3(64), 0, s, partials, ppg, groups, CH, count, dscale, st,
                       weight, training, coef, dweight, dbias);
    return launch_status("bn_bwd_finalize");
}

// --------------------------------------------------------------------------- //
// bn_output apply + pair-sum + AvgPool2d(stride)        (axialnet.py:179-187)
// --------------------------------------------------------------------------- //
// (BF: the storage type of `stk` at compile time -- with a runtime flag inside ld_act every load sits in a branch of its own)
template <bool BF>
__device__ __forceinline__ void axial_out_fwd_body(const float* __restrict__ stk, BnStats st,
                                                   float* __restrict__ y, int N, int C, int H, int W,
                                                   int OC, int stride, int npg, int relu, const FinSrc& src) {
    constexpr int bf16 = BF ? 1 : 0;
    const int Ho = H / stride, Wo = W / stride;
    const size_t total = (size_t)N * C * Ho * Wo;
    const size_t idx = (size_t)blockIdx.x * MEDT_THREADS + threadIdx.x;
    const int per = OC / C;                     // 2 (sv|sve pair) or 1 (wopos)
    float fsc[2] = {0.f, 0.f}, fsh[2] = {0.f, 0.f};
    if (src.on) {
        // bn_output finalised HERE (fin_inline.h; the launcher guarantees whole workgroups of ONE output channel and one BatchNorm
        // group): wave t sums the partial rows of stacked channel c * per + t; the first workgroup of the channel saves the result
        MEDT_STATIC_SHARED double fsum[4];
        const size_t idx0 = (size_t)blockIdx.x * MEDT_THREADS;
        const int cb = (int)((idx0 / ((size_t)Wo * Ho)) % C);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (wave < per) {
            double a, b;
            fin_wave_sums(reinterpret_cast<const double*>(src.f.partials), src.f.ppg, OC, cb * per + wave, lane, a, b);
            if (lane == 0) { fsum[2 * wave] = a; fsum[2 * wave + 1] = b; }
        }
        // (lane 0 of wave t finishes channel t -- one run of the double arithmetic per workgroup -- and hands scale | shift over)
        MEDT_STATIC_SHARED float fss[4];
        if (wave < per && lane == 0) {
            const int ch = cb * per + wave;
            const FinVals v = fin_vals(fsum[2 * wave], fsum[2 * wave + 1], src.f.count, src.eps, src.f.weight[ch], src.f.bias[ch]);
            fss[2 * wave] = v.scale;
            fss[2 * wave + 1] = v.shift;
            const bool first = idx0 % ((size_t)Wo * Ho) == 0 && idx0 / ((size_t)Wo * Ho * C) == 0;
            if (first) fin_save(src, ch, v);
        }
        __syncthreads();
        for (int t = 0; t < per; ++t) { fsc[t] = fss[2 * t]; fsh[t] = fss[2 * t + 1]; }
    }
    if (idx >= total) return;
    const int wo = (int)(idx % Wo);
    const int ho = (int)((idx / Wo) % Ho);
    const int c  = (int)((idx / ((size_t)Wo * Ho)) % C);
    const int n  = (int)(idx / ((size_t)Wo * Ho * C));
    const int grp = n / npg;
    float acc = 0.f;
    for (int t = 0; t < per; ++t) {
        const int ch = c * per + t;
        const float sc = src.on ? fsc[t] : st.scale[grp * OC + ch], sh = src.on ? fsh[t] : st.shift[grp * OC + ch];
        const size_t srcp = ((size_t)n * OC + ch) * H * W;
        float a = 0.f;
        for (int dh = 0; dh < stride; ++dh)
            for (int dw = 0; dw < stride; ++dw)
                a += fmaf(sc, ld_act(stk, srcp + (size_t)(ho * stride + dh) * W + wo * stride + dw, bf16), sh);
        acc += a;
    }
    acc *= 1.f / (float)(stride * stride);
    y[idx] = relu ? fmaxf(acc, 0.f) : acc;
}

__global__ __launch_bounds__(MEDT_THREADS) void axial_out_fwd_kernel(const float* __restrict__ stk, BnStats st,
                                                                     float* __restrict__ y, int N, int C, int H, int W,
                                                                     int OC, int stride, int npg, int relu, int bf16, FinSrc src) {
    if (bf16) axial_out_fwd_body<true>(stk, st, y, N, C, H, W, OC, stride, npg, relu, src);
    else axial_out_fwd_body<false>(stk, st, y, N, C, H, W, OC, stride, npg, relu, src);
}

bool axial_out_fwd_inlines(const medt_axial_desc& d) {
    const int Ho = d.H / d.stride, Wo = d.W / d.stride, OC = d.has_pos ? 2 * d.C : d.C;
    return d.bn_groups == 1 && (Ho * Wo) % MEDT_THREADS == 0 && OC / d.C <= 2;
}

int axial_out_fwd(const medt_axial_desc& d, const float* stacked, BnStats st, float* y, hipStream_t s, const FinSrc* src) {
    const int OC = d.has_pos ? 2 * d.C : d.C;
    const size_t total = (size_t)d.N * d.C * (d.H / d.stride) * (d.W / d.stride);
    if (src && src->on && !axial_out_fwd_inlines(d)) { set_error("axial_out_fwd: shape cannot finalise bn_output in the kernel"); return MEDT_EINVAL; }
    hipLaunchKernelGGL(axial_out_fwd_kernel, dim3((unsigned)((total + MEDT_THREADS - 1) / MEDT_THREADS)),
                       dim3(MEDT_THREADS), 0, s, stacked, st, y, d.N, d.C, d.H, d.W, OC, d.stride, d.N / d.bn_groups,
                       d.out_relu, d.act_dtype, src ? *src : no_fin_src());
    return launch_status("axial_out_fwd");
}

// partials[group][part][OC][2] = [sum dstk, sum dstk*xhat], dstk = dy at the pooled position (x 1/s^2 later);
// lanes over the flattened (image, pixel) positions of one group and one stacked channel
__global__ __launch_bounds__(MEDT_THREADS) void axial_out_bwd_stats_kernel(const float* __restrict__ stk,
                                                                           const float* __restrict__ dy, BnStats st,
                                                                           float* __restrict__ partials, int C, int H,
                                                                           int W, int OC, int stride, int npg, int bf16, TablesJob tj,
                                                                           const float* __restrict__ ymask) {
    MEDT_STATIC_SHARED float red[MEDT_WAVES * 2];
    // (fin_inline.h: the fix kernel's tables, no launch of their own -- the launch's first tj.blocks workgroups in (y, x) order: round 6,
    //  so that the deep layers fit too: 28 / 88 table blocks for hq = 4 / 8 on a grid of 8 x 256)
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    if (lin < tj.blocks) {
        MEDT_STATIC_SHARED float tl[512];
        sim_tables_block(lin, tj.relative, tj.tables, tj.HQ, tj.L, tl);
    }
    const int HW = H * W, Ho = H / stride, Wo = W / stride;
    const int per_group = npg * HW, ppg = (per_group + MEDT_THREADS - 1) / MEDT_THREADS;
    const int grp = blockIdx.x / ppg, part = blockIdx.x - grp * ppg, ch = blockIdx.y;
    const int q = part * MEDT_THREADS + threadIdx.x;
    const int c = ch / (OC / C);
    float v[2] = {0.f, 0.f};
    if (q < per_group) {
        const int ni = q / HW, p = q - ni * HW, n = grp * npg + ni;
        const int h = p / W, w = p - h * W;
        const int ho = h / stride, wo = w / stride;
        if (ho < Ho && wo < Wo) {
            const size_t di = ((size_t)(n * C + c) * Ho + ho) * Wo + wo;
            float d = dy[di];
            if (ymask) d = ymask[di] > 0.f ? d : 0.f;            // (fused output ReLU: its backward on load, see the sweep)
            const float xh = (ld_act(stk, ((size_t)n * OC + ch) * HW + p, bf16) - st.mean[grp * OC + ch]) * st.rstd[grp * OC + ch];
            v[0] = d;
            v[1] = d * xh;
        }
    }
    block_sum<2>(v, red, partials + ((size_t)blockIdx.x * OC + ch) * 2);
}

bool axial_out_bwd_stats_tables_ok(const medt_axial_desc& d, int blocks, int L) {
#ifdef MEDT_AB_TABLES_X                 // (A/B build: the table blocks in the grid's first row only)
    return blocks <= d.bn_groups * cdiv((d.N / d.bn_groups) * d.H * d.W, MEDT_THREADS) && L <= 128;
#else
    const long OC = d.has_pos ? 2 * d.C : d.C;
    return blocks <= (long)d.bn_groups * cdiv((d.N / d.bn_groups) * d.H * d.W, MEDT_THREADS) * OC && L <= 128;
#endif
}

int axial_out_bwd_stats(const medt_axial_desc& d, const float* stacked, const float* dy, BnStats st, float* partials,
                        hipStream_t s, const TablesJob* tjp, const float* ymask) {
    const int OC = d.has_pos ? 2 * d.C : d.C;
    const int npg = d.N / d.bn_groups, ppg = cdiv(npg * d.H * d.W, MEDT_THREADS);
    const TablesJob tj = tjp ? *tjp : TablesJob{nullptr, nullptr, 0, 0, 0};
    if (tj.blocks && !axial_out_bwd_stats_tables_ok(d, tj.blocks, tj.L)) { set_error("axial_out_bwd_stats: no room for the table blocks"); return MEDT_EINVAL; }
    hipLaunchKernelGGL(axial_out_bwd_stats_kernel, dim3(d.bn_groups * ppg, OC), dim3(MEDT_THREADS), 0, s, stacked, dy, st,
                       partials, d.C, d.H, d.W, OC, d.stride, npg, d.act_dtype, tj, ymask);
    return launch_status("axial_out_bwd_stats");
}

__global__ __launch_bounds__(MEDT_THREADS) void bn_bwd_apply_raw_bf16_kernel(float* __restrict__ d,
                                                                             const unsigned short* __restrict__ raw,
                                                                             const float* __restrict__ coef, int CH, int HW,
                                                                             int npg, size_t total) {
    const size_t idx = (size_t)blockIdx.x * MEDT_THREADS + threadIdx.x;
    if (idx >= total) return;
    const int ch = (int)((idx / HW) % CH), n = (int)(idx / ((size_t)HW * CH));
    const float* cf = coef + ((size_t)(n / npg) * CH + ch) * 3;
    d[idx] = fmaf(cf[0], d[idx], fmaf(cf[1], bf16_bits_to_f32(raw[idx]), cf[2]));
}

// bf16 storage (BASELINE configs[1]), round 5: bn_bwd_finalize AND the application of bn_qkv's backward to dqkv in ONE launch (the
// bn_fin_apply pattern of the forward): every workgroup re-derives the coefficients of ITS (BatchNorm group, channel) from the
// partial rows (one wave, double accumulation, bn_bwd_coef as in the finalize kernel), applies them to its slice of the plane
//     d <- coef0 * d + coef1 * raw(bf16) + coef2
// and the first workgroup of every channel also does the finalize kernel's job (coefficients of all groups, parameter gradients).
// Round 4 ran bn_bwd_finalize and bn_bwd_apply_raw_bf16 as two launches per attention layer: bf16 storage was 4-5 % SLOWER than
// fp32 (one more dependent launch per layer on the chain); with this kernel both storage types have the same launch count.
#define MEDT_BFA_PPT 4
__global__ __launch_bounds__(MEDT_THREADS) void bn_bwd_fin_apply_bf16_kernel(
    const float* __restrict__ partials, int ppg, int groups, int CH, double count, float dscale, BnStats st,
    const float* __restrict__ weight, int training, float* __restrict__ coef, float* __restrict__ dweight,
    float* __restrict__ dbias, float* __restrict__ d, const unsigned short* __restrict__ raw, int HW, int npg) {
    MEDT_STATIC_SHARED float cf[4];
    const int ch = blockIdx.y % CH, n = blockIdx.y / CH, grp = n / npg;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        double s1 = 0.0, s2 = 0.0;
        for (int p = lane; p < ppg; p += 64) {
            const float* q = partials + ((size_t)(grp * ppg + p) * CH + ch) * 2;
            s1 += (double)q[0];
            s2 += (double)q[1];
        }
        s1 = wave_sum_d(s1) * dscale;
        s2 = wave_sum_d(s2) * dscale;
        if (lane == 0)
            bn_bwd_coef(s1, s2, count, dscale, st.mean[grp * CH + ch], st.rstd[grp * CH + ch], weight[ch], training, cf);
    }
    __syncthreads();
    const float c0 = cf[0], c1 = cf[1], c2 = cf[2];
    const size_t base = (size_t)blockIdx.y * HW;
#pragma unroll
    for (int k = 0; k < MEDT_BFA_PPT; ++k) {
        const int i = (blockIdx.x * MEDT_BFA_PPT + k) * MEDT_THREADS + threadIdx.x;
        if (i < HW) d[base + i] = fmaf(c0, d[base + i], fmaf(c1, bf16_bits_to_f32(raw[base + i]), c2));
    }
    // the finalize kernel's own outputs, once per channel (the weight-gradient jobs of other storage modes read `coef`; the
    // parameter gradients are this layer's bn_qkv.weight / .bias gradients)
    if (n == 0 && blockIdx.x == 0 && threadIdx.x < 64)
        bn_bwd_finalize_body(ch, partials, ppg, groups, CH, count, dscale, st, weight, training, coef, dweight, dbias);
}

int bn_bwd_fin_apply_bf16(const float* partials, int ppg, int groups, int CH, double count, float dscale, BnStats st,
                          const float* weight, int training, float* coef, float* dweight, float* dbias, float* d,
                          const float* raw_bf16, int N, int HW, hipStream_t s) {
    if ((long)N * CH > 65535) { set_error("bn_bwd_fin_apply_bf16: %d x %d planes exceed the grid", N, CH); return MEDT_EUNSUPPORTED; }
    hipLaunchKernelGGL(bn_bwd_fin_apply_bf16_kernel, dim3(cdiv(HW, MEDT_THREADS * MEDT_BFA_PPT), N * CH), dim3(MEDT_THREADS), 0, s,
                       partials, ppg, groups, CH, count, dscale, st, weight, training, coef, dweight, dbias, d,
                       reinterpret_cast<const unsigned short*>(raw_bf16), HW, N / groups);
    return launch_status("bn_bwd_fin_apply_bf16");
}

int bn_bwd_apply_raw_bf16(float* d, const float* raw_bf16, const float* coef, int N, int CH, int HW, int groups, hipStream_t s) {
    const size_t total = (size_t)N * CH * HW;
    hipLaunchKernelGGL(bn_bwd_apply_raw_bf16_kernel, dim3((unsigned)((total + MEDT_THREADS - 1) / MEDT_THREADS)),
                       dim3(MEDT_THREADS), 0, s, d, reinterpret_cast<const unsigned short*>(raw_bf16), coef, CH, HW,
                       N / groups, total);
    return launch_status("bn_bwd_apply_raw_bf16");
}

}  // namespace medt
