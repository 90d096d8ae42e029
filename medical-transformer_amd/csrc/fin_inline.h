// fin_inline.h -- BatchNorm finalisation by the CONSUMER: no bn_finalize / bn_bwd_finalize launch between two kernels of a layer.
//
// Inside the networks an attention layer is a chain of ~5-us launches, and six of its fifteen were finalisations: 30 - 70 one-wave
// workgroups that reduce <= 64 partial rows per channel into mean / rstd / scale / shift (forward) or the three backward
// coefficients -- 4.8 us of launch ramp and one dependent load round trip plus the 1.4 us kernel boundary, for a few hundred
// additions.  Finalising in the PRODUCER's last workgroup costs more than it saves on this part (eight XCDs, L2s not coherent
// with each other: profiles/r06_ticket_ubench.txt).  Here the kernel that CONSUMES the statistics re-derives them in its
// prologue from the producer's partial rows -- every workgroup for the channels it is about to use (the load round trip replaces
// the one it spent on the finalised values) -- and ONE designated workgroup per channel set also writes what outlives the launch:
// the saved statistics the backward pass reads, the running statistics, the parameter gradients.  Every workgroup runs the same
// reduction in the same order, so the values used and the values saved are the same bits.
//
// Scope: training mode, one BatchNorm group (the global branch of MedT, the unets), few partial rows (<= 256).  Everything else
// keeps the finalisation launches (pointwise.hip).  MEDT_INLINE_FIN=0 turns the consumer-side path off.
#pragma once
#include "defer.h"

namespace medt {

struct FinSrc {                      // forward: BnFin = partial rows [ppg][CH][2] DOUBLES, the parameters, where the statistics are saved
    BnFin f;
    float momentum, eps;
    int on;                          // 0: the finalised statistics are in memory (a bn_finalize launch ran)
};
struct BfinSrc {                     // backward: BfinJob = partial rows [ppg][CH][2] FLOATS, saved statistics, weight, coef / gradient outputs
    BfinJob j;
    int on;
};

bool inline_fin_enabled();           // MEDT_INLINE_FIN (defer.hip)
inline FinSrc no_fin_src() { FinSrc s; s.f = BnFin{}; s.momentum = 0.f; s.eps = 0.f; s.on = 0; return s; }
inline BfinSrc no_bfin_src() { BfinSrc s; s.j = BfinJob{}; s.on = 0; return s; }
inline bool inline_fin_ok(int training, int groups, int rows) { return inline_fin_enabled() && training == 1 && groups == 1 && rows <= 256; }

struct FinVals { float mean, rstd, scale, shift; double meand, var; };

// (sum, sum of squares) -> the statistics, the arithmetic of bn_finalize_body (pointwise.hip)
__device__ __forceinline__ FinVals fin_vals(double s, double ss, double count, float eps, float g, float b) {
    FinVals v;
    const double mean = s / count;
    double var = ss / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    v.meand = mean; v.var = var;
    v.mean = (float)mean;
    v.rstd = (float)rstd;
    v.scale = (float)(g * rstd);
    v.shift = (float)(b - mean * g * rstd);
    return v;
}

// The designated thread's part: saved statistics, running statistics (one group), batch counter.
__device__ __forceinline__ void fin_save(const FinSrc& s, int ch, const FinVals& v) {
    const BnFin& f = s.f;
    f.out.mean[ch] = v.mean;
    f.out.rstd[ch] = v.rstd;
    f.out.scale[ch] = v.scale;
    f.out.shift[ch] = v.shift;
    if (f.running_mean) f.running_mean[ch] = (float)((1.0 - s.momentum) * (double)f.running_mean[ch] + s.momentum * v.meand);
    if (f.running_var)
        f.running_var[ch] = (float)((1.0 - s.momentum) * (double)f.running_var[ch] + s.momentum * v.var * (f.count / (f.count - 1.0)));
    if (ch == 0 && f.nbt) *f.nbt += 1;
}

// Few rows (bn_similarity: <= 4): every thread sums the rows of channel ch itself, in row order (wave-uniform addresses).
__device__ __forceinline__ FinVals fin_channel_serial(const FinSrc& s, int ch) {
    const double* p = reinterpret_cast<const double*>(s.f.partials);
    double a = 0.0, b = 0.0;
    for (int r = 0; r < s.f.ppg; ++r) {
        a += p[((size_t)r * s.f.CH + ch) * 2];
        b += p[((size_t)r * s.f.CH + ch) * 2 + 1];
    }
    return fin_vals(a, b, s.f.count, s.eps, s.f.weight[ch], s.f.bias[ch]);
}

// The same for up to 64 channels at once: lane l takes channel chs(l) (its own serial row sum and ONE run of the double
// arithmetic); the caller broadcasts what it needs with readlane.
__device__ __forceinline__ FinVals fin_channel_lane(const FinSrc& s, int ch) { return fin_channel_serial(s, ch); }

__device__ __forceinline__ float fin_bcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__device__ __forceinline__ double fin_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Eight lanes per channel (lane = 8 * slot + sub): the rows sub, sub + 8, ... of channel `ch` summed in double, all eight lanes of
// the slot get the result (three DPP steps on the two halves of the double: no LDS crossbar round trips).
template <int CTRL>
__device__ __forceinline__ double fin_dpp_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true), hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return v + __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double fin_sum8(double v) {
    v = fin_dpp_add<0xB1>(v);          // quad_perm [1,0,3,2]
    v = fin_dpp_add<0x4E>(v);          // quad_perm [2,3,0,1]
    v = fin_dpp_add<0x141>(v);         // row_half_mirror: the other quad of the eight
    return v;
}
template <class T>
__device__ __forceinline__ void fin_slot_sums(const T* partials, int rows, int CH, int ch, int sub, double& s0, double& s1) {
    double a = 0.0, b = 0.0;
    for (int r0 = sub; r0 < rows; r0 += 64) {                 // eight rows per lane and trip: sixteen loads in flight
        T x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = min(r0 + 8 * u, rows - 1);
            x[u] = partials[((size_t)r * CH + ch) * 2];
            y[u] = partials[((size_t)r * CH + ch) * 2 + 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (r0 + 8 * u < rows) { a += (double)x[u]; b += (double)y[u]; }
    }
    s0 = fin_sum8(a);
    s1 = fin_sum8(b);
}

// Many rows: one WAVE sums channel ch (lane = row, strided by 64, then the xor tree) -- all of its lanes get the result.
template <class T>
__device__ __forceinline__ void fin_wave_sums(const T* partials, int rows, int CH, int ch, int lane, double& s0, double& s1) {
    double a = 0.0, b = 0.0;
    for (int r = lane; r < rows; r += 64) {
        a += (double)partials[((size_t)r * CH + ch) * 2];
        b += (double)partials[((size_t)r * CH + ch) * 2 + 1];
    }
    s0 = fin_wave_sum_d(a);
    s1 = fin_wave_sum_d(b);
}

// ---- bn_similarity backward (axial_core.hip: sim_bwd_finalize_kernel) --------------------------------------------------------------
// coef[SC][3] = (e, u, w) with dS_x = e * dZ + u * S_x + w
__device__ __forceinline__ void sim_coef(double a0, double sxh, double count, double mean, double rstd, double w,
                                         int training, float* cf) {
    const double e = w * rstd;
    cf[0] = (float)e;
    if (training) {
        const double m1 = a0 / count, m2 = sxh / count;
        const double u = -e * rstd * m2;
        cf[1] = (float)u;
        cf[2] = (float)(-e * m1 - u * mean);
    } else {
        cf[1] = 0.f;
        cf[2] = 0.f;
    }
}

struct SimBSrc {                     // partial rows [rows][G][4] FLOATS = sum dZ * {1, S_qk, S_qr, S_kr} (one BatchNorm group)
    const float* partials;
    int rows, G, SC, training, on;
    double count;
    BnStats ss;
    const float* weight;
    float *coef, *dweight, *dbias;
};
inline SimBSrc no_simb_src() { SimBSrc s{}; s.on = 0; return s; }

__device__ __forceinline__ double fin_bcast_d(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// The three channels (qk, qr, kr) of head hg, by every wave that calls it: slot v = (lane >> 3) & 3 sums value v of the partial rows,
// slots 1..3 run the double arithmetic of their channel, cf[x][0..2] is broadcast to all lanes.  `writer`: this wave also writes the
// coefficients (the relfix kernel behind reads them) and bn_similarity's parameter gradients.
__device__ __forceinline__ void sim_coef_inline(const SimBSrc& s, int hg, int lane, bool writer, float (&cf)[3][3]) {
    const int slot = lane >> 3, sub = lane & 7, vi = slot & 3;
    double a = 0.0;
    for (int r0 = sub; r0 < s.rows; r0 += 64) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = s.partials[((size_t)min(r0 + 8 * u, s.rows - 1) * s.G + hg) * 4 + vi];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (r0 + 8 * u < s.rows) a += (double)t[u];
    }
    a = fin_sum8(a);
    const double a0 = fin_bcast_d(a, 0);
    const int x = vi > 0 ? vi - 1 : 0, ch = x * s.G + hg;
    const double mean = s.ss.mean[ch], rstd = s.ss.rstd[ch];
    const double sxh = rstd * (a - mean * a0);                // sum dZ * xhat
    float c3[3];
    sim_coef(a0, sxh, s.count, mean, rstd, s.weight[ch], s.training, c3);
    if (writer && sub == 0 && slot >= 1 && slot <= 3) {
        s.coef[(size_t)ch * 3] = c3[0]; s.coef[(size_t)ch * 3 + 1] = c3[1]; s.coef[(size_t)ch * 3 + 2] = c3[2];
        s.dweight[ch] = (float)sxh;
        s.dbias[ch] = (float)a0;
    }
#pragma unroll
    for (int xx = 0; xx < 3; ++xx)
#pragma unroll
        for (int k = 0; k < 3; ++k) cf[xx][k] = fin_bcast(c3[k], 8 * (xx + 1));
}

}  // namespace medt
