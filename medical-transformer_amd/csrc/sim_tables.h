// sim_tables.h -- sliding-window sums of an attention layer's relative table (closed-form bn_similarity statistics,
// axial_stats.hip).  Device-inline so the work can ride as extra blocks of another launch (bn_finalize of bn_qkv).
#pragma once
#include "medt_common.h"

namespace medt {

// tables[(side*L + i)*NR + r]: r < HQ: U_r[i] = sum_{d=i}^{i+L-1} R[r,d];  r >= HQ: pair (a <= b) in row-major order,
// T_ab[i] = sum_d R[a,d] R[b,d], off-diagonal pairs carry the factor 2 of the symmetric double sum.
// side 0 = q rows of `relative`, 1 = k rows.  One block per (side, r); lds: 2L-1 floats.
__device__ __forceinline__ void sim_tables_block(int block, const float* __restrict__ relative, float* __restrict__ tables, int HQ, int L,
                                 float* lds /* 2*(2L-1) floats */) {
    const int NR = HQ + HQ * (HQ + 1) / 2, TL = 2 * L - 1;
    const int side = block / NR, r = block - side * NR;
    int a = r, b = -1;
    if (r >= HQ) {
        int p = r - HQ;
        a = 0;
        while (p >= HQ - a) { p -= HQ - a; ++a; }
        b = a + p;
    }
    const float* Ra = relative + (size_t)(side * HQ + a) * TL;
    const float* Rb = b >= 0 ? relative + (size_t)(side * HQ + b) * TL : nullptr;
    for (int d = threadIdx.x; d < TL; d += blockDim.x) lds[d] = Ra[d] * (Rb ? Rb[d] : 1.f);   // products, once
    __syncthreads();
    const double w = (b > a) ? 2.0 : 1.0;
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        double acc = 0.0;
        for (int d = i; d < i + L; ++d) acc += (double)lds[d];
        tables[((size_t)side * L + i) * NR + r] = (float)(w * acc);
    }
}


}  // namespace medt
