// axial_bwd.hip -- the attention backward of the position-encoded layers as ONE L x L sweep.
//
// Replaces the autograd of reference lib/models/axialnet.py:155-178.  In training mode that
// backward has a global barrier in the middle: bn_similarity's backward needs mean(dY) and mean(dY * xhat) over
// (B*, L, L) before dq / dk / the table gradients can be formed, which is why the generic kernels of axial_core.hip
// recompute the L x L softmax twice (pass A: the two means; pass B: the gradients).  Here (algebra spelled out and
// checked against autograd in tests/test_bwd_algebra.py):
//
//     dS_x = e_x dZ + u_x S_x + w_x       x in {qk, qr, kr};  e_x = gamma_x rstd_x is known BEFORE the sweep
//
//   * the sweep accumulates everything that is linear in dZ -- row sums Dqk = sum_j dZ k_j, Dqr = sum_j dZ Rq[i-j],
//     column sums Eqk = sum_i dZ q_i, Dkr = sum_i dZ Rk[j-i], dv, and the table gradients along the diagonals;
//   * the barrier quantities fall out of those accumulators: sum dZ S_qk = q . Dqk, sum dZ S_qr = f_qr q . Dqr,
//     sum dZ S_kr = f_kr k . Dkr, sum dZ = 0 -- no second L x L pass;
//   * the u / w terms do not involve dZ: they are closed forms in per-sequence Gram matrices of q and k and sliding-window
//     sums of the relative table (attn_bwd_fix_kernel: one elementwise pass over dq | dk; attn_bwd_relfix_kernel: the
//     table and gate gradients from per-position Gram sums).  In running-statistics mode u = w = 0 and the sweep is final.
//
// Work decomposition of the sweep (gfx950, wave64), chosen so that NO operand of an (i, j) pair is fetched in the inner
// loop and the table gradients need no atomics (bit-reproducible):
//   * LS lanes share one sequence, every lane owns D = L / LS key columns j: k, v and the column accumulators
//     (Eqk, Dkr, dv) are lane-private registers for the whole sweep;
//   * all lanes of a sequence walk the query rows i = 0 .. L-1 together: the row's q | dsv | dse | lse | delta record is
//     one LDS broadcast per row, the row accumulators (Dqk, Dqr) are all-reduced over the LS lanes with DPP once per row;
//   * the pair (i, j) uses table entry d = i - j + L - 1.  When i advances, every column's d advances by one: the
//     D table operands AND their gradient accumulators shift one slot, the last slot moves to the next lane
//     (DPP row_shr:1) -- a chain through the LS lanes that a diagonal enters at lane 0 (fresh entry from LDS) and
//     leaves at the last lane, complete, after having met every row it intersects.  The slots are renamed at compile
//     time (the loop is unrolled by D), so a shift costs one DPP move per register per row.
// Per (i, j) pair that is 18 FMAs + 1 exp2 (gp = 2) against ~7 instructions of per-row bookkeeping amortised over D pairs.
#include "axial_tiles.h"
#include "sim_tables.h"
#include "defer.h"
#include "fin_inline.h"
#include <stdlib.h>
#include <type_traits>

#ifndef MEDT_ABL
#define MEDT_ABL 0            // timing experiments only (scripts/r3_ablate.sh): 0 = the real kernel
#endif

namespace medt {

namespace {

template <int GP, int L_, int LS_>
struct Sw {
    static constexpr int HQ = GP / 2, NCH = 2 * GP, L = L_, LS = LS_, D = L / LS, SPW = 64 / LS, TL = 2 * L - 1;
    static constexpr int NT = 2 * HQ + GP;                   // diagonal record: Rq | Rk(reversed) | Rv
    static constexpr int TREC = (NT + 3) & ~3;
    static constexpr int RREC = (HQ + 2 * GP + 2 + 3) & ~3;  // row record: q | dsv | dse | lse | delta
    static constexpr int CREC = TREC;                        // column record in: k | v ; out: dq | dk | dv
    static constexpr int NP = HQ * (HQ + 1) / 2;
    static constexpr int NPG = 2 * (NP + HQ);                // Gq pairs (a <= b) | Sq | Gk pairs | Sk
    // row records of one sequence: + 8 floats so that the records the sequences of a wave read together (same row, one
    // broadcast address per sequence) fall into different LDS banks
    static constexpr int RS = L * RREC + 8;
    static_assert(L % LS == 0 && (LS == 8 || LS == 16 || LS == 32), "lanes per sequence");
};

typedef medt_f4 f4;

template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ float dpp(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROWMASK, 0xf, false));
}
// every lane has a valid source (rotations / permutations inside a row): bound_ctrl lets the compiler fold the move into
// the consuming VOP2 (v_add_f32_dpp), one instruction per reduction step
template <int CTRL>
__device__ __forceinline__ float dppv(float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(src), CTRL, 0xf, 0xf, true));
}

// sum over the LS lanes of a sequence, result in all of them
template <int LS>
__device__ __forceinline__ float seq_allsum(float v) {
    if (LS >= 16) {
        v += dppv<0x128>(v);           // row_ror:8
        v += dppv<0x124>(v);           // row_ror:4
        v += dppv<0x122>(v);           // row_ror:2
        v += dppv<0x121>(v);           // row_ror:1
        if (LS == 32) {                // a sequence = two 16-lane rows: rows 1 <-> 0 and 3 <-> 2 trade places (gfx950 lane swap)
            auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
            v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
    } else {                           // LS == 8: the two halves of a 16-lane row are different sequences
        v += dppv<0xB1>(v);            // quad_perm [1,0,3,2]
        v += dppv<0x4E>(v);            // quad_perm [2,3,0,1]
        v += dppv<0x141>(v);           // row_half_mirror: the other quad of the half row (all its lanes hold the quad sum)
    }
    return v;
}

// x <- the value of the previous lane of the sequence; the first lane of every sequence takes `fresh`
template <int LS>
__device__ __forceinline__ float chain_shift(float x, float fresh, bool head) {
    if (LS == 32) {                    // the chain crosses the row boundary inside a sequence: wave_shr:1 (lane 0 keeps `fresh`)
        const float r = dpp<0x138>(fresh, x);
        return head ? fresh : r;
    }
    float r = dpp<0x111>(fresh, x);    // row_shr:1 (lane 0 of each 16-lane row keeps `fresh`)
    if (LS < 16) r = head ? fresh : r;
    return r;
}
// the same with fresh = 0 (accumulators): bound_ctrl zero-fills lane 0 of the row, no `old` operand to set up
template <int LS>
__device__ __forceinline__ float chain_shift0(float x, bool head) {
    if (LS == 32) {
        const float r = dppv<0x138>(x);                      // wave_shr:1, lane 0 zero-filled
        return head ? 0.f : r;
    }
    float r = dppv<0x111>(x);
    if (LS < 16) r = head ? 0.f : r;
    return r;
}

// lane-wise sum over the sequences of the wave (lane b of every sequence), result in all lanes
template <int LS>
__device__ __forceinline__ float seqs_sum(float v) {
    if (LS == 8) v += dppv<0x128>(v);                        // row_ror:8
    // gfx950 lane swaps (VALU, no LDS crossbar round trip): row 1 <-> row 0 / row 3 <-> row 2, then the wave halves
    if (LS < 32) {                                           // (LS == 32: the two sequences of the wave are its halves)
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(h[0]) + __uint_as_float(h[1]);
}

// The parked records are WAVE-PRIVATE: written by the tail lanes of a wave's own sequences into those sequences' column
// records and read back (the fold after the row loop) by lanes of the same wave -- same-wave LDS operations complete in order,
// no barrier or s_waitcnt is needed between the store and that read.  A consumer in another wave would need both.
// v[k] -> LDS word (addr + k * STRIDE_BYTES), k < 4, from the lanes in `mask` only.  EXEC is narrowed inside the statement, so
// there is no branch: the row steps of an iteration stay one scheduling region, and the LDS pipe sees 4 lanes instead of 64.
template <int STRIDE_BYTES>
__device__ __forceinline__ void lds_store4_masked(unsigned addr, float v0, float v1, float v2, float v3, unsigned long long mask) {
#ifdef MEDT_LANE_EMU              // (CPU lane emulator: the same stores, lane by lane; addr is a byte offset into the LDS array)
    if ((mask >> (threadIdx.x & 63)) & 1) {
        extern __shared__ __attribute__((aligned(16))) float smem[];
        float* p = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + addr);
        p[0] = v0; p[STRIDE_BYTES / 4] = v1; p[2 * STRIDE_BYTES / 4] = v2; p[3 * STRIDE_BYTES / 4] = v3;
    }
    return;
#endif
    unsigned long long save;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_and_b64 exec, exec, %1\n\t"
                 "ds_write_b32 %2, %3 offset:%7\n\t"
                 "ds_write_b32 %2, %4 offset:%8\n\t"
                 "ds_write_b32 %2, %5 offset:%9\n\t"
                 "ds_write_b32 %2, %6 offset:%10\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(save)
                 : "s"(mask), "v"(addr), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "n"(0), "n"(STRIDE_BYTES), "n"(2 * STRIDE_BYTES),
                   "n"(3 * STRIDE_BYTES)
                 : "scc", "memory");                              // (LDS stores the compiler must not move loads across)
}

struct SweepArgs {
    AxialGeom g;
    const float *qkv_raw, *stacked, *lse, *dy, *relative, *out_coef;
    BnStats qs, ss;
    GatePtrs gates;
    int pool;
    float *dqkv, *part_qb, *part_sb, *rel_part, *pg_part, *gram, *gate_raw;
    float* raw32;                  // bf16 storage only: qkv_raw widened to float32 once, for the 1x1 dgrad / wgrad behind bn_qkv's backward
    int tiles, nparts;             // tiles of S_T sequences per BN group; workgroups per BN group
    int qb_rpg;                    // rows per BN group of part_qb (the sweep's nparts rows first, then the fix kernel's)
    BfinSrc ob;                    // on: bn_output's backward coefficients are derived here from axial_out_bwd_stats' partial rows (fin_inline.h)
    const float* ymask;            // layers with a fused output ReLU: the layer's output y -- dy counts where y > 0 (no relu_mask launch in front)
};

// Sum K per-thread values over the workgroup (nw waves, fixed order); thread k < K stores value k to out[k].
template <int K>
__device__ __forceinline__ void wg_sum(float (&v)[K], float* red, float* out, int nw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float s = wave_sum(v[k]);
        if (lane == 0) red[wave * K + k] = s;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        float s = 0.f;
        for (int w = 0; w < nw; ++w) s += red[w * K + threadIdx.x];
        out[threadIdx.x] = s;
    }
}

template <int GP, int L, int LS, bool GATES>
__global__ __launch_bounds__(MEDT_THREADS, (L > 64 || GP >= 8) ? 1 : 2) void attn_bwd_sweep_kernel(SweepArgs a) {
    using C = Sw<GP, L, LS>;
    constexpr int HQ = C::HQ, NCH = C::NCH, D = C::D, SPW = C::SPW, TL = C::TL, NT = C::NT, TREC = C::TREC,
                  RREC = C::RREC, CREC = C::CREC, NP = C::NP, NPG = C::NPG, RS = C::RS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const AxialGeom& g = a.g;
    const int nw = blockDim.x >> 6, S_T = nw * SPW, nthreads = blockDim.x;
    float* rowrec = smem;                                     // [S_T][RS]         row records, RREC floats per position
    float* colrec = rowrec + S_T * RS;                        // [S_T][L][CREC]    (in: k | v ; out: dq | dk | dv)
    float* tab = colrec + S_T * L * CREC;                     // [TL + 1][TREC]
    // per wave: table-gradient accumulators [2][NT][D][LS] (low half: d = j, high half: d = 2L-2-j; lane-contiguous) and
    // per-position Gram sums [NPG][D][LS]
    float* wacc = tab + (TL + 1) * TREC;
    float* pg = wacc + nw * 2 * NT * L;
    float* red = pg + nw * L * NPG;                           // [4 * 32]
    const int grp = blockIdx.x / a.nparts, part = blockIdx.x - grp * a.nparts, hg = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sg = lane / LS, cb = lane % LS;                 // sequence of the wave, column block of the sequence
    const bool head = cb == 0;
    const unsigned long long tail_mask = LS == 32 ? 0x8000000080000000ull                              // last lane of every sequence
                                         : (LS == 16 ? 0x8000800080008000ull : 0x8080808080808080ull);
    const float f_qr = gate(a.gates.f_qr), f_kr = gate(a.gates.f_kr), f_sve = gate(a.gates.f_sve), f_sv = gate(a.gates.f_sv);
    const float e_qk = a.ss.scale[grp * g.SC + hg], e_qr = a.ss.scale[grp * g.SC + g.G + hg],
                e_kr = a.ss.scale[grp * g.SC + 2 * g.G + hg];
    const float s_qk = e_qk * MEDT_LOG2E, s_qr = e_qr * f_qr * MEDT_LOG2E, s_kr = e_kr * f_kr * MEDT_LOG2E;
    float sc[NCH], sh[NCH], vmean[GP], vrstd[GP], cf[NCH][3];
    // sums over everything this workgroup sees
    float T_qk = 0.f, T_qr = 0.f, T_kr = 0.f, g_pe = 0.f, g_pv = 0.f;
    float vst[2 * GP];                                        // bn_qkv backward partials of the v channels
#pragma unroll
    for (int k = 0; k < 2 * GP; ++k) vst[k] = 0.f;
    const int Ho = g.H / a.pool, Wo = g.W / a.pool;
    // uniform bases of the per-lane 32-bit offsets (saddr + voffset addressing: no 64-bit address arithmetic per load)
    const size_t es = g.bf16 ? 2 : 4;
    const char* qbase = reinterpret_cast<const char*>(a.qkv_raw) + (size_t)hg * NCH * g.HW * es;
    const char* sbase = reinterpret_cast<const char*>(a.stacked) + (size_t)hg * NCH * g.HW * es;
    const float* dybase = a.dy + (size_t)hg * GP * Ho * Wo;
    const float* lbase = a.lse + (size_t)hg * g.HW;
    float* obase = a.dqkv + (size_t)hg * NCH * g.HW;
    // (sequence, position) of element kk of this thread in a tile of nseq sequences; false: idle slot of a ragged tile
    auto locate = [&](int kk, int seq0, int nseq, int& ls, int& i, int& n, int& h, int& w) -> bool {
        const int e = threadIdx.x + kk * nthreads;
        const bool ok = e < nseq * L;
        if (g.axis == 1 || !ok) { ls = e / L; i = e % L; } else { i = e / nseq; ls = e - i * nseq; }
        const int b = grp * g.spg + seq0 + (ok ? ls : 0);
        n = b / g.Bo;
        const int sq = b - n * g.Bo;
        h = g.axis == 1 ? sq : i;
        w = g.axis == 1 ? i : sq;
        return ok;
    };
    // Tile staging in two halves: `issue` = the tile's global loads into registers (unconditional: clamped offsets), `commit`
    // = bn_qkv / bn_output-backward arithmetic + the LDS records.  With TPF the loads of tile u+1 are issued before the sweep
    // of tile u and committed after it: their HBM latency hides under the sweep (44 registers at gp = 2; gp = 4 has no room).
    constexpr bool QA_REC = RREC - (HQ + 2 * GP + 2) >= HQ;   // room in the row record for q * s_qk (one multiply less per row)
    // (measured on the C=16 L=64 B*=16384 shape: 0.87 ms with the tile prefetch against 0.72 ms without -- kept off)
    constexpr bool TPF = false && GP == 2 && D <= 4;
    // (round 6) the wide instances (the launches inside the networks) keep the tile's raw v values in registers for bn_qkv's backward
    // sums at the end instead of reading them again: one memory round trip less in the epilogue of a one-tile workgroup
    constexpr bool KEEP_V = !TPF && LS == 32 && MEDT_ABL != 31;
    struct TileRegs { float raw[D][NCH], stk[D][NCH], dyv[D][GP], lse[D]; };
    auto issue_t = [&](int tile_, TileRegs& r, auto bf) {
        constexpr bool BF = decltype(bf)::value;              // (compile-time storage type: no branch per load)
        constexpr unsigned ES = BF ? 2u : 4u;
        const int seq0_ = tile_ * S_T, nseq_ = min(S_T, g.spg - seq0_);
#pragma unroll
        for (int kk = 0; kk < D; ++kk) {
            int ls, i, n, h, w;
            const bool ok = locate(kk, seq0_, nseq_, ls, i, n, h, w);
            const int pix = ok ? h * g.W + w : 0;
            const unsigned qoff = ((unsigned)n * 2u * g.C * g.HW + pix) * ES;
            const int ho = min(h / a.pool, Ho - 1), wo = min(w / a.pool, Wo - 1);
            const unsigned doff = ok ? (((unsigned)n * g.C * Ho + ho) * Wo + wo) * 4u : 0u;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const unsigned o = qoff + (unsigned)(ch * g.HW) * ES;
                if (BF) {
                    r.raw[kk][ch] = bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(qbase + o));
                    r.stk[kk][ch] = bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(sbase + o));
                } else {
                    r.raw[kk][ch] = *reinterpret_cast<const float*>(qbase + o);
                    r.stk[kk][ch] = *reinterpret_cast<const float*>(sbase + o);               // OC == 2C: same layout
                }
            }
#pragma unroll
            for (int c = 0; c < GP; ++c)
                r.dyv[kk][c] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(dybase) + doff + (unsigned)(c * Ho * Wo) * 4u);
            r.lse[kk] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(lbase) + ((unsigned)n * g.G * g.HW + pix) * 4u);
        }
        if (a.ymask) {                // (wave-uniform) the output ReLU's backward on load: one more load per dy element, same offsets
            const char* ybase = reinterpret_cast<const char*>(a.ymask + (size_t)hg * GP * Ho * Wo);
            float ym[D][GP];
#pragma unroll
            for (int kk = 0; kk < D; ++kk) {
                int ls, i, n, h, w;
                const bool ok = locate(kk, seq0_, nseq_, ls, i, n, h, w);
                const int ho = min(h / a.pool, Ho - 1), wo = min(w / a.pool, Wo - 1);
                const unsigned doff = ok ? (((unsigned)n * g.C * Ho + ho) * Wo + wo) * 4u : 0u;
#pragma unroll
                for (int c = 0; c < GP; ++c) ym[kk][c] = *reinterpret_cast<const float*>(ybase + doff + (unsigned)(c * Ho * Wo) * 4u);
            }
#pragma unroll
            for (int kk = 0; kk < D; ++kk)
#pragma unroll
                for (int c = 0; c < GP; ++c) r.dyv[kk][c] = ym[kk][c] > 0.f ? r.dyv[kk][c] : 0.f;
        }
    };
    auto issue = [&](int tile_, TileRegs& r) {
        if (g.bf16) issue_t(tile_, r, std::true_type{}); else issue_t(tile_, r, std::false_type{});
    };
    auto commit = [&](int tile_, const TileRegs& r) {
        const int seq0_ = tile_ * S_T, nseq_ = min(S_T, g.spg - seq0_);
#pragma unroll
        for (int kk = 0; kk < D; ++kk) {
            int ls, i, n, h, w;
            const bool ok = locate(kk, seq0_, nseq_, ls, i, n, h, w);
            const bool in = h / a.pool < Ho && w / a.pool < Wo;
            float rbuf[RREC], cbuf[CREC];
#pragma unroll
            for (int k = 0; k < RREC; ++k) rbuf[k] = 0.f;
#pragma unroll
            for (int k = 0; k < CREC; ++k) cbuf[k] = 0.f;
            float delta = 0.f;
#pragma unroll
            for (int c = 0; c < HQ; ++c) rbuf[c] = fmaf(r.raw[kk][c], sc[c], sh[c]);
#pragma unroll
            for (int c = 0; c < HQ + GP; ++c) cbuf[c] = fmaf(r.raw[kk][HQ + c], sc[HQ + c], sh[HQ + c]);
#pragma unroll
            for (int k2 = 0; k2 < NCH; ++k2) {                // gradient wrt the stacked sv | sve values (bn_output backward)
                const float d0 = in ? r.dyv[kk][k2 >> 1] : 0.f;
                const float ds = fmaf(cf[k2][0], d0, fmaf(cf[k2][1], r.stk[kk][k2], cf[k2][2]));
                delta = fmaf(ds, r.stk[kk][k2], delta);
                rbuf[HQ + (k2 & 1) * GP + (k2 >> 1)] = GATES ? ds : ds * ((k2 & 1) ? f_sve : f_sv);
            }
            rbuf[HQ + 2 * GP] = r.lse[kk];
            rbuf[HQ + 2 * GP + 1] = delta;
            if (QA_REC)
#pragma unroll
                for (int c = 0; c < HQ; ++c) rbuf[HQ + 2 * GP + 2 + c] = rbuf[c] * s_qk;
            if (!ok) {                                        // (idle sequences of a ragged tile get all-zero records)
#pragma unroll
                for (int k = 0; k < RREC; ++k) rbuf[k] = 0.f;
#pragma unroll
                for (int k = 0; k < CREC; ++k) cbuf[k] = 0.f;
            }
            if (a.raw32 && ok) {      // (bf16 storage: the layer's backward-data / weight-gradient kernels read fp32 -- see medt_api.hip)
                float* r32 = a.raw32 + (size_t)hg * NCH * g.HW + ((size_t)n * 2 * g.C * g.HW + h * g.W + w);
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) r32[(size_t)ch * g.HW] = r.raw[kk][ch];
            }
            f4* rr = reinterpret_cast<f4*>(rowrec + ls * RS + i * RREC);             // 16-byte records: vector stores
            f4* cr = reinterpret_cast<f4*>(colrec + (ls * L + i) * CREC);
#pragma unroll
            for (int k = 0; k < RREC / 4; ++k) rr[k] = f4{rbuf[4 * k], rbuf[4 * k + 1], rbuf[4 * k + 2], rbuf[4 * k + 3]};
#pragma unroll
            for (int k = 0; k < CREC / 4; ++k) cr[k] = f4{cbuf[4 * k], cbuf[4 * k + 1], cbuf[4 * k + 2], cbuf[4 * k + 3]};
        }
    };
    TileRegs pf;
    // (only the gp <= 4 instances that run inside the networks -- 32 lanes per sequence or two columns per lane: the narrow instances of
    //  the bandwidth shapes would spill, and the gp >= 8 instances at 256 registers measured SLOWER with it: <16,16,16> 58.7 -> 69.4 us)
    constexpr bool EARLY = GP <= 4 && (LS == 32 || D <= 2);
    if (EARLY && MEDT_ABL != 21 && MEDT_ABL != 32 && part < a.tiles) issue(part, pf);     // the first tile's loads: in flight during the whole set-up below
    MEDT_SCHED_FENCE();
    // ---- tables: record d = { Rq[c][d] | Rk[c][2L-2-d] | Rv[c][d] } ---------------------------------------------
    // (round 6: inside the networks a workgroup sees ONE tile, and its prologue was four dependent memory round trips -- table entries
    //  one per loop trip, coefficients, the partial rows of the consumer-side finalisation, the tile -- 11 us of a 33-us launch without
    //  its row loop, profiles/r06_sweep_phases.txt.  Now the first tile's loads are issued first, the table entries behind them as one
    //  batch of independent loads, and everything else follows while they fly)
    constexpr int NTB = ((TL + 1) * TREC + 63) / 64;          // table entries per thread of a one-wave workgroup
    float tb[NTB];
#pragma unroll
    for (int k = 0; k < NTB; ++k) {
        const int e = threadIdx.x + k * nthreads, d = e / TREC, r = e - d * TREC;
        float v = 0.f;
        if (MEDT_ABL != 20 && e < (TL + 1) * TREC && d < TL && r < NT)
            v = (r >= HQ && r < GP) ? a.relative[r * TL + (TL - 1 - d)] : a.relative[r * TL + d];
        tb[k] = v;
    }
    MEDT_SCHED_FENCE();
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        sc[ch] = a.qs.scale[grp * 2 * g.C + hg * NCH + ch];
        sh[ch] = a.qs.shift[grp * 2 * g.C + hg * NCH + ch];
        if (!a.ob.on || MEDT_ABL == 10) {
#pragma unroll
            for (int k = 0; k < 3; ++k) cf[ch][k] = a.out_coef[((size_t)grp * g.OC + hg * NCH + ch) * 3 + k];
        }
    }
    if (a.ob.on && MEDT_ABL != 10) {
        // bn_output's backward finalised HERE (fin_inline.h: no bn_bwd_finalize launch in front of the sweep).  Eight lanes per
        // channel, eight channels of the head at a time: every wave sums the partial rows (one load round trip), runs the double
        // arithmetic once for all of them and broadcasts the three coefficients per channel; the first workgroup of the head writes
        // the coefficients and the parameter gradients (one BatchNorm group).
        const BfinJob& j = a.ob.j;
        const int ln = threadIdx.x & 63, slot = ln >> 3, sub = ln & 7;
#pragma unroll
        for (int c0 = 0; c0 < NCH; c0 += 8) {
            const int chl = min(c0 + slot, NCH - 1), chg = hg * NCH + chl;
            double s1, s2;
            fin_slot_sums(j.partials, j.ppg, g.OC, chg, sub, s1, s2);
            s1 *= j.dscale;
            s2 *= j.dscale;
            float c3[3];
            bn_bwd_coef(s1, s2, j.count, j.dscale, j.st.mean[chg], j.st.rstd[chg], j.weight[chg], j.training, c3);
            if (blockIdx.x == 0 && threadIdx.x < 64 && sub == 0 && c0 + slot < NCH) {
#pragma unroll
                for (int k = 0; k < 3; ++k) j.coef[(size_t)chg * 3 + k] = c3[k];
                if (j.dweight) j.dweight[chg] = (float)s2;
                if (j.dbias) j.dbias[chg] = (float)s1;
            }
#pragma unroll
            for (int cc = 0; cc < 8; ++cc)
                if (c0 + cc < NCH) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) cf[c0 + cc][k] = fin_bcast(c3[k], cc * 8);
                }
        }
    }
#pragma unroll
    for (int c = 0; c < GP; ++c) {
        vmean[c] = a.qs.mean[grp * 2 * g.C + hg * NCH + GP + c];
        vrstd[c] = a.qs.rstd[grp * 2 * g.C + hg * NCH + GP + c];
    }
#pragma unroll
    for (int k = 0; k < NTB; ++k) {
        const int e = threadIdx.x + k * nthreads;
        if (MEDT_ABL != 20 && e < (TL + 1) * TREC) tab[e] = tb[k];
    }
    for (int e = threadIdx.x; e < (MEDT_ABL == 27 ? 0 : nw * (2 * NT * L + L * NPG)); e += nthreads) wacc[e] = 0.f;        // wacc | pg contiguous
    for (int tile = part; tile < a.tiles; tile += a.nparts) {
        const int seq0 = tile * S_T, nseq = min(S_T, g.spg - seq0);
        __syncthreads();                                      // the previous tile's outputs have been read
        if (MEDT_ABL == 21) { float* z = reinterpret_cast<float*>(&pf); for (int k = 0; k < (int)(sizeof(pf) / 4); ++k) z[k] = 0.01f * k; }
        else if (!TPF && (tile != part || !EARLY || MEDT_ABL == 32)) issue(tile, pf);
        commit(tile, pf);
        __syncthreads();
        if (TPF && tile + a.nparts < a.tiles) issue(tile + a.nparts, pf);          // in flight during the sweep below
        // ---- the sweep ----------------------------------------------------------------------------------------------
        const int ls = wave * SPW + sg;                       // this lane's sequence of the tile
        const float* rrow = rowrec + ls * RS;
        float* crow = colrec + ls * L * CREC;
        float kc[D][HQ], kb[D][HQ], vc[D][GP], Eqk[D][HQ], Dkr[D][HQ], dv[D][GP], dq_own[D][HQ];
        float tq[D][HQ], tk[D][HQ], tv[D][GP], aq[D][HQ], ak[D][HQ], av[D][GP];
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const float* cr = crow + (cb * D + s) * CREC;
            const float* tr = tab + (L - 1 - cb * D - s) * TREC;          // row 0: d = -j + L - 1
#pragma unroll
            for (int c = 0; c < HQ; ++c) {
                kc[s][c] = cr[c];
                kb[s][c] = cr[c] * s_kr;
                Eqk[s][c] = 0.f; Dkr[s][c] = 0.f; dq_own[s][c] = 0.f;
                tq[s][c] = tr[c]; tk[s][c] = tr[HQ + c];
                aq[s][c] = 0.f; ak[s][c] = 0.f;
            }
#pragma unroll
            for (int c = 0; c < GP; ++c) {
                vc[s][c] = cr[HQ + c];
                dv[s][c] = 0.f;
                tv[s][c] = tr[GP + c];
                av[s][c] = 0.f;
            }
        }
        // per-sequence Gram matrices / sums of q and k (the fix kernel's operands) and the per-position products
        if (MEDT_ABL != 22) {
            float pr[D][NPG], gsum[NPG];
#pragma unroll
            for (int m = 0; m < NPG; ++m) gsum[m] = 0.f;
#pragma unroll
            for (int s = 0; s < D; ++s) {
                const float* rq = rrow + (cb * D + s) * RREC;
                float qv[HQ];
#pragma unroll
                for (int c = 0; c < HQ; ++c) qv[c] = rq[c];
                int m = 0;
#pragma unroll
                for (int x = 0; x < HQ; ++x)
#pragma unroll
                    for (int y = x; y < HQ; ++y, ++m) { pr[s][m] = qv[x] * qv[y]; pr[s][NP + HQ + m] = kc[s][x] * kc[s][y]; }
#pragma unroll
                for (int c = 0; c < HQ; ++c) { pr[s][NP + c] = qv[c]; pr[s][2 * NP + HQ + c] = kc[s][c]; }
#pragma unroll
                for (int m2 = 0; m2 < NPG; ++m2) gsum[m2] += pr[s][m2];
            }
#pragma unroll
            for (int m = 0; m < NPG; ++m) gsum[m] = seq_allsum<LS>(gsum[m]);
            if (head && ls < nseq) {
                float* gr = a.gram + ((size_t)(grp * g.spg + seq0 + ls) * g.G + hg) * NPG;
#pragma unroll
                for (int m = 0; m < NPG; ++m) gr[m] = gsum[m];
            }
            float* pgw = pg + wave * L * NPG + cb;
#pragma unroll
            for (int s = 0; s < D; ++s)
#pragma unroll
                for (int m = 0; m < NPG; ++m) {
                    const float t = seqs_sum<LS>(pr[s][m]);
                    if (sg == 0) pgw[(m * D + s) * LS] += t;
                }
        }
        float* waccw = wacc + wave * 2 * NT * L + cb;
#ifdef MEDT_LANE_EMU
        const unsigned park_addr = (unsigned)(reinterpret_cast<const char*>(crow) - reinterpret_cast<const char*>(smem));
#else
        const unsigned park_addr = (unsigned)(size_t)crow;     // LDS byte address (low half of the flat pointer)
#endif
        // the row record (and the table entry that enters the chain behind it) of row i + 1 is fetched while row i is
        // computed: the loads sit in front of the row's exit store, whose address the compiler cannot tell apart
        // (gp = 4: 20 more registers; measured at 32 lanes per sequence in round 6, where they fit: no gain -- the row loop of a wave that
        //  has its SIMD to itself is bound by its 185 instructions per row, not by the LDS round trip)
        constexpr bool PREFETCH = GP == 2 && MEDT_ABL != 5;
        float nrec[RREC], nfr[TREC];
#pragma unroll
        for (int k = 0; k < RREC; ++k) nrec[k] = rrow[k];
#pragma unroll
        for (int k = 0; k < TREC; ++k) nfr[k] = tab[L * TREC + k];
#pragma unroll 1
        for (int it = 0; it < ((MEDT_ABL == 6 || MEDT_ABL >= 20) ? 0 : LS); ++it) {
            const bool owner = cb == it;                      // this lane's columns are the rows of this iteration
#pragma unroll
            for (int t = 0; t < D; ++t) {
                float rec[RREC], fr[TREC];
                if (MEDT_ABL == 8) {
#pragma unroll
                    for (int k = 0; k < RREC; ++k) rec[k] = 0.01f * k + 1e-3f * it;
#pragma unroll
                    for (int k = 0; k < TREC; ++k) fr[k] = 0.02f * k;
                } else if (PREFETCH) {
#pragma unroll
                    for (int k = 0; k < RREC; ++k) rec[k] = nrec[k];
#pragma unroll
                    for (int k = 0; k < TREC; ++k) fr[k] = nfr[k];
                } else {
                    const float* rr = rrow + (it * D + t) * RREC;
                    const float* tf = tab + (it * D + t + L) * TREC;
#pragma unroll
                    for (int k = 0; k < RREC; ++k) rec[k] = rr[k];
#pragma unroll
                    for (int k = 0; k < TREC; ++k) fr[k] = tf[k];
                }
                if (PREFETCH && MEDT_ABL != 8) {
                    const int inext = min(it * D + t + 1, L - 1);
                    const float* rr = rrow + inext * RREC;
                    const float* tf = tab + (inext + L) * TREC;
#pragma unroll
                    for (int k = 0; k < RREC; ++k) nrec[k] = rr[k];
#pragma unroll
                    for (int k = 0; k < TREC; ++k) nfr[k] = tf[k];
                }
                float q[HQ], qa[HQ], qb[HQ], dsv[GP], dse[GP], Dqk[HQ], Dqr[HQ];
#pragma unroll
                for (int c = 0; c < HQ; ++c) { q[c] = rec[c]; qa[c] = QA_REC ? rec[HQ + 2 * GP + 2 + c] : q[c] * s_qk; qb[c] = q[c] * s_qr; Dqk[c] = 0.f; Dqr[c] = 0.f; }
#pragma unroll
                for (int c = 0; c < GP; ++c) { dsv[c] = rec[HQ + c]; dse[c] = rec[HQ + GP + c]; }
                const float nlse = -rec[HQ + 2 * GP], ndelta = -rec[HQ + 2 * GP + 1];
#pragma unroll
                for (int s = 0; s < D; ++s) {
                    const int p = (s - t + D) % D;            // physical slot of the diagonal column s meets in this row
                    float z = nlse;
#pragma unroll
                    for (int c = 0; c < HQ; ++c) z = fmaf(qa[c], kc[s][c], fmaf(qb[c], tq[p][c], fmaf(kb[s][c], tk[p][c], z)));
                    const float P = MEDT_ABL == 4 ? z : __builtin_amdgcn_exp2f(z);
                    float dZ;
                    if (GATES) {
                        float tvv = 0.f, tee = 0.f;
#pragma unroll
                        for (int c = 0; c < GP; ++c) { tvv = fmaf(dsv[c], vc[s][c], tvv); tee = fmaf(dse[c], tv[p][c], tee); }
                        g_pv = fmaf(P, tvv, g_pv);
                        g_pe = fmaf(P, tee, g_pe);
                        dZ = P * fmaf(f_sv, tvv, fmaf(f_sve, tee, ndelta));
                    } else {
                        float t2 = ndelta;
#pragma unroll
                        for (int c = 0; c < GP; ++c) t2 = fmaf(dsv[c], vc[s][c], fmaf(dse[c], tv[p][c], t2));
                        dZ = P * t2;
                    }
#pragma unroll
                    for (int c = 0; c < HQ; ++c) {
                        Dqk[c] = fmaf(dZ, kc[s][c], Dqk[c]);
                        Dqr[c] = fmaf(dZ, tq[p][c], Dqr[c]);
                        Eqk[s][c] = fmaf(dZ, q[c], Eqk[s][c]);
                        Dkr[s][c] = fmaf(dZ, tk[p][c], Dkr[s][c]);
                        aq[p][c] = fmaf(dZ, q[c], aq[p][c]);
                        ak[p][c] = fmaf(dZ, kc[s][c], ak[p][c]);
                    }
#pragma unroll
                    for (int c = 0; c < GP; ++c) {
                        dv[s][c] = fmaf(P, dsv[c], dv[s][c]);
                        av[p][c] = fmaf(P, dse[c], av[p][c]);
                    }
                }
                // row totals over the LS lanes of the sequence; the lane that owns column j = i keeps dq of the row
#pragma unroll
                for (int c = 0; c < HQ; ++c) {
                    T_qk = fmaf(q[c], Dqk[c], T_qk);            // (this lane's columns only: the lanes are summed at the end)
                    T_qr = fmaf(q[c], Dqr[c], T_qr);
                    const float dq0 = fmaf(e_qk, Dqk[c], (f_qr * e_qr) * Dqr[c]);
                    const float dqv = MEDT_ABL == 3 ? dq0 : seq_allsum<LS>(dq0);
                    dq_own[t][c] = owner ? dqv : dq_own[t][c];
                }
                // the diagonal in slot D-1 leaves the lane: the last lane's is complete (d = i) and is parked in the column
                // record of position i of its sequence (dead between the initial column loads and the out records); every
                // other one moves to the next lane, lane 0 of the sequence takes the next table entry
                const int px = (2 * D - 1 - t) % D;
                if ((t < D - 1 || it < LS - 1) && MEDT_ABL != 7) {
                    if (MEDT_ABL != 1 && MEDT_ABL != 8) {
                        // value-major parking: parked[m][i] -- the fold below reads consecutive words from consecutive lanes
                        float ev[TREC];
#pragma unroll
                        for (int m = 0; m < TREC; ++m) ev[m] = 0.f;
#pragma unroll
                        for (int c = 0; c < HQ; ++c) { ev[c] = aq[px][c]; ev[HQ + c] = ak[px][c]; }
#pragma unroll
                        for (int c = 0; c < GP; ++c) ev[GP + c] = av[px][c];
                        const unsigned pa = park_addr + (unsigned)(it * D + t) * 4u;
#pragma unroll
                        for (int m0 = 0; m0 < NT; m0 += 4)
                            lds_store4_masked<L * 4>(pa + m0 * L * 4, ev[m0], ev[m0 + 1], ev[m0 + 2], ev[m0 + 3], tail_mask);
                    }
                    if (MEDT_ABL != 2)
#pragma unroll
                    for (int c = 0; c < HQ; ++c) {
                        tq[px][c] = chain_shift<LS>(tq[px][c], fr[c], head);
                        tk[px][c] = chain_shift<LS>(tk[px][c], fr[HQ + c], head);
                        aq[px][c] = chain_shift0<LS>(aq[px][c], head);
                        ak[px][c] = chain_shift0<LS>(ak[px][c], head);
                    }
                    if (MEDT_ABL != 2)
#pragma unroll
                    for (int c = 0; c < GP; ++c) {
                        tv[px][c] = chain_shift<LS>(tv[px][c], fr[GP + c], head);
                        av[px][c] = chain_shift0<LS>(av[px][c], head);
                    }
                }
            }
        }
        asm volatile("" ::: "memory");                         // (the parked values below were stored by asm statements)
        // ---- table gradients of the tile -> the wave's accumulators: the diagonals parked by the last lanes (d = 0 .. L-2,
        // this lane folds d = its own D positions) and the ones still in the chain (d = 2L-2-j; slot s sits in physical
        // (s + 1) % D after the last row); the sequences of the wave are summed lane-wise, sequence 0's lanes accumulate
#pragma unroll
        for (int s = 0; s < (MEDT_ABL == 23 ? 0 : D); ++s) {
            const int p = (s + 1) % D;
            const int j = cb * D + s;
            const float* pk = crow + j;                           // parked[m][j]
            float ex[NT], ch[NT];
#pragma unroll
            for (int m = 0; m < NT; ++m) ex[m] = j < L - 1 ? pk[m * L] : 0.f;
            MEDT_SCHED_FENCE();                                  // (every parked value is read before the out records overwrite them)
#pragma unroll
            for (int m = 0; m < NT; ++m) ex[m] = seqs_sum<LS>(ex[m]);
#pragma unroll
            for (int c = 0; c < HQ; ++c) { ch[c] = seqs_sum<LS>(aq[p][c]); ch[HQ + c] = seqs_sum<LS>(ak[p][c]); }
#pragma unroll
            for (int c = 0; c < GP; ++c) ch[GP + c] = seqs_sum<LS>(av[p][c]);
            if (sg == 0) {
#pragma unroll
                for (int m = 0; m < NT; ++m) {
                    waccw[(m * D + s) * LS] += ex[m];                     // d = j
                    waccw[NT * L + (m * D + s) * LS] += ch[m];            // d = 2L-2-j
                }
            }
        }
        // ---- column results: dq of the lane's own positions, dk, dv -> the out records (alias of the column records)
#pragma unroll
        for (int s = 0; s < D; ++s) {
            float* cr = crow + (cb * D + s) * CREC;
#pragma unroll
            for (int c = 0; c < HQ; ++c) {
                T_kr = fmaf(kc[s][c], Dkr[s][c], T_kr);
                cr[c] = dq_own[s][c];
                cr[HQ + c] = fmaf(e_qk, Eqk[s][c], (f_kr * e_kr) * Dkr[s][c]);
            }
#pragma unroll
            for (int c = 0; c < GP; ++c) cr[GP + c] = GATES ? f_sv * dv[s][c] : dv[s][c];
        }
        __syncthreads();
        // ---- write dqkv (NCHW), bn_qkv backward partials of the v channels (raw v re-read: L2-resident) --------------------
#pragma unroll
        for (int kk = 0; kk < (MEDT_ABL == 24 ? 0 : D); ++kk) {
            int ls2, i2, n, h, w;
            if (locate(kk, seq0, nseq, ls2, i2, n, h, w)) {
                const float* cr = colrec + (ls2 * L + i2) * CREC;
                const int pix = h * g.W + w;
                const unsigned ooff = ((unsigned)n * 2u * g.C * g.HW + pix) * 4u;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch)
                    *reinterpret_cast<float*>(reinterpret_cast<char*>(obase) + ooff + (unsigned)(ch * g.HW) * 4u) = cr[ch];
#pragma unroll
                for (int c = 0; c < GP; ++c) {
                    float rv;
                    if constexpr (KEEP_V) {
                        rv = pf.raw[kk][GP + c];              // (the tile's own registers: same thread, same element as in `issue`)
                    } else {
                        const unsigned o = ((unsigned)n * 2u * g.C * g.HW + pix + (unsigned)((GP + c) * g.HW)) * (unsigned)es;
                        rv = g.bf16 ? bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(qbase + o))
                                    : *reinterpret_cast<const float*>(qbase + o);
                    }
                    const float d = cr[GP + c];
                    vst[2 * c] += d;
                    vst[2 * c + 1] = fmaf(d, (rv - vmean[c]) * vrstd[c], vst[2 * c + 1]);
                }
            }
        }
    }
    // ---- workgroup results ------------------------------------------------------------------------------------------
    __syncthreads();
    const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;       // (head, group, part)
    if (MEDT_ABL != 11 && MEDT_ABL != 25) {   // table gradients: waves in fixed order, the per-head scales, `relative`'s layout (Rk rows reversed back)
        float* rp = a.rel_part + blk * NCH * TL;
        const float sq = f_qr * e_qr, sk = f_kr * e_kr, sv = GATES ? f_sve : 1.f;
        // (round 6: a row r of the table per trip of the outer loop -- no division by TL = 2L - 1 per element, the LDS reads of a
        //  trip independent of each other; same sums in the same order)
#pragma unroll
        for (int r = 0; r < NCH; ++r) {
            const float scale = r < HQ ? sq : (r < GP ? sk : sv);
            for (int d = threadIdx.x; d < TL; d += nthreads) {
                const int dd = (r >= HQ && r < GP) ? TL - 1 - d : d;
                const int half = dd >= L - 1, j = half ? 2 * L - 2 - dd : dd;
                const int idx = half * NT * L + (r * D + j % D) * LS + j / D;
                float s = 0.f;
                for (int w = 0; w < nw; ++w) s += wacc[w * 2 * NT * L + idx];
                rp[r * TL + d] = s * scale;
            }
        }
        float* pp = a.pg_part + blk * L * NPG;
#pragma unroll
        for (int m = 0; m < NPG; ++m)
            for (int j = threadIdx.x; j < L; j += nthreads) {
                float s = 0.f;
                for (int w = 0; w < nw; ++w) s += pg[w * L * NPG + (m * D + j % D) * LS + j / D];
                pp[j * NPG + m] = s;
            }
    }
    if (MEDT_ABL != 26) {
        float v[4] = {0.f, T_qk, f_qr * T_qr, f_kr * T_kr};     // sum dZ * {1, S_qk, S_qr, S_kr}
        wg_sum<4>(v, red, a.part_sb + ((size_t)blockIdx.x * g.G + hg) * 4, nw);
    }
    if (MEDT_ABL != 26) {
        // bn_qkv backward partials [group][row][2C][2]: this head's v channels; its q | k columns are written as zeros
        // (attn_bwd_fix_kernel's rows carry those)
        float* dst = a.part_qb + ((size_t)(grp * a.qb_rpg + part) * 2 * g.C + hg * NCH) * 2;
        if (threadIdx.x < 2 * GP) dst[threadIdx.x] = 0.f;
        wg_sum<2 * GP>(vst, red, dst + 2 * GP, nw);
    }
    if (GATES) {
        float v[4] = {T_qr, T_kr, g_pe, g_pv};                // sum dZ rq, sum dZ rk (ungated), sum P dPe, sum P dPv
        wg_sum<4>(v, red, a.gate_raw + blk * 4, nw);
    }
}

// --------------------------------------------------------------------------- //
// The u / w terms of dq | dk (closed forms, see the header) + the bn_qkv backward partials of the q | k channels.
// Elementwise: one lane per position of the BN group, one head per blockIdx.y.
// --------------------------------------------------------------------------- //
struct FixArgs {
    AxialGeom g;
    const float *qkv_raw, *sim_coef, *tables, *gram;
    BnStats qs;
    GatePtrs gates;
    float *dqkv, *part_qb;
    int fparts, qb_rpg, qb_row0, apply;
    SimBSrc sb;                    // on: bn_similarity's backward coefficients are derived here from the sweep's partial rows (fin_inline.h)
};


// (BF: the storage type of qkv_raw at compile time -- with the runtime flag inside ld_act every load sat in its own branch, one global
//  round trip per channel: +7 us per launch with bf16 storage, round 6)
template <int HQ, bool BF, int FIX_PPT>
__device__ __forceinline__ void attn_bwd_fix_body(const FixArgs& a, float* red) {
    constexpr int GP = 2 * HQ, NCH = 2 * GP, NP = HQ * (HQ + 1) / 2, NPG = 2 * (NP + HQ), NR = HQ + NP;
    const AxialGeom& g = a.g;
    const int grp = blockIdx.x / a.fparts, part = blockIdx.x - grp * a.fparts, hg = blockIdx.y;
    const int per_group = g.npg * g.HW;
    float v[4 * HQ];
#pragma unroll
    for (int k = 0; k < 4 * HQ; ++k) v[k] = 0.f;
    // bn_similarity's backward coefficients (e, u, w) of the head's three channels: finalised HERE from the sweep's partial rows
    // (fin_inline.h: no sim_bwd_finalize launch between the sweep and this kernel; the head's first workgroup writes them for the
    // relfix kernel behind, and bn_similarity's parameter gradients), or read where the finalisation kernel left them
    float scf[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    if (a.sb.on) {
        sim_coef_inline(a.sb, hg, (int)(threadIdx.x & 63), blockIdx.x == 0 && threadIdx.x < 64, scf);
    } else if (a.apply) {
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int k = 0; k < 3; ++k) scf[x][k] = a.sim_coef[((size_t)grp * g.SC + x * g.G + hg) * 3 + k];
    }
    // FIX_PPT positions per thread (a thread's positions are 256 apart: every load instruction stays coalesced, four
    // independent chains of loads are in flight per lane)
#pragma unroll
    for (int pp = 0; pp < FIX_PPT; ++pp) {
    const int qpos = (part * FIX_PPT + pp) * MEDT_THREADS + threadIdx.x;
    if (qpos < per_group) {
        const int ni = qpos / g.HW, pix = qpos - ni * g.HW, n = grp * g.npg + ni;
        const int h = pix / g.W, w = pix - h * g.W;
        const int i = g.axis == 1 ? w : h, sq = g.axis == 1 ? h : w;
        const size_t off = ((size_t)n * 2 * g.C + hg * NCH) * g.HW + pix;
        const int cbase = grp * 2 * g.C + hg * NCH;
        float raw[GP], x[GP], d[GP];
#pragma unroll
        for (int c = 0; c < GP; ++c) {
            raw[c] = ld_act(a.qkv_raw, off + (size_t)c * g.HW, BF ? 1 : 0);
            d[c] = a.dqkv[off + (size_t)c * g.HW];
            x[c] = fmaf(raw[c], a.qs.scale[cbase + c], a.qs.shift[cbase + c]);
        }
        if (a.apply) {
            const float f_qr = gate(a.gates.f_qr), f_kr = gate(a.gates.f_kr);
            const float u_qk = scf[0][1], w_qk = scf[0][2];
            const float u_qr = f_qr * f_qr * scf[1][1], w_qr = f_qr * scf[1][2], u_kr = f_kr * f_kr * scf[2][1], w_kr = f_kr * scf[2][2];
            const float* gr = a.gram + ((size_t)(n * g.Bo + sq) * g.G + hg) * NPG;      // Gq pairs | Sq | Gk pairs | Sk
            const float* tQ = a.tables + (size_t)i * NR;                             // U_c | T pairs (x2 off the diagonal)
            const float* tK = a.tables + (size_t)(g.L + i) * NR;
            float Gq[HQ][HQ], Gk[HQ][HQ], TQ[HQ][HQ], TK[HQ][HQ];
            int m = 0;
#pragma unroll
            for (int p = 0; p < HQ; ++p)
#pragma unroll
                for (int r = p; r < HQ; ++r, ++m) {
                    Gq[p][r] = Gq[r][p] = gr[m];
                    Gk[p][r] = Gk[r][p] = gr[NP + HQ + m];
                    const float hf = r > p ? 0.5f : 1.f;
                    TQ[p][r] = TQ[r][p] = hf * tQ[HQ + m];
                    TK[p][r] = TK[r][p] = hf * tK[HQ + m];
                }
#pragma unroll
            for (int c = 0; c < HQ; ++c) {
                float fq = fmaf(w_qk, gr[2 * NP + HQ + c], w_qr * tQ[c]);          // w_qk Sk_c + f w_qr UQ_c[i]
                float fk = fmaf(w_qk, gr[NP + c], w_kr * tK[c]);                   // w_qk Sq_c + f w_kr UK_c[i]
#pragma unroll
                for (int e = 0; e < HQ; ++e) {
                    fq = fmaf(x[e], fmaf(u_qk, Gk[e][c], u_qr * TQ[e][c]), fq);
                    fk = fmaf(x[HQ + e], fmaf(u_qk, Gq[e][c], u_kr * TK[e][c]), fk);
                }
                d[c] += fq;
                d[HQ + c] += fk;
            }
#pragma unroll
            for (int c = 0; c < GP; ++c) a.dqkv[off + (size_t)c * g.HW] = d[c];
        }
#pragma unroll
        for (int c = 0; c < GP; ++c) {
            v[2 * c] += d[c];
            v[2 * c + 1] = fmaf(d[c], (raw[c] - a.qs.mean[cbase + c]) * a.qs.rstd[cbase + c], v[2 * c + 1]);
        }
    }
    }
    float* dst = a.part_qb + ((size_t)(grp * a.qb_rpg + a.qb_row0 + part) * 2 * g.C + hg * NCH) * 2;
    if (threadIdx.x < 2 * GP) dst[2 * GP + threadIdx.x] = 0.f;            // (the sweep's rows carry the v channels)
    block_sum<4 * HQ>(v, red, dst);
}

template <int HQ, int PPT>
__global__ __launch_bounds__(MEDT_THREADS) void attn_bwd_fix_kernel(FixArgs a) {
    MEDT_STATIC_SHARED float red[MEDT_WAVES * 4 * HQ];
    if (a.g.bf16) attn_bwd_fix_body<HQ, true, PPT>(a, red);
    else attn_bwd_fix_body<HQ, false, PPT>(a, red);
}

// --------------------------------------------------------------------------- //
// The u / w terms of the relative-table gradient and the gate gradients, one workgroup per (BN group, head):
// written as one extra row per workgroup of the partial slabs the (deferred) row reductions sum.
// --------------------------------------------------------------------------- //
using RelfixArgs = RelfixJob;       // defer.h: the same record is what a bound queue stores until the flush

constexpr int RELFIX_THREADS = 1024, RELFIX_SLICES = 4;      // the partial rows are summed by 4 slices of 256 threads

template <int HQ>
__device__ __forceinline__ void relfix_body(const RelfixArgs& a, const int blk, float* pgs, double (*gred)[4]) {
    constexpr int GP = 2 * HQ, NCH = 2 * GP, NP = HQ * (HQ + 1) / 2, NPG = 2 * (NP + HQ), MEDT_RT = RELFIX_THREADS;
    // pgs: [L][NPG], then [RELFIX_SLICES - 1][L][NPG] slice sums
    const struct { int L, G, SC; double sim_count; } g = {a.L, a.G, a.SC, a.sim_count};
    const int L = g.L, TL = 2 * L - 1;
    const int grp = blk / g.G, hg = blk - grp * g.G;
    const size_t blk0 = (size_t)hg * a.sweep_gridx + (size_t)grp * a.nparts;
    if (a.qb_on) {
        // rider: bn_qkv's backward coefficients and parameter gradients of this head's channels (one BatchNorm group), one wave per
        // channel -- the arithmetic of bn_bwd_finalize_body (pointwise.hip); no launch of its own between the fix kernel and the 1x1
        // backward-data kernel
        const BfinJob& j = a.qb;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int c = wave; c < a.qb_nch; c += MEDT_RT / 64) {
            const int ch = hg * a.qb_nch + c;
            double s1, s2;
            fin_wave_sums(j.partials, j.ppg, j.CH, ch, lane, s1, s2);
            s1 *= j.dscale;
            s2 *= j.dscale;
            if (lane == 0) {
                bn_bwd_coef(s1, s2, j.count, j.dscale, j.st.mean[ch], j.st.rstd[ch], j.weight[ch], j.training, j.coef + (size_t)ch * 3);
                if (j.dweight) j.dweight[ch] = (float)s2;
                if (j.dbias) j.dbias[ch] = (float)s1;
            }
        }
    }
    {   // per-position Gram sums of this (group, head): slice q sums the parts [q * per, (q + 1) * per), eight loads in
        // flight, then the slices are added in fixed order
        const int slice = threadIdx.x / MEDT_THREADS, t0 = threadIdx.x - slice * MEDT_THREADS;
        const int per = (a.nparts + RELFIX_SLICES - 1) / RELFIX_SLICES;
        const int p0 = slice * per, p1 = min(a.nparts, p0 + per);
        for (int e = t0; e < L * NPG; e += MEDT_THREADS) {
            const float* src = a.pg_part + blk0 * L * NPG + e;
            float s = 0.f;
            int p = p0;
            for (; p + 8 <= p1; p += 8) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = src[(size_t)(p + k) * L * NPG];
#pragma unroll
                for (int k = 0; k < 8; ++k) s += v[k];
            }
            for (; p < p1; ++p) s += src[(size_t)p * L * NPG];
            pgs[slice * L * NPG + e] = s;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < L * NPG; e += MEDT_RT)
            pgs[e] = (pgs[e] + pgs[L * NPG + e]) + (pgs[2 * L * NPG + e] + pgs[3 * L * NPG + e]);
    }
    double graw[4] = {0.0, 0.0, 0.0, 0.0};
    if (a.gate_rows) {                                        // gate sums of this (group, head): parts over the threads
        for (int p = threadIdx.x; p < a.nparts; p += MEDT_RT) {
            const float* q = a.gate_raw + (blk0 + p) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) graw[k] += q[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) graw[k] += __shfl_xor(graw[k], o, 64);
            if ((threadIdx.x & 63) == 0) gred[threadIdx.x >> 6][k] = graw[k];
        }
    }
    __syncthreads();
    const float f_qr = gate(a.gates.f_qr), f_kr = gate(a.gates.f_kr);
    const float* cr = a.sim_coef + ((size_t)grp * g.SC + g.G + hg) * 3;
    const float* ck = a.sim_coef + ((size_t)grp * g.SC + 2 * g.G + hg) * 3;
    float* out = a.rel_rows + (size_t)blk * NCH * TL;
    for (int e = threadIdx.x; e < NCH * TL; e += MEDT_RT) {
        const int r = e / TL, d = e - r * TL;
        float res = 0.f;
        if (r < GP && a.training) {
            const int side = r / HQ, c = r - side * HQ;
            const float f = side ? f_kr : f_qr;
            const float* cf = side ? ck : cr;
            const double cu = (double)f * f * cf[1], cw = (double)f * cf[2];
            const int lo = max(0, d - L + 1), hi = min(L - 1, d);
            const int base = side * (NP + HQ);
            double acc = 0.0;
            float Rd[HQ];
#pragma unroll
            for (int x = 0; x < HQ; ++x) Rd[x] = a.relative[(size_t)(side * HQ + x) * TL + d];
            for (int i = lo; i <= hi; ++i) {
                const float* pp = pgs + i * NPG + base;
                double t = 0.0;
#pragma unroll
                for (int x = 0; x < HQ; ++x) {
                    const int lo2 = x < c ? x : c, hi2 = x < c ? c : x;
                    const int m = lo2 * HQ - lo2 * (lo2 - 1) / 2 + (hi2 - lo2);       // pair index of (lo2 <= hi2)
                    t += (double)Rd[x] * pp[m];
                }
                acc += cu * t + cw * pp[NP + c];
            }
            res = (float)acc;
        }
        out[e] = res;
    }
    if (a.gate_rows && threadIdx.x == 0) {
        double t_qr = 0.0, t_kr = 0.0, pe = 0.0, pv = 0.0;
        for (int w = 0; w < MEDT_RT / 64; ++w) { t_qr += gred[w][0]; t_kr += gred[w][1]; pe += gred[w][2]; pv += gred[w][3]; }
        double gq = (double)cr[0] * t_qr, gk = (double)ck[0] * t_kr;
        if (a.training) {
            const double count = g.sim_count;
            const int cq = grp * g.SC + g.G + hg, ckk = grp * g.SC + 2 * g.G + hg;
            const double mq = a.ss.mean[cq], rq = a.ss.rstd[cq], mk = a.ss.mean[ckk], rk = a.ss.rstd[ckk];
            const double s1q = count * mq, s2q = count * (1.0 / (rq * rq) - (double)a.eps + mq * mq);
            const double s1k = count * mk, s2k = count * (1.0 / (rk * rk) - (double)a.eps + mk * mk);
            // d f = e T + (u S2 + w S1) / f with S1, S2 the sums of the GATED logits from the saved statistics.  At f == 0 the
            // gated logits are identically 0: mean(dY xhat) = 0 and mean(dY) = 0 (softmax rows), so u = w = 0 EXACTLY and e T is
            // the whole gradient -- the skipped term is 0, not an approximation (tests: test_layer_zero_gate_gradients).
            if (f_qr != 0.f) gq += ((double)cr[1] * s2q + (double)cr[2] * s1q) / f_qr;
            if (f_kr != 0.f) gk += ((double)ck[1] * s2k + (double)ck[2] * s1k) / f_kr;
        }
        float* go = a.gate_rows + (size_t)blk * 4;
        go[0] = (float)gq; go[1] = (float)gk; go[2] = (float)pe; go[3] = (float)pv;
    }
}

template <int HQ>
__global__ __launch_bounds__(RELFIX_THREADS) void attn_bwd_relfix_kernel(RelfixArgs a) {
    extern __shared__ float pgs[];
    MEDT_STATIC_SHARED double gred[RELFIX_THREADS / 64][4];
    relfix_body<HQ>(a, blockIdx.x, pgs, gred);
}

__global__ __launch_bounds__(MEDT_THREADS) void bwd_tables_kernel(const float* __restrict__ relative, float* __restrict__ tables,
                                                                  int HQ, int L) {
    extern __shared__ float lds[];
    sim_tables_block(blockIdx.x, relative, tables, HQ, L, lds);
}

template <int GP, int L, int LS>
size_t sweep_lds_bytes(int nw) {
    using C = Sw<GP, L, LS>;
    const int S_T = nw * C::SPW;
    return ((size_t)S_T * (C::RS + L * C::CREC) + (size_t)(C::TL + 1) * C::TREC + (size_t)nw * (2 * C::NT * L + L * C::NPG) +
            128) * sizeof(float);
}

static bool sweep_enabled() {
    static const bool on = [] { const char* e = getenv("MEDT_BWD_SWEEP"); return !(e && e[0] == '0'); }();
    return on;
}

}  // namespace

// every (channels per head, length, lanes per sequence) the sweep is compiled for
#define MEDT_SWEEP_INSTANCES(X)                                                                                          \
    X(2, 32, 8) X(2, 64, 16) X(2, 128, 16) X(4, 32, 8) X(4, 64, 16) X(4, 128, 32) X(8, 32, 16) X(8, 16, 16) X(16, 16, 16)         \
    X(2, 32, 16) X(2, 64, 32) X(2, 128, 32) X(4, 32, 16) X(4, 64, 32)

// Plan: lanes per sequence, waves per workgroup, tiles, persistent workgroups.  Returns false when the generic two-pass
// kernels of axial_core.hip have to run (other lengths, per-sequence gates, MEDT_BWD_SWEEP=0).
bool axial_bwd_sweep_plan(const AxialGeom& g, int gate_stride, SweepPlan* p) {
    if (!g.pos || gate_stride != 0 || !sweep_enabled() || !fast_path_enabled()) return false;
    int ls = 0;
    if (g.gp == 2 && (g.L == 32 || g.L == 64 || g.L == 128)) ls = g.L == 32 ? 8 : 16;
    else if (g.gp == 4 && (g.L == 32 || g.L == 64)) ls = g.L == 32 ? 8 : 16;
    // round 5: layer2.0 of MedT / the unets at 256 px.  32 lanes per sequence (a sequence = two 16-lane rows, D = 4 key columns
    // per lane): the register budget of <4, 64, 16>; the diagonal chain crosses the row boundary with wave_shr:1
    else if (g.gp == 4 && g.L == 128) ls = 32;
    else if (g.gp == 8 && (g.L == 32 || g.L == 16)) ls = 16;       // round 4: the deep layers of axialunet / gatedaxialunet
    else if (g.gp == 16 && g.L == 16) ls = 16;                     // (layer4 at 128 px: 256 VGPRs + 70 AGPRs, no scratch)
    if (!ls) return false;
    // the sweep / fix kernels address qkv_raw, stacked and dqkv with 32-bit byte offsets (saddr + voffset): tensors of 4 GiB
    // and more take the generic kernels (size_t arithmetic)
    if ((size_t)g.N * 2 * g.C * g.HW * 4 > 0xffffffffull) return false;
    static const int env_nw = 0;
    static const int env_cap = 2048;
    // Round 5: few sequences (the layers inside the networks: 128 - 256 sequences x 8 heads) leave half of the chip's 1024 SIMDs
    // without a wave and every wave alone on its SIMD -- the launch is one wave's latency.  Twice the lanes per sequence then:
    // half the key columns per lane (the row loop's body halves, its per-row bookkeeping does not) on twice the waves.
    // MEDT_BWD_WIDE=0: the lane counts tuned on the bandwidth shape (rounds 3 / 4) everywhere.
    // gp = 4: the wide instances everywhere -- <4,64,16> / <4,32,8> need more than the 256 registers two resident workgroups leave
    // a lane (244 / 228 bytes of scratch), <4,64,32> / <4,32,16> take 164 (measured: gatedaxialunet bs 8 4.071 -> 4.022 ms/step, MedT-256
    // 2.989 -> 2.980).  gp = 2 keeps the narrow lanes on big launches (the bandwidth shape: 0.61 vs 0.75 ms, round 3).
    // MEDT_BWD_WIDE=0: the lane counts of rounds 3 / 4 everywhere; =2: wide wherever an instance exists.
    static const int wide = [] { const char* e = getenv("MEDT_BWD_WIDE"); return e ? atoi(e) : 1; }();
    const bool has_wide = (g.gp == 2 || g.gp == 4) && ((g.L == 64 && ls == 16) || (g.L == 32 && ls == 8) || (g.gp == 2 && g.L == 128 && ls == 16));
    if (wide && has_wide && (wide == 2 || g.gp == 4 || (long)g.groups * g.G * cdiv(g.spg, 64 / ls) < 1024)) ls *= 2;
    const int spw = 64 / ls;
    int nw = g.axis == 0 ? 4 : 2;
    // small problems: as many workgroups as there are sequences to give them
    while (nw > 1 && (long)g.groups * g.G * cdiv(g.spg, nw * spw) < 1024) nw >>= 1;
    if (env_nw == 1 || env_nw == 2 || env_nw == 4) nw = env_nw;
    p->LS = ls;
    p->nw = nw;
    p->S_T = nw * spw;
    p->tiles = cdiv(g.spg, p->S_T);
    int cap = env_cap / (g.groups * g.G);
    if (cap < 1) cap = 1;
    p->nparts = p->tiles < cap ? p->tiles : cap;
    // the fix kernel: four positions per thread (four independent load chains) -- or ONE where that leaves fewer than 32 workgroups (the
    // deep layers of the unets: 2048 positions per group = 2 parts x 8 heads = 16 workgroups of a 14 - 30 us launch; round 6:
    // gatedaxialunet bs 8 3.885 -> 3.779 ms/step, profiles/r06_fix_ppt_ab.txt.  At 32 workgroups -- MedT's 32 x 32 layers -- the kernel
    // itself gains, 10.7 -> 7.0 us, and the step does not: the threshold stays below them) and the extra partial rows still fit the
    // consumer-side finalisation of bn_qkv's backward (<= 256 rows)
    p->fix_ppt = 4;
    p->fparts = cdiv(g.npg * g.HW, MEDT_THREADS * 4);
#ifndef MEDT_AB_FIX_PPT4                // (A/B build: round 5's four positions per thread everywhere)
    if ((long)g.groups * g.G * p->fparts < 32 && p->nparts + cdiv(g.npg * g.HW, MEDT_THREADS) <= 256) {
        p->fix_ppt = 1;
        p->fparts = cdiv(g.npg * g.HW, MEDT_THREADS);
    }
#endif
    const int hq = g.hq, np = hq * (hq + 1) / 2;
    p->npg_floats = 2 * (np + hq);
    p->lds = 0;
#define MEDT_SWEEP_LDS(GPv, Lv, LSv) if (g.gp == GPv && g.L == Lv && ls == LSv) p->lds = sweep_lds_bytes<GPv, Lv, LSv>(nw);
    MEDT_SWEEP_INSTANCES(MEDT_SWEEP_LDS)
#undef MEDT_SWEEP_LDS
    if (!p->lds) return false;
    return p->lds <= 160 * 1024;
}

int axial_bwd_tables(const AxialGeom& g, const float* relative, float* tables, hipStream_t s) {
    hipLaunchKernelGGL(bwd_tables_kernel, dim3(sim_tables_blocks(g)), dim3(MEDT_THREADS), (2 * g.L - 1) * 2 * sizeof(float), s,
                       relative, tables, g.hq, g.L);
    return launch_status("bwd_tables");
}

int axial_attn_bwd_sweep(const AxialGeom& g, const SweepPlan& p, const float* qkv_raw, BnStats qkv, BnStats sim,
                         const float* relative, GatePtrs gates, const float* stacked, const float* lse, const float* dy,
                         const float* out_coef, int stride, float* dqkv, float* part_qb, int qb_rpg, float* part_sb,
                         float* rel_part, float* pg_part, float* gram, float* gate_raw, hipStream_t s, float* raw32,
                         const BfinSrc* ob, const float* ymask) {
    if (abl_skip("sweep")) return MEDT_OK;
    SweepArgs a;
    a.ob = ob ? *ob : no_bfin_src();
    a.ymask = ymask;
    a.g = g;
    a.qkv_raw = qkv_raw; a.stacked = stacked; a.lse = lse; a.dy = dy; a.relative = relative; a.out_coef = out_coef;
    a.qs = qkv; a.ss = sim; a.gates = gates; a.pool = stride;
    a.dqkv = dqkv; a.part_qb = part_qb; a.part_sb = part_sb; a.rel_part = rel_part; a.pg_part = pg_part; a.gram = gram;
    a.gate_raw = gate_raw;
    a.raw32 = raw32;
    a.tiles = p.tiles; a.nparts = p.nparts; a.qb_rpg = qb_rpg;
    const dim3 grid(g.groups * p.nparts, g.G), block(64 * p.nw);
    const bool gt = gate_raw != nullptr;
#define MEDT_SWEEP(GPv, Lv, LSv)                                                                                     \
    do {                                                                                                             \
        if (gt) hipLaunchKernelGGL((attn_bwd_sweep_kernel<GPv, Lv, LSv, true>), grid, block, p.lds, s, a);            \
        else hipLaunchKernelGGL((attn_bwd_sweep_kernel<GPv, Lv, LSv, false>), grid, block, p.lds, s, a);              \
    } while (0)
    bool launched = false;
#define MEDT_SWEEP_CASE(GPv, Lv, LSv) if (!launched && g.gp == GPv && g.L == Lv && p.LS == LSv) { MEDT_SWEEP(GPv, Lv, LSv); launched = true; }
    MEDT_SWEEP_INSTANCES(MEDT_SWEEP_CASE)
#undef MEDT_SWEEP_CASE
    if (!launched) { set_error("attn_bwd_sweep: no instantiation for gp=%d L=%d LS=%d", g.gp, g.L, p.LS); return MEDT_EUNSUPPORTED; }
#undef MEDT_SWEEP
    return launch_status("attn_bwd_sweep_kernel");
}

int axial_attn_bwd_fix(const AxialGeom& g, const SweepPlan& p, const float* qkv_raw, BnStats qkv, const float* sim_coef,
                       const float* tables, const float* gram, GatePtrs gates, int apply, float* dqkv, float* part_qb,
                       int qb_rpg, hipStream_t s, const SimBSrc* sb) {
    FixArgs a;
    a.sb = sb ? *sb : no_simb_src();
    a.g = g; a.qkv_raw = qkv_raw; a.sim_coef = sim_coef; a.tables = tables; a.gram = gram; a.qs = qkv; a.gates = gates;
    a.dqkv = dqkv; a.part_qb = part_qb; a.fparts = p.fparts; a.qb_rpg = qb_rpg; a.qb_row0 = p.nparts; a.apply = apply;
    const dim3 grid(g.groups * p.fparts, g.G), block(MEDT_THREADS);
#define MEDT_FIX(HQv)                                                                                   \
    do {                                                                                                \
        if (p.fix_ppt == 1) hipLaunchKernelGGL((attn_bwd_fix_kernel<HQv, 1>), grid, block, 0, s, a);     \
        else hipLaunchKernelGGL((attn_bwd_fix_kernel<HQv, 4>), grid, block, 0, s, a);                    \
    } while (0)
    if (g.hq == 1) MEDT_FIX(1);
    else if (g.hq == 2) MEDT_FIX(2);
    else if (g.hq == 4) MEDT_FIX(4);
    else MEDT_FIX(8);
#undef MEDT_FIX
    return launch_status("attn_bwd_fix_kernel");
}

int axial_attn_bwd_relfix(const AxialGeom& g, const SweepPlan& p, const float* relative, const float* sim_coef, BnStats sim,
                          GatePtrs gates, const float* pg_part, const float* gate_raw, int training, float eps,
                          float* rel_rows, float* gate_rows, hipStream_t s, const BfinSrc* qb) {
    RelfixArgs a;
    a.qb = qb ? qb->j : BfinJob{};
    a.qb_on = qb ? qb->on : 0;
    a.qb_nch = 4 * g.hq;
    a.relative = relative; a.sim_coef = sim_coef; a.pg_part = pg_part; a.gate_raw = gate_raw; a.ss = sim;
    a.gates = gates; a.rel_rows = rel_rows; a.gate_rows = gate_raw ? gate_rows : nullptr; a.nparts = p.nparts;
    a.sweep_gridx = g.groups * p.nparts; a.training = training; a.eps = eps;
    a.L = g.L; a.G = g.G; a.SC = g.SC; a.hq = g.hq; a.sim_count = g.sim_count; a.blocks = g.groups * g.G;
    a.lds = (unsigned)((size_t)RELFIX_SLICES * g.L * p.npg_floats * sizeof(float));
    const dim3 grid(a.blocks), block(RELFIX_THREADS);
    if (g.hq == 1) hipLaunchKernelGGL((attn_bwd_relfix_kernel<1>), grid, block, a.lds, s, a);
    else if (g.hq == 2) hipLaunchKernelGGL((attn_bwd_relfix_kernel<2>), grid, block, a.lds, s, a);
    else if (g.hq == 4) hipLaunchKernelGGL((attn_bwd_relfix_kernel<4>), grid, block, a.lds, s, a);
    else hipLaunchKernelGGL((attn_bwd_relfix_kernel<8>), grid, block, a.lds, s, a);
    return launch_status("attn_bwd_relfix_kernel");
}

}  // namespace medt
