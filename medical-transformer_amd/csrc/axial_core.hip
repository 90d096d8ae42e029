// axial_core.hip -- the L x L stages of the axial-attention layer, fused.
//
// Replaces, per (sequence b, head g) of reference lib/models/axialnet.py:
//   :155-159  index_select + 3 einsums (qk, qr, kr)          -> recomputed in registers, never stored
//   :163-167  gates, cat, BatchNorm2d(3G) on (B*,3G,L,L)     -> statistics pass + affine inside the main pass
//   :170-176  softmax, sv / sve einsums, gates               -> same pass, row per lane
//   :178      cat(sv,sve).view                               -> interleaved channel store
// and the autograd backward of all of it (two recompute passes around the
// bn_similarity backward barrier).
//
// Work decomposition (gfx950, wave64): one 256-thread workgroup owns S_T = 256/L whole
// sequences of one head; thread t owns query row i = t % L of sequence t / L ("row per lane":
// softmax max/sum and the P.V products need no cross-lane traffic).  q/k/v of the tile are
// staged once through LDS (coalesced along whichever of (sequence, position) is contiguous in
// NCHW), bn_qkv's affine is applied in place, keys/values are then read as LDS broadcasts and
// the three relative tables by a skewed index d = i - j + L - 1 (consecutive lanes ->
// consecutive addresses).  All tensors stay NCHW: the reference's permute+contiguous copies
// (:143-148, :181-184) become address arithmetic.
#include "axial_tiles.h"
#include "fin_inline.h"
#include "sim_tables.h"
#include <stdlib.h>

namespace medt {

bool fast_path_enabled() {
    static const bool on = [] { const char* e = getenv("MEDT_DISABLE_FAST"); return !(e && e[0] == '1'); }();
    return on;
}

// --------------------------------------------------------------------------- //
// geometry
// --------------------------------------------------------------------------- //
int axial_geom(const medt_axial_desc& d, AxialGeom* g) {
    if (d.N <= 0 || d.C <= 0 || d.H <= 0 || d.W <= 0 || d.G <= 0 || d.C % d.G) {
        set_error("axial: bad shape N=%d C=%d H=%d W=%d G=%d", d.N, d.C, d.H, d.W, d.G);
        return MEDT_EINVAL;
    }
    if (d.bn_groups < 1 || d.N % d.bn_groups || (d.stride != 1 && d.stride != 2) || (d.axis != 0 && d.axis != 1)) {
        set_error("axial: bad bn_groups=%d / stride=%d / axis=%d", d.bn_groups, d.stride, d.axis);
        return MEDT_EINVAL;
    }
    // every tensor of the layer below 2^31 elements: the kernels index with 32-bit element counts in places, and the size
    // arithmetic of the workspace must not wrap (found by tests/test_abi_fuzz.py: H = 2^30 used to be "accepted")
    if ((double)d.N * 2.0 * d.C * d.H * d.W >= 2147483648.0) {
        set_error("axial: N*2C*H*W = %.3g elements: tensors of 2^31 elements and more are unsupported", (double)d.N * 2.0 * d.C * d.H * d.W);
        return MEDT_EUNSUPPORTED;
    }
    const int gp = d.C / d.G;
    if (gp != 2 && gp != 4 && gp != 8 && gp != 16) {
        set_error("axial: group_planes=%d not in {2,4,8,16}", gp);
        return MEDT_EUNSUPPORTED;
    }
    g->N = d.N; g->C = d.C; g->H = d.H; g->W = d.W; g->G = d.G; g->gp = gp; g->hq = gp / 2;
    g->axis = d.axis; g->pos = d.has_pos ? 1 : 0;
    if (d.act_dtype != 0 && d.act_dtype != 1) { set_error("axial: act_dtype %d unsupported (0 f32, 1 bf16)", d.act_dtype); return MEDT_EUNSUPPORTED; }
    if (d.act_dtype == 1 && !d.has_pos) {
        set_error("axial: bfloat16 storage is implemented for the position-encoded layers only");
        return MEDT_EUNSUPPORTED;
    }
    g->bf16 = d.act_dtype;
    g->L = d.axis ? d.W : d.H;
    g->Bo = d.axis ? d.H : d.W;
    if (g->L > MEDT_THREADS) {
        set_error("axial: sequence length %d > %d unsupported", g->L, MEDT_THREADS);
        return MEDT_EUNSUPPORTED;
    }
    g->OC = g->pos ? 2 * d.C : d.C;
    g->OCg = g->pos ? 2 * gp : gp;
    g->SC = g->pos ? 3 * d.G : d.G;
    g->groups = d.bn_groups;
    g->npg = d.N / d.bn_groups;
    g->spg = g->npg * g->Bo;
    g->S_T = MEDT_THREADS / g->L;
    g->tpg = cdiv(g->spg, g->S_T);
    g->HW = d.H * d.W;
    g->fast3 = 0;
    g->fparts = g->tpg;
    g->nt = 1;
    g->bound_path = 0;
    g->rows4 = 0;
    g->nt4 = 1;
    g->oparts = g->tpg;
    {
        static const float shift = [] { const char* e = getenv("MEDT_DEBUG_BOUND_SHIFT"); return e ? (float)atof(e) : 0.f; }();
        g->bound_shift = shift;
    }
    if (d.gate_mode < 0 || d.gate_mode > 2) { set_error("axial: gate_mode %d unsupported (0 raw, 1 sigmoid, 2 per sequence)", d.gate_mode); return MEDT_EUNSUPPORTED; }
    if (g->pos && fast_path_enabled() && d.gate_mode != 2) {       // per-sequence gates: generic kernels
        int nt = fast3_max_subtiles(gp, g->L, g->axis);
        if (nt > 0) {
            // Persistent workgroups over super-tiles of nt*S_T sequences.  Small problems (the model's own
            // shapes) keep nt = 1 so there are enough workgroups; big ones amortise the table staging over
            // up to NT sub-tiles and cap the grid at ~8 workgroups per CU.
            while (nt > 1 && (long)g->groups * d.G * cdiv(g->spg, g->S_T * nt) < 2048) nt >>= 1;
            static const int env_nt = 0;
            static const int env_cap = 2048;
            if (env_nt > 0 && env_nt < nt) nt = env_nt;
            const int nsup = cdiv(g->spg, g->S_T * nt);
            int cap = env_cap / (g->groups * d.G);
            if (cap < 1) cap = 1;
            g->fast3 = 1;
            g->nt = nt;
            // the bound-referenced kernel saves ~25 % of the issue slots but costs a memset + a repair launch:
            // worth it from ~64M (i,j) pairs per launch (big batches / 256-px inputs), not at bs=4 x 128 px
            static const int force = [] { const char* e = getenv("MEDT_BOUND_PATH"); return e ? atoi(e) : -1; }();
            g->bound_path = force >= 0 ? force : ((double)g->groups * g->spg * d.G * g->L * g->L >= 64e6);
            g->fparts = nsup < cap ? nsup : cap;
            g->oparts = g->fparts;
            // gp = 2: from ~2048 workgroups of the wider tile on, four rows per lane
            static const int force4 = [] { const char* e = getenv("MEDT_ROWS4"); return e ? atoi(e) : -1; }();
            if (gp == 2 && force4 != 0) {
                const int st4 = fast4_subtile_sequences(g->L);
                int nt4 = fast4_max_subtiles(g->axis);
                while (nt4 > 1 && (long)g->groups * d.G * cdiv(g->spg, st4 * nt4) < 2048) nt4 >>= 1;
                const long blocks4 = (long)g->groups * d.G * cdiv(g->spg, st4 * nt4);
                if (force4 == 1 || blocks4 >= 2048) {
                    const int nsup4 = cdiv(g->spg, st4 * nt4);
                    g->rows4 = 1;
                    g->nt4 = nt4;
                    g->oparts = nsup4 < cap ? nsup4 : cap;
                }
            }
        }
    }
    g->sim_count = (double)g->spg * g->L * g->L;
    g->row_count = (double)g->spg * g->L;
    return MEDT_OK;
}

static inline int region_stride(const AxialGeom& g) { return (2 * g.gp + 1) * g.L + 1; }

size_t axial_core_lds_bytes(const AxialGeom& g, bool backward) {
    const int TL = 2 * g.L - 1;
    size_t fl = (size_t)g.S_T * region_stride(g) + 256;                       // qkv region + reduction scratch
    if (g.pos) fl += (size_t)2 * g.gp * TL;                                   // tables
    if (backward) {
        fl += (size_t)g.S_T * (2 * g.gp * g.L + 1);                           // stacked -> dsv|dsve
        fl += (size_t)g.S_T * ((g.gp + 2) * g.L + 1);                         // dy, lse, delta
        if (g.pos) fl += (size_t)2 * g.gp * TL;                               // table-gradient accumulators
    }
    return fl * sizeof(float);
}

// --------------------------------------------------------------------------- //
// forward main pass
// --------------------------------------------------------------------------- //
template <int GP, bool POS, int AXIS>
__global__ __launch_bounds__(MEDT_THREADS) void attn_fwd_kernel(AxialGeom g, const float* __restrict__ qkv_raw,
                                                                BnStats qs, BnStats ss,
                                                                const float* __restrict__ relative, GatePtrs gates,
                                                                float* __restrict__ stacked, float* __restrict__ lse_out,
                                                                float* __restrict__ out_partials) {
    constexpr int HQ = GP / 2, NCH = 2 * GP, OCG = POS ? 2 * GP : GP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = g.L, TL = 2 * L - 1, RS = (NCH + 1) * L + 1;
    float* reg = smem;
    float* red = reg + g.S_T * RS;
    float* tq = red + 256;
    float* tk = tq + HQ * TL;
    float* tv = tk + HQ * TL;
    const int grp = blockIdx.x / g.tpg, tile = blockIdx.x - grp * g.tpg, hg = blockIdx.y;
    TileCtx t{L, g.Bo, g.W, g.HW, grp * g.spg + tile * g.S_T, min(g.S_T, g.spg - tile * g.S_T)};
    tile_load<AXIS>(reg, RS, 0, qkv_raw, 2 * g.C, hg * NCH, NCH, t, g.bf16);
    const int gseq = t.seq0 + min((int)threadIdx.x / L, t.nseq - 1);              // this thread's sequence (per-sequence gates)
    const float f_qr = gate_at(gates.f_qr, gates.stride, gseq), f_kr = gate_at(gates.f_kr, gates.stride, gseq);
    const float f_sve = gate_at(gates.f_sve, gates.stride, gseq), f_sv = gate_at(gates.f_sv, gates.stride, gseq);
    const float a_qk = ss.scale[grp * g.SC + hg] * MEDT_LOG2E;
    const float a_qr = POS ? ss.scale[grp * g.SC + g.G + hg] * f_qr * MEDT_LOG2E : 0.f;
    // the staged Rk table carries bn_similarity's scale (and the gate, when it is one scalar for the whole tile);
    // per-sequence gates multiply the key on use instead
    const float kmul = gates.stride ? f_kr : 1.f;
    const float a_kr = POS ? ss.scale[grp * g.SC + 2 * g.G + hg] * (gates.stride ? 1.f : f_kr) * MEDT_LOG2E : 0.f;
    if (POS) {
        for (int e = threadIdx.x; e < HQ * TL; e += MEDT_THREADS) {
            const int c = e / TL, d = e - c * TL;
            tq[e] = relative[c * TL + d];
            tk[e] = a_kr * relative[(HQ + c) * TL + (TL - 1 - d)];
        }
        for (int e = threadIdx.x; e < GP * TL; e += MEDT_THREADS) tv[e] = relative[GP * TL + e];
    }
    __syncthreads();
    const int ls = threadIdx.x / L, i = threadIdx.x - ls * L;
    const bool active = ls < t.nseq;
    const float* sc = qs.scale + grp * 2 * g.C + hg * NCH;
    const float* sh = qs.shift + grp * 2 * g.C + hg * NCH;
    if (active) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int idx = ls * RS + ch * L + i;
            reg[idx] = fmaf(reg[idx], sc[ch], sh[ch]);
        }
    }
    __syncthreads();
    float outv[OCG];
    float lse = 0.f;
#pragma unroll
    for (int k = 0; k < OCG; ++k) outv[k] = 0.f;
    if (active) {
        float qa[HQ], qb[HQ];
#pragma unroll
        for (int c = 0; c < HQ; ++c) {
            const float q = reg[ls * RS + c * L + i];
            qa[c] = q * a_qk;
            qb[c] = q * a_qr;
        }
        const float* kp = reg + ls * RS + HQ * L;
        const float* vp = reg + ls * RS + GP * L;
        float m = -INFINITY;
        for (int j = 0; j < L; ++j) {
            const int d = i - j + L - 1;
            float z = 0.f;
#pragma unroll
            for (int c = 0; c < HQ; ++c) {
                const float kc = kp[c * L + j];
                z = fmaf(qa[c], kc, z);
                if (POS) {
                    z = fmaf(qb[c], tq[c * TL + d], z);
                    z = fmaf(kc * kmul, tk[c * TL + d], z);
                }
            }
            m = fmaxf(m, z);
        }
        float l = 0.f, accv[GP], acce[GP];
#pragma unroll
        for (int c = 0; c < GP; ++c) { accv[c] = 0.f; acce[c] = 0.f; }
        for (int j = 0; j < L; ++j) {
            const int d = i - j + L - 1;
            float z = 0.f;
#pragma unroll
            for (int c = 0; c < HQ; ++c) {
                const float kc = kp[c * L + j];
                z = fmaf(qa[c], kc, z);
                if (POS) {
                    z = fmaf(qb[c], tq[c * TL + d], z);
                    z = fmaf(kc * kmul, tk[c * TL + d], z);
                }
            }
            const float p = __builtin_amdgcn_exp2f(z - m);
            l += p;
#pragma unroll
            for (int c = 0; c < GP; ++c) {
                accv[c] = fmaf(p, vp[c * L + j], accv[c]);
                if (POS) acce[c] = fmaf(p, tv[c * TL + d], acce[c]);
            }
        }
        const float inv = 1.f / l;
        lse = m + __log2f(l);
#pragma unroll
        for (int c = 0; c < GP; ++c) {
            if (POS) {
                outv[2 * c] = f_sv * accv[c] * inv;
                outv[2 * c + 1] = f_sve * acce[c] * inv;
            } else {
                outv[c] = accv[c] * inv;
            }
        }
    }
    __syncthreads();                                   // all reads of reg done
    if (active) {
#pragma unroll
        for (int k = 0; k < OCG; ++k) reg[ls * RS + k * L + i] = outv[k];
        reg[ls * RS + NCH * L + i] = lse;
    }
    __syncthreads();
    tile_store<AXIS>(reg, RS, 0, stacked, g.OC, hg * OCG, OCG, t, g.bf16);
    if (lse_out) tile_store<AXIS>(reg, RS, NCH, lse_out, g.G, hg, 1, t);
    if (out_partials) {
        float v[2 * OCG];
#pragma unroll
        for (int k = 0; k < OCG; ++k) { v[2 * k] = outv[k]; v[2 * k + 1] = outv[k] * outv[k]; }
        block_sum_d<2 * OCG>(v, red, reinterpret_cast<double*>(out_partials) + ((size_t)blockIdx.x * g.OC + hg * OCG) * 2);
    }
}

// --------------------------------------------------------------------------- //
// backward: shared staging
// --------------------------------------------------------------------------- //
// LDS map (floats):  reg  [S_T][RS]      q|k|v normalised (+1 spare channel)
//                    red  [256]
//                    tq|tk|tv            raw tables, tk reversed        (POS)
//                    dtq|dtk|dtv         table-gradient accumulators    (POS, pass B)
//                    g2   [S_T][R2]      stacked -> dsv|dsve (gradient wrt stacked, pre bn_output)
//                    g3   [S_T][R3]      dy (gp ch) | lse | delta
template <int GP, bool POS>
struct BwdLds {
    static constexpr int HQ = GP / 2, NCH = 2 * GP, OCG = POS ? 2 * GP : GP;
    float *reg, *red, *tq, *tk, *tv, *dtq, *dtk, *dtv, *g2, *g3;
    int RS, R2, R3, TL;
    __device__ BwdLds(float* smem, const AxialGeom& g, int L) {
        TL = 2 * L - 1;
        RS = (NCH + 1) * L + 1;
        R2 = NCH * L + 1;
        R3 = (GP + 2) * L + 1;
        reg = smem;
        red = reg + g.S_T * RS;
        float* p = red + 256;
        tq = p; tk = tq + HQ * TL; tv = tk + HQ * TL;
        if (POS) p = tv + GP * TL;
        dtq = p; dtk = dtq + HQ * TL; dtv = dtk + HQ * TL;
        if (POS) p = dtv + GP * TL;
        g2 = p;
        g3 = g2 + g.S_T * R2;
    }
};

// Stage everything the two backward passes need; on return (after the trailing barrier):
//   reg = normalised q|k|v, g2 = dsv|dsve (wrt gated stacked values), g3 = [dy.. | lse | delta]
// rawv[] receives this thread's own raw qkv elements (position idx of its sequence).
template <int GP, bool POS, int AXIS>
__device__ __forceinline__ void bwd_stage(const AxialGeom& g, BwdLds<GP, POS>& S, const TileCtx& t, int grp, int hg,
                                          const float* __restrict__ qkv_raw, const BnStats& qs,
                                          const float* __restrict__ relative, const float* __restrict__ stacked,
                                          const float* __restrict__ lse, const float* __restrict__ dy,
                                          const float* __restrict__ out_coef, int pool, float (&rawv)[2 * GP], int L) {
    constexpr int HQ = GP / 2, NCH = 2 * GP, OCG = POS ? 2 * GP : GP;
    const int TL = S.TL;
    tile_load<AXIS>(S.reg, S.RS, 0, qkv_raw, 2 * g.C, hg * NCH, NCH, t, g.bf16);
    tile_load<AXIS>(S.g2, S.R2, 0, stacked, g.OC, hg * OCG, OCG, t, g.bf16);
    tile_load_pooled<AXIS>(S.g3, S.R3, 0, dy, g.C, hg * GP, GP, g.H, pool, t);
    tile_load<AXIS>(S.g3, S.R3, GP, lse, g.G, hg, 1, t);
    if (POS) {
        for (int e = threadIdx.x; e < HQ * TL; e += MEDT_THREADS) {
            const int c = e / TL, d = e - c * TL;
            S.tq[e] = relative[c * TL + d];
            S.tk[e] = relative[(HQ + c) * TL + (TL - 1 - d)];
        }
        for (int e = threadIdx.x; e < GP * TL; e += MEDT_THREADS) S.tv[e] = relative[GP * TL + e];
    }
    __syncthreads();
    const int ls = threadIdx.x / L, i = threadIdx.x - ls * L;
    if (ls < t.nseq) {
        const float* sc = qs.scale + grp * 2 * g.C + hg * NCH;
        const float* sh = qs.shift + grp * 2 * g.C + hg * NCH;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int idx = ls * S.RS + ch * L + i;
            rawv[ch] = S.reg[idx];
            S.reg[idx] = fmaf(rawv[ch], sc[ch], sh[ch]);
        }
        // gradient wrt the stacked (pre-bn_output) values: c0*dy + c1*stacked + c2 ; delta = sum dstk*stk
        const float* cf = out_coef + ((size_t)grp * g.OC + hg * OCG) * 3;
        float delta = 0.f;
#pragma unroll
        for (int k = 0; k < OCG; ++k) {
            const int idx = ls * S.R2 + k * L + i;
            const float sv = S.g2[idx];
            const float d = S.g3[ls * S.R3 + (POS ? (k >> 1) : k) * L + i];
            const float ds = fmaf(cf[k * 3 + 0], d, fmaf(cf[k * 3 + 1], sv, cf[k * 3 + 2]));
            delta = fmaf(ds, sv, delta);
            S.g2[idx] = ds;
        }
        S.g3[ls * S.R3 + (GP + 1) * L + i] = delta;
    } else {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) rawv[ch] = 0.f;
    }
    __syncthreads();
}

// Sum over the L lanes of a sequence (L a power of two <= 64, segments aligned to L), result in every lane.
// 16-lane rows are all-reduced with DPP row rotations; rows are combined through SGPRs (v_readlane): no LDS traffic.
__device__ __forceinline__ float seg_allsum(float v, int L) {
    if (L >= 16) {
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
        if (L == 16) return v;
        const int vi = __float_as_int(v);
        const float r0 = __int_as_float(__builtin_amdgcn_readlane(vi, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(vi, 16));
        const float r2 = __int_as_float(__builtin_amdgcn_readlane(vi, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(vi, 48));
        if (L == 64) return (r0 + r1) + (r2 + r3);
        return (threadIdx.x & 32) ? r2 + r3 : r0 + r1;                                                    // L == 32
    }
    for (int o = L >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Everything one (i, j) pair contributes, shared by the row- and column-oriented loops.
struct PairTerms {
    float tqk, rq, rk;      // sum_c q k ; sum_c q Rq[d] ; sum_c k Rk[d']   (ungated)
    float P, dPv, dPe, dZ;
};

// --------------------------------------------------------------------------- //
// backward pass A: sum over (b,i,j) of dZ * {1, S_qk, S_qr, S_kr} per head
// --------------------------------------------------------------------------- //
// LC: compile-time sequence length (0 = runtime g.L): index arithmetic folds into immediates and the sweeps unroll
template <int GP, bool POS, int AXIS, int LC = 0>
__global__ __launch_bounds__(MEDT_THREADS) void attn_bwd_stats_kernel(
    AxialGeom g, const float* __restrict__ qkv_raw, BnStats qs, BnStats ss, const float* __restrict__ relative,
    GatePtrs gates, const float* __restrict__ stacked, const float* __restrict__ lse, const float* __restrict__ dy,
    const float* __restrict__ out_coef, int pool, float* __restrict__ partials) {
    constexpr int HQ = GP / 2, NCH = 2 * GP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = LC ? LC : g.L;
    BwdLds<GP, POS> S(smem, g, L);
    const int TL = S.TL;
    const int grp = blockIdx.x / g.tpg, tile = blockIdx.x - grp * g.tpg, hg = blockIdx.y;
    TileCtx t{L, g.Bo, g.W, g.HW, grp * g.spg + tile * g.S_T, min(g.S_T, g.spg - tile * g.S_T)};
    float rawv[NCH];
    bwd_stage<GP, POS, AXIS>(g, S, t, grp, hg, qkv_raw, qs, relative, stacked, lse, dy, out_coef, pool, rawv, L);
    const int gseq = t.seq0 + min((int)threadIdx.x / L, t.nseq - 1);              // this thread's sequence (per-sequence gates)
    const float f_qr = gate_at(gates.f_qr, gates.stride, gseq), f_kr = gate_at(gates.f_kr, gates.stride, gseq);
    const float f_sve = gate_at(gates.f_sve, gates.stride, gseq), f_sv = gate_at(gates.f_sv, gates.stride, gseq);
    const float e_qk = ss.scale[grp * g.SC + hg] * MEDT_LOG2E;
    const float e_qr = POS ? ss.scale[grp * g.SC + g.G + hg] * MEDT_LOG2E : 0.f;
    const float e_kr = POS ? ss.scale[grp * g.SC + 2 * g.G + hg] * MEDT_LOG2E : 0.f;
    const int ls = threadIdx.x / L, i = threadIdx.x - ls * L;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (ls < t.nseq) {
        float q[HQ], dsv[GP], dse[GP];
#pragma unroll
        for (int c = 0; c < HQ; ++c) q[c] = S.reg[ls * S.RS + c * L + i];
#pragma unroll
        for (int c = 0; c < GP; ++c) {
            dsv[c] = S.g2[ls * S.R2 + (POS ? 2 * c : c) * L + i];
            dse[c] = POS ? S.g2[ls * S.R2 + (2 * c + 1) * L + i] : 0.f;
        }
        const float lse_i = S.g3[ls * S.R3 + GP * L + i], delta = S.g3[ls * S.R3 + (GP + 1) * L + i];
        const float* kp = S.reg + ls * S.RS + HQ * L;
        const float* vp = S.reg + ls * S.RS + GP * L;
#pragma unroll 4
        for (int j = 0; j < L; ++j) {
            const int d = i - j + L - 1;
            float tqk = 0.f, rq = 0.f, rk = 0.f;
#pragma unroll
            for (int c = 0; c < HQ; ++c) {
                const float kc = kp[c * L + j];
                tqk = fmaf(q[c], kc, tqk);
                if (POS) {
                    rq = fmaf(q[c], S.tq[c * TL + d], rq);
                    rk = fmaf(kc, S.tk[c * TL + d], rk);
                }
            }
            const float tqr = f_qr * rq, tkr = f_kr * rk;
            const float z = fmaf(e_qk, tqk, fmaf(e_qr, tqr, e_kr * tkr));
            const float P = __builtin_amdgcn_exp2f(z - lse_i);
            float dPv = 0.f, dPe = 0.f;
#pragma unroll
            for (int c = 0; c < GP; ++c) {
                dPv = fmaf(dsv[c], vp[c * L + j], dPv);
                if (POS) dPe = fmaf(dse[c], S.tv[c * TL + d], dPe);
            }
            const float dP = POS ? fmaf(f_sv, dPv, f_sve * dPe) : dPv;
            const float dZ = P * (dP - delta);
            acc[0] += dZ;
            acc[1] = fmaf(dZ, tqk, acc[1]);
            if (POS) {
                acc[2] = fmaf(dZ, tqr, acc[2]);
                acc[3] = fmaf(dZ, tkr, acc[3]);
            }
        }
    }
    block_sum<4>(acc, S.red, partials + ((size_t)blockIdx.x * g.G + hg) * 4);
}

// (sim_coef -- coef[grp][SC][3] = (e, u, w) with dS_x = e*dZ + u*S_x + w: fin_inline.h)
__global__ __launch_bounds__(64) void sim_bwd_finalize_kernel(const float* __restrict__ partials, int tpg, int groups,
                                                              int G, int SC, double count, BnStats ss,
                                                              const float* __restrict__ weight, int training,
                                                              float* __restrict__ coef, float* __restrict__ dweight,
                                                              float* __restrict__ dbias, TablesJob tj) {
    if ((int)blockIdx.x >= SC) {         // appended blocks: sliding-window table sums for the fix kernel of axial_bwd.hip
        MEDT_STATIC_SHARED float lds[512];
        sim_tables_block(blockIdx.x - SC, tj.relative, tj.tables, tj.HQ, tj.L, lds);
        return;
    }
    const int ch = blockIdx.x, lane = threadIdx.x;          // ch = x*G + hg
    const int x = ch / G, hg = ch - x * G;
    double dg = 0.0, db = 0.0;
    if (groups <= 64 && !(groups & (groups - 1))) {
        const int slots = 64 / groups, g = lane % groups, slot = lane / groups;
        double a0 = 0.0, ax = 0.0;
        for (int p = slot; p < tpg; p += slots) {
            const float* q = partials + ((size_t)(g * tpg + p) * G + hg) * 4;
            a0 += (double)q[0];
            ax += (double)q[1 + x];
        }
        for (int o = groups; o < 64; o <<= 1) { a0 += __shfl_xor(a0, o, 64); ax += __shfl_xor(ax, o, 64); }
        const int gi = lane < groups ? lane : 0;
        const double mean = ss.mean[gi * SC + ch], rstd = ss.rstd[gi * SC + ch];
        const double sxh = rstd * (ax - mean * a0);          // sum dZ * xhat of this lane's group
        if (lane < groups) sim_coef(a0, sxh, count, mean, rstd, weight[ch], training, coef + ((size_t)lane * SC + ch) * 3);
        dg = lane < groups ? sxh : 0.0;
        db = lane < groups ? a0 : 0.0;
        for (int o = 32; o > 0; o >>= 1) { dg += __shfl_xor(dg, o, 64); db += __shfl_xor(db, o, 64); }
    } else {
        for (int grp = 0; grp < groups; ++grp) {
            double a0 = 0.0, ax = 0.0;
            for (int p = lane; p < tpg; p += 64) {
                const float* q = partials + ((size_t)(grp * tpg + p) * G + hg) * 4;
                a0 += (double)q[0];
                ax += (double)q[1 + x];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o, 64); ax += __shfl_xor(ax, o, 64); }
            const double mean = ss.mean[grp * SC + ch], rstd = ss.rstd[grp * SC + ch];
            const double sxh = rstd * (ax - mean * a0);
            dg += sxh;
            db += a0;
            if (lane == 0) sim_coef(a0, sxh, count, mean, rstd, weight[ch], training, coef + ((size_t)grp * SC + ch) * 3);
        }
    }
    if (lane == 0) {
        dweight[ch] = (float)dg;
        dbias[ch] = (float)db;
    }
}

int axial_sim_bwd_finalize(const AxialGeom& g, const float* partials, BnStats sim, const float* weight, int training,
                           float* coef, float* dweight, float* dbias, hipStream_t s, const TablesJob* tables) {
    const TablesJob tj = tables ? *tables : TablesJob{nullptr, nullptr, 0, 0, 0};
    hipLaunchKernelGGL(sim_bwd_finalize_kernel, dim3(g.SC + tj.blocks), dim3(64), 0, s, partials, g.tpg, g.groups, g.G, g.SC,
                       g.sim_count, sim, weight, training, coef, dweight, dbias, tj);
    return launch_status("sim_bwd_finalize");
}

// --------------------------------------------------------------------------- //
// backward pass B: dq (row-oriented), dk / dv (column-oriented), table and gate gradients
// --------------------------------------------------------------------------- //
template <int GP, bool POS, int AXIS, int LC = 0>
__global__ __launch_bounds__(MEDT_THREADS) void attn_bwd_kernel(
    AxialGeom g, const float* __restrict__ qkv_raw, BnStats qs, BnStats ss, const float* __restrict__ sim_coef,
    const float* __restrict__ relative, GatePtrs gates, const float* __restrict__ stacked,
    const float* __restrict__ lse, const float* __restrict__ dy, const float* __restrict__ out_coef, int pool,
    float* __restrict__ dqkv, float* __restrict__ qkv_partials, float* __restrict__ rel_partials,
    float* __restrict__ gate_partials) {
    constexpr int HQ = GP / 2, NCH = 2 * GP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = LC ? LC : g.L;
    BwdLds<GP, POS> S(smem, g, L);
    const int TL = S.TL;
    const int grp = blockIdx.x / g.tpg, tile = blockIdx.x - grp * g.tpg, hg = blockIdx.y;
    TileCtx t{L, g.Bo, g.W, g.HW, grp * g.spg + tile * g.S_T, min(g.S_T, g.spg - tile * g.S_T)};
    if (POS)
        for (int e = threadIdx.x; e < NCH * TL; e += MEDT_THREADS) S.dtq[e] = 0.f;      // dtq|dtk|dtv contiguous
    float rawv[NCH];
    bwd_stage<GP, POS, AXIS>(g, S, t, grp, hg, qkv_raw, qs, relative, stacked, lse, dy, out_coef, pool, rawv, L);
    const int gseq = t.seq0 + min((int)threadIdx.x / L, t.nseq - 1);              // this thread's sequence (per-sequence gates)
    const float f_qr = gate_at(gates.f_qr, gates.stride, gseq), f_kr = gate_at(gates.f_kr, gates.stride, gseq);
    const float f_sve = gate_at(gates.f_sve, gates.stride, gseq), f_sv = gate_at(gates.f_sv, gates.stride, gseq);
    const float* cq = sim_coef + ((size_t)grp * g.SC + hg) * 3;
    const float* cr = sim_coef + ((size_t)grp * g.SC + g.G + hg) * 3;
    const float* ck = sim_coef + ((size_t)grp * g.SC + 2 * g.G + hg) * 3;
    const float b_qk = cq[0], u_qk = cq[1], w_qk = cq[2];
    const float b_qr = POS ? cr[0] : 0.f, u_qr = POS ? cr[1] : 0.f, w_qr = POS ? cr[2] : 0.f;
    const float b_kr = POS ? ck[0] : 0.f, u_kr = POS ? ck[1] : 0.f, w_kr = POS ? ck[2] : 0.f;
    const float e_qk = b_qk * MEDT_LOG2E, e_qr = b_qr * MEDT_LOG2E, e_kr = b_kr * MEDT_LOG2E;
    const int ls = threadIdx.x / L, idx = threadIdx.x - ls * L;
    const bool active = ls < t.nseq;
    // Wrapped-diagonal second pass (no LDS float atomics in the inner loop, which run ~1 lane/clk on gfx950):
    // needs the L lanes of a sequence inside one wave so column accumulators can rotate between lanes.
    const bool diag = POS && L <= 64 && (L & (L - 1)) == 0;
    float dqkv_v[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) dqkv_v[ch] = 0.f;
    float gacc[4] = {0.f, 0.f, 0.f, 0.f};       // f_qr, f_kr, f_sve, f_sv
    if (active) {
        const float* qp = S.reg + ls * S.RS;
        const float* kp = qp + HQ * L;
        const float* vp = qp + GP * L;
        const float* g2 = S.g2 + ls * S.R2;
        const float* lsep = S.g3 + ls * S.R3 + GP * L;
        const float* delp = lsep + L;
        // ---------------- row-oriented: thread owns query row i = idx ----------------
        // (only when the wrapped-diagonal sweep below cannot run: it produces dq and the gate gradients as well)
        if (!diag) {
            const int i = idx;
            float q[HQ], dsv[GP], dse[GP], dq[HQ];
#pragma unroll
            for (int c = 0; c < HQ; ++c) { q[c] = qp[c * L + i]; dq[c] = 0.f; }
#pragma unroll
            for (int c = 0; c < GP; ++c) {
                dsv[c] = g2[(POS ? 2 * c : c) * L + i];
                dse[c] = POS ? g2[(2 * c + 1) * L + i] : 0.f;
            }
            const float lse_i = lsep[i], delta = delp[i];
            for (int j = 0; j < L; ++j) {
                const int d = i - j + L - 1;
                float tqk = 0.f, rq = 0.f, rk = 0.f;
#pragma unroll
                for (int c = 0; c < HQ; ++c) {
                    const float kc = kp[c * L + j];
                    tqk = fmaf(q[c], kc, tqk);
                    if (POS) {
                        rq = fmaf(q[c], S.tq[c * TL + d], rq);
                        rk = fmaf(kc, S.tk[c * TL + d], rk);
                    }
                }
                const float tqr = f_qr * rq, tkr = f_kr * rk;
                const float z = fmaf(e_qk, tqk, fmaf(e_qr, tqr, e_kr * tkr));
                const float P = __builtin_amdgcn_exp2f(z - lse_i);
                float dPv = 0.f, dPe = 0.f;
#pragma unroll
                for (int c = 0; c < GP; ++c) {
                    dPv = fmaf(dsv[c], vp[c * L + j], dPv);
                    if (POS) dPe = fmaf(dse[c], S.tv[c * TL + d], dPe);
                }
                const float dP = POS ? fmaf(f_sv, dPv, f_sve * dPe) : dPv;
                const float dZ = P * (dP - delta);
                const float dSqk = fmaf(b_qk, dZ, fmaf(u_qk, tqk, w_qk));
                if (POS) {
                    const float dSqr = fmaf(b_qr, dZ, fmaf(u_qr, tqr, w_qr));
                    const float dSkr = fmaf(b_kr, dZ, fmaf(u_kr, tkr, w_kr));
                    const float gq = f_qr * dSqr, gk = f_kr * dSkr, gv = f_sve * P;
                    gacc[0] = fmaf(dSqr, rq, gacc[0]);
                    gacc[1] = fmaf(dSkr, rk, gacc[1]);
                    gacc[2] = fmaf(P, dPe, gacc[2]);
                    gacc[3] = fmaf(P, dPv, gacc[3]);
#pragma unroll
                    for (int c = 0; c < HQ; ++c) {
                        const float kc = kp[c * L + j];
                        dq[c] = fmaf(dSqk, kc, fmaf(gq, S.tq[c * TL + d], dq[c]));
                        if (!diag) {
                            atomicAdd(&S.dtq[c * TL + d], gq * q[c]);
                            atomicAdd(&S.dtk[c * TL + d], gk * kc);
                        }
                    }
                    if (!diag) {
#pragma unroll
                        for (int c = 0; c < GP; ++c) atomicAdd(&S.dtv[c * TL + d], gv * dse[c]);
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < HQ; ++c) dq[c] = fmaf(dSqk, kp[c * L + j], dq[c]);
                }
            }
#pragma unroll
            for (int c = 0; c < HQ; ++c) dqkv_v[c] = dq[c];
        }
        // ---------------- wrapped diagonals: lane delta visits (i, j = (i - delta) mod L) ----------------
        // All pairs a lane visits share one of two table entries (d = delta+L-1 while i >= delta, delta-1
        // after the wrap), so the relative-table gradients accumulate in lane-private registers; the dk/dv
        // accumulators belong to a column and hop to the next lane every step (one cross-lane move each).
        if (diag) {
            const int dl = idx, Lm = L - 1;
            const int d_hi = dl + L - 1, d_lo = dl > 0 ? dl - 1 : 0;
            const int src_lane = (threadIdx.x & 63 & ~Lm) | ((dl - 1) & Lm);
            float tq_hi[HQ], tq_lo[HQ], tk_hi[HQ], tk_lo[HQ], tv_hi[GP], tv_lo[GP];
            float aq_hi[HQ], aq_lo[HQ], ak_hi[HQ], ak_lo[HQ], av_hi[GP], av_lo[GP];
            float dk[HQ], dv[GP], dq_d[HQ];
#pragma unroll
            for (int c = 0; c < HQ; ++c) {
                dq_d[c] = 0.f;
                tq_hi[c] = S.tq[c * TL + d_hi]; tq_lo[c] = S.tq[c * TL + d_lo];
                tk_hi[c] = S.tk[c * TL + d_hi]; tk_lo[c] = S.tk[c * TL + d_lo];
                aq_hi[c] = aq_lo[c] = ak_hi[c] = ak_lo[c] = 0.f;
                dk[c] = 0.f;
            }
#pragma unroll
            for (int c = 0; c < GP; ++c) {
                tv_hi[c] = S.tv[c * TL + d_hi]; tv_lo[c] = S.tv[c * TL + d_lo];
                av_hi[c] = av_lo[c] = 0.f;
                dv[c] = 0.f;
            }
#pragma unroll 2
            for (int i = 0; i < L; ++i) {
                const int j = (i - dl) & Lm;
                const bool hi = i >= dl;
                float tqk = 0.f, rq = 0.f, rk = 0.f, kj[HQ], qi[HQ];
#pragma unroll
                for (int c = 0; c < HQ; ++c) {
                    qi[c] = qp[c * L + i];
                    kj[c] = kp[c * L + j];
                    tqk = fmaf(qi[c], kj[c], tqk);
                    rq = fmaf(qi[c], hi ? tq_hi[c] : tq_lo[c], rq);
                    rk = fmaf(kj[c], hi ? tk_hi[c] : tk_lo[c], rk);
                }
                const float tqr = f_qr * rq, tkr = f_kr * rk;
                const float z = fmaf(e_qk, tqk, fmaf(e_qr, tqr, e_kr * tkr));
                const float P = __builtin_amdgcn_exp2f(z - lsep[i]);
                float dPv = 0.f, dPe = 0.f, dsv_i[GP], dse_i[GP];
#pragma unroll
                for (int c = 0; c < GP; ++c) {
                    dsv_i[c] = g2[(2 * c) * L + i];
                    dse_i[c] = g2[(2 * c + 1) * L + i];
                    dPv = fmaf(dsv_i[c], vp[c * L + j], dPv);
                    dPe = fmaf(dse_i[c], hi ? tv_hi[c] : tv_lo[c], dPe);
                }
                const float dZ = P * (fmaf(f_sv, dPv, f_sve * dPe) - delp[i]);
                const float dSqk = fmaf(b_qk, dZ, fmaf(u_qk, tqk, w_qk));
                const float dSqr = fmaf(b_qr, dZ, fmaf(u_qr, tqr, w_qr)), dSkr = fmaf(b_kr, dZ, fmaf(u_kr, tkr, w_kr));
                const float gq = f_qr * dSqr, gk = f_kr * dSkr;
                const float gv = f_sve * P, pv = f_sv * P;
                // gate gradients: every (i, j) pair is visited exactly once by this sweep
                gacc[0] = fmaf(dSqr, rq, gacc[0]);
                gacc[1] = fmaf(dSkr, rk, gacc[1]);
                gacc[2] = fmaf(P, dPe, gacc[2]);
                gacc[3] = fmaf(P, dPv, gacc[3]);
                // dq of row i: every lane of the sequence holds one column's term -> sum over the L lanes, kept by lane i
#pragma unroll
                for (int c = 0; c < HQ; ++c) {
                    float t = fmaf(dSqk, kj[c], gq * (hi ? tq_hi[c] : tq_lo[c]));
                    t = seg_allsum(t, L);
                    if (dl == i) dq_d[c] = t;
                }
#pragma unroll
                for (int c = 0; c < HQ; ++c) {
                    dk[c] = fmaf(dSqk, qi[c], fmaf(gk, hi ? tk_hi[c] : tk_lo[c], dk[c]));
                    const float cq = gq * qi[c], ck2 = gk * kj[c];
                    aq_hi[c] += hi ? cq : 0.f;  aq_lo[c] += hi ? 0.f : cq;
                    ak_hi[c] += hi ? ck2 : 0.f; ak_lo[c] += hi ? 0.f : ck2;
                }
#pragma unroll
                for (int c = 0; c < GP; ++c) {
                    dv[c] = fmaf(pv, dsv_i[c], dv[c]);
                    const float cv = gv * dse_i[c];
                    av_hi[c] += hi ? cv : 0.f;  av_lo[c] += hi ? 0.f : cv;
                }
                if (i + 1 < L) {                          // hand the column accumulators to the next lane
#pragma unroll
                    for (int c = 0; c < HQ; ++c) dk[c] = __shfl(dk[c], src_lane, 64);
#pragma unroll
                    for (int c = 0; c < GP; ++c) dv[c] = __shfl(dv[c], src_lane, 64);
                }
            }
#pragma unroll
            for (int c = 0; c < HQ; ++c) dqkv_v[c] = dq_d[c];
            // after step L-1 lane delta holds column (L-1-delta) mod L: park the values, the owner lane reads them
            const int owner = (threadIdx.x & 63 & ~Lm) | ((L - 1 - dl) & Lm);
#pragma unroll
            for (int c = 0; c < HQ; ++c) dqkv_v[HQ + c] = __shfl(dk[c], owner, 64);
#pragma unroll
            for (int c = 0; c < GP; ++c) dqkv_v[GP + c] = __shfl(dv[c], owner, 64);
#pragma unroll
            for (int c = 0; c < HQ; ++c) {
                atomicAdd(&S.dtq[c * TL + d_hi], aq_hi[c]);
                atomicAdd(&S.dtk[c * TL + d_hi], ak_hi[c]);
                if (dl > 0) {
                    atomicAdd(&S.dtq[c * TL + d_lo], aq_lo[c]);
                    atomicAdd(&S.dtk[c * TL + d_lo], ak_lo[c]);
                }
            }
#pragma unroll
            for (int c = 0; c < GP; ++c) {
                atomicAdd(&S.dtv[c * TL + d_hi], av_hi[c]);
                if (dl > 0) atomicAdd(&S.dtv[c * TL + d_lo], av_lo[c]);
            }
        } else
        // ---------------- column-oriented: thread owns key column j = idx ----------------
        {
            const int j = idx;
            float k[HQ], v[GP], dk[HQ], dv[GP];
#pragma unroll
            for (int c = 0; c < HQ; ++c) { k[c] = kp[c * L + j]; dk[c] = 0.f; }
#pragma unroll
            for (int c = 0; c < GP; ++c) { v[c] = vp[c * L + j]; dv[c] = 0.f; }
            for (int i = 0; i < L; ++i) {
                const int d = i - j + L - 1;
                float tqk = 0.f, rq = 0.f, rk = 0.f;
#pragma unroll
                for (int c = 0; c < HQ; ++c) {
                    const float qc = qp[c * L + i];
                    tqk = fmaf(qc, k[c], tqk);
                    if (POS) {
                        rq = fmaf(qc, S.tq[c * TL + d], rq);
                        rk = fmaf(k[c], S.tk[c * TL + d], rk);
                    }
                }
                const float tqr = f_qr * rq, tkr = f_kr * rk;
                const float z = fmaf(e_qk, tqk, fmaf(e_qr, tqr, e_kr * tkr));
                const float P = __builtin_amdgcn_exp2f(z - lsep[i]);
                float dPv = 0.f, dPe = 0.f;
#pragma unroll
                for (int c = 0; c < GP; ++c) {
                    dPv = fmaf(g2[(POS ? 2 * c : c) * L + i], v[c], dPv);
                    if (POS) dPe = fmaf(g2[(2 * c + 1) * L + i], S.tv[c * TL + d], dPe);
                }
                const float dP = POS ? fmaf(f_sv, dPv, f_sve * dPe) : dPv;
                const float dZ = P * (dP - delp[i]);
                const float dSqk = fmaf(b_qk, dZ, fmaf(u_qk, tqk, w_qk));
                const float gk = POS ? f_kr * fmaf(b_kr, dZ, fmaf(u_kr, tkr, w_kr)) : 0.f;
                const float pv = (POS ? f_sv : 1.f) * P;
#pragma unroll
                for (int c = 0; c < HQ; ++c) {
                    dk[c] = fmaf(dSqk, qp[c * L + i], dk[c]);
                    if (POS) dk[c] = fmaf(gk, S.tk[c * TL + d], dk[c]);
                }
#pragma unroll
                for (int c = 0; c < GP; ++c) dv[c] = fmaf(pv, g2[(POS ? 2 * c : c) * L + i], dv[c]);
            }
#pragma unroll
            for (int c = 0; c < HQ; ++c) dqkv_v[HQ + c] = dk[c];
#pragma unroll
            for (int c = 0; c < GP; ++c) dqkv_v[GP + c] = dv[c];
        }
    }
    __syncthreads();                                    // all reads of reg / tables / atomics done
    if (active) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) S.reg[ls * S.RS + ch * L + idx] = dqkv_v[ch];
    }
    __syncthreads();
    tile_store<AXIS>(S.reg, S.RS, 0, dqkv, 2 * g.C, hg * NCH, NCH, t);
    const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    if (POS) {
        float* rp = rel_partials + blk * NCH * TL;
        for (int e = threadIdx.x; e < HQ * TL; e += MEDT_THREADS) {
            const int c = e / TL, d = e - c * TL;
            rp[e] = S.dtq[e];
            rp[(HQ + c) * TL + d] = S.dtk[c * TL + (TL - 1 - d)];
        }
        for (int e = threadIdx.x; e < GP * TL; e += MEDT_THREADS) rp[GP * TL + e] = S.dtv[e];
        if (gate_partials) {
            if (gates.stride == 0) {
                block_sum<4>(gacc, S.red, gate_partials + blk * 4);
            } else {                                 // per-sequence gates: one row of 4 per (sequence, head)
                __syncthreads();
                for (int e = threadIdx.x; e < g.S_T * 4; e += MEDT_THREADS) S.red[e] = 0.f;
                __syncthreads();
                if (active) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) atomicAdd(&S.red[ls * 4 + k], gacc[k]);
                }
                __syncthreads();
                for (int e = threadIdx.x; e < t.nseq * 4; e += MEDT_THREADS)
                    gate_partials[((size_t)(t.seq0 + (e >> 2)) * g.G + hg) * 4 + (e & 3)] = S.red[e];
            }
        }
    }
    // bn_qkv backward statistics: sum d, sum d * xhat  per channel of this head
    {
        const float* mean = qs.mean + grp * 2 * g.C + hg * NCH;
        const float* rstd = qs.rstd + grp * 2 * g.C + hg * NCH;
        float v[2 * NCH];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            v[2 * ch] = dqkv_v[ch];
            v[2 * ch + 1] = dqkv_v[ch] * ((rawv[ch] - mean[ch]) * rstd[ch]);
        }
        block_sum<2 * NCH>(v, S.red, qkv_partials + ((size_t)blockIdx.x * 2 * g.C + hg * NCH) * 2);
    }
}

// --------------------------------------------------------------------------- //
// launchers
// --------------------------------------------------------------------------- //
#define MEDT_DISPATCH(KERNEL, ...)                                                                            \
    do {                                                                                                      \
        const dim3 grid(g.groups * g.tpg, g.G), block(MEDT_THREADS);                                          \
        if (lds > 160 * 1024) { set_error(#KERNEL ": needs %zu B of LDS", lds); return MEDT_EUNSUPPORTED; }    \
        switch (g.gp * 4 + g.pos * 2 + g.axis) {                                                              \
            case 2 * 4 + 0: hipLaunchKernelGGL((KERNEL<2, false, 0>), grid, block, lds, s, __VA_ARGS__); break;  \
            case 2 * 4 + 1: hipLaunchKernelGGL((KERNEL<2, false, 1>), grid, block, lds, s, __VA_ARGS__); break;  \
            case 2 * 4 + 2: hipLaunchKernelGGL((KERNEL<2, true, 0>), grid, block, lds, s, __VA_ARGS__); break;   \
            case 2 * 4 + 3: hipLaunchKernelGGL((KERNEL<2, true, 1>), grid, block, lds, s, __VA_ARGS__); break;   \
            case 4 * 4 + 0: hipLaunchKernelGGL((KERNEL<4, false, 0>), grid, block, lds, s, __VA_ARGS__); break;  \
            case 4 * 4 + 1: hipLaunchKernelGGL((KERNEL<4, false, 1>), grid, block, lds, s, __VA_ARGS__); break;  \
            case 4 * 4 + 2: hipLaunchKernelGGL((KERNEL<4, true, 0>), grid, block, lds, s, __VA_ARGS__); break;   \
            case 4 * 4 + 3: hipLaunchKernelGGL((KERNEL<4, true, 1>), grid, block, lds, s, __VA_ARGS__); break;   \
            case 8 * 4 + 0: hipLaunchKernelGGL((KERNEL<8, false, 0>), grid, block, lds, s, __VA_ARGS__); break;  \
            case 8 * 4 + 1: hipLaunchKernelGGL((KERNEL<8, false, 1>), grid, block, lds, s, __VA_ARGS__); break;  \
            case 8 * 4 + 2: hipLaunchKernelGGL((KERNEL<8, true, 0>), grid, block, lds, s, __VA_ARGS__); break;   \
            case 8 * 4 + 3: hipLaunchKernelGGL((KERNEL<8, true, 1>), grid, block, lds, s, __VA_ARGS__); break;   \
            case 16 * 4 + 0: hipLaunchKernelGGL((KERNEL<16, false, 0>), grid, block, lds, s, __VA_ARGS__); break; \
            case 16 * 4 + 1: hipLaunchKernelGGL((KERNEL<16, false, 1>), grid, block, lds, s, __VA_ARGS__); break; \
            case 16 * 4 + 2: hipLaunchKernelGGL((KERNEL<16, true, 0>), grid, block, lds, s, __VA_ARGS__); break;  \
            case 16 * 4 + 3: hipLaunchKernelGGL((KERNEL<16, true, 1>), grid, block, lds, s, __VA_ARGS__); break;  \
            default: set_error(#KERNEL ": no instantiation"); return MEDT_EUNSUPPORTED;                       \
        }                                                                                                     \
        return launch_status(#KERNEL);                                                                        \
    } while (0)

// The one-launch exact kernel of axial_fast.hip runs: it can finalise bn_similarity itself (fin_inline.h).
bool axial_attn_fwd_inlines(const AxialGeom& g, GatePtrs gates, const unsigned* flag) {
    return fast_path_enabled() && gates.stride == 0 && g.pos && !(g.L & 3) && g.fast3 && !g.rows4 && !(g.bound_path && flag);
}

int axial_attn_fwd(const AxialGeom& g, const float* qkv_raw, BnStats qkv, BnStats sim, const float* relative,
                   GatePtrs gates, float* stacked, float* lse, float* out_partials, unsigned* flag, hipStream_t s,
                   const FinSrc* simsrc) {
    if (abl_skip("attn_fwd")) return MEDT_OK;
    if (simsrc && simsrc->on && !axial_attn_fwd_inlines(g, gates, flag)) { set_error("attn_fwd: no kernel to finalise bn_similarity in"); return MEDT_EINVAL; }
    if (fast_path_enabled() && gates.stride == 0) {            // (per-sequence gates: generic kernel)
        const int rc = g.bf16 ? axial_attn_fwd_fast_bf16(g, qkv_raw, qkv, sim, relative, gates, stacked, lse, out_partials, flag, s, simsrc)
                              : axial_attn_fwd_fast(g, qkv_raw, qkv, sim, relative, gates, stacked, lse, out_partials, flag, s, simsrc);
        if (rc <= 0) return rc;
    }
    const size_t lds = axial_core_lds_bytes(g, false);
    MEDT_DISPATCH(attn_fwd_kernel, g, qkv_raw, qkv, sim, relative, gates, stacked, lse, out_partials);
}

// compile-time-L instantiations of the two backward passes (position-encoded layers, L = 16 / 32 / 64)
#define MEDT_LC_CASE(KERNEL, GPv, AXv, LCv, ...) \
    case (GPv * 2 + AXv) * 256 + LCv: hipLaunchKernelGGL((KERNEL<GPv, true, AXv, LCv>), grid, block, lds, s, __VA_ARGS__); break;
#define MEDT_LC_L(KERNEL, GPv, AXv, ...) \
    MEDT_LC_CASE(KERNEL, GPv, AXv, 16, __VA_ARGS__) MEDT_LC_CASE(KERNEL, GPv, AXv, 32, __VA_ARGS__) MEDT_LC_CASE(KERNEL, GPv, AXv, 64, __VA_ARGS__)
#define MEDT_DISPATCH_LC(KERNEL, ...)                                                                        \
    do {                                                                                                     \
        if (g.pos && (g.L == 16 || g.L == 32 || g.L == 64) && lds <= 160 * 1024) {                          \
            const dim3 grid(g.groups * g.tpg, g.G), block(MEDT_THREADS);                                    \
            switch ((g.gp * 2 + g.axis) * 256 + g.L) {                                                      \
                MEDT_LC_L(KERNEL, 2, 0, __VA_ARGS__) MEDT_LC_L(KERNEL, 2, 1, __VA_ARGS__)                    \
                MEDT_LC_L(KERNEL, 4, 0, __VA_ARGS__) MEDT_LC_L(KERNEL, 4, 1, __VA_ARGS__)                    \
                MEDT_LC_L(KERNEL, 8, 0, __VA_ARGS__) MEDT_LC_L(KERNEL, 8, 1, __VA_ARGS__)                    \
                MEDT_LC_L(KERNEL, 16, 0, __VA_ARGS__) MEDT_LC_L(KERNEL, 16, 1, __VA_ARGS__)                  \
                default: set_error(#KERNEL ": no compile-time-L instantiation"); return MEDT_EUNSUPPORTED;   \
            }                                                                                                \
            return launch_status(#KERNEL);                                                                   \
        }                                                                                                    \
    } while (0)

int axial_attn_bwd_stats(const AxialGeom& g, const float* qkv_raw, BnStats qkv, BnStats sim, const float* relative,
                         GatePtrs gates, const float* stacked, const float* lse, const float* dy,
                         const float* out_coef, int stride, float* partials, hipStream_t s) {
    const size_t lds = axial_core_lds_bytes(g, true);
    MEDT_DISPATCH_LC(attn_bwd_stats_kernel, g, qkv_raw, qkv, sim, relative, gates, stacked, lse, dy, out_coef, stride,
                     partials);
    MEDT_DISPATCH(attn_bwd_stats_kernel, g, qkv_raw, qkv, sim, relative, gates, stacked, lse, dy, out_coef, stride,
                  partials);
}

int axial_attn_bwd(const AxialGeom& g, const float* qkv_raw, BnStats qkv, BnStats sim, const float* sim_coef,
                   const float* relative, GatePtrs gates, const float* stacked, const float* lse, const float* dy,
                   const float* out_coef, int stride, float* dqkv, float* qkv_partials, float* rel_partials,
                   float* gate_partials, hipStream_t s) {
    const size_t lds = axial_core_lds_bytes(g, true);
    MEDT_DISPATCH_LC(attn_bwd_kernel, g, qkv_raw, qkv, sim, sim_coef, relative, gates, stacked, lse, dy, out_coef, stride,
                     dqkv, qkv_partials, rel_partials, gate_partials);
    MEDT_DISPATCH(attn_bwd_kernel, g, qkv_raw, qkv, sim, sim_coef, relative, gates, stacked, lse, dy, out_coef, stride,
                  dqkv, qkv_partials, rel_partials, gate_partials);
}

}  // namespace medt
