// medt_kernels.h -- internal launchers shared between the .hip translation units.
#pragma once
#include "medt_common.h"

namespace medt {

struct Queue;         // defer.h: recorded (deferred, grouped) launches

// ---- pointwise.hip ----------------------------------------------------------
// 1x1 convolution on NCHW:  y[n,o,p] = sum_c w[o,c] * x[n,c,p].
// partials (optional): per-(n, pixel-tile) [sum, sum of squares] of every output channel,
// laid out [n][ptile][Cout][2]  ==  [group][part][Cout][2] with part = (n in group, ptile).
int conv1x1_ptiles(int HW);
int conv1x1_fwd(const float* x, const float* w, float* y, float* partials,
                int N, int Cin, int Cout, int HW, hipStream_t s);
// dx[n,c,p] = sum_o w[o,c] * val[n,o,p],  val = coef ? c0*dy + c1*raw + c2 : dy   (coef [group][Cout][3])
int conv1x1_bwd_data(const float* dy, const float* raw, const float* coef, const float* w, float* dx,
                     int N, int Cin, int Cout, int HW, int groups, hipStream_t s);
// Batch-norm statistics: partials [group][parts_per_group][CH][2] -> mean/rstd/scale/shift per (group, ch),
// running-stat recurrence over the groups in order.  training==0: fold the running stats instead.
struct TablesJob { const float* relative; float* tables; int HQ, L, blocks; };   // optional: see sim_tables_blocks()
int bn_finalize(const float* partials, int parts_per_group, int groups, int CH, double count,
                const medt_bn_ptrs& bn, float momentum, float eps, int training, BnStats out, hipStream_t s,
                const TablesJob* tables = nullptr);
// Backward: partials [group][parts][CH][2] = [sum d, sum d*xhat] (d still to be multiplied by dscale).
// Writes coef[group][CH][3] with  dx = c0*d + c1*x + c2,  and dweight[CH], dbias[CH].
int bn_finalize3(const float* p0, int CH0, double n0, const medt_bn_ptrs& bn0, BnStats o0,
                 const float* p1, int CH1, double n1, const medt_bn_ptrs& bn1, BnStats o1,
                 const float* p2, int CH2, double n2, const medt_bn_ptrs& bn2, BnStats o2,
                 int ppg, int groups, float momentum, float eps, int training, hipStream_t s);
int bn_bwd_finalize(const float* partials, int parts_per_group, int groups, int CH, double count, float dscale,
                    BnStats st, const float* weight, int training, float* coef, float* dweight, float* dbias,
                    hipStream_t s);

// bn_output apply + pair-sum + AvgPool2d(stride):  stacked (N,OC,H,W) -> y (N,C,H/s,W/s)
struct FinSrc;
// src (optional, fin_inline.h; only where axial_out_fwd_inlines()): bn_output finalised by the kernel from the attention kernel's partial rows
bool axial_out_fwd_inlines(const medt_axial_desc& d);
int axial_out_fwd(const medt_axial_desc& d, const float* stacked, BnStats st, float* y, hipStream_t s, const FinSrc* src = nullptr);
// d[i] = c0*d[i] + c1*raw[i] + c2 per (group, channel) with raw stored as bfloat16: materialises the bn_qkv backward
// so the fp32 1x1 dgrad / wgrad kernels run without their (raw, coef) operands
int bn_bwd_apply_raw_bf16(float* d, const float* raw_bf16, const float* coef, int N, int CH, int HW, int groups, hipStream_t s);
int bn_bwd_fin_apply_bf16(const float* partials, int ppg, int groups, int CH, double count, float dscale, BnStats st,
                          const float* weight, int training, float* coef, float* dweight, float* dbias, float* d,
                          const float* raw_bf16, int N, int HW, hipStream_t s);
// partials [n][ptile][OC][2] of [sum dstk, sum dstk*xhat] with dstk = dy (un-pooled, unscaled)
// tj (optional): the launch's first tj->blocks workgroups also build the sliding-window tables of the layer's relative table (what
// sim_bwd_finalize's appended blocks build otherwise -- axial_out_bwd_stats_tables_ok() says whether the grid has room)
bool axial_out_bwd_stats_tables_ok(const medt_axial_desc& d, int blocks, int L);
// ymask (optional): the layer's output y of a fused output ReLU -- dy counts where y > 0 (no relu_mask launch in front)
int axial_out_bwd_stats(const medt_axial_desc& d, const float* stacked, const float* dy, BnStats st,
                        float* partials, hipStream_t s, const TablesJob* tj = nullptr, const float* ymask = nullptr);

// out[k] = sum_p in[p][k]
int reduce_rows(const float* in, int P, int K, float* out, hipStream_t s);

// ---- conv.hip ------------------------------------------------------------------
// partials (optional): [group][part][Cout][2], part = 256-position chunk of the group's npg*Ho*Wo positions
int conv2d_parts_per_group(int N, int groups, int HoWo);       // VALU kernel (256 positions per part)
int conv_parts_per_group(int N, int groups, int HoWo, int Cin, int Cout, int K, int stride, int H = 0, int W = 0, int pad = 0);   // whichever kernel runs
// thin-channel 3x3 stride-1 layers on the matrix cores (conv_mfma.hip, round 6): one partial row per workgroup tile
bool conv_thin_ok(int N, int groups, int Cin, int H, int W, int Cout, int K, int stride, int pad);
int conv_thin_parts_per_group(int N, int groups, int Cin, int H, int W, int Cout, int K, int stride, int pad);
// add (optional, y's shape): y = conv + add (the fan-in addend of a backward-data call)
int conv_thin_fwd(const float* x, const float* w, const float* bias, const float* add, float* y, float* partials, int N, int groups,
                  int Cin, int H, int W, int Cout, int relu, hipStream_t s);
// the LDS-patch MFMA kernel of the 7x7 stride-2 stems (conv_mfma.hip): 128 positions per part
bool conv_stem7_ok(int Cin, int H, int W, int Cout, int K, int stride, int pad);
int conv_stem7_parts_per_group(int N, int groups, int HoWo);
int conv_stem7_fwd(const float* x, const float* w, const float* bias, float* y, float* partials, int N, int H, int W, int Cout,
                   int relu, hipStream_t s);
// y_bf16: store y as bfloat16 (VALU path only: the 1x1 qkv_transform of a bf16-storage attention layer)
int conv2d_fwd(const float* x, const float* w, const float* bias, float* y, float* partials, float* scratch, int N,
               int Cin, int H, int W, int Cout, int K, int stride, int pad, int relu, int groups, hipStream_t s,
               int y_bf16 = 0);
size_t conv2d_fwd_scratch_floats(int N, int groups, int Cin, int H, int W, int Cout, int K, int stride, int pad);
// wt_scratch: Cout*Cin*K*K floats (used by the MFMA path for the flipped weights; may be NULL -> VALU path)
// add (optional): dx = dgrad + add -- the other gradient contributions of a fanned-out input, summed in the epilogue
int conv2d_bwd_data(const float* dy, const float* w, float* dx, float* wt_scratch, float* ksplit_scratch, int N, int Cin,
                    int H, int W, int Cout, int K, int stride, int pad, hipStream_t s, const float* add = nullptr, bool wt_ready = false);
bool conv2d_bwd_data_flips(int N, int Cin, int H, int W, int Cout, int K, int stride, int pad);
int conv_flip_weights(const float* w, float* wt, int Cout, int Cin, int K, hipStream_t s);
size_t conv2d_bwd_data_scratch_floats(int N, int Cin, int H, int W, int Cout, int K, int stride, int pad);
// dw[o,c,kh,kw] = sum_{n,ho,wo} val * x[...], val = coef ? c0*dy + c1*raw + c2 : dy (coef [group][Cout][3]);
// scratch: conv2d_bwd_weight_splits() * Cout*Cin*K*K floats
int conv2d_bwd_weight_splits(int N, int Cin, int Cout, int K, int Ho, int Wo);
// defer.h: when given, the launch (and the reduction of its partial slabs) is recorded, not issued
int conv2d_bwd_weight(const float* dy, const float* raw, const float* coef, const float* x, float* dw, float* scratch,
                      int N, int Cin, int H, int W, int Cout, int K, int stride, int pad, int groups, hipStream_t s,
                      Queue* q = nullptr);
int channel_sum_splits_for(int N, int HW);      // workgroups per channel (<= 16); 1: the sums go straight to their destination
int channel_sum(const float* x, float* out, float* scratch /* 16*C floats */, int N, int C, int HW, hipStream_t s);

// ---- conv_mfma.hip (fp32 matrix-core implicit GEMM; chosen by conv_use_mfma) -----------
bool conv_use_mfma(int Cin, int Cout, int K, int stride, long positions);
int conv_mfma_parts_per_group(int N, int groups, int HoWo);
// scratch: conv_mfma_scratch_floats() floats when the split-K variant may run (NULL forbids it)
bool conv_rows16_ok(int Cin, int H, int W, int K, int stride, int pad);          // conv_mfma.hip: the LDS-patch kernels of the 16-wide maps
size_t conv_mfma_scratch_floats(int N, int groups, int HoWo, int Cin, int Cout, int K, bool rows16 = false);   // rows16: conv_rows16_ok() of the problem
int conv_mfma_fwd(const float* x, const float* w, const float* bias, float* y, float* partials, float* scratch, int N,
                  int Cin, int H, int W, int Cout, int K, int stride, int pad, int relu, int groups, hipStream_t s);
int conv_mfma_bwd_data_s1(const float* dy, const float* w, float* wt_scratch, float* ksplit_scratch, float* dx, int N,
                          int Cin, int H, int W, int Cout, int K, int pad, hipStream_t s, bool wt_ready = false);
int conv_wgrad_mfma(const float* dy, const float* raw, const float* coef, const float* x, float* scratch, int N, int Cin,
                    int H, int W, int Cout, int Ho, int Wo, int K, int stride, int pad, int QS, int splits, int npg,
                    hipStream_t s);

// ---- elementwise.hip -----------------------------------------------------------
int bn_apply_act(const float* z, BnStats st, const float* res, float* y, int N, int C, int HW, int groups, int relu,
                 hipStream_t s);
int bn_act_bwd_stats(const float* dy, const float* y, const float* z, BnStats st, float* g, float* partials, int N, int C,
                     int HW, int groups, int relu, hipStream_t s);
int bn_bwd_apply(const float* g, const float* z, const float* coef, float* dz, int N, int C, int HW, int groups,
                 hipStream_t s);
int relu_mask(const float* a, const float* y, float* out, size_t total, hipStream_t s);
int seg_counts(const float* logits, const int64_t* target, int* counts, int N, int K, int HW, float threshold, hipStream_t s);
int relu_fwd(const float* x, float* y, size_t total, hipStream_t s);
int up2x_relu_add_fwd(const float* x, const float* skip, float* y, int NC, int H, int W, hipStream_t s);
int up2x_relu_bwd(const float* x, const float* dy, float* dx, int NC, int H, int W, hipStream_t s);
int patch_gather(const float* x, float* xp, int N, int C, int S, int P, int G, hipStream_t s);
int logo_merge_fwd(const float* x, const float* yp, float* y, int N, int C, int S, int P, int G, hipStream_t s);
int logo_merge_bwd(const float* dy, float* dx, float* dyp, int N, int C, int S, int P, int G, hipStream_t s);
int ce_parts(size_t npix);
int ce_fwd(const float* logits, const int64_t* target, float* partials, float* loss_out, int N, int K, int HW, int ignore,
           hipStream_t s);
int ce_bwd(const float* logits, const int64_t* target, const float* loss_out, const float* dloss, float* dlogits, int N,
           int K, int HW, int ignore, hipStream_t s);
// gated_sig: eff[k] = sigmoid(*f[k]) (k: f_qr, f_kr, f_sve, f_sv);  dgate[k] = d_eff[k] * eff[k] * (1 - eff[k])
int gate_sigmoid_fwd(const float* f_qr, const float* f_kr, const float* f_sve, const float* f_sv, float* eff, hipStream_t s);
int gate_sigmoid_bwd(const float* d_eff, const float* eff, float* dgate, hipStream_t s);
// gate_mode 2: per-sequence gate gradients [B*][G][4] (f_qr, f_kr, f_sve, f_sv) -> dgates [B*][4] in the (qr, kr, sv, sve)
// column order of the gate tensor
int gate_seq_reduce(const float* partials, float* dgates, int nseq, int G, hipStream_t s);
// AxialAttention_gated_data's gate MLP (reference lib/models/model_codes.py:371-380) and its backward
int gate_mlp_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* xn, float* h,
                 float* o, float* gates, int N, int C, int H, int W, int axis, hipStream_t s);
int gate_mlp_bwd(const float* dgates, const float* gates, const float* o, const float* h, const float* xn, const float* w1,
                 const float* w2, float* d_o, float* dh, float* dxn, float* dw1, float* db1, float* dw2, float* db2,
                 float* dx, int N, int C, int H, int W, int axis, hipStream_t s);
int adam_step(float* p, const float* g, float* m, float* v, float* state, size_t n, float lr, float b1, float b2,
              float eps, float wd, float gscale, hipStream_t s);

// ---- axial_core.hip ---------------------------------------------------------
struct AxialGeom {
    int N, C, H, W, G, gp, hq, L, Bo, axis, pos, OC, OCg, SC;
    int bf16;               // 1: qkv_raw / stacked are stored as bfloat16 (medt_axial_desc.act_dtype)
    int groups, npg;        // BN groups, images per group
    int spg;                // sequences per group = npg * Bo
    int S_T;                // sequences per workgroup tile
    int tpg;                // tiles per group
    int HW;
    int fast3;              // 1: compile-time-L persistent forward kernels (axial_fast.hip)
    int fparts;             // partial-statistics slots per group written by the forward L x L kernels
    int nt;                 // sub-tiles (of S_T sequences) per super-tile in the persistent kernels
    int rows4, nt4, oparts; // 4-rows-per-lane forward kernel (gp = 2, large problems): its sub-tiles and out-partials slots
    int bound_path;         // 1: bound-referenced softmax + repair pass (large problems only: two extra graph nodes)
    float bound_shift;      // test hook (MEDT_DEBUG_BOUND_SHIFT): added to the softmax reference bound to force the exact fallback
    double sim_count;       // elements per bn_similarity channel per group = spg * L * L
    double row_count;       // elements per bn_qkv / bn_output channel per group = spg * L
};
int  axial_geom(const medt_axial_desc& d, AxialGeom* g);   // validates, returns MEDT_E*
size_t axial_core_lds_bytes(const AxialGeom& g, bool backward);

// stride 0: one scalar per gate; stride 4 (gate_mode 2, AxialAttention_gated_data): element [b*stride] belongs to sequence b
struct GatePtrs { const float *f_qr, *f_kr, *f_sve, *f_sv; int stride; };

// axial_fast.hip: 16-byte-LDS-read variants for has_pos && L % 4 == 0; return 1 when not applicable
struct FinSrc;                   // fin_inline.h: statistics the kernel finalises itself from the producer's partial rows
struct BfinSrc;
int axial_attn_fwd_fast(const AxialGeom& g, const float* qkv_raw, BnStats qkv, BnStats sim, const float* relative,
                        GatePtrs gates, float* stacked, float* lse, float* out_partials, unsigned* flag, hipStream_t s,
                        const FinSrc* simsrc = nullptr);
// same kernels compiled for bfloat16 storage of qkv_raw / stacked (axial_fast.hip with -DMEDT_FAST_BF16=1)
int axial_attn_fwd_fast_bf16(const AxialGeom& g, const float* qkv_raw, BnStats qkv, BnStats sim, const float* relative,
                             GatePtrs gates, float* stacked, float* lse, float* out_partials, unsigned* flag,
                             hipStream_t s, const FinSrc* simsrc = nullptr);
int fast3_max_subtiles(int gp, int L, int axis);
int fast4_subtile_sequences(int L);
int fast4_max_subtiles(int axis);
// block_small.hip: a whole AxialBlock_wopos forward in one workgroup per BatchNorm group
bool wopos_block_ok(const medt_block_desc& d);
size_t wopos_block_part_doubles(const medt_block_desc& d);
int wopos_block_fwd(const medt_block_desc& d, const medt_block_params& p, const float* x, float* y,
                    const medt_block_saved& sv, double* parts, hipStream_t s);
// the stride-2 first block + downsample path (round 6; layer4_p.0 of MedT-128)
bool wopos_block_s2_ok(const medt_block_desc& d);
size_t wopos_block_s2_part_doubles(const medt_block_desc& d);
int wopos_block_s2_fwd(const medt_block_desc& d, const medt_block_s2_params& p, const float* x, float* y,
                       const medt_block_s2_saved& sv, double* parts, hipStream_t s);
// ... and its backward (default; MEDT_BLOCK_BWD=0 disables): launch + recorded / immediate parameter-gradient jobs
bool wopos_block_bwd_ok(const medt_block_desc& d);
size_t wopos_block_bwd_ws_bytes(const medt_block_desc& d);
int wopos_block_bwd(const medt_block_desc& d, const medt_block_params& p, const float* x, const float* y, const float* dy,
                    const float* dx_add, const medt_block_saved& sv, float* dx, const medt_block_grads& gr, void* ws,
                    size_t ws_bytes, hipStream_t s);

// conv_small.hip: 1x1 conv + BatchNorm blocks whose BN group fits one workgroup
bool conv_small_ok(const medt_conv_desc& d);
int conv_small_fwd(const medt_conv_desc& d, const float* x, const float* w, const medt_bn_ptrs& bn, const float* res,
                   float* z, float* y, float* partials, hipStream_t s);
bool bn_dgrad1x1_small_ok(const medt_conv_desc& d);    // BatchNorm backward + the 1x1 backward-data behind it, one launch
int bn_dgrad1x1_small(const medt_conv_desc& d, const float* dy, const float* y, const float* z, BnStats st,
                      const float* gamma, const float* w, const float* dx_add, float* g, float* dz, float* partials,
                      float* dx, hipStream_t s);
int bn_act_bwd_small(const medt_conv_desc& d, const float* dy, const float* y, const float* z, BnStats st,
                     const float* weight, float* g, float* dz, float* partials, int HoWo, hipStream_t s);
// axial_small.hip: a whole position-free layer per (BN group, head) workgroup
int bn_fin_apply(const float* z, const float* partials, int ppg, double count, const medt_bn_ptrs& bn, float eps, int training,
                 BnStats st, const float* res, float* y, int N, int C, int HW, int groups, int relu, hipStream_t s);
bool conv_fwd_ws_ok(int Cin, int Cout, int K, int stride, long positions);      // conv.hip: wave-split forward (64-position parts)
int bn_chan_threads(const medt_conv_desc& d, int HoWo);       // conv_small.hip: one workgroup per (group, channel); 0 = no
int bn_act_bwd_chan(const medt_conv_desc& d, const float* dy, const float* y, const float* z, BnStats st,
                    const float* weight, float* g, float* dz, float* partials, int HoWo, hipStream_t s);
bool wopos_small_ok(const AxialGeom& g, const medt_axial_desc& d);
int wopos_small_fwd(const AxialGeom& g, const medt_axial_desc& d, const medt_axial_params& p, const float* x, float* y,
                    float* qkv_raw, float* stacked, float* lse, float* part_q, float* part_s, float* part_o,
                    hipStream_t s);
bool wopos_small_bwd_ok(const AxialGeom& g, const medt_axial_desc& d);
int wopos_small_bwd(const AxialGeom& g, const medt_axial_desc& d, const medt_axial_params& p, const float* y,
                    const float* dy, const float* qkv_raw, const float* stacked, const float* lse, BnStats sq, BnStats ss,
                    BnStats so, float* dqkv, float* part_ob, float* part_sb, float* part_qb, float* coef_qkv,
                    hipStream_t s);
int wopos_small_bwd_finalize(const AxialGeom& g, const medt_axial_desc& d, const medt_axial_params& p,
                             const float* part_ob, const float* part_sb, const float* part_qb, BnStats sq, BnStats ss,
                             BnStats so, const medt_axial_grads& gr, hipStream_t s, Queue* q);
bool fast_path_enabled();       // MEDT_DISABLE_FAST=1 forces the generic kernels (A/B checks)

// axial_stats.hip: bn_similarity batch statistics in closed form (one read of q and k, no L x L pass).
// partials [group][sim_stats_parts()][SC][2]; tables: sim_tables_floats() floats, the sliding-window sums of the
// relative table, built by sim_tables_blocks() extra blocks appended to the bn_qkv bn_finalize launch that precedes the
// statistics kernel in the layer (no launch of their own).
size_t sim_tables_floats(const AxialGeom& g);
int sim_tables_blocks(const AxialGeom& g);
int sim_stats_parts(const AxialGeom& g);
int axial_logit_stats(const AxialGeom& g, const float* qkv_raw, BnStats qkv, const float* relative, GatePtrs gates,
                      const float* tables, float* partials, hipStream_t s);
// fused attention: stacked, lse, bn_output partials [group][tile][OC][2] (may be NULL)
// simsrc (optional, fin_inline.h; only where axial_attn_fwd_inlines()): bn_similarity is finalised by the kernel itself from the
// statistics kernel's partial rows -- `sim` is not read, no bn_finalize launch in front
bool axial_attn_fwd_inlines(const AxialGeom& g, GatePtrs gates, const unsigned* flag);
int axial_attn_fwd(const AxialGeom& g, const float* qkv_raw, BnStats qkv, BnStats sim, const float* relative,
                   GatePtrs gates, float* stacked, float* lse, float* out_partials, unsigned* flag, hipStream_t s,
                   const FinSrc* simsrc = nullptr);
// backward pass A: partials [group][tile][G][4] = sum dZ*{S_qk,S_qr,S_kr,1}
int axial_attn_bwd_stats(const AxialGeom& g, const float* qkv_raw, BnStats qkv, BnStats sim, const float* relative,
                         GatePtrs gates, const float* stacked, const float* lse, const float* dy,
                         const float* out_coef, int stride, float* partials, hipStream_t s);
// sim backward coefficients [group][SC][3] (e,u,w) + dweight/dbias of bn_similarity
// tables (optional): sim_tables_blocks() extra blocks build the sliding-window table sums the single-sweep backward's fix
// kernel reads (no launch of their own)
int axial_sim_bwd_finalize(const AxialGeom& g, const float* partials, BnStats sim, const float* weight, int training,
                           float* coef, float* dweight, float* dbias, hipStream_t s, const TablesJob* tables = nullptr);
// backward pass B: dqkv (wrt normalised qkv), bn_qkv bwd partials [group][tile][2C][2],
// relative-table partials [blocks][2gp*(2L-1)], gate partials [blocks][4]
int axial_attn_bwd(const AxialGeom& g, const float* qkv_raw, BnStats qkv, BnStats sim, const float* sim_coef,
                   const float* relative, GatePtrs gates, const float* stacked, const float* lse, const float* dy,
                   const float* out_coef, int stride, float* dqkv, float* qkv_partials, float* rel_partials,
                   float* gate_partials, hipStream_t s);


// ---- axial_bwd.hip: the single-sweep backward of the position-encoded layers (gp <= 4, L in {32, 64, 128}) --------------
struct SweepPlan {
    int LS, nw, S_T;        // lanes per sequence, waves per workgroup, sequences per workgroup tile
    int tiles, nparts;      // tiles per BN group, (persistent) workgroups per BN group
    int fparts;             // parts per BN group of the fix kernel: 256 * fix_ppt positions each
    int fix_ppt;            // positions per thread of the fix kernel: 4, or 1 where that leaves fewer than 32 workgroups (round 6)
    int npg_floats;         // per-position / per-sequence Gram record: Gq pairs | Sq | Gk pairs | Sk
    size_t lds;
};
bool axial_bwd_sweep_plan(const AxialGeom& g, int gate_stride, SweepPlan* p);
// sliding-window sums of the relative table (sim_tables.h layout) for the fix kernel
int axial_bwd_tables(const AxialGeom& g, const float* relative, float* tables, hipStream_t s);
// the sweep: dqkv (e-terms of dq | dk, final dv), bn_qkv partial rows [0, nparts) of every group (v channels),
// part_sb [group][nparts][G][4], rel_part [G * groups * nparts][2gp * TL], pg_part [same blocks][L * npg_floats],
// gram [B*][G][npg_floats], gate_raw [same blocks][4] (NULL: no gate gradients)
int axial_attn_bwd_sweep(const AxialGeom& g, const SweepPlan& p, const float* qkv_raw, BnStats qkv, BnStats sim,
                         const float* relative, GatePtrs gates, const float* stacked, const float* lse, const float* dy,
                         const float* out_coef, int stride, float* dqkv, float* part_qb, int qb_rpg, float* part_sb,
                         float* rel_part, float* pg_part, float* gram, float* gate_raw, hipStream_t s, float* raw32 = nullptr,
                         const BfinSrc* ob = nullptr,        // ob (fin_inline.h): bn_output's backward finalised by the sweep, out_coef not read
                         const float* ymask = nullptr);      // ymask: the layer's output y (fused output ReLU) -- dy counts where y > 0
// u / w terms of dq | dk (apply != 0: training mode) and the bn_qkv partial rows [nparts, nparts + fparts) (q | k channels)
int axial_attn_bwd_fix(const AxialGeom& g, const SweepPlan& p, const float* qkv_raw, BnStats qkv, const float* sim_coef,
                       const float* tables, const float* gram, GatePtrs gates, int apply, float* dqkv, float* part_qb,
                       int qb_rpg, hipStream_t s, const struct SimBSrc* sb = nullptr);   // sb (fin_inline.h): sim_coef derived in the kernel
// u / w terms of the table gradients -> rel_rows [groups * G][2gp * TL]; gate gradients -> gate_rows [groups * G][4]
int axial_attn_bwd_relfix(const AxialGeom& g, const SweepPlan& p, const float* relative, const float* sim_coef, BnStats sim,
                          GatePtrs gates, const float* pg_part, const float* gate_raw, int training, float eps,
                          float* rel_rows, float* gate_rows, hipStream_t s, const BfinSrc* qb = nullptr);
// qb (fin_inline.h, one BatchNorm group): the launch also finalises bn_qkv's backward (coefficients + parameter gradients), head by head

}  // namespace medt
