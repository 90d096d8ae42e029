/* medt_abi.h -- C ABI of libmedt_hip.so, the MI355X (gfx950) implementation of
 * Medical-Transformer's gated axial-attention hot path.
 *
 * The reference has no FFI / plugin interface for this path: it is ~5.6 kLoC of
 * Python whose hot path lives behind a *Python module surface*
 * (lib/models/axialnet.py).  The drop-in boundary is therefore that module
 * surface (mirrored in medical-transformer_amd/lib/), and THIS header is the
 * boundary underneath it: what the mirrored modules' forward/backward bind
 * through ctypes instead of the stock torch ops the reference dispatches.
 * Each entry point cites the reference lines it replaces.
 *
 * Conventions
 *   - plain C: no C++ or torch types cross the boundary.
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch's
 *     allocator); the library never allocates, frees or retains device memory.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); all work
 *     is enqueued on it, nothing synchronises -> the calls are hipGraph-capturable.
 *   - tensors are contiguous NCHW float32 unless stated.
 *   - alignment: tensors are expected 16-byte aligned (what every device allocator returns; sub-tensor views at channel or
 *     image granularity keep it because H*W is a multiple of 4 for every shape of the networks).  Kernels that stage 16-byte
 *     pieces through LDS check the pointers per call and fall back to element-wise movers when a tensor is not.
 *   - return 0 on success, <0 on error (MEDT_E*); medt_last_error() returns a
 *     thread-local message.  No global mutable state: safe to call concurrently
 *     from several host threads on different streams (nn.DataParallel's
 *     thread-per-replica model, reference train.py:104-107).
 */
#ifndef MEDT_ABI_H
#define MEDT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MEDT_ABI_VERSION 9

#define MEDT_OK            0
#define MEDT_EINVAL       -1   /* bad descriptor / null pointer / size mismatch            */
#define MEDT_EUNSUPPORTED -2   /* geometry outside what the kernels are built for            */
#define MEDT_ELAUNCH      -3   /* hipLaunchKernel / hipGetLastError reported a failure        */
#define MEDT_EWORKSPACE   -4   /* workspace smaller than medt_*_workspace_bytes()             */

int         medt_abi_version(void);
const char* medt_last_error(void);

/* ------------------------------------------------------------------------- *
 * Deferred, grouped execution of the work no later LAYER waits for (optional).
 *   The reference's loss.backward() / optimizer.step() (train.py:159-161) makes every Conv2d weight gradient, bias
 *   gradient and BatchNorm bookkeeping update its own kernel launch in the middle of the dependent chain.  With a
 *   queue bound to a stream, the layer entry points below RECORD those launches instead (weight / bias gradients and
 *   the reductions of their partial slabs, the relative-table / gate reductions, the saved-statistics + running-stat
 *   finalisation and BatchNorm parameter gradients of the fused small-layer kernels) and medt_queue_flush() issues
 *   everything recorded as a few grouped launches -- call it after the forward pass (backward reads the saved
 *   statistics) and after the backward pass (the optimizer reads the gradients).  The caller must keep every buffer
 *   it passed to a recording call (inputs, outputs, saved tensors, workspace) alive and unmodified until the flush.
 *   Same kernel bodies as the immediate launches; results are identical except that weight gradients are summed over
 *   differently sized position chunks (fp32 rounding, ~1e-7 relative).
 *   Without a bound queue every call launches immediately (the default).
 * ------------------------------------------------------------------------- */
void*  medt_queue_create(void);
int    medt_queue_destroy(void* queue);
int    medt_queue_bind(void* queue, void* stream);      /* queue == NULL: unbind the stream */
size_t medt_queue_pending(const void* queue);            /* recorded, not yet flushed */
int    medt_queue_flush(void* queue, void* stream);      /* enqueue everything recorded on `stream` */
/* ... with the dedicated MFMA weight-gradient launches on `aux_stream` (another stream of the caller's, idle by now; NULL or == stream:
 * medt_queue_flush), forked from and joined back into `stream` by events: after the call everything is ordered on `stream` (ABI v9) */
int    medt_queue_flush2(void* queue, void* stream, void* aux_stream);
int    medt_queue_discard(void* queue);                  /* drop everything recorded WITHOUT launching it (error paths: the
                                                            buffers the jobs point into are about to be released) */

/* ------------------------------------------------------------------------- *
 * Axial attention layer
 *   replaces AxialAttention.forward          lib/models/axialnet.py:52-92
 *            AxialAttention_dynamic.forward  lib/models/axialnet.py:142-189
 *            AxialAttention_wopos.forward    lib/models/axialnet.py:222-253
 *   and their autograd backward.
 * ------------------------------------------------------------------------- */
typedef struct medt_axial_desc {
    int32_t N, C, H, W;     /* input (N,C,H,W); in_planes == out_planes == C                     */
    int32_t G;              /* heads ("groups", always 8 in the reference); gp = C/G, gp even    */
    int32_t axis;           /* 0: attend along H (width=False, :146); 1: along W (width=True)    */
    int32_t has_pos;        /* 1: relative-position tables + BN2d(3G) (:155-167); 0: wopos       */
    int32_t stride;         /* AvgPool2d(stride) after the layer (:186-187); 1 or 2              */
    int32_t training;       /* 1: batch statistics + running-stat update; 0: running statistics  */
    int32_t bn_groups;      /* BN statistic groups along N (N % bn_groups == 0).  1 normally; 16
                               when the 16 LoGo patches (:661-700) are stacked patch-major on N:
                               each group is normalised with its own statistics and the running
                               stats receive the groups' updates in order.        */
    float   eps;            /* 1e-5 */
    float   momentum;       /* 0.1  */
    int32_t out_relu;       /* 1: fuse the block's ReLU after the width layer (:333) into the output pass */
    int32_t gate_mode;      /* 0: f_* multiply as stored (axialnet.py:163-164,175-176);
                               1: sigmoid(f_*) multiplies -- AxialAttention_gated_sig, lib/models/model_codes.py:279-280,
                                  292-293; the gate gradients returned are then wrt the stored (pre-sigmoid) values;
                               2: one set of gates PER SEQUENCE -- AxialAttention_gated_data, model_codes.py:406-407,420-421:
                                  params.f_qr points to a (B*, 4) tensor with columns (qr, kr, sv, sve), B* = N*W (axis 0) or
                                  N*H (axis 1), sequence b = n*Bo + s; f_kr / f_sve / f_sv are ignored; grads.gates receives the
                                  (B*, 4) gradient in the same layout; the generic (not the bandwidth-tuned) kernels run */
    int32_t act_dtype;      /* storage type of saved->qkv_raw and saved->stacked: 0 float32 (the reference's arithmetic and
                               storage), 1 bfloat16 (BASELINE.json configs[1]: half the attention path's HBM bytes;
                               accumulation, statistics, x / y / dx and every gradient stay float32).  has_pos only. */
} medt_axial_desc;

typedef struct medt_bn_ptrs {
    const float* weight;            /* (CH)                                            */
    const float* bias;              /* (CH)                                            */
    float*       running_mean;      /* (CH)  updated in place when training            */
    float*       running_var;       /* (CH)                                            */
    int64_t*     num_batches_tracked; /* 0-d int64, += bn_groups when training; may be NULL */
} medt_bn_ptrs;

typedef struct medt_axial_params {
    const float* w_qkv;             /* qkv_transform.weight (2C, C[,1])                 :114 */
    medt_bn_ptrs bn_qkv;            /* BatchNorm1d(2C)                                  :116 */
    medt_bn_ptrs bn_similarity;     /* BatchNorm2d(3G)  (G when !has_pos)               :117 */
    medt_bn_ptrs bn_output;         /* BatchNorm1d(2C)  (C when !has_pos)               :118 */
    const float* relative;          /* (2gp, 2L-1); NULL when !has_pos                  :131 */
    const float* f_qr;              /* 0-d gates (:124-127); NULL == 1.0 (AxialAttention)    */
    const float* f_kr;
    const float* f_sve;
    const float* f_sv;
} medt_axial_params;

/* Activations kept between forward and backward (caller-allocated). */
typedef struct medt_axial_saved {
    void*  qkv_raw;   /* (N, 2C, H, W)  qkv_transform output before bn_qkv; float32 or bfloat16 (desc.act_dtype) */
    void*  stacked;   /* (N, OC, H, W)  sv|sve before bn_output, channel 2(g*gp+c)+{0:sv,1:sve};
                         OC = 2C (has_pos) or C (wopos: sv only); float32 or bfloat16 (desc.act_dtype)  */
    float* lse;       /* (N, G, H, W)   log2-domain log-sum-exp of every softmax row           */
    float* stats;     /* medt_axial_stats_floats() floats: per-BN mean / rstd / scale / shift  */
} medt_axial_saved;

typedef struct medt_axial_grads {   /* all written (not accumulated) by medt_axial_layer_bwd */
    float* w_qkv;                   /* (2C, C)                                   */
    float* bn_qkv_weight;  float* bn_qkv_bias;        /* (2C)                    */
    float* bn_sim_weight;  float* bn_sim_bias;        /* (3G) or (G)             */
    float* bn_out_weight;  float* bn_out_bias;        /* (2C) or (C)             */
    float* relative;                /* (2gp, 2L-1) or NULL                       */
    float* gates;                   /* 4 floats [f_qr, f_kr, f_sve, f_sv] or NULL (skip) */
} medt_axial_grads;

size_t medt_axial_stats_floats(const medt_axial_desc*);
size_t medt_axial_workspace_bytes(const medt_axial_desc*);   /* max over fwd and bwd */

/* x (N,C,H,W) -> y (N,C,H/stride,W/stride).  `saved` may have NULL lse when !training
 * is used for inference only (qkv_raw, stacked and stats are always needed as scratch). */
int medt_axial_layer_fwd(const medt_axial_desc*, const medt_axial_params*, const float* x, float* y,
                         const medt_axial_saved*, void* workspace, size_t workspace_bytes, void* stream);

/* dy (N,C,H/stride,W/stride) -> dx (N,C,H,W) + parameter gradients.  Training-mode
 * statistics are differentiated through (the reference's autograd does); with
 * desc.training == 0 the BatchNorms are the affine maps of their running stats. */
int medt_axial_layer_bwd(const medt_axial_desc*, const medt_axial_params*, const float* x, const float* y,
                         const float* dy, const medt_axial_saved*, float* dx, const medt_axial_grads*,
                         void* workspace, size_t workspace_bytes, void* stream);
/* y: forward output, needed iff out_relu. */

/* The stages of the attention core on their own, for benchmarks / profiling (bench.py's roofline legs): the bn_similarity
 * statistics pass (closed form: one read of q,k) and the fused logits + softmax + sv|sve pass, given the BN scale/shift
 * already in saved->stats.  Same kernels the layer entry points launch.  Call medt_axial_layer_fwd once with the same
 * descriptor and workspace first: it leaves the sliding-window tables the statistics kernel reads in the workspace. */
int medt_axial_core_stats(const medt_axial_desc*, const medt_axial_params*, const medt_axial_saved*,
                          void* workspace, size_t workspace_bytes, void* stream);
int medt_axial_core_fwd(const medt_axial_desc*, const medt_axial_params*, const medt_axial_saved*,
                        void* workspace, size_t workspace_bytes, void* stream);
/* The two L x L backward passes (bn_similarity backward statistics; dq/dk/dv + relative-table and gate gradients) on
 * their own, for benchmarks: call medt_axial_layer_bwd once with the same descriptor and workspace first (it leaves the
 * bn_output / bn_similarity backward coefficients these passes read in the workspace).  Same kernels as the layer call. */
int medt_axial_core_bwd(const medt_axial_desc*, const medt_axial_params*, const medt_axial_saved*, const float* dy,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * A whole AxialBlock_wopos forward as ONE launch (lib/models/axialnet.py:368-391):
 *   y = relu( bn2(conv_up( relu(width_block(hight_block( relu(bn1(conv_down(x))) ))) )) + x )
 * for the deep blocks of the LoGo local branch (:661-700), whose BatchNorm group (4 images on 4x4 maps) fits one
 * workgroup: stride 1, no downsample, in_planes == out_planes == C, attention width `width`.  The saved tensors are
 * exactly those medt_conv_block_fwd / medt_axial_layer_fwd would have produced for the four stages, so the backward runs
 * through medt_conv_block_bwd / medt_axial_layer_bwd unchanged.  medt_wopos_block_workspace_bytes() returns 0 for shapes
 * the fused kernel is not built for (the caller then uses the per-stage entry points).
 * ------------------------------------------------------------------------- */
typedef struct medt_block_desc {
    int32_t N, C, width, H, W;   /* x, y: (N,C,H,W); the attention layers work on (N,width,H,W)      */
    int32_t G;                   /* heads                                                            */
    int32_t training, bn_groups; /* as in medt_axial_desc                                            */
    float   eps, momentum;
} medt_block_desc;

typedef struct medt_block_params {
    const float*      w_down;    /* conv_down.weight (width, C)                                 :355 */
    medt_bn_ptrs      bn1;       /* BatchNorm2d(width)                                          :356 */
    medt_axial_params height;    /* hight_block: w_qkv + the three BatchNorms (relative / f_* NULL)  */
    medt_axial_params width;     /* width_block                                                      */
    const float*      w_up;      /* conv_up.weight (C, width)                                   :359 */
    medt_bn_ptrs      bn2;       /* BatchNorm2d(C)                                              :360 */
} medt_block_params;

typedef struct medt_block_saved {
    float* z1;                   /* (N,width,H,W) conv_down output before bn1                        */
    float* y1;                   /* (N,width,H,W) relu(bn1(z1)) = input of the height layer          */
    float* stats1;               /* medt_conv_stats_floats() of the conv_down block                  */
    medt_axial_saved height;     /* as medt_axial_layer_fwd fills it for the height layer            */
    float* y_h;                  /* (N,width,H,W) height layer output = input of the width layer     */
    medt_axial_saved width;
    float* y_w;                  /* (N,width,H,W) relu(width layer output) = input of conv_up        */
    float* z2;                   /* (N,C,H,W) conv_up output before bn2                              */
    float* stats2;               /* medt_conv_stats_floats() of the conv_up block                    */
} medt_block_saved;

size_t medt_wopos_block_workspace_bytes(const medt_block_desc*);      /* 0: not a shape of the fused kernel */
int medt_wopos_block_fwd(const medt_block_desc*, const medt_block_params*, const float* x, float* y,
                         const medt_block_saved*, void* workspace, size_t workspace_bytes, void* stream);

/* The STRIDE-2 first block of a layer with its downsample path as one launch (round 6, ABI v9): AxialBlock_wopos(C -> width ->
 * 2*width, stride 2) + downsample = Sequential(conv1x1(C, 2*width, stride 2), BatchNorm2d) -- lib/models/axialnet.py:368-391 with
 * :596-606 -- x (N,C,H,W) -> y (N,2*width,H/2,W/2).  `blk` fields as above except: width.stats / y_w describe the width layer
 * WITH its AvgPool2d(2) (y_w: (N,width,H/2,W/2), pooled and ReLU'd), z2 / stats2: (N,2*width,H/2,W/2).  zd / yd / statsd: the
 * downsample block's conv output, its BatchNorm output (the identity added behind bn2) and its saved statistics
 * (medt_conv_stats_floats of that block).  Fused shape: 4x4 maps, C = width = 128, 4 images per BatchNorm group (layer4_p.0 of
 * MedT at 128 px); medt_wopos_block_s2_workspace_bytes() returns 0 otherwise (or with MEDT_BLOCK_S2=0). */
typedef struct medt_block_s2_params {
    medt_block_params blk;
    const float*      w_ds;      /* downsample[0].weight (2*width, C)                           :600 */
    medt_bn_ptrs      bn_ds;     /* downsample[1]                                               :601 */
} medt_block_s2_params;
typedef struct medt_block_s2_saved {
    medt_block_saved blk;
    float *zd, *yd, *statsd;
} medt_block_s2_saved;
size_t medt_wopos_block_s2_workspace_bytes(const medt_block_desc*);
int medt_wopos_block_s2_fwd(const medt_block_desc*, const medt_block_s2_params*, const float* x, float* y,
                            const medt_block_s2_saved*, void* workspace, size_t workspace_bytes, void* stream);

/* The same block's BACKWARD as one launch: dy -> dx through bn2, conv_up, the two attention layers, bn1 and conv_down,
 * the identity's gradient and `dx_add` (the other consumers' contribution to d(x), may be NULL) summed in the last phase;
 * every parameter gradient in `grads` is written (not accumulated) -- when a queue is bound to the stream the weight
 * gradients and the BatchNorm parameter reductions are recorded for the grouped flush exactly like the per-stage entry
 * points record theirs.  `saved` is what medt_wopos_block_fwd (or the four per-stage forwards) filled, `y` the block output.
 * medt_wopos_block_bwd_workspace_bytes() returns 0 when this path is not available for the shape or not enabled
 * (on by default, MEDT_BLOCK_BWD=0 disables; the caller then runs medt_conv_block_bwd / medt_axial_layer_bwd for the four stages).
 * State: verified against the reference fixture on the CPU lane emulator (tests/test_lane_emu.py) and on the MI355X
 * (tests/test_block_gpu.py); default since round 5 (profiles/r05_never_run_kernels.txt, profiles/r05_step_ab.json). */
typedef struct medt_block_grads {
    float *w_down, *bn1_weight, *bn1_bias;
    medt_axial_grads height, width;          /* relative / gates: NULL (position-free layers)                 */
    float *w_up, *bn2_weight, *bn2_bias;
} medt_block_grads;
size_t medt_wopos_block_bwd_workspace_bytes(const medt_block_desc*);
int medt_wopos_block_bwd(const medt_block_desc*, const medt_block_params*, const float* x, const float* y, const float* dy,
                         const medt_block_saved*, float* dx, const float* dx_add, const medt_block_grads*,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * Convolution block:  y = act( BN( conv2d(x, w) + bias ) + res )
 *   replaces the nn.Conv2d / nn.BatchNorm2d / ReLU / residual-add sequences of
 *   AxialBlock*.forward (lib/models/axialnet.py:285-300, 327-342, 373-389), the stems
 *   (:475-483, :623-632, :666-678), downsample (:450-454), decoders / adjust (:434-439, :493-502).
 * ------------------------------------------------------------------------- */
typedef struct medt_conv_desc {
    int32_t N, Cin, H, W, Cout;
    int32_t K, stride, pad;       /* square kernels: 1, 3 or 7                                   */
    int32_t has_bias;             /* nn.Conv2d bias                                              */
    int32_t has_bn;               /* BatchNorm2d on the conv output                              */
    int32_t has_res;              /* residual added after BN (block identity / downsample)       */
    int32_t relu;                 /* ReLU last                                                   */
    int32_t training, bn_groups;  /* as in medt_axial_desc.  `training` also means "a backward pass follows this forward":
                                     for has_bn == 0 it changes no result, but a training-mode forward of a 3x3 layer whose
                                     backward-data runs on the MFMA kernel leaves that kernel's flipped weights in `stats`
                                     (below) -- pass the same value to medt_conv_block_bwd                                */
    float   eps, momentum;
    int32_t lean;                 /* scheduling hint, no effect on results.  1: the caller runs CU-filling kernels on ANOTHER
                                     stream at the same time (MedT's global branch at 256 px: persistent attention kernels that
                                     hold ~all of a CU's LDS): keep to the per-stage kernels, whose workgroups co-reside with
                                     anything, instead of the fused BatchNorm-backward + dgrad kernel (tens of KB of LDS per
                                     workgroup -- measured: +0.5 ms on the 5.7 ms MedT-256 step, -0.02 ms on the MedT-128 step) */
} medt_conv_desc;

/* Floats the caller must allocate for `stats` -- ALWAYS size it with this call, never by formula (ABI v9).  Layout:
 *   [0, 4*bn_groups*Cout)                    saved BatchNorm statistics (mean, rstd, scale, shift) when has_bn, else empty;
 *   [align_up(that, 64), + Cout*Cin*K*K)     only for training-mode K == 3 layers whose backward-data takes the MFMA kernel
 *                                            (has_bn or not): the flipped / transposed weights, written by the forward (or by
 *                                            the flush of the queue bound to its stream) and read by medt_conv_block_bwd.
 * 0 means "no stats buffer needed" (pass NULL or a dummy).  The same buffer goes to medt_conv_block_fwd and _bwd. */
size_t medt_conv_stats_floats(const medt_conv_desc*);
size_t medt_conv_workspace_bytes(const medt_conv_desc*);   /* max over fwd and bwd              */

/* z: conv output (N,Cout,Ho,Wo), kept for backward when has_bn (pass z == y otherwise). */
int medt_conv_block_fwd(const medt_conv_desc*, const float* x, const float* w, const float* bias,
                        const medt_bn_ptrs* bn, const float* res, float* z, float* y, float* stats,
                        void* workspace, size_t workspace_bytes, void* stream);
/* dx, dbias, dres may be NULL (not needed).  dw, dbn_weight, dbn_bias are written, not accumulated.
 * dx_add (optional, same shape as dx): dx = conv-backward + dx_add.  The input of an AxialBlock feeds conv_down AND the
 * residual / downsample path (axialnet.py:327,339-340) and the decoder skips (:494-500); the reference's autograd sums
 * those gradients with separate add kernels, here the last producer adds the others in its epilogue. */
int medt_conv_block_bwd(const medt_conv_desc*, const float* x, const float* w, const medt_bn_ptrs* bn,
                        const float* z, const float* y, const float* stats, const float* dy,
                        float* dx, float* dw, float* dbias, float* dbn_weight, float* dbn_bias, float* dres,
                        const float* dx_add, void* workspace, size_t workspace_bytes, void* stream);

/* The gate network of AxialAttention_gated_data (lib/models/model_codes.py:371-380):
 *   xn = mean over the sequence of x (B*, C);  h = relu(fcn1(xn));  o = relu(fcn2(h));  gates = sigmoid(o)   (B*, 4)
 * xn, h (B*, C) and o (B*, 4) are kept for the backward.  bwd: scratch = B* * (4 + 2C) floats; dw1 (C,C), db1 (C), dw2 (4,C),
 * db2 (4) and dx (N,C,H,W: the gradient that reaches x THROUGH the gates) are written, not accumulated. */
int medt_gate_mlp_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* xn, float* h,
                      float* o, float* gates, int N, int C, int H, int W, int axis, void* stream);
int medt_gate_mlp_bwd(const float* dgates, const float* gates, const float* o, const float* h, const float* xn,
                      const float* w1, const float* w2, float* scratch, float* dw1, float* db1, float* dw2, float* db2,
                      float* dx, int N, int C, int H, int W, int axis, void* stream);

/* y = relu(bilinear_x2(x)) + skip       F.interpolate(scale_factor=(2,2), mode='bilinear') + relu + torch.add
 * (lib/models/axialnet.py:493-501, 650-652, 690-698).  x (NC,H,W) -> y (NC,2H,2W); skip may be NULL.
 * Backward: dskip = dy (aliased by the caller), dx via medt_up2x_relu_bwd (needs x to rebuild the ReLU mask). */
int medt_up2x_relu_add_fwd(const float* x, const float* skip, float* y, int NC, int H, int W, void* stream);
int medt_up2x_relu_bwd(const float* x, const float* dy, float* dx, int NC, int H, int W, void* stream);

/* LoGo local branch plumbing (lib/models/axialnet.py:658-702): gather the G x G grid of P-px patches of
 * x (N,C,S,S) into (G*G*N, C, P, P), patch-major;  y = x + x_loc with x_loc = x overwritten by the patches. */
int medt_patch_gather(const float* x, float* xp, int N, int C, int S, int P, int G, void* stream);
int medt_logo_merge_fwd(const float* x, const float* yp, float* y, int N, int C, int S, int P, int G, void* stream);
int medt_logo_merge_bwd(const float* dy, float* dx, float* dyp, int N, int C, int S, int P, int G, void* stream);

/* LogNLLLoss.forward == F.cross_entropy(mean, ignore_index) (metrics.py:17-20).  logits (N,K,HW) float,
 * target (N,HW) int64.  loss_out: 3 floats [mean loss, number of counted pixels, number of targets outside [0,K) that
 * are not ignore_index].  F.cross_entropy raises on such targets; a kernel cannot, so they are excluded from the mean
 * and COUNTED -- the host side (medt_amd.ops.cross_entropy, TrainStep.check_targets) raises from the count.  All pixels
 * ignored -> 0/0 = NaN, as in torch.  partials: medt_ce_partials() floats. */
size_t medt_ce_partials(int N, int HW);
int medt_ce_fwd(const float* logits, const int64_t* target, float* partials, float* loss_out, int N, int K, int HW,
                int ignore_index, void* stream);
int medt_ce_bwd(const float* logits, const int64_t* target, const float* loss_out, const float* dloss, float* dlogits,
                int N, int K, int HW, int ignore_index, void* stream);

/* torch.optim.Adam(lr, betas, eps, weight_decay) over one flat buffer (train.py:111-112,161).
 * state: 3 device floats [step, 1-b1^step, 1-b2^step], zero-initialised; advanced on the device so a captured
 * hipGraph replays the right bias correction.  g is multiplied by gscale first (1/world_size after all-reduce). */
int medt_adam_step(float* p, const float* g, float* m, float* v, float* state, size_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, float gscale, void* stream);

/* out = a * (y > 0) elementwise (ReLU backward by output sign). */
int medt_relu_mask(const float* a, const float* y, float* out, size_t n, void* stream);

/* Scoring without MATLAB (performancemetrics_monuseg.m:19-84, performancemetrics_glas.m): per-image confusion
 * counts of the prediction test.py writes (logits[:,1] >= threshold, test.py:131-137) against target > 0.
 * logits (N,K,HW) float, target (N,HW) int64, counts (N,4) int32 = {tp, fp, fn, tn}; the call zeroes counts itself. */
int medt_seg_counts(const float* logits, const int64_t* target, int32_t* counts, int N, int K, int HW, float threshold,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MEDT_ABI_H */
