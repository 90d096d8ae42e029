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

#define MEDT_ABI_VERSION 1

#define MEDT_OK            0
#define MEDT_EINVAL       -1   /* bad descriptor / null pointer / size mismatch            */
#define MEDT_EUNSUPPORTED -2   /* geometry outside what the kernels are built for            */
#define MEDT_ELAUNCH      -3   /* hipLaunchKernel / hipGetLastError reported a failure        */
#define MEDT_EWORKSPACE   -4   /* workspace smaller than medt_*_workspace_bytes()             */

int         medt_abi_version(void);
const char* medt_last_error(void);

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
    float* qkv_raw;   /* (N, 2C, H, W)  qkv_transform output before bn_qkv                     */
    float* stacked;   /* (N, OC, H, W)  sv|sve before bn_output, channel 2(g*gp+c)+{0:sv,1:sve};
                         OC = 2C (has_pos) or C (wopos: sv only)                               */
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
int medt_axial_layer_bwd(const medt_axial_desc*, const medt_axial_params*, const float* x, const float* dy,
                         const medt_axial_saved*, float* dx, const medt_axial_grads*,
                         void* workspace, size_t workspace_bytes, void* stream);

/* The two L x L stages on their own, for benchmarks / profiling (bench.py's roofline leg).
 * qkv_raw -> [logit statistics partials] and qkv_raw -> stacked, lse, given the BN
 * scale/shift already in saved->stats.  Same kernels the layer entry points launch. */
int medt_axial_core_stats(const medt_axial_desc*, const medt_axial_params*, const medt_axial_saved*,
                          void* workspace, size_t workspace_bytes, void* stream);
int medt_axial_core_fwd(const medt_axial_desc*, const medt_axial_params*, const medt_axial_saved*,
                        void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MEDT_ABI_H */
