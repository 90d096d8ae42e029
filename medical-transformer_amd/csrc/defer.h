// defer.h -- deferred, grouped execution of the work nothing in the layer chain waits for.
//
// At BASELINE.json's batch sizes a training step is ~600 dependent launches of a few microseconds each
// (SURVEY.md H2): latency, not bandwidth.  A third of those launches produce results that no later *layer* consumes --
// weight / bias gradients and the reductions of their split partial slabs (only the optimizer reads them), the
// saved-statistics + running-stat bookkeeping of the fused small-layer kernels (only backward reads them), the
// BatchNorm parameter gradients of the fused small backward.  With a queue bound to the caller's stream
// (medt_queue_bind) the entry points record those launches as jobs instead of issuing them, and medt_queue_flush
// issues everything recorded as a handful of GROUPED launches (one kernel, many problems: job table in the kernel
// arguments, block -> job by prefix sums), at the end of the forward and of the backward pass.  The caller keeps
// every buffer it handed to a deferring call alive until the flush.
#pragma once
#include "medt_kernels.h"
#include <vector>

namespace medt {

struct BnFin {                       // one BatchNorm statistics finalisation (see bn_finalize)
    const float* partials;
    int ppg, CH;
    double count;
    const float *weight, *bias;
    float *running_mean, *running_var;
    int64_t* nbt;
    BnStats out;
};
struct FinJob { BnFin f; int groups, training; float momentum, eps; };
struct BfinJob {                     // bn_bwd_finalize
    const float* partials;
    int ppg, groups, CH, training;
    double count;
    float dscale;
    BnStats st;
    const float* weight;
    float *coef, *dweight, *dbias;
};
struct SmallFinArgs {                // wopos_small_bwd_finalize: BatchNorm parameter gradients of a fused small layer
    const float *part_ob, *part_sb, *part_qb;
    BnStats so, ss, sq;
    float *d_out_w, *d_out_b, *d_sim_w, *d_sim_b, *d_qkv_w, *d_qkv_b;
    int C, G, groups, training;
    double row_count, sim_count;
    float dscale_out;
};
struct RJob { const float* src; float* dst; int P, K; };                         // reduce_rows
struct FlipJob { const float* w; float* wt; int Cout, Cin, K; };                  // conv_flip_weights (forward pass, for the MFMA backward-data)
struct CJob { const float* x; float* part; int N, C, HW, splits; };              // channel_sum: part[splits][C] (splits == 1: the gradient itself)
struct WJob {                        // conv_wgrad_body<K, 64, 64> over a (gx, gy, gz) grid of (o-tile, k-tile, position chunk)
    const float *dy, *raw, *coef, *x;
    float* scratch;
    int N, Cin, H, W, Cout, Ho, Wo, stride, pad, QS, npg, gx, gy, gz, K;
    int v4;                          // 1: the 16-byte position-axis body (conv_wgrad_v4_ok; QS is a multiple of 128 then)
};

struct MJob {                        // conv_wgrad_mfma: the dedicated MFMA weight-gradient kernels (LDS-patch kernel of the 16-wide
    const float *dy, *raw, *coef, *x;    // maps, few-tile split-K kernels), one launch per job at the flush
    float* scratch;
    int N, Cin, H, W, Cout, Ho, Wo, K, stride, pad, QS, splits, npg;
};

struct RelfixJob {                   // attn_bwd_relfix (axial_bwd.hip) kernel arguments: u / w terms of one layer's relative-table and gate gradients
    const float *relative, *sim_coef, *pg_part, *gate_raw;
    BnStats ss;
    GatePtrs gates;
    float *rel_rows, *gate_rows;     // [groups * G][2gp * TL], [groups * G][4]
    double sim_count;
    int L, G, SC, hq, nparts, sweep_gridx, training, blocks;
    unsigned lds;
    float eps;
    // rider (round 6, fin_inline.h): bn_qkv's backward finalisation of this head's channels -- the kernel sits between the fix
    // kernel that completes the partial rows and the 1x1 backward-data / weight-gradient kernels that apply the coefficients
    BfinJob qb;
    int qb_on, qb_nch;
};

struct Queue {
    std::vector<FinJob> fin;
    std::vector<BfinJob> bfin;
    std::vector<SmallFinArgs> sfin;
    std::vector<CJob> csum;
    std::vector<WJob> wgrad;
    std::vector<MJob> mwgrad;
    std::vector<RJob> reduce;
    std::vector<FlipJob> flip;
    size_t pending() const {
        return flip.size() + fin.size() + bfin.size() + sfin.size() + csum.size() + wgrad.size() + mwgrad.size() + reduce.size();
    }
};

Queue* queue_for(hipStream_t s);     // the queue bound to this stream, or nullptr (immediate launches)

BnFin make_fin(const float* partials, int ppg, int CH, double count, const medt_bn_ptrs& bn, BnStats out);

// grouped launchers (each may issue several launches when the job table exceeds one kernel-argument block)
int conv_flip_weights_grouped(const FlipJob* jobs, int n, hipStream_t s);     // conv_mfma.hip
int bn_finalize_grouped(const FinJob* jobs, int n, hipStream_t s);
int bn_bwd_finalize_grouped(const BfinJob* jobs, int n, hipStream_t s);
int wopos_small_bwd_finalize_grouped(const SmallFinArgs* jobs, int n, hipStream_t s);
int reduce_rows_grouped(const RJob* jobs, int n, hipStream_t s);
int channel_sum_grouped(const CJob* jobs, int n, hipStream_t s);
int conv_wgrad_grouped(const WJob* jobs, int n, hipStream_t s);            // MFMA tiles (conv_mfma.hip)
bool conv_wgrad_v4_ok(const float* dy, const float* raw, const float* x, int N, int Cin, int H, int W, int Cout, int Ho, int Wo,
                      int K, int stride, int pad);
bool conv_wgrad_rows16_ok(int Cin, int H, int W, int Ho, int Wo, int K, int stride, int pad, int QS);
int conv_wgrad_rows16_grouped(const MJob* const* jobs, int n, hipStream_t s);   // LDS-patch kernel of the 16-wide maps, <= 4 problems per launch
int conv_wgrad_mfma_batch(const MJob* const* jobs, int n, hipStream_t s);       // generic MFMA tile kernel (K = 1 | 3), <= 4 problems per launch

// Job table passed by value in the kernel arguments (< 4 KB): block b belongs to job j with start[j] <= b < start[j+1].
template <class J, int MAXJ>
struct JobBatch {
    int n;
    int start[MAXJ + 1];
    J job[MAXJ];
};
template <class B>
__device__ __forceinline__ int find_job(const B& b, int block) {       // block-uniform binary search
    int lo = 0, hi = b.n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (block >= b.start[mid]) lo = mid; else hi = mid;
    }
    return lo;
}

}  // namespace medt
