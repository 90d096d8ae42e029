// medt_api.hip -- the extern "C" surface declared in include/medt_abi.h.
// Each entry point validates its descriptor, carves the caller's workspace and enqueues
// the kernel chain on the caller's stream; nothing here allocates or synchronises.
#include "medt_kernels.h"
#include <string.h>

namespace medt {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return MEDT_ELAUNCH;
    }
    return MEDT_OK;
}

struct LayerStats {
    BnStats qkv, sim, out;
    LayerStats(float* p, const AxialGeom& g) {
        const int nq = g.groups * 2 * g.C, ns = g.groups * g.SC, no = g.groups * g.OC;
        qkv = BnStats(p, nq);
        sim = BnStats(p + 4 * (size_t)nq, ns);
        out = BnStats(p + 4 * (size_t)(nq + ns), no);
    }
};

struct FwdWs {
    float *part_qkv, *part_sim, *part_out;
    FwdWs(Carver& c, const AxialGeom& g) {
        const int pt = conv1x1_ptiles(g.HW);
        part_qkv = c.take<float>((size_t)g.N * pt * 2 * g.C * 2);
        part_sim = c.take<float>((size_t)g.groups * g.tpg * g.SC * 2);
        part_out = c.take<float>((size_t)g.groups * g.tpg * g.OC * 2);
    }
};

struct BwdWs {
    float *part_ob, *coef_out, *part_sb, *coef_sim, *dqkv, *part_qb, *coef_qkv, *rel_part, *gate_part, *dw_scratch;
    size_t nblocks;
    BwdWs(Carver& c, const AxialGeom& g) {
        const int pt = conv1x1_ptiles(g.HW), TL = 2 * g.L - 1;
        nblocks = (size_t)g.groups * g.tpg * g.G;
        part_ob = c.take<float>((size_t)g.N * pt * g.OC * 2);
        coef_out = c.take<float>((size_t)g.groups * g.OC * 3);
        part_sb = c.take<float>((size_t)g.groups * g.tpg * g.G * 4);
        coef_sim = c.take<float>((size_t)g.groups * g.SC * 3);
        dqkv = c.take<float>((size_t)g.N * 2 * g.C * g.HW);
        part_qb = c.take<float>((size_t)g.groups * g.tpg * 2 * g.C * 2);
        coef_qkv = c.take<float>((size_t)g.groups * 2 * g.C * 3);
        rel_part = c.take<float>(g.pos ? nblocks * 2 * g.gp * TL : 0);
        gate_part = c.take<float>(g.pos ? nblocks * 4 : 0);
        dw_scratch = c.take<float>((size_t)conv1x1_bwd_weight_splits(g.N, g.HW) * 2 * g.C * g.C);
    }
};

static int check_common(const medt_axial_desc* d, const medt_axial_params* p, const medt_axial_saved* sv, AxialGeom* g) {
    if (!d || !p || !sv) { set_error("null descriptor / params / saved"); return MEDT_EINVAL; }
    int rc = axial_geom(*d, g);
    if (rc) return rc;
    if (!p->w_qkv || !p->bn_qkv.weight || !p->bn_qkv.bias || !p->bn_similarity.weight || !p->bn_similarity.bias ||
        !p->bn_output.weight || !p->bn_output.bias) { set_error("null parameter pointer"); return MEDT_EINVAL; }
    if (!d->training && (!p->bn_qkv.running_mean || !p->bn_qkv.running_var || !p->bn_similarity.running_mean ||
                         !p->bn_similarity.running_var || !p->bn_output.running_mean || !p->bn_output.running_var)) {
        set_error("eval mode needs running statistics"); return MEDT_EINVAL;
    }
    if (g->pos && !p->relative) { set_error("has_pos without relative table"); return MEDT_EINVAL; }
    if (!sv->qkv_raw || !sv->stacked || !sv->stats) { set_error("null saved buffer"); return MEDT_EINVAL; }
    return MEDT_OK;
}

}  // namespace medt

using namespace medt;

extern "C" {

int medt_abi_version(void) { return MEDT_ABI_VERSION; }
const char* medt_last_error(void) { return g_err; }

size_t medt_axial_stats_floats(const medt_axial_desc* d) {
    AxialGeom g;
    if (!d || axial_geom(*d, &g)) return 0;
    return (size_t)4 * g.groups * (2 * g.C + g.SC + g.OC);
}

size_t medt_axial_workspace_bytes(const medt_axial_desc* d) {
    AxialGeom g;
    if (!d || axial_geom(*d, &g)) return 0;
    Carver cf(nullptr, 0), cb(nullptr, 0);
    FwdWs f(cf, g);
    BwdWs b(cb, g);
    return align_up(cf.off > cb.off ? cf.off : cb.off, 256) + 256;
}

int medt_axial_core_stats(const medt_axial_desc* d, const medt_axial_params* p, const medt_axial_saved* sv, void* ws,
                          size_t ws_bytes, void* stream) {
    AxialGeom g;
    int rc = check_common(d, p, sv, &g);
    if (rc) return rc;
    Carver c(ws, ws_bytes);
    FwdWs w(c, g);
    if (!ws || !c.ok()) { set_error("workspace too small: need %zu", c.off); return MEDT_EWORKSPACE; }
    LayerStats st(sv->stats, g);
    GatePtrs gates{p->f_qr, p->f_kr, p->f_sve, p->f_sv};
    return axial_logit_stats(g, sv->qkv_raw, st.qkv, p->relative, gates, w.part_sim, (hipStream_t)stream);
}

int medt_axial_core_fwd(const medt_axial_desc* d, const medt_axial_params* p, const medt_axial_saved* sv, void* ws,
                        size_t ws_bytes, void* stream) {
    AxialGeom g;
    int rc = check_common(d, p, sv, &g);
    if (rc) return rc;
    Carver c(ws, ws_bytes);
    FwdWs w(c, g);
    if (!ws || !c.ok()) { set_error("workspace too small: need %zu", c.off); return MEDT_EWORKSPACE; }
    LayerStats st(sv->stats, g);
    GatePtrs gates{p->f_qr, p->f_kr, p->f_sve, p->f_sv};
    return axial_attn_fwd(g, sv->qkv_raw, st.qkv, st.sim, p->relative, gates, sv->stacked, sv->lse,
                          d->training ? w.part_out : nullptr, (hipStream_t)stream);
}

int medt_axial_layer_fwd(const medt_axial_desc* d, const medt_axial_params* p, const float* x, float* y,
                         const medt_axial_saved* sv, void* ws, size_t ws_bytes, void* stream) {
    AxialGeom g;
    int rc = check_common(d, p, sv, &g);
    if (rc) return rc;
    if (!x || !y) { set_error("null x / y"); return MEDT_EINVAL; }
    Carver c(ws, ws_bytes);
    FwdWs w(c, g);
    if (!ws || !c.ok()) { set_error("workspace too small: need %zu", c.off); return MEDT_EWORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    LayerStats st(sv->stats, g);
    GatePtrs gates{p->f_qr, p->f_kr, p->f_sve, p->f_sv};
    const int tr = d->training ? 1 : 0, pt = conv1x1_ptiles(g.HW);
    // qkv_transform (1x1 conv over channels, axis-agnostic on NCHW) + bn_qkv batch statistics      :151
    if ((rc = conv1x1_fwd(x, p->w_qkv, sv->qkv_raw, tr ? w.part_qkv : nullptr, g.N, g.C, 2 * g.C, g.HW, s))) return rc;
    if ((rc = bn_finalize(w.part_qkv, g.npg * pt, g.groups, 2 * g.C, g.row_count, p->bn_qkv, d->momentum, d->eps, tr,
                          st.qkv, s))) return rc;
    // bn_similarity batch statistics over the (never materialised) logits                         :166-167
    if (tr && (rc = axial_logit_stats(g, sv->qkv_raw, st.qkv, p->relative, gates, w.part_sim, s))) return rc;
    if ((rc = bn_finalize(w.part_sim, g.tpg, g.groups, g.SC, g.sim_count, p->bn_similarity, d->momentum, d->eps, tr,
                          st.sim, s))) return rc;
    // logits + softmax + gated sv|sve, bn_output batch statistics                                 :157-178
    if ((rc = axial_attn_fwd(g, sv->qkv_raw, st.qkv, st.sim, p->relative, gates, sv->stacked, sv->lse,
                             tr ? w.part_out : nullptr, s))) return rc;
    if ((rc = bn_finalize(w.part_out, g.tpg, g.groups, g.OC, g.row_count, p->bn_output, d->momentum, d->eps, tr,
                          st.out, s))) return rc;
    // bn_output + pair-sum + AvgPool                                                              :179-187
    return axial_out_fwd(*d, sv->stacked, st.out, y, s);
}

int medt_axial_layer_bwd(const medt_axial_desc* d, const medt_axial_params* p, const float* x, const float* dy,
                         const medt_axial_saved* sv, float* dx, const medt_axial_grads* gr, void* ws, size_t ws_bytes,
                         void* stream) {
    AxialGeom g;
    int rc = check_common(d, p, sv, &g);
    if (rc) return rc;
    if (!x || !dy || !dx || !gr || !sv->lse) { set_error("null x / dy / dx / grads / lse"); return MEDT_EINVAL; }
    if (!gr->w_qkv || !gr->bn_qkv_weight || !gr->bn_qkv_bias || !gr->bn_sim_weight || !gr->bn_sim_bias ||
        !gr->bn_out_weight || !gr->bn_out_bias || (g.pos && !gr->relative)) {
        set_error("null gradient pointer"); return MEDT_EINVAL;
    }
    Carver c(ws, ws_bytes);
    BwdWs w(c, g);
    if (!ws || !c.ok()) { set_error("workspace too small: need %zu", c.off); return MEDT_EWORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    LayerStats st(sv->stats, g);
    GatePtrs gates{p->f_qr, p->f_kr, p->f_sve, p->f_sv};
    const int tr = d->training ? 1 : 0, pt = conv1x1_ptiles(g.HW), TL = 2 * g.L - 1;
    // AvgPool + bn_output backward (statistics, then coefficients applied on load downstream)
    if ((rc = axial_out_bwd_stats(*d, sv->stacked, dy, st.out, w.part_ob, s))) return rc;
    if ((rc = bn_bwd_finalize(w.part_ob, g.npg * pt, g.groups, g.OC, g.row_count, 1.f / (float)(d->stride * d->stride),
                              st.out, p->bn_output.weight, tr, w.coef_out, gr->bn_out_weight, gr->bn_out_bias, s)))
        return rc;
    // bn_similarity backward statistics (pass A), coefficients
    if ((rc = axial_attn_bwd_stats(g, sv->qkv_raw, st.qkv, st.sim, p->relative, gates, sv->stacked, sv->lse, dy,
                                   w.coef_out, d->stride, w.part_sb, s))) return rc;
    if ((rc = axial_sim_bwd_finalize(g, w.part_sb, st.sim, p->bn_similarity.weight, tr, w.coef_sim, gr->bn_sim_weight,
                                     gr->bn_sim_bias, s))) return rc;
    // attention backward (pass B)
    if ((rc = axial_attn_bwd(g, sv->qkv_raw, st.qkv, st.sim, w.coef_sim, p->relative, gates, sv->stacked, sv->lse, dy,
                             w.coef_out, d->stride, w.dqkv, w.part_qb, w.rel_part, gr->gates ? w.gate_part : nullptr,
                             s))) return rc;
    // bn_qkv backward, qkv_transform backward
    if ((rc = bn_bwd_finalize(w.part_qb, g.tpg, g.groups, 2 * g.C, g.row_count, 1.f, st.qkv, p->bn_qkv.weight, tr,
                              w.coef_qkv, gr->bn_qkv_weight, gr->bn_qkv_bias, s))) return rc;
    if ((rc = conv1x1_bwd_data(w.dqkv, sv->qkv_raw, w.coef_qkv, p->w_qkv, dx, g.N, g.C, 2 * g.C, g.HW, g.groups, s)))
        return rc;
    if ((rc = conv1x1_bwd_weight(w.dqkv, sv->qkv_raw, w.coef_qkv, x, gr->w_qkv, w.dw_scratch, g.N, g.C, 2 * g.C, g.HW,
                                 g.groups, s))) return rc;
    if (g.pos) {
        if ((rc = reduce_rows(w.rel_part, (int)w.nblocks, 2 * g.gp * TL, gr->relative, s))) return rc;
        if (gr->gates && (rc = reduce_rows(w.gate_part, (int)w.nblocks, 4, gr->gates, s))) return rc;
    }
    return MEDT_OK;
}

}  // extern "C"
