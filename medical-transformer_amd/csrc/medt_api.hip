// medt_api.hip -- the extern "C" surface declared in include/medt_abi.h.
// Each entry point validates its descriptor, carves the caller's workspace and enqueues
// the kernel chain on the caller's stream; nothing here allocates or synchronises.
#include "defer.h"
#include "fin_inline.h"
#include <string.h>

namespace medt {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// TIMING EXPERIMENTS ONLY, compiled in with -DMEDT_ABLATE (scripts/r6_skip.sh builds libmedt_ablate.so and loads it through
// MEDT_LIB_OVERRIDE): MEDT_SKIP=<families> makes the named entry points return without launching.  The product library answers false.
bool abl_skip(const char* family) {
#ifndef MEDT_ABLATE
    (void)family;
    return false;
#else
    static const char* env = [] {
        const char* e = getenv("MEDT_SKIP");
        if (e && *e)
            fprintf(stderr, "libmedt_hip: MEDT_SKIP=%s -- TIMING EXPERIMENT: these kernel families are NOT launched, every result of "
                            "this process is garbage\n", e);
        return e;
    }();
    if (!env || !*env) return false;
    const size_t n = strlen(family);
    for (const char* p = env; *p;) {
        const char* e = strchr(p, ',');
        const size_t len = e ? (size_t)(e - p) : strlen(p);
        if (len == n && strncmp(p, family, n) == 0) return true;
        p += len + (e ? 1 : 0);
    }
    return false;
#endif
}

int lds_opt_in(const void* kernel, unsigned char (&done)[64], const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    if (dev < 64 && done[dev]) return MEDT_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("%s: opt-in for more than 64 KB of dynamic LDS refused on device %d: %s", what, dev, hipGetErrorString(e));
        return MEDT_ELAUNCH;
    }
    if (dev < 64) done[dev] = 1;
    return MEDT_OK;
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
    float *part_qkv, *part_sim, *part_out, *qkv_ksplit, *tables, *gate_eff;
    unsigned* flag;
    FwdWs(Carver& c, const AxialGeom& g) {
        flag = c.take<unsigned>(4);
        gate_eff = c.take<float>(4);
        const size_t kq = conv2d_fwd_scratch_floats(g.N, g.groups, g.C, g.H, g.W, 2 * g.C, 1, 1, 0);
        qkv_ksplit = c.take<float>(kq);
        if (!kq) qkv_ksplit = nullptr;
        // forward statistics partials are [..][CH][2] DOUBLES (block_sum_d, medt_common.h): twice the floats
        part_qkv = c.take<float>((size_t)g.groups * conv_parts_per_group(g.N, g.groups, g.HW, g.C, 2 * g.C, 1, 1) * 2 * g.C * 2 * 2);
        part_sim = c.take<float>((size_t)g.groups * sim_stats_parts(g) * g.SC * 2 * 2);
        tables = c.take<float>(sim_tables_floats(g));
        part_out = c.take<float>((size_t)g.groups * g.tpg * g.OC * 2 * 2);
    }
};

struct BwdWs {
    float *part_ob, *coef_out, *part_sb, *coef_sim, *dqkv, *part_qb, *coef_qkv, *rel_part, *gate_part, *dw_scratch,
        *dy_masked, *gate_eff, *gate_tmp;
    float *tables, *gram, *pg_part, *gate_raw, *gate_rows;     // single-sweep backward (axial_bwd.hip)
    float* raw32;                // bf16 storage + sweep: qkv_raw as float32 (written by the sweep, read by the 1x1 dgrad / wgrad)
    size_t nblocks;
    bool sweep;                  // the single-sweep backward runs (else the two generic passes of axial_core.hip)
    SweepPlan plan;
    size_t sweep_blocks;         // workgroups of the sweep = rows of rel_part / pg_part / gate_raw it writes
    int qb_rpg;                  // rows per BN group of part_qb
    BwdWs(Carver& c, const AxialGeom& g, int stride, int out_relu, int gate_mode) {
        gate_eff = c.take<float>(4);
        gate_tmp = c.take<float>(4);
        const int ppg = conv2d_parts_per_group(g.N, g.groups, g.HW), TL = 2 * g.L - 1;
        sweep = axial_bwd_sweep_plan(g, gate_mode == 2 ? 4 : 0, &plan);
        sweep_blocks = sweep ? (size_t)g.groups * plan.nparts * g.G : 0;
        qb_rpg = sweep ? plan.nparts + plan.fparts : g.tpg;
        nblocks = (size_t)g.groups * g.tpg * g.G;
        const size_t rel_rows = sweep ? sweep_blocks + (size_t)g.groups * g.G : nblocks;
        part_ob = c.take<float>((size_t)g.groups * ppg * g.OC * 2);
        coef_out = c.take<float>((size_t)g.groups * g.OC * 3);
        // (rows per group: the generic pass writes one per forward tile, the sweep one per persistent workgroup -- with 16 lanes
        //  per sequence on short sequences, gp = 8, a group has more sweep workgroups than forward tiles)
        part_sb = c.take<float>((size_t)g.groups * (sweep && plan.nparts > g.tpg ? plan.nparts : g.tpg) * g.G * 4);
        coef_sim = c.take<float>((size_t)g.groups * g.SC * 3);
        dqkv = c.take<float>((size_t)g.N * 2 * g.C * g.HW);
        part_qb = c.take<float>((size_t)g.groups * (qb_rpg > g.tpg ? qb_rpg : g.tpg) * 2 * g.C * 2);
        coef_qkv = c.take<float>((size_t)g.groups * 2 * g.C * 3);
        rel_part = c.take<float>(g.pos ? (rel_rows > nblocks ? rel_rows : nblocks) * 2 * g.gp * TL : 0);
        const size_t nseqh = (size_t)g.groups * g.spg * g.G;          // per-sequence gates: one row per (sequence, head)
        gate_part = c.take<float>(g.pos ? (nblocks > nseqh ? nblocks : nseqh) * 4 : 0);
        dw_scratch = c.take<float>((size_t)conv2d_bwd_weight_splits(g.N, g.C, 2 * g.C, 1, g.H, g.W) * 2 * g.C * g.C);
        dy_masked = c.take<float>(out_relu ? (size_t)g.N * g.C * (g.H / stride) * (g.W / stride) : 0);
        tables = c.take<float>(sweep ? sim_tables_floats(g) : 0);
        gram = c.take<float>(sweep ? (size_t)g.groups * g.spg * g.G * plan.npg_floats : 0);
        pg_part = c.take<float>(sweep ? sweep_blocks * g.L * plan.npg_floats : 0);
        gate_raw = c.take<float>(sweep ? sweep_blocks * 4 : 0);
        gate_rows = c.take<float>(sweep ? (size_t)g.groups * g.G * 4 : 0);
        static const bool raw32_on = true;
        raw32 = c.take<float>(sweep && g.bf16 && raw32_on ? (size_t)g.N * 2 * g.C * g.HW : 0);
        if (!(sweep && g.bf16 && raw32_on)) raw32 = nullptr;
    }
};

// The attention backward between bn_output's and bn_qkv's: dy -> dqkv (+ bn_qkv backward partials, relative-table and gate
// partial rows, bn_similarity's parameter gradients).  Either the single sweep (+ the closed-form u / w terms) of
// axial_bwd.hip, or the two generic L x L passes of axial_core.hip.
static int attention_core_bwd(const AxialGeom& g, const medt_axial_desc* d, const medt_axial_params* p, const BwdWs& w,
                              const float* qkv_raw, const float* stacked, const float* lse, const float* dy, const LayerStats& st,
                              GatePtrs gates, bool want_gates, float* d_sim_w, float* d_sim_b, hipStream_t s,
                              const BfinSrc* ob = nullptr, bool sim_inline = false, const BfinSrc* qb = nullptr,
                              const float* ymask = nullptr) {
    const int tr = d->training ? 1 : 0;
    int rc;
    if (w.sweep) {
        if ((rc = axial_attn_bwd_sweep(g, w.plan, qkv_raw, st.qkv, st.sim, p->relative, gates, stacked, lse, dy, w.coef_out,
                                       d->stride, w.dqkv, w.part_qb, w.qb_rpg, w.part_sb, w.rel_part, w.pg_part, w.gram,
                                       want_gates ? w.gate_raw : nullptr, s, w.raw32, ob, ymask))) return rc;
        AxialGeom gs = g;
        gs.tpg = w.plan.nparts;                               // part_sb rows per group
        SimBSrc sb = no_simb_src();
        if (sim_inline) {
            // (round 6, fin_inline.h: the fix kernel derives bn_similarity's backward coefficients from the sweep's partial rows and
            //  writes them for the relfix kernel; the tables were built by the first workgroups of axial_out_bwd_stats)
            sb.partials = w.part_sb; sb.rows = w.plan.nparts; sb.G = g.G; sb.SC = g.SC; sb.training = tr; sb.on = 1;
            sb.count = g.sim_count; sb.ss = st.sim; sb.weight = p->bn_similarity.weight;
            sb.coef = w.coef_sim; sb.dweight = d_sim_w; sb.dbias = d_sim_b;
        } else {
            // (+ the sliding-window table sums of the fix kernel as extra blocks of this launch)
            const TablesJob tj{p->relative, w.tables, g.hq, g.L, tr ? sim_tables_blocks(g) : 0};
            if ((rc = axial_sim_bwd_finalize(gs, w.part_sb, st.sim, p->bn_similarity.weight, tr, w.coef_sim, d_sim_w, d_sim_b, s, &tj)))
                return rc;
        }
        if ((rc = axial_attn_bwd_fix(g, w.plan, qkv_raw, st.qkv, w.coef_sim, w.tables, w.gram, gates, tr, w.dqkv, w.part_qb,
                                     w.qb_rpg, s, sim_inline ? &sb : nullptr))) return rc;
        return axial_attn_bwd_relfix(g, w.plan, p->relative, w.coef_sim, st.sim, gates, w.pg_part,
                                     want_gates ? w.gate_raw : nullptr, tr, d->eps,
                                     w.rel_part + w.sweep_blocks * 2 * g.gp * (2 * g.L - 1), w.gate_rows, s, qb);
    }
    // bn_similarity backward statistics (pass A), coefficients, attention backward (pass B)
    if ((rc = axial_attn_bwd_stats(g, qkv_raw, st.qkv, st.sim, p->relative, gates, stacked, lse, dy, w.coef_out, d->stride,
                                   w.part_sb, s))) return rc;
    if ((rc = axial_sim_bwd_finalize(g, w.part_sb, st.sim, p->bn_similarity.weight, tr, w.coef_sim, d_sim_w, d_sim_b, s)))
        return rc;
    return axial_attn_bwd(g, qkv_raw, st.qkv, st.sim, w.coef_sim, p->relative, gates, stacked, lse, dy, w.coef_out, d->stride,
                          w.dqkv, w.part_qb, w.rel_part, want_gates ? w.gate_part : nullptr, s);
}

// The gates the kernels multiply with: the stored scalars, or (gate_mode 1) their sigmoids computed into `eff`.
static int effective_gates(const medt_axial_desc* d, const medt_axial_params* p, float* eff, hipStream_t s, GatePtrs* out) {
    *out = GatePtrs{p->f_qr, p->f_kr, p->f_sve, p->f_sv, 0};
    if (d->gate_mode == 0 || !p->f_qr) return MEDT_OK;
    if (d->gate_mode == 2) {      // f_qr = the (B*, 4) gate tensor, columns (qr, kr, sv, sve); the other pointers are ignored
        *out = GatePtrs{p->f_qr, p->f_qr + 1, p->f_qr + 3, p->f_qr + 2, 4};
        return MEDT_OK;
    }
    if (d->gate_mode != 1) { set_error("axial: gate_mode %d unsupported (0: raw, 1: sigmoid, 2: per sequence)", d->gate_mode); return MEDT_EUNSUPPORTED; }
    if (!p->f_kr || !p->f_sve || !p->f_sv) { set_error("axial: gate_mode 1 needs all four gates"); return MEDT_EINVAL; }
    int rc = gate_sigmoid_fwd(p->f_qr, p->f_kr, p->f_sve, p->f_sv, eff, s);
    if (rc) return rc;
    *out = GatePtrs{eff, eff + 1, eff + 2, eff + 3, 0};
    return MEDT_OK;
}

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
    BwdWs b(cb, g, d->stride, d->out_relu, d->gate_mode);
    return align_up(cf.off > cb.off ? cf.off : cb.off, 256) + 256;
}

int medt_axial_core_stats(const medt_axial_desc* d, const medt_axial_params* p, const medt_axial_saved* sv, void* ws,
                          size_t ws_bytes, void* stream) {
    AxialGeom g;
    int rc = check_common(d, p, sv, &g);
    if (rc) return rc;
    float* qkv_raw = (float*)sv->qkv_raw;         // float32 or bfloat16 storage (g.bf16)
    float* stacked = (float*)sv->stacked;
    Carver c(ws, ws_bytes);
    FwdWs w(c, g);
    if (!ws || !c.ok()) { set_error("workspace too small: need %zu", c.off); return MEDT_EWORKSPACE; }
    LayerStats st(sv->stats, g);
    GatePtrs gates;
    if ((rc = effective_gates(d, p, w.gate_eff, (hipStream_t)stream, &gates))) return rc;
    // (the sliding-window tables are the ones the preceding medt_axial_layer_fwd left in this workspace)
    return axial_logit_stats(g, qkv_raw, st.qkv, p->relative, gates, w.tables, w.part_sim, (hipStream_t)stream);
}

int medt_axial_core_fwd(const medt_axial_desc* d, const medt_axial_params* p, const medt_axial_saved* sv, void* ws,
                        size_t ws_bytes, void* stream) {
    AxialGeom g;
    int rc = check_common(d, p, sv, &g);
    if (rc) return rc;
    float* qkv_raw = (float*)sv->qkv_raw;         // float32 or bfloat16 storage (g.bf16)
    float* stacked = (float*)sv->stacked;
    Carver c(ws, ws_bytes);
    FwdWs w(c, g);
    if (!ws || !c.ok()) { set_error("workspace too small: need %zu", c.off); return MEDT_EWORKSPACE; }
    LayerStats st(sv->stats, g);
    GatePtrs gates;
    if ((rc = effective_gates(d, p, w.gate_eff, (hipStream_t)stream, &gates))) return rc;
    return axial_attn_fwd(g, qkv_raw, st.qkv, st.sim, p->relative, gates, stacked, sv->lse,
                          d->training ? w.part_out : nullptr, w.flag, (hipStream_t)stream);
}

int medt_axial_core_bwd(const medt_axial_desc* d, const medt_axial_params* p, const medt_axial_saved* sv, const float* dy,
                        void* ws, size_t ws_bytes, void* stream) {
    AxialGeom g;
    int rc = check_common(d, p, sv, &g);
    if (rc) return rc;
    float* qkv_raw = (float*)sv->qkv_raw;
    float* stacked = (float*)sv->stacked;
    if (!dy || !sv->lse) { set_error("core_bwd: null dy / lse"); return MEDT_EINVAL; }
    if (d->out_relu) { set_error("core_bwd: out_relu layers are not supported by the benchmark entry"); return MEDT_EUNSUPPORTED; }
    Carver c(ws, ws_bytes);
    BwdWs w(c, g, d->stride, d->out_relu, d->gate_mode);
    if (!ws || !c.ok()) { set_error("workspace too small: need %zu", c.off); return MEDT_EWORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    LayerStats st(sv->stats, g);
    GatePtrs gates;
    if ((rc = effective_gates(d, p, w.gate_eff, s, &gates))) return rc;
    // everything between bn_output's and bn_qkv's backward: the single sweep + its closed-form corrections, or the two
    // generic L x L passes; coef_out is whatever the preceding medt_axial_layer_bwd left in the workspace, bn_similarity's
    // parameter gradients land in scratch (part_ob is dead by then)
    return attention_core_bwd(g, d, p, w, qkv_raw, stacked, sv->lse, dy, st, gates, p->f_qr != nullptr, w.part_ob,
                              w.part_ob + g.SC, s);
}

int medt_axial_layer_fwd(const medt_axial_desc* d, const medt_axial_params* p, const float* x, float* y,
                         const medt_axial_saved* sv, void* ws, size_t ws_bytes, void* stream) {
    AxialGeom g;
    int rc = check_common(d, p, sv, &g);
    if (rc) return rc;
    float* qkv_raw = (float*)sv->qkv_raw;         // float32 or bfloat16 storage (g.bf16)
    float* stacked = (float*)sv->stacked;
    if (!x || !y) { set_error("null x / y"); return MEDT_EINVAL; }
    if (d->training && g.row_count <= 1.0) {      // nn.BatchNorm raises here too; the unbiased running variance divides by count-1
        set_error("axial: training-mode BatchNorm needs more than 1 value per channel (got %g)", g.row_count);
        return MEDT_EINVAL;
    }
    Carver c(ws, ws_bytes);
    FwdWs w(c, g);
    if (!ws || !c.ok()) { set_error("workspace too small: need %zu", c.off); return MEDT_EWORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    LayerStats st(sv->stats, g);
    GatePtrs gates;
    if ((rc = effective_gates(d, p, w.gate_eff, s, &gates))) return rc;
    const int tr = d->training ? 1 : 0, ppg = conv_parts_per_group(g.N, g.groups, g.HW, g.C, 2 * g.C, 1, 1);
    if (wopos_small_ok(g, *d)) {
        // tiny position-free layers (MedT's local branch): the whole layer in one workgroup per (BN group, head),
        // then one finalisation launch for the saved statistics and the ordered running-stat updates
        if ((rc = wopos_small_fwd(g, *d, *p, x, y, qkv_raw, stacked, sv->lse, w.part_qkv, w.part_sim, w.part_out,
                                  s))) return rc;
        // (saved statistics + ordered running-stat updates: y is final already, nothing in the forward chain reads
        // them -> recorded for the grouped flush when a queue is bound)
        if (Queue* q = queue_for(s)) {
            q->fin.push_back(FinJob{make_fin(w.part_qkv, 1, 2 * g.C, g.row_count, p->bn_qkv, st.qkv), g.groups, tr, d->momentum, d->eps});
            q->fin.push_back(FinJob{make_fin(w.part_sim, 1, g.SC, g.sim_count, p->bn_similarity, st.sim), g.groups, tr, d->momentum, d->eps});
            q->fin.push_back(FinJob{make_fin(w.part_out, 1, g.OC, g.row_count, p->bn_output, st.out), g.groups, tr, d->momentum, d->eps});
            return MEDT_OK;
        }
        return bn_finalize3(w.part_qkv, 2 * g.C, g.row_count, p->bn_qkv, st.qkv, w.part_sim, g.SC, g.sim_count,
                            p->bn_similarity, st.sim, w.part_out, g.OC, g.row_count, p->bn_output, st.out, 1, g.groups,
                            d->momentum, d->eps, tr, s);
    }
    // qkv_transform (1x1 conv over channels, axis-agnostic on NCHW) + bn_qkv batch statistics      :151
    if ((rc = conv2d_fwd(x, p->w_qkv, nullptr, qkv_raw, tr ? w.part_qkv : nullptr, w.qkv_ksplit, g.N, g.C, g.H, g.W,
                         2 * g.C, 1, 1, 0, 0, g.groups, s, g.bf16))) return rc;
    // (+ the sliding-window tables of the statistics kernel below as extra blocks of this launch)
    const TablesJob tj{p->relative, w.tables, g.hq, g.L, tr ? sim_tables_blocks(g) : 0};
    if ((rc = bn_finalize(w.part_qkv, ppg, g.groups, 2 * g.C, g.row_count, p->bn_qkv, d->momentum, d->eps, tr,
                          st.qkv, s, &tj))) return rc;
    // (round 6, measured and removed: bn_qkv finalised by the statistics kernel itself -- every workgroup for the channels of its
    //  head from the qkv convolution's partial rows, the tables riding on the convolution's launch -- one launch fewer per layer
    //  and SLOWER: the load round trip + double arithmetic in front of the statistics kernel's own loads cost more than the 5-us
    //  launch they replace (gatedaxialunet bs 8: 3.95 vs 3.86 ms, MedT 128: 1.881 vs 1.874; profiles/r06_qkv_inline_ab.txt))
    // bn_similarity batch statistics over the (never materialised) logits                         :166-167
    if (tr && (rc = axial_logit_stats(g, qkv_raw, st.qkv, p->relative, gates, w.tables, w.part_sim, s))) return rc;
    // Round 6 (fin_inline.h): inside the networks (training mode, one BatchNorm group, few partial rows) bn_similarity and bn_output
    // are finalised by the kernels that consume them -- the attention kernel and the output pass -- instead of by a bn_finalize
    // launch each: 7 launches per layer become 5.
    auto fin_src_of = [&](const float* partials, int parts, int CH, double count, const medt_bn_ptrs& bn, BnStats out) {
        FinSrc fs = no_fin_src();
        fs.f = make_fin(partials, parts, CH, count, bn, out);
        fs.momentum = d->momentum; fs.eps = d->eps; fs.on = 1;
        return fs;
    };
    const bool sim_inl = inline_fin_ok(tr, g.groups, sim_stats_parts(g)) && axial_attn_fwd_inlines(g, gates, w.flag);
    const FinSrc sim_src = fin_src_of(w.part_sim, sim_stats_parts(g), g.SC, g.sim_count, p->bn_similarity, st.sim);
    if (!sim_inl && (rc = bn_finalize(w.part_sim, sim_stats_parts(g), g.groups, g.SC, g.sim_count, p->bn_similarity, d->momentum,
                                      d->eps, tr, st.sim, s))) return rc;
    // logits + softmax + gated sv|sve, bn_output batch statistics                                 :157-178
    if ((rc = axial_attn_fwd(g, qkv_raw, st.qkv, st.sim, p->relative, gates, stacked, sv->lse,
                             tr ? w.part_out : nullptr, w.flag, s, sim_inl ? &sim_src : nullptr))) return rc;
    const bool out_inl = inline_fin_ok(tr, g.groups, g.oparts) && axial_out_fwd_inlines(*d);
    const FinSrc out_src = fin_src_of(w.part_out, g.oparts, g.OC, g.row_count, p->bn_output, st.out);
    if (!out_inl && (rc = bn_finalize(w.part_out, g.oparts, g.groups, g.OC, g.row_count, p->bn_output, d->momentum, d->eps, tr,
                                      st.out, s))) return rc;
    // bn_output + pair-sum + AvgPool                                                              :179-187
    return axial_out_fwd(*d, stacked, st.out, y, s, out_inl ? &out_src : nullptr);
}

int medt_axial_layer_bwd(const medt_axial_desc* d, const medt_axial_params* p, const float* x, const float* y,
                         const float* dy, const medt_axial_saved* sv, float* dx, const medt_axial_grads* gr, void* ws,
                         size_t ws_bytes, void* stream) {
    AxialGeom g;
    int rc = check_common(d, p, sv, &g);
    if (rc) return rc;
    float* qkv_raw = (float*)sv->qkv_raw;         // float32 or bfloat16 storage (g.bf16)
    float* stacked = (float*)sv->stacked;
    if (!x || !dy || !dx || !gr || !sv->lse) { set_error("null x / dy / dx / grads / lse"); return MEDT_EINVAL; }
    if (!gr->w_qkv || !gr->bn_qkv_weight || !gr->bn_qkv_bias || !gr->bn_sim_weight || !gr->bn_sim_bias ||
        !gr->bn_out_weight || !gr->bn_out_bias || (g.pos && !gr->relative)) {
        set_error("null gradient pointer"); return MEDT_EINVAL;
    }
    if (d->out_relu && !y) { set_error("out_relu backward needs the forward output y"); return MEDT_EINVAL; }
    Carver c(ws, ws_bytes);
    BwdWs w(c, g, d->stride, d->out_relu, d->gate_mode);
    if (!ws || !c.ok()) { set_error("workspace too small: need %zu", c.off); return MEDT_EWORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    LayerStats st(sv->stats, g);
    GatePtrs gates;
    if ((rc = effective_gates(d, p, w.gate_eff, s, &gates))) return rc;
    const int tr = d->training ? 1 : 0, ppg = conv2d_parts_per_group(g.N, g.groups, g.HW), TL = 2 * g.L - 1;
    if (wopos_small_bwd_ok(g, *d)) {
        // tiny position-free layers: dy -> gradient at the qkv_transform output in one workgroup per (BN group, head);
        // one finalisation launch then sums the per-group partials into the BatchNorm parameter gradients and writes
        // bn_qkv's backward coefficients for the conv kernels
        if ((rc = wopos_small_bwd(g, *d, *p, y, dy, qkv_raw, stacked, sv->lse, st.qkv, st.sim, st.out, w.dqkv,
                                  w.part_ob, w.part_sb, w.part_qb, w.coef_qkv, s))) return rc;
        if ((rc = wopos_small_bwd_finalize(g, *d, *p, w.part_ob, w.part_sb, w.part_qb, st.qkv, st.sim, st.out, *gr, s,
                                           queue_for(s)))) return rc;
        if ((rc = conv1x1_bwd_data(w.dqkv, qkv_raw, w.coef_qkv, p->w_qkv, dx, g.N, g.C, 2 * g.C, g.HW, g.groups, s)))
            return rc;
        return conv2d_bwd_weight(w.dqkv, qkv_raw, w.coef_qkv, x, gr->w_qkv, w.dw_scratch, g.N, g.C, g.H, g.W, 2 * g.C,
                                 1, 1, 0, g.groups, s, queue_for(s));
    }
    const float* ymask = nullptr;
    if (d->out_relu) {               // fused ReLU after the layer: mask the incoming gradient by the output sign
#ifndef MEDT_AB_NO_MASK_FUSE           // (A/B build: medt_amd.build.build_ab)
        // (round 6: the single sweep and the statistics kernel in front of it apply the mask as they load dy -- no relu_mask launch)
        if (w.sweep) ymask = y;
#endif
        if (!ymask) {
            if ((rc = relu_mask(dy, y, w.dy_masked, (size_t)g.N * g.C * (g.H / d->stride) * (g.W / d->stride), s))) return rc;
            dy = w.dy_masked;
        }
    }
    // AvgPool + bn_output backward (statistics, then coefficients applied on load downstream)
    const float out_dscale = 1.f / (float)(d->stride * d->stride);
    // (round 6: where the fix kernel finalises bn_similarity's backward itself, the tables it reads ride on this launch)
    const bool sim_inl = w.sweep && g.pos && inline_fin_ok(tr, g.groups, w.plan.nparts) &&
                         axial_out_bwd_stats_tables_ok(*d, sim_tables_blocks(g), g.L);
    const TablesJob btj{p->relative, w.tables, g.hq, g.L, sim_tables_blocks(g)};
    if ((rc = axial_out_bwd_stats(*d, stacked, dy, st.out, w.part_ob, s, sim_inl ? &btj : nullptr, ymask))) return rc;
    // (round 6, fin_inline.h: where the single sweep runs inside the networks, IT derives bn_output's backward coefficients from the
    //  partial rows and its first workgroup per head writes them and the parameter gradients -- no bn_bwd_finalize launch)
    BfinSrc ob = no_bfin_src();
    if (w.sweep && inline_fin_ok(tr, g.groups, ppg)) {
        ob.j = BfinJob{w.part_ob, ppg, g.groups, g.OC, tr, g.row_count, out_dscale, st.out, p->bn_output.weight, w.coef_out,
                       gr->bn_out_weight, gr->bn_out_bias};
        ob.on = 1;
    } else if ((rc = bn_bwd_finalize(w.part_ob, ppg, g.groups, g.OC, g.row_count, out_dscale, st.out, p->bn_output.weight, tr,
                                     w.coef_out, gr->bn_out_weight, gr->bn_out_bias, s))) return rc;
    // softmax / bn_similarity / logits backward: dqkv, the partial rows of bn_qkv's backward, of the tables and of the gates
    // (round 6, fin_inline.h: where the single sweep runs inside the networks, bn_qkv's backward finalisation rides on the relfix
    //  launch -- one workgroup per head, between the fix kernel that completes the partial rows and the 1x1 kernels that apply the
    //  coefficients -- except with bf16 storage without the widened copy, where finalisation and application are one launch)
    BfinSrc qb = no_bfin_src();
    if (w.sweep && (!g.bf16 || w.raw32) && inline_fin_ok(tr, g.groups, w.qb_rpg)) {
        qb.j = BfinJob{w.part_qb, w.qb_rpg, g.groups, 2 * g.C, tr, g.row_count, 1.f, st.qkv, p->bn_qkv.weight, w.coef_qkv,
                       gr->bn_qkv_weight, gr->bn_qkv_bias};
        qb.on = 1;
    }
    if ((rc = attention_core_bwd(g, d, p, w, qkv_raw, stacked, sv->lse, dy, st, gates, gr->gates != nullptr,
                                 gr->bn_sim_weight, gr->bn_sim_bias, s, ob.on ? &ob : nullptr, sim_inl, qb.on ? &qb : nullptr, ymask))) return rc;
    // bn_qkv backward, qkv_transform backward
    const float *bq_raw = qkv_raw, *bq_coef = w.coef_qkv;
    static const bool bf16_fused = true;
    if (g.bf16 && w.raw32) {
        // bf16 storage, round 5: the sweep has left qkv_raw widened to float32 in the workspace (one extra store per element of a
        // VALU-bound kernel): bn_qkv's backward is then applied on load by the 1x1 dgrad / wgrad exactly as with fp32 storage --
        // no extra launch, no extra pass over dqkv (gatedaxialunet bs 8: bf16 was 4-6 % slower than fp32 with the separate pass)
        if (!qb.on && (rc = bn_bwd_finalize(w.part_qb, w.qb_rpg, g.groups, 2 * g.C, g.row_count, 1.f, st.qkv, p->bn_qkv.weight, tr,
                                         w.coef_qkv, gr->bn_qkv_weight, gr->bn_qkv_bias, s))) return rc;
        bq_raw = w.raw32;
    } else if (g.bf16 && bf16_fused && (long)g.N * 2 * g.C <= 65535) {
        // bf16 storage: the finalisation and the bn_qkv backward materialised in fp32 in ONE launch, then the plain 1x1 dgrad / wgrad
        if ((rc = bn_bwd_fin_apply_bf16(w.part_qb, w.qb_rpg, g.groups, 2 * g.C, g.row_count, 1.f, st.qkv, p->bn_qkv.weight, tr,
                                        w.coef_qkv, gr->bn_qkv_weight, gr->bn_qkv_bias, w.dqkv, qkv_raw, g.N, g.HW, s))) return rc;
        bq_raw = nullptr;
        bq_coef = nullptr;
    } else {
        if (!qb.on && (rc = bn_bwd_finalize(w.part_qb, w.qb_rpg, g.groups, 2 * g.C, g.row_count, 1.f, st.qkv, p->bn_qkv.weight, tr,
                                            w.coef_qkv, gr->bn_qkv_weight, gr->bn_qkv_bias, s))) return rc;
        if (g.bf16) {   // (MEDT_BF16_FIN_APPLY=0: round 4's two launches)
            if ((rc = bn_bwd_apply_raw_bf16(w.dqkv, qkv_raw, w.coef_qkv, g.N, 2 * g.C, g.HW, g.groups, s))) return rc;
            bq_raw = nullptr;
            bq_coef = nullptr;
        }
    }
    if ((rc = conv1x1_bwd_data(w.dqkv, bq_raw, bq_coef, p->w_qkv, dx, g.N, g.C, 2 * g.C, g.HW, g.groups, s)))
        return rc;
    Queue* q = queue_for(s);          // parameter gradients: recorded for the grouped flush when a queue is bound
    if ((rc = conv2d_bwd_weight(w.dqkv, bq_raw, bq_coef, x, gr->w_qkv, w.dw_scratch, g.N, g.C, g.H, g.W, 2 * g.C, 1,
                                1, 0, g.groups, s, q))) return rc;
    if (g.pos) {
        // partial rows -> gradients (single sweep: one row per workgroup + one correction row per (group, head))
        const int rel_rows = w.sweep ? (int)w.sweep_blocks + g.groups * g.G : (int)w.nblocks;
        const float* gate_src = w.sweep ? w.gate_rows : w.gate_part;
        const int gate_rows = w.sweep ? g.groups * g.G : (int)w.nblocks;
        if (q) q->reduce.push_back(RJob{w.rel_part, gr->relative, rel_rows, 2 * g.gp * TL});
        else if ((rc = reduce_rows(w.rel_part, rel_rows, 2 * g.gp * TL, gr->relative, s))) return rc;
        if (gr->gates && d->gate_mode == 2) {
            if ((rc = gate_seq_reduce(w.gate_part, gr->gates, g.groups * g.spg, g.G, s))) return rc;
        } else if (gr->gates) {
            const bool sig = d->gate_mode == 1 && p->f_qr;
            if (q && !sig) q->reduce.push_back(RJob{gate_src, gr->gates, gate_rows, 4});
            else {
                if ((rc = reduce_rows(gate_src, gate_rows, 4, sig ? w.gate_tmp : gr->gates, s))) return rc;
                if (sig && (rc = gate_sigmoid_bwd(w.gate_tmp, w.gate_eff, gr->gates, s))) return rc;
            }
        }
    }
    return MEDT_OK;
}


// ------------------------------------------------------------------------- //
// convolution block
// ------------------------------------------------------------------------- //
}  // extern "C"  (helpers below are C++)

namespace medt {
struct ConvGeom { int Ho, Wo, HoWo, ppg, ppg_bwd, splits; size_t out_elems; };   // ppg: forward kernel's partial slots per BN group
static int conv_geom(const medt_conv_desc* d, ConvGeom* g) {
    if (!d || d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->H <= 0 || d->W <= 0 || d->stride < 1 || d->pad < 0) {
        set_error("conv: bad descriptor"); return MEDT_EINVAL;
    }
    if (d->K != 1 && d->K != 3 && d->K != 7) { set_error("conv: kernel size %d unsupported (1, 3, 7)", d->K); return MEDT_EUNSUPPORTED; }
    if (d->has_bn && (d->bn_groups < 1 || d->N % d->bn_groups)) { set_error("conv: bad bn_groups"); return MEDT_EINVAL; }
    if (d->has_bn && d->has_bias) { set_error("conv: bias followed by BatchNorm is not on the reference's path"); return MEDT_EUNSUPPORTED; }
    if (!d->has_bn && d->has_res) { set_error("conv: residual without BatchNorm is not on the reference's path"); return MEDT_EUNSUPPORTED; }
    g->Ho = (d->H + 2 * d->pad - d->K) / d->stride + 1;
    g->Wo = (d->W + 2 * d->pad - d->K) / d->stride + 1;
    if (g->Ho <= 0 || g->Wo <= 0) { set_error("conv: empty output"); return MEDT_EINVAL; }
    // (tests/test_abi_fuzz.py: H = 2^30 used to pass and wrap the workspace size)
    if ((double)d->N * d->Cin * d->H * d->W >= 2147483648.0 || (double)d->N * d->Cout * g->Ho * g->Wo >= 2147483648.0 ||
        (double)d->Cout * d->Cin * d->K * d->K >= 2147483648.0) {
        set_error("conv: tensors of 2^31 elements and more are unsupported"); return MEDT_EUNSUPPORTED;
    }
    g->HoWo = g->Ho * g->Wo;
    g->ppg = conv_stem7_ok(d->Cin, d->H, d->W, d->Cout, d->K, d->stride, d->pad)
                 ? conv_stem7_parts_per_group(d->N, d->has_bn ? d->bn_groups : 1, g->HoWo)
                 : conv_parts_per_group(d->N, d->has_bn ? d->bn_groups : 1, g->HoWo, d->Cin, d->Cout, d->K, d->stride, d->H, d->W, d->pad);
    g->ppg_bwd = conv2d_parts_per_group(d->N, d->has_bn ? d->bn_groups : 1, g->HoWo);       // bn_act_bwd_stats: 256 positions / part
    g->splits = conv2d_bwd_weight_splits(d->N, d->Cin, d->Cout, d->K, g->Ho, g->Wo);
    g->out_elems = (size_t)d->N * d->Cout * g->HoWo;
    return MEDT_OK;
}
struct ConvWs {
    float *partials, *coef, *bias_scratch, *gbuf, *dz, *dw_scratch, *wt, *ksplit, *ksplit_fwd, *ksplit_bwd;
    ConvWs(Carver& c, const medt_conv_desc* d, const ConvGeom& g) {
        wt = c.take<float>((size_t)d->Cout * d->Cin * d->K * d->K);
        const size_t kf = conv2d_fwd_scratch_floats(d->N, d->has_bn ? d->bn_groups : 1, d->Cin, d->H, d->W, d->Cout, d->K,
                                                    d->stride, d->pad);
        const size_t kb = conv2d_bwd_data_scratch_floats(d->N, d->Cin, d->H, d->W, d->Cout, d->K, d->stride, d->pad);
        ksplit = c.take<float>(kf > kb ? kf : kb);
        if (!(kf > 0 || kb > 0)) ksplit = nullptr;
        ksplit_fwd = kf > 0 ? ksplit : nullptr;
        ksplit_bwd = kb > 0 ? ksplit : nullptr;
        // (forward: [..][Cout][2] doubles; backward: [..][Cout][2] floats)
        partials = c.take<float>(d->has_bn ? (size_t)d->bn_groups * (2 * g.ppg > g.ppg_bwd ? 2 * g.ppg : g.ppg_bwd) * d->Cout * 2 : 0);
        coef = c.take<float>(d->has_bn ? (size_t)d->bn_groups * d->Cout * 3 : 0);
        bias_scratch = c.take<float>((size_t)16 * d->Cout);
        gbuf = c.take<float>(g.out_elems);
        dz = c.take<float>(d->has_bn ? g.out_elems : 0);
        dw_scratch = c.take<float>((size_t)g.splits * d->Cout * d->Cin * d->K * d->K);
    }
};
}  // namespace medt

extern "C" {

// Round 5: a TRAINING-mode forward leaves the flipped / transposed weights the MFMA backward-data kernel reads behind the
// layer's saved BatchNorm statistics (`stats` is the one buffer both passes get): the flip leaves the backward chain (four
// launches of MedT's local branch) for the forward pass's grouped flush.  Both sides derive the layout from the descriptor.
static bool conv_preflip(const medt_conv_desc* d) {
    static const bool on = true;
    // (3x3 only: the layers that take the MFMA backward-data in practice; the 1x1 blocks that adopt a one-launch block kernel's outputs
    //  -- medt_amd/block.py -- never call the forward entry and bring their own, smaller statistics block)
    return on && d->training && d->K == 3 && conv2d_bwd_data_flips(d->N, d->Cin, d->H, d->W, d->Cout, d->K, d->stride, d->pad);
}
static size_t conv_bn_stats_floats(const medt_conv_desc* d) { return d->has_bn ? (size_t)4 * d->bn_groups * d->Cout : 0; }
static float* conv_preflip_ptr(const medt_conv_desc* d, float* stats) { return stats + align_up(conv_bn_stats_floats(d), 64); }

size_t medt_conv_stats_floats(const medt_conv_desc* d) {
    if (!d) return 0;
    const size_t bn = (d->has_bn && d->bn_groups > 0 && d->Cout > 0) ? (size_t)4 * d->bn_groups * d->Cout : 0;
    ConvGeom g;
    if (conv_geom(d, &g)) return bn;           // (invalid descriptors: the entry points refuse them)
    return conv_preflip(d) ? align_up(bn, 64) + (size_t)d->Cout * d->Cin * d->K * d->K : bn;
}

size_t medt_conv_workspace_bytes(const medt_conv_desc* d) {
    ConvGeom g;
    if (conv_geom(d, &g)) return 0;
    Carver c(nullptr, 0);
    ConvWs w(c, d, g);
    return align_up(c.off, 256) + 256;
}

int medt_conv_block_fwd(const medt_conv_desc* d, const float* x, const float* w, const float* bias,
                        const medt_bn_ptrs* bn, const float* res, float* z, float* y, float* stats, void* ws,
                        size_t ws_bytes, void* stream) {
    ConvGeom g;
    int rc = conv_geom(d, &g);
    if (rc) return rc;
    if (!x || !w || !y || (d->has_bias && !bias) || (d->has_bn && (!bn || !z || !stats)) || (d->has_res && !res)) {
        set_error("conv fwd: null pointer"); return MEDT_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    Carver c(ws, ws_bytes);
    ConvWs cw(c, d, g);
    if (!ws || !c.ok()) { set_error("conv workspace too small: need %zu", c.off); return MEDT_EWORKSPACE; }
    if (stats && conv_preflip(d)) {            // (see medt_conv_stats_floats; nothing in the forward chain reads it)
        float* wt = conv_preflip_ptr(d, stats);
        if (Queue* q = queue_for(s)) q->flip.push_back(FlipJob{w, wt, d->Cout, d->Cin, d->K});
        else if ((rc = conv_flip_weights(w, wt, d->Cout, d->Cin, d->K, s))) return rc;
    }
    if (!d->has_bn)
        return conv2d_fwd(x, w, d->has_bias ? bias : nullptr, y, nullptr, cw.ksplit_fwd, d->N, d->Cin, d->H, d->W, d->Cout,
                          d->K, d->stride, d->pad, d->relu, 1, s);
    const int tr = d->training ? 1 : 0;
    if (!tr && (!bn->running_mean || !bn->running_var)) { set_error("conv fwd: eval mode needs running statistics"); return MEDT_EINVAL; }
    if (tr && (double)(d->N / d->bn_groups) * g.HoWo <= 1.0) {
        set_error("conv fwd: training-mode BatchNorm needs more than 1 value per channel"); return MEDT_EINVAL;
    }
    BnStats st(stats, d->bn_groups * d->Cout);
    if (conv_small_ok(*d)) {
        // small BN groups (local branch): conv + exact block-level statistics + normalise/residual/ReLU in one kernel,
        // then the saved statistics and the ordered running-stat updates
        if ((rc = conv_small_fwd(*d, x, w, *bn, res, z, y, cw.partials, s))) return rc;
        if (Queue* q = queue_for(s)) {
            q->fin.push_back(FinJob{make_fin(cw.partials, 1, d->Cout, (double)(d->N / d->bn_groups) * g.HoWo, *bn, st),
                                    d->bn_groups, tr, d->momentum, d->eps});
            return MEDT_OK;
        }
        return bn_finalize(cw.partials, 1, d->bn_groups, d->Cout, (double)(d->N / d->bn_groups) * g.HoWo, *bn, d->momentum,
                           d->eps, tr, st, s);
    }
    if ((rc = conv2d_fwd(x, w, nullptr, z, tr ? cw.partials : nullptr, cw.ksplit_fwd, d->N, d->Cin, d->H, d->W, d->Cout, d->K,
                         d->stride, d->pad, 0, d->bn_groups, s))) return rc;
    static const bool fused_fin = true;
    const double count = (double)(d->N / d->bn_groups) * g.HoWo;
    if (!fused_fin) {
        if ((rc = bn_finalize(cw.partials, g.ppg, d->bn_groups, d->Cout, count, *bn, d->momentum, d->eps, tr, st, s))) return rc;
        return bn_apply_act(z, st, d->has_res ? res : nullptr, y, d->N, d->Cout, g.HoWo, d->bn_groups, d->relu, s);
    }
    // statistics + normalise / residual / ReLU in one launch; the running statistics follow off the layer chain
    if ((rc = bn_fin_apply(z, cw.partials, g.ppg, count, *bn, d->eps, tr, st, d->has_res ? res : nullptr, y, d->N, d->Cout,
                           g.HoWo, d->bn_groups, d->relu, s))) return rc;
    if (!tr) return MEDT_OK;
    if (Queue* q = queue_for(s)) {
        q->fin.push_back(FinJob{make_fin(cw.partials, g.ppg, d->Cout, count, *bn, st), d->bn_groups, 2, d->momentum, d->eps});
        return MEDT_OK;
    }
    return bn_finalize(cw.partials, g.ppg, d->bn_groups, d->Cout, count, *bn, d->momentum, d->eps, 2, st, s);
}

int medt_conv_block_bwd(const medt_conv_desc* d, const float* x, const float* w, const medt_bn_ptrs* bn, const float* z,
                        const float* y, const float* stats, const float* dy, float* dx, float* dw, float* dbias,
                        float* dbn_weight, float* dbn_bias, float* dres, const float* dx_add, void* ws, size_t ws_bytes,
                        void* stream) {
    ConvGeom g;
    int rc = conv_geom(d, &g);
    if (rc) return rc;
    if (!x || !w || !dy || !dw || (d->relu && !y) || (d->has_bn && (!bn || !z || !stats || !dbn_weight || !dbn_bias)) ||
        (d->has_bias && !dbias)) { set_error("conv bwd: null pointer"); return MEDT_EINVAL; }
    Carver c(ws, ws_bytes);
    ConvWs cw(c, d, g);
    if (!ws || !c.ok()) { set_error("conv workspace too small: need %zu", c.off); return MEDT_EWORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    const float* grad_out;              // gradient wrt the convolution output
    bool dx_done = false;               // the BatchNorm-backward kernel also produced dx
    if (d->has_bn) {
        BnStats st(const_cast<float*>(stats), d->bn_groups * d->Cout);
        float* gb = (d->has_res && dres) ? dres : cw.gbuf;        // d(res) == the ReLU-masked incoming gradient
        const bool small = conv_small_ok(*d);
        // (the one-workgroup kernel moves float4s: 16-byte aligned tensors only)
        const bool aligned = ((((uintptr_t)dy | (uintptr_t)y | (uintptr_t)z | (uintptr_t)dres | (uintptr_t)cw.dz) & 15) == 0);
        if (small || (aligned && bn_chan_threads(*d, g.HoWo))) {
            // one wave (small blocks) or one workgroup per (group, channel): mask, sums, coefficients and dz in one
            // kernel; the finalisation only produces the parameter gradients (one partial slot per group)
            if (small && dx && aligned && !d->lean && bn_dgrad1x1_small_ok(*d)) {
                // ... and the 1x1 backward-data behind it in the same launch (conv_small.hip, round 4)
                rc = bn_dgrad1x1_small(*d, dy, y, z, st, bn->weight, w, dx_add, (d->has_res && dres) ? dres : nullptr,
                                       cw.dz, cw.partials, dx, s);
                dx_done = true;
            } else if (small)
                rc = bn_act_bwd_small(*d, dy, y, z, st, bn->weight, (d->has_res && dres) ? dres : nullptr, cw.dz,
                                      cw.partials, g.HoWo, s);
            else
                rc = bn_act_bwd_chan(*d, dy, y, z, st, bn->weight, (d->has_res && dres) ? dres : nullptr, cw.dz,
                                     cw.partials, g.HoWo, s);
            if (rc) return rc;
            // (only the BatchNorm parameter gradients come out of this one: dz is already final)
            if (Queue* q = queue_for(s))
                q->bfin.push_back(BfinJob{cw.partials, 1, d->bn_groups, d->Cout, d->training ? 1 : 0,
                                          (double)(d->N / d->bn_groups) * g.HoWo, 1.f, st, bn->weight, cw.coef, dbn_weight,
                                          dbn_bias});
            else if ((rc = bn_bwd_finalize(cw.partials, 1, d->bn_groups, d->Cout, (double)(d->N / d->bn_groups) * g.HoWo, 1.f,
                                           st, bn->weight, d->training ? 1 : 0, cw.coef, dbn_weight, dbn_bias, s))) return rc;
        } else {
        if ((rc = bn_act_bwd_stats(dy, y, z, st, gb, cw.partials, d->N, d->Cout, g.HoWo, d->bn_groups, d->relu, s))) return rc;
        if ((rc = bn_bwd_finalize(cw.partials, g.ppg_bwd, d->bn_groups, d->Cout,
                                  (double)(d->N / d->bn_groups) * g.HoWo, 1.f, st, bn->weight, d->training ? 1 : 0, cw.coef,
                                  dbn_weight, dbn_bias, s))) return rc;
        if ((rc = bn_bwd_apply(gb, z, cw.coef, cw.dz, d->N, d->Cout, g.HoWo, d->bn_groups, s))) return rc;
        }
        grad_out = cw.dz;
    } else if (d->relu) {
        if ((rc = relu_mask(dy, y, cw.gbuf, g.out_elems, s))) return rc;
        grad_out = cw.gbuf;
    } else {
        grad_out = dy;
    }
    // (training mode: the forward pass left the flipped weights behind the saved statistics)
    const bool preflipped = stats && conv_preflip(d);
    if (dx && !dx_done && (rc = conv2d_bwd_data(grad_out, w, dx, preflipped ? conv_preflip_ptr(d, const_cast<float*>(stats)) : cw.wt, cw.ksplit_bwd,
                                                d->N, d->Cin, d->H, d->W, d->Cout, d->K, d->stride, d->pad, s, dx_add, preflipped))) return rc;
    Queue* q = queue_for(s);          // parameter gradients: recorded for the grouped flush when a queue is bound
    if (d->has_bias) {
        if (q) {
            const int splits = channel_sum_splits_for(d->N, g.HoWo);
            q->csum.push_back(CJob{grad_out, splits == 1 ? dbias : cw.bias_scratch, d->N, d->Cout, g.HoWo, splits});
            if (splits > 1) q->reduce.push_back(RJob{cw.bias_scratch, dbias, splits, d->Cout});
        } else if ((rc = channel_sum(grad_out, dbias, cw.bias_scratch, d->N, d->Cout, g.HoWo, s))) return rc;
    }
    return conv2d_bwd_weight(grad_out, nullptr, nullptr, x, dw, cw.dw_scratch, d->N, d->Cin, d->H, d->W, d->Cout, d->K, d->stride,
                             d->pad, 1, s, q);
}

size_t medt_wopos_block_workspace_bytes(const medt_block_desc* d) {
    if (!d || !wopos_block_ok(*d)) return 0;
    return align_up(wopos_block_part_doubles(*d) * sizeof(double), 256) + 256;
}

int medt_wopos_block_fwd(const medt_block_desc* d, const medt_block_params* p, const float* x, float* y,
                         const medt_block_saved* sv, void* ws, size_t ws_bytes, void* stream) {
    if (!d || !p || !x || !y || !sv) { set_error("block fwd: null argument"); return MEDT_EINVAL; }
    if (!wopos_block_ok(*d)) { set_error("block fwd: shape not supported by the fused kernel"); return MEDT_EUNSUPPORTED; }
    if (!p->w_down || !p->w_up || !p->height.w_qkv || !p->width.w_qkv || !sv->z1 || !sv->y1 || !sv->stats1 || !sv->y_h ||
        !sv->y_w || !sv->z2 || !sv->stats2 || !sv->height.qkv_raw || !sv->height.stacked || !sv->height.lse ||
        !sv->height.stats || !sv->width.qkv_raw || !sv->width.stacked || !sv->width.lse || !sv->width.stats) {
        set_error("block fwd: null pointer"); return MEDT_EINVAL;
    }
    const medt_bn_ptrs* bns[8] = {&p->bn1, &p->height.bn_qkv, &p->height.bn_similarity, &p->height.bn_output,
                                  &p->width.bn_qkv, &p->width.bn_similarity, &p->width.bn_output, &p->bn2};
    for (int b = 0; b < 8; ++b) {
        if (!bns[b]->weight || !bns[b]->bias) { set_error("block fwd: null BatchNorm parameter"); return MEDT_EINVAL; }
        if (!d->training && (!bns[b]->running_mean || !bns[b]->running_var)) {
            set_error("block fwd: eval mode needs running statistics"); return MEDT_EINVAL;
        }
    }
    Carver c(ws, ws_bytes);
    double* parts = c.take<double>(wopos_block_part_doubles(*d));
    if (!ws || !c.ok()) { set_error("block workspace too small: need %zu", c.off); return MEDT_EWORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    int rc = wopos_block_fwd(*d, *p, x, y, *sv, parts, s);
    if (rc) return rc;
    // saved statistics + ordered running-stat updates of the eight BatchNorms: recorded for the grouped flush when a queue
    // is bound (nothing in the forward chain reads them), else issued now
    const int tr = d->training ? 1 : 0, gs = d->bn_groups, W = d->width, G = d->G;
    const double rows = (double)(d->N / gs) * d->H * d->W, sims = rows * d->H;       // H == W: sequence length either axis
    AxialGeom gh;
    {
        medt_axial_desc ad{d->N, W, d->H, d->W, G, 0, 0, 1, d->training, gs, d->eps, d->momentum, 0, 0, 0};
        if ((rc = axial_geom(ad, &gh))) return rc;
    }
    LayerStats sh(sv->height.stats, gh), sw(sv->width.stats, gh);
    const int chs[8] = {W, 2 * W, G, W, 2 * W, G, W, d->C};
    const double cnt[8] = {rows, rows, sims, rows, rows, sims, rows, rows};
    BnStats outs[8] = {BnStats(sv->stats1, gs * W), sh.qkv, sh.sim, sh.out, sw.qkv, sw.sim, sw.out, BnStats(sv->stats2, gs * d->C)};
    const float* pp = reinterpret_cast<const float*>(parts);
    Queue* q = queue_for(s);
    for (int b = 0; b < 8; ++b) {
        if (q) q->fin.push_back(FinJob{make_fin(pp, 1, chs[b], cnt[b], *bns[b], outs[b]), gs, tr, d->momentum, d->eps});
        else if ((rc = bn_finalize(pp, 1, gs, chs[b], cnt[b], *bns[b], d->momentum, d->eps, tr, outs[b], s))) return rc;
        pp += (size_t)gs * chs[b] * 2 * 2;           // doubles
    }
    return MEDT_OK;
}

size_t medt_wopos_block_s2_workspace_bytes(const medt_block_desc* d) {
    if (!d || !wopos_block_s2_ok(*d)) return 0;
    return align_up(wopos_block_s2_part_doubles(*d) * sizeof(double), 256) + 256;
}

int medt_wopos_block_s2_fwd(const medt_block_desc* d, const medt_block_s2_params* p, const float* x, float* y,
                            const medt_block_s2_saved* sv, void* ws, size_t ws_bytes, void* stream) {
    if (!d || !p || !x || !y || !sv) { set_error("block s2 fwd: null argument"); return MEDT_EINVAL; }
    if (!wopos_block_s2_ok(*d)) { set_error("block s2 fwd: shape not supported by the fused kernel"); return MEDT_EUNSUPPORTED; }
    const medt_block_params& bp = p->blk;
    const medt_block_saved& b0 = sv->blk;
    if (!bp.w_down || !bp.w_up || !bp.height.w_qkv || !bp.width.w_qkv || !p->w_ds || !b0.z1 || !b0.y1 || !b0.stats1 || !b0.y_h ||
        !b0.y_w || !b0.z2 || !b0.stats2 || !b0.height.qkv_raw || !b0.height.stacked || !b0.height.lse || !b0.height.stats ||
        !b0.width.qkv_raw || !b0.width.stacked || !b0.width.lse || !b0.width.stats || !sv->zd || !sv->yd || !sv->statsd) {
        set_error("block s2 fwd: null pointer"); return MEDT_EINVAL;
    }
    const medt_bn_ptrs* bns[9] = {&bp.bn1, &bp.height.bn_qkv, &bp.height.bn_similarity, &bp.height.bn_output,
                                  &bp.width.bn_qkv, &bp.width.bn_similarity, &bp.width.bn_output, &bp.bn2, &p->bn_ds};
    for (int b = 0; b < 9; ++b) {
        if (!bns[b]->weight || !bns[b]->bias) { set_error("block s2 fwd: null BatchNorm parameter"); return MEDT_EINVAL; }
        if (!d->training && (!bns[b]->running_mean || !bns[b]->running_var)) {
            set_error("block s2 fwd: eval mode needs running statistics"); return MEDT_EINVAL;
        }
    }
    Carver c(ws, ws_bytes);
    double* parts = c.take<double>(wopos_block_s2_part_doubles(*d));
    if (!ws || !c.ok()) { set_error("block s2 workspace too small: need %zu", c.off); return MEDT_EWORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    int rc = wopos_block_s2_fwd(*d, *p, x, y, *sv, parts, s);
    if (rc) return rc;
    // saved statistics + ordered running-stat updates of the nine BatchNorms (recorded when a queue is bound, like medt_wopos_block_fwd)
    const int tr = d->training ? 1 : 0, gs = d->bn_groups, W = d->width, G = d->G, CO = 2 * d->width;
    const double rows = (double)(d->N / gs) * d->H * d->W, sims = rows * d->H, rows2 = rows / 4;
    AxialGeom gh;
    {
        medt_axial_desc ad{d->N, W, d->H, d->W, G, 0, 0, 1, d->training, gs, d->eps, d->momentum, 0, 0, 0};
        if ((rc = axial_geom(ad, &gh))) return rc;
    }
    LayerStats sh(b0.height.stats, gh), sw(b0.width.stats, gh);
    const int chs[9] = {W, 2 * W, G, W, 2 * W, G, W, CO, CO};
    const double cnt[9] = {rows, rows, sims, rows, rows, sims, rows, rows2, rows2};
    BnStats outs[9] = {BnStats(b0.stats1, gs * W), sh.qkv, sh.sim, sh.out, sw.qkv, sw.sim, sw.out, BnStats(b0.stats2, gs * CO),
                       BnStats(sv->statsd, gs * CO)};
    const float* pp = reinterpret_cast<const float*>(parts);
    Queue* q = queue_for(s);
    for (int b = 0; b < 9; ++b) {
        if (q) q->fin.push_back(FinJob{make_fin(pp, 1, chs[b], cnt[b], *bns[b], outs[b]), gs, tr, d->momentum, d->eps});
        else if ((rc = bn_finalize(pp, 1, gs, chs[b], cnt[b], *bns[b], d->momentum, d->eps, tr, outs[b], s))) return rc;
        pp += (size_t)gs * chs[b] * 2 * 2;           // doubles
    }
    return MEDT_OK;
}

size_t medt_wopos_block_bwd_workspace_bytes(const medt_block_desc* d) {
    if (!d || !wopos_block_bwd_ok(*d)) return 0;
    return wopos_block_bwd_ws_bytes(*d);
}

int medt_wopos_block_bwd(const medt_block_desc* d, const medt_block_params* p, const float* x, const float* y, const float* dy,
                         const medt_block_saved* sv, float* dx, const float* dx_add, const medt_block_grads* gr, void* ws,
                         size_t ws_bytes, void* stream) {
    if (!d || !p || !x || !y || !dy || !sv || !dx || !gr) { set_error("block bwd: null argument"); return MEDT_EINVAL; }
    if (!wopos_block_bwd_ok(*d)) { set_error("block bwd: shape not supported by the fused kernel, or MEDT_BLOCK_BWD != 1"); return MEDT_EUNSUPPORTED; }
    if (!p->w_down || !p->w_up || !p->height.w_qkv || !p->width.w_qkv || !sv->z1 || !sv->y1 || !sv->stats1 || !sv->y_h ||
        !sv->y_w || !sv->z2 || !sv->stats2 || !sv->height.qkv_raw || !sv->height.stacked || !sv->height.lse ||
        !sv->height.stats || !sv->width.qkv_raw || !sv->width.stacked || !sv->width.lse || !sv->width.stats) {
        set_error("block bwd: null pointer"); return MEDT_EINVAL;
    }
    const medt_bn_ptrs* bns[8] = {&p->bn1, &p->height.bn_qkv, &p->height.bn_similarity, &p->height.bn_output,
                                  &p->width.bn_qkv, &p->width.bn_similarity, &p->width.bn_output, &p->bn2};
    for (int b = 0; b < 8; ++b)
        if (!bns[b]->weight) { set_error("block bwd: null BatchNorm weight"); return MEDT_EINVAL; }
    const medt_axial_grads* ag[2] = {&gr->height, &gr->width};
    for (int l = 0; l < 2; ++l)
        if (!ag[l]->w_qkv || !ag[l]->bn_qkv_weight || !ag[l]->bn_qkv_bias || !ag[l]->bn_sim_weight || !ag[l]->bn_sim_bias ||
            !ag[l]->bn_out_weight || !ag[l]->bn_out_bias) { set_error("block bwd: null gradient pointer"); return MEDT_EINVAL; }
    if (!gr->w_down || !gr->bn1_weight || !gr->bn1_bias || !gr->w_up || !gr->bn2_weight || !gr->bn2_bias) {
        set_error("block bwd: null gradient pointer"); return MEDT_EINVAL;
    }
    return wopos_block_bwd(*d, *p, x, y, dy, dx_add, *sv, dx, *gr, ws, ws_bytes, (hipStream_t)stream);
}

int medt_gate_mlp_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* xn, float* h,
                      float* o, float* gates, int N, int C, int H, int W, int axis, void* stream) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !xn || !h || !o || !gates || N < 1 || C < 1 || C > 4096 || H < 1 || W < 1) {
        set_error("gate_mlp fwd: bad arguments"); return MEDT_EINVAL;
    }
    return gate_mlp_fwd(x, w1, b1, w2, b2, xn, h, o, gates, N, C, H, W, axis ? 1 : 0, (hipStream_t)stream);
}
int medt_gate_mlp_bwd(const float* dgates, const float* gates, const float* o, const float* h, const float* xn,
                      const float* w1, const float* w2, float* scratch, float* dw1, float* db1, float* dw2, float* db2,
                      float* dx, int N, int C, int H, int W, int axis, void* stream) {
    if (!dgates || !gates || !o || !h || !xn || !w1 || !w2 || !scratch || !dw1 || !db1 || !dw2 || !db2 || !dx) {
        set_error("gate_mlp bwd: null pointer"); return MEDT_EINVAL;
    }
    const size_t nseq = (size_t)N * (axis ? H : W);
    float *d_o = scratch, *dh = d_o + nseq * 4, *dxn = dh + nseq * C;        // scratch: nseq * (4 + 2C) floats
    return gate_mlp_bwd(dgates, gates, o, h, xn, w1, w2, d_o, dh, dxn, dw1, db1, dw2, db2, dx, N, C, H, W, axis ? 1 : 0,
                        (hipStream_t)stream);
}

int medt_up2x_relu_add_fwd(const float* x, const float* skip, float* y, int NC, int H, int W, void* stream) {
    if (!x || !y || NC <= 0 || H <= 0 || W <= 0) { set_error("up2x fwd: bad arguments"); return MEDT_EINVAL; }
    return up2x_relu_add_fwd(x, skip, y, NC, H, W, (hipStream_t)stream);
}
int medt_up2x_relu_bwd(const float* x, const float* dy, float* dx, int NC, int H, int W, void* stream) {
    if (!x || !dy || !dx || NC <= 0 || H <= 0 || W <= 0) { set_error("up2x bwd: bad arguments"); return MEDT_EINVAL; }
    return up2x_relu_bwd(x, dy, dx, NC, H, W, (hipStream_t)stream);
}
int medt_patch_gather(const float* x, float* xp, int N, int C, int S, int P, int G, void* stream) {
    if (!x || !xp || G * P > S) { set_error("patch_gather: bad arguments"); return MEDT_EINVAL; }
    return patch_gather(x, xp, N, C, S, P, G, (hipStream_t)stream);
}
int medt_logo_merge_fwd(const float* x, const float* yp, float* y, int N, int C, int S, int P, int G, void* stream) {
    if (!x || !yp || !y || G * P > S) { set_error("logo_merge: bad arguments"); return MEDT_EINVAL; }
    return logo_merge_fwd(x, yp, y, N, C, S, P, G, (hipStream_t)stream);
}
int medt_logo_merge_bwd(const float* dy, float* dx, float* dyp, int N, int C, int S, int P, int G, void* stream) {
    if (!dy || !dx || !dyp || G * P > S) { set_error("logo_merge bwd: bad arguments"); return MEDT_EINVAL; }
    return logo_merge_bwd(dy, dx, dyp, N, C, S, P, G, (hipStream_t)stream);
}
size_t medt_ce_partials(int N, int HW) { return (size_t)3 * ce_parts((size_t)N * HW); }
int medt_ce_fwd(const float* logits, const int64_t* target, float* partials, float* loss_out, int N, int K, int HW,
                int ignore_index, void* stream) {
    if (!logits || !target || !partials || !loss_out || K < 1) { set_error("ce fwd: bad arguments"); return MEDT_EINVAL; }
    return ce_fwd(logits, target, partials, loss_out, N, K, HW, ignore_index, (hipStream_t)stream);
}
int medt_ce_bwd(const float* logits, const int64_t* target, const float* loss_out, const float* dloss, float* dlogits,
                int N, int K, int HW, int ignore_index, void* stream) {
    if (!logits || !target || !loss_out || !dlogits) { set_error("ce bwd: bad arguments"); return MEDT_EINVAL; }
    return ce_bwd(logits, target, loss_out, dloss, dlogits, N, K, HW, ignore_index, (hipStream_t)stream);
}
int medt_adam_step(float* p, const float* g, float* m, float* v, float* state, size_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, float gscale, void* stream) {
    if (!p || !g || !m || !v || !state) { set_error("adam: null pointer"); return MEDT_EINVAL; }
    if (n == 0) return MEDT_OK;
    return adam_step(p, g, m, v, state, n, lr, beta1, beta2, eps, weight_decay, gscale, (hipStream_t)stream);
}
int medt_relu_mask(const float* a, const float* y, float* out, size_t n, void* stream) {
    if (!a || !y || !out) { set_error("relu_mask: null pointer"); return MEDT_EINVAL; }
    return relu_mask(a, y, out, n, (hipStream_t)stream);
}
int medt_seg_counts(const float* logits, const int64_t* target, int32_t* counts, int N, int K, int HW, float threshold,
                    void* stream) {
    if (!logits || !target || !counts || K < 2 || N < 1 || HW < 1) { set_error("seg_counts: bad arguments"); return MEDT_EINVAL; }
    return seg_counts(logits, target, counts, N, K, HW, threshold, (hipStream_t)stream);
}

}  // extern "C"
