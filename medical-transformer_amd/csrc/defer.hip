// defer.hip -- the queue registry and flush of defer.h, and its C entry points (include/medt_abi.h: medt_queue_*).
#include "defer.h"
#include "fin_inline.h"
#include <stdlib.h>
#include <mutex>
#include <stdio.h>

namespace medt {

static std::mutex g_mu;
static std::vector<std::pair<hipStream_t, Queue*>> g_bound;

Queue* queue_for(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& e : g_bound)
        if (e.first == s) return e.second;
    return nullptr;
}

// fin_inline.h: consumer-side BatchNorm finalisation (MEDT_INLINE_FIN=0: the finalisation launches everywhere)
bool inline_fin_enabled() {
    static const bool on = [] { const char* e = getenv("MEDT_INLINE_FIN"); return !(e && e[0] == '0'); }();
    return on;
}

}  // namespace medt

using namespace medt;

extern "C" {

void* medt_queue_create(void) { return new Queue(); }

int medt_queue_destroy(void* q) {
    if (!q) return MEDT_OK;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (size_t i = g_bound.size(); i-- > 0;)
            if (g_bound[i].second == (Queue*)q) g_bound.erase(g_bound.begin() + i);
    }
    delete (Queue*)q;
    return MEDT_OK;
}

int medt_queue_bind(void* q, void* stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = 0; i < g_bound.size(); ++i)
        if (g_bound[i].first == (hipStream_t)stream) {
            if (q) g_bound[i].second = (Queue*)q;
            else g_bound.erase(g_bound.begin() + i);
            return MEDT_OK;
        }
    if (q) g_bound.emplace_back((hipStream_t)stream, (Queue*)q);
    return MEDT_OK;
}

size_t medt_queue_pending(const void* q) { return q ? ((const Queue*)q)->pending() : 0; }

// aux (optional): a second stream of the caller's that is idle by now -- the dedicated MFMA weight-gradient kernels (the LDS-patch
// problems of the 16-wide maps, the few-tile problems) are issued THERE, side by side with the grouped launches on `stream`
// (event fork / join: capturable), and the slab reductions follow the join.  Round 6: the local branch's flush tail was 152 us of
// serial launches behind the step's longest chain while the other stream had been idle for ~90 us.
static int queue_flush(Queue& q, hipStream_t s, hipStream_t aux) {
    int rc = MEDT_OK;
    if (abl_skip("flush")) rc = -1000;                      // (timing experiments: drop everything recorded)
    std::vector<const MJob*> r16, rest;
    for (const MJob& m : q.mwgrad) {
        if (conv_wgrad_rows16_ok(m.Cin, m.H, m.W, m.Ho, m.Wo, m.K, m.stride, m.pad, m.QS)) r16.push_back(&m);
        else if (m.K == 1 || m.K == 3) rest.push_back(&m);
    }
    const bool fork = aux && aux != s && !rc && !(r16.empty() && rest.empty());
    static hipEvent_t ev[2] = {nullptr, nullptr};
    if (fork && !ev[0]) {
        if (hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) != hipSuccess) { set_error("queue flush: hipEventCreate failed"); return MEDT_ELAUNCH; }
    }
    hipStream_t sm = s;                                     // stream of the dedicated MFMA weight gradients
    if (fork) {
        if (hipEventRecord(ev[0], s) != hipSuccess || hipStreamWaitEvent(aux, ev[0], 0) != hipSuccess) {
            set_error("queue flush: event fork failed"); return MEDT_ELAUNCH;
        }
        sm = aux;
        // the LDS-patch problems of the 16-wide maps side by side in one launch (they fill each other's load gaps); so the others
        if (!rc && !r16.empty()) rc = conv_wgrad_rows16_grouped(r16.data(), (int)r16.size(), sm);
        if (!rc && !rest.empty()) rc = conv_wgrad_mfma_batch(rest.data(), (int)rest.size(), sm);
        if (hipEventRecord(ev[1], aux) != hipSuccess) { set_error("queue flush: event join failed"); return MEDT_ELAUNCH; }
    }
    // order: statistics bookkeeping; first-stage sums and weight gradients; then the reductions of their partial slabs
#ifdef MEDT_AB_FLIP_EACH                // (A/B build: one launch per recorded flip)
    for (const FlipJob& f : q.flip)
        if (!rc) rc = conv_flip_weights(f.w, f.wt, f.Cout, f.Cin, f.K, s);
#else
    if (!rc && !q.flip.empty()) rc = conv_flip_weights_grouped(q.flip.data(), (int)q.flip.size(), s);
#endif
    if (!rc && !q.fin.empty()) rc = bn_finalize_grouped(q.fin.data(), (int)q.fin.size(), s);
    if (!rc && !q.bfin.empty()) rc = bn_bwd_finalize_grouped(q.bfin.data(), (int)q.bfin.size(), s);
    if (!rc && !q.sfin.empty()) rc = wopos_small_bwd_finalize_grouped(q.sfin.data(), (int)q.sfin.size(), s);
    if (!rc && !q.csum.empty()) rc = channel_sum_grouped(q.csum.data(), (int)q.csum.size(), s);
    if (!rc && !q.wgrad.empty()) rc = conv_wgrad_grouped(q.wgrad.data(), (int)q.wgrad.size(), s);
    if (fork) {
        if (hipStreamWaitEvent(s, ev[1], 0) != hipSuccess) { set_error("queue flush: event join failed"); return MEDT_ELAUNCH; }
    } else {
        if (!rc && !r16.empty()) rc = conv_wgrad_rows16_grouped(r16.data(), (int)r16.size(), s);
        if (!rc && !rest.empty()) rc = conv_wgrad_mfma_batch(rest.data(), (int)rest.size(), s);
    }
    if (!rc && !q.reduce.empty()) rc = reduce_rows_grouped(q.reduce.data(), (int)q.reduce.size(), s);
    q.flip.clear(); q.fin.clear(); q.bfin.clear(); q.sfin.clear(); q.csum.clear(); q.wgrad.clear(); q.mwgrad.clear(); q.reduce.clear();
    return rc == -1000 ? MEDT_OK : rc;
}

int medt_queue_flush(void* qv, void* stream) {
    if (!qv) { set_error("queue flush: null queue"); return MEDT_EINVAL; }
    return queue_flush(*(Queue*)qv, (hipStream_t)stream, nullptr);
}

int medt_queue_flush2(void* qv, void* stream, void* aux_stream) {
    if (!qv) { set_error("queue flush: null queue"); return MEDT_EINVAL; }
    return queue_flush(*(Queue*)qv, (hipStream_t)stream, (hipStream_t)aux_stream);
}

int medt_queue_discard(void* qv) {
    if (!qv) { set_error("queue discard: null queue"); return MEDT_EINVAL; }
    Queue& q = *(Queue*)qv;
    q.flip.clear(); q.fin.clear(); q.bfin.clear(); q.sfin.clear(); q.csum.clear(); q.wgrad.clear(); q.mwgrad.clear(); q.reduce.clear();
    return MEDT_OK;
}

}  // extern "C"
