// The loop of SimRank.fit behind the C ABI (create_plan / step / download): a PLAN owns
// everything one single-GPU fit needs — the graph in the solver's node order, the three panel-blocked
// matrices of an update, evidence counts, prior, the striped convergence counters — and runs
//
//     for k in range(iterations):                 SimRank.py:129-140 (:351-362, :443-454 with evidence / prior)
//         if converged(old, new): break
//         new = E * C * W.S.W^T (+ lbd A); diag <- 1
//
// as two launches per update (leg 1: fused_trans_kernel, leg 2: upper-triangle gather with the fused
// epilogue and count), with update k + 1 queued BEFORE the count of update k is read (graphs below 16384 nodes,
// where an update is short; common.h kSpeculateBelow): the host never
// leaves the device idle to learn whether it may go on, and when the count says "converged" the
// speculative update is simply not adopted (it wrote the buffer of the iterate before last).
// Python's driver.Solver does the same choreography for every world size; this is the single-rank case
// for callers that bind the library directly (examples/reference_hip_stub.py).
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "common.h"

struct simrank_plan {
    int64_t n = 0, rows_pad = 0;
    size_t mat_bytes = 0;
    simrank_graph* g = nullptr;
    float* S[2] = {nullptr, nullptr};         // ping-pong iterates, panel-blocked
    float* Tt = nullptr;                      // (W.S)^T
    uint8_t* ev = nullptr;                    // evidence counts (SimRank++), panel-blocked u8
    float* prior = nullptr;                   // panel-blocked, solver order
    int32_t* inv = nullptr;                   // device: position of caller's node i in the solver's order
    int32_t* ord_dev = nullptr;               // device: caller's node at position r (ids of the columns, top-k)
    std::vector<int32_t> ord;                 // host copy
    unsigned long long* counters = nullptr;   // device, SIMRANK_CHANGED_SLOTS
    unsigned long long* host_counters[2] = {nullptr, nullptr};   // pinned; update u lands in slot u & 1
    hipEvent_t counted[2] = {nullptr, nullptr};
    hipStream_t stream = nullptr;
    float coef = 0.8f, lbd = 0.f;
    int32_t restrict_support = 0;
    int32_t half = 0;                         // 1: S and Tt are fp16 on 64-column panels (half.hip), value x kHalfScale
    int32_t asym = 0;                         // 1: the prior is not symmetric, so the iterates are not: leg 2 = leg 1's launch again
                                              //    (its product stored transposed), then the epilogue as a pass of its own
    int cur = 0;                              // S[cur] is the current iterate
    int32_t updates = 0;                      // updates applied since the last reset
    int32_t identity_leg1 = 1;                // the first update's leg 1 without gathers (S_0 = I; SIMRANK_IDENTITY_LEG1=0: off)
    int32_t at_identity = 0;                  // S[cur] is the identity (simrank_plan_reset), no update queued since
    // simrank_plan_set_timing: three events per update (before leg 1, between the legs, after leg 2) on the plan's stream
    std::vector<hipEvent_t> ev_pool;          // created ahead of the timed region
    std::vector<hipEvent_t> ev_used;          // 3 per timed update, in order
    bool timing = false;
};

namespace simrank {

constexpr float kHalfScale = 16384.0f;        // what fp16-held matrices are scaled by (include/simrank_hip.h, SCALE)

static int stamp(simrank_plan* p) {
    if (!p->timing || p->ev_pool.empty()) return SIMRANK_OK;
    hipEvent_t e = p->ev_pool.back();
    p->ev_pool.pop_back();
    p->ev_used.push_back(e);
    SR_HIP(hipEventRecord(e, p->stream));
    return SIMRANK_OK;
}

static int leg_pair(simrank_plan* p, double eps, int32_t exact_count, int slot) {
    const int nx = p->cur ^ 1;
    const bool timed = p->timing && p->ev_pool.size() >= 3;
    if (timed) { const int rs = stamp(p); if (rs) return rs; }
    // (the first update of a fit multiplies by the identity: W^T is written directly — the same bits without a gather)
    const bool from_identity = p->at_identity && p->identity_leg1;
    p->at_identity = 0;
    int rc = from_identity ? (p->half ? identity_leg1_blocked_h16(p->g, reinterpret_cast<uint16_t*>(p->Tt), p->rows_pad, kHalfScale, p->stream)
                                      : identity_leg1_blocked(p->g, p->Tt, p->rows_pad, p->stream))
             : p->half ? simrank_spmm_blocked_h16(p->g, p->S[p->cur], p->rows_pad, p->n, p->Tt, p->rows_pad, 1, nullptr,
                                                  0, kHalfScale, p->stream)
                     : simrank_spmm_blocked(p->g, p->S[p->cur], p->rows_pad, p->n, p->Tt, p->rows_pad, 1, nullptr,
                                            p->stream);
    if (rc) return rc;
    if (timed) { const int rs = stamp(p); if (rs) return rs; }
    simrank_epilogue ep{};
    ep.coef = p->coef;
    ep.lbd = p->lbd;
    ep.evidence = p->ev;
    ep.ld_evidence = 32;
    ep.apriori = p->prior;
    ep.ld_apriori = 32;
    ep.previous = p->S[p->cur];
    ep.ld_previous = 32;
    ep.eps = eps;
    ep.n_changed = p->counters;
    ep.diag_col0 = 0;
    ep.set_diag = 1;
    ep.symmetric = 1;
    ep.restrict_support = p->restrict_support;
    ep.count_any = exact_count ? 0 : 1;
    if (p->asym) {
        // S is not symmetric (SimRank.py:453 with a prior that is not): W . Tt is the TRANSPOSE of W S W^T, so leg 2 is leg 1's
        // launch on Tt — X -> (W X)^T, the one-launch kernel again — and the epilogue (coefficient, evidence, prior, diagonal,
        // exact count) runs over the stored product in place
        ep.symmetric = 0;
        ep.restrict_support = 0;
        rc = simrank_spmm_blocked(p->g, p->Tt, p->rows_pad, p->n, p->S[nx], p->rows_pad, 1, nullptr, p->stream);
        if (!rc) rc = simrank_epilogue_apply_blocked(p->S[nx], p->S[nx], p->n, p->n, p->rows_pad, &ep, p->stream);
    } else
    rc = p->half ? simrank_spmm_blocked_h16(p->g, p->Tt, p->rows_pad, p->n, p->S[nx], p->rows_pad, 0, &ep, p->rows_pad,
                                            kHalfScale, p->stream)
                 : simrank_spmm_blocked(p->g, p->Tt, p->rows_pad, p->n, p->S[nx], p->rows_pad, 0, &ep, p->stream);
    if (rc) return rc;
    if (timed) { const int rs = stamp(p); if (rs) return rs; }
    SR_HIP(hipMemcpyAsync(p->host_counters[slot], p->counters, sizeof(unsigned long long) * SIMRANK_CHANGED_SLOTS,
                          hipMemcpyDeviceToHost, p->stream));
    SR_HIP(hipEventRecord(p->counted[slot], p->stream));
    return SIMRANK_OK;
}

// the count of the update that used `slot`: waits for that update only, not for what was queued behind it
static int read_count(simrank_plan* p, int slot, unsigned long long* sum) {
    SR_HIP(hipEventSynchronize(p->counted[slot]));
    unsigned long long t = 0;
    for (int i = 0; i < SIMRANK_CHANGED_SLOTS; ++i) t += p->host_counters[slot][i];
    *sum = t;
    return SIMRANK_OK;
}

}  // namespace simrank

using namespace simrank;

extern "C" {

int simrank_plan_destroy(simrank_plan* p) {
    if (!p) return SIMRANK_OK;
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    (void)pool_free(p->S[0]); (void)pool_free(p->S[1]); (void)pool_free(p->Tt); (void)pool_free(p->ev);
    (void)pool_free(p->prior); (void)pool_free(p->inv); (void)pool_free(p->ord_dev); (void)pool_free(p->counters);
    for (int i = 0; i < 2; ++i) {
        if (p->host_counters[i]) (void)hipHostFree(p->host_counters[i]);
        if (p->counted[i]) (void)hipEventDestroy(p->counted[i]);
    }
    for (hipEvent_t e : p->ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->ev_used) (void)hipEventDestroy(e);
    simrank_graph_destroy(p->g);
    delete p;
    return SIMRANK_OK;
}

int simrank_plan_create(int64_t n, int64_t nnz, const int32_t* rowptr, const int32_t* col, const float* rowscale,
                        const simrank_plan_options* opt, void* stream, simrank_plan** out) {
    SR_REQUIRE(out, "out is NULL");
    *out = nullptr;
    const bool timed = std::getenv("SIMRANK_TIME_BUILD") != nullptr;     // diagnostic: phase durations on stderr
    const auto t_start = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (timed)
            std::fprintf(stderr, "simrank_plan_create: %6.1f ms  %s\n",
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(), what);
    };
    PlanPrep pp;
    int rc = plan_prepare(n, nnz, rowptr, col, rowscale, opt, &pp);     // validation, node order, renamed pattern (planprep.hip)
    if (rc) return rc;
    lap("validated, ordered, renamed");
    const std::vector<int32_t>& ord = pp.ord;
    const std::vector<int32_t>& inv = pp.inv;
    simrank_plan* p = new simrank_plan;
    p->n = n;
    p->stream = as_stream(stream);
    p->coef = opt->coef;
    p->lbd = opt->lbd;
    p->rows_pad = (n + 7) / 8 * 8 + 8;
    p->half = opt->storage_fp16 ? 1 : 0;
    p->asym = pp.asym ? 1 : 0;
    if (const char* e = std::getenv("SIMRANK_IDENTITY_LEG1")) p->identity_leg1 = (*e == '0') ? 0 : 1;
    const int64_t panels = (n + 31) / 32;
    // (fp16: 64-column panels of 2-byte elements — a row segment is 128 bytes either way)
    p->mat_bytes = p->half ? size_t((n + 63) / 64) * size_t(p->rows_pad) * 128
                           : size_t(panels) * size_t(p->rows_pad) * 32 * sizeof(float);
    const size_t prior_bytes = size_t(panels) * size_t(p->rows_pad) * 32 * sizeof(float);
    auto fail = [&](int code) { simrank_plan_destroy(p); return code; };
    const size_t ev_bytes = size_t(panels) * size_t(p->rows_pad) * 32;
    {
        Tuning t = tuning_snapshot();
        if (p->half) {
            t.fuse_unit = int64_t(1) << 20;      // (half.hip runs whole blocks: no units whose sums meet in memory)
            // one fp16 MFMA term instead of three bf16 ones, but an operand segment serves 64 columns, so the gathers got
            // cheaper still: the break-even moves up by one (4 is 3 % faster than 3, 2 is 12 % slower), and groups of four
            // blocks without a set beat three (7 % at config 5) — while the knobs are at their defaults (as driver.Side did)
            // (round 6, fuse_min = 0 — quads that pay: the same step up, 256 entries per quad instead of 192)
            if (t.fuse_min == 3) {
                t.fuse_min = 4;
                if (t.fuse_group == 3) t.fuse_group = 4;
            } else if (t.fuse_min == 0 && t.fuse_pays < 0) {
                t.fuse_pays = 256;
                if (t.fuse_group == 3) t.fuse_group = 4;
            }
        }
        // leg 1 = the one-launch kernel (fp16-held: always; f32: whenever spmm.hip's conditions hold), so the dense-block plan
        // could only serve the upper-triangle leg 2: built only if that leg would take it
        // (and only where that leg IS the upper-triangle one — knob on, 64 nodes or more — or stores transposed: asymmetric priors)
        if (t.fuse == 1 && ((t.triangle && n >= 64) || p->asym) &&
            (p->half || ((opt->dense_terms == 0 || opt->dense_terms == 3) && n <= t.fuse_max_rows &&
                         (p->rows_pad + 1) * 128 < (int64_t(1) << 31))))
            t.dense_lazy = 1;
        // The evidence counts (SimRank.py:311-320: common in-neighbours of the pattern; 1 - 2^-count in the epilogue) read the
        // CSR / CSC arrays only: they are queued as soon as those are on the device and run while the host threads still
        // build the tile, dense-block and one-launch plans (14 ms beside 30 at config 5).
        std::function<int(simrank_graph*)> counts = [&](simrank_graph* g) -> int {
            if (!opt->evidence) return SIMRANK_OK;
            hipError_t e = pool_hip_alloc((void**)&p->ev, ev_bytes);
            if (e == hipSuccess) e = hipMemsetAsync(p->ev, 0, ev_bytes, p->stream);
            if (e != hipSuccess) {
                set_error("evidence counts: %s", hipGetErrorString(e));
                (void)hipGetLastError();
                return e == hipErrorOutOfMemory ? SIMRANK_ERR_ALLOC : SIMRANK_ERR_HIP;
            }
            return simrank_evidence_counts_blocked(g, 0, n, p->ev, p->rows_pad, p->stream);
        };
        rc = graph_create_with(t, n, n, nnz, pp.rp.data(), pp.cl.data(), pp.rs.data(), &p->g, &counts);
    }
    if (rc) return fail(rc);
    lap("graph object (evidence counts queued)");
    if (opt->dense_terms == 1) {                 // one fp16 operand term on the matrix cores (config 5's literal reading)
        rc = simrank_graph_set_dense_terms(p->g, 1);
        if (rc) return fail(rc);
    }
    if (p->half && !p->g->fused) {
        set_error("storage_fp16 needs the one-launch plan (tuning fuse = 1) and a graph that has one");
        return fail(SIMRANK_ERR_INVALID);
    }
#define PLAN_HIP(call)                                                                            \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            (void)hipGetLastError();                                                              \
            return fail(e_ == hipErrorOutOfMemory ? SIMRANK_ERR_ALLOC : SIMRANK_ERR_HIP);         \
        }                                                                                         \
    } while (0)
    // (no memset: simrank_plan_reset fills S[0] — zeros and the diagonal —, every update writes all of Tt and of the other
    // iterate before anything reads them, and the padding rows and columns of a panel are read by nobody: lanes that
    // hold columns past the edge compute on whatever is there and never store.  Three 17 GiB memsets were 10 ms of a
    // config-5 set-up.)
    for (float** b : {&p->S[0], &p->S[1], &p->Tt}) PLAN_HIP(pool_hip_alloc((void**)b, p->mat_bytes));
    PLAN_HIP(pool_hip_alloc((void**)&p->counters, sizeof(unsigned long long) * SIMRANK_CHANGED_SLOTS));
    for (int i = 0; i < 2; ++i) {
        PLAN_HIP(hipHostMalloc((void**)&p->host_counters[i], sizeof(unsigned long long) * SIMRANK_CHANGED_SLOTS, hipHostMallocPortable));
        PLAN_HIP(hipEventCreateWithFlags(&p->counted[i], hipEventDisableTiming));
    }
    PLAN_HIP(pool_hip_alloc((void**)&p->inv, size_t(n) * sizeof(int32_t)));
    PLAN_HIP(hipMemcpyAsync(p->inv, inv.data(), size_t(n) * sizeof(int32_t), hipMemcpyHostToDevice, p->stream));
    p->ord = ord;
    PLAN_HIP(pool_hip_alloc((void**)&p->ord_dev, size_t(n) * sizeof(int32_t)));
    PLAN_HIP(hipMemcpyAsync(p->ord_dev, p->ord.data(), size_t(n) * sizeof(int32_t), hipMemcpyHostToDevice, p->stream));
    PLAN_HIP(hipStreamSynchronize(p->stream));          // (inv is a host vector about to go away)
    lap("matrices allocated, orders uploaded, stream drained");
    if (opt->evidence) {
        int64_t live = 0, total = 1;
        rc = simrank_evidence_live_segments(p->ev, 32, p->rows_pad, n, n, &live, &total, p->stream);
        if (rc) return fail(rc);
        p->restrict_support = 2 * live < total ? 1 : 0;
    }
    if (opt->apriori) {
        // host n x n (caller's order) -> device row-major -> panel-blocked in the solver's order
        float* tmp = nullptr;
        int32_t* ord_dev = nullptr;
        PLAN_HIP(pool_hip_alloc((void**)&tmp, size_t(n) * size_t(n) * sizeof(float)));
        hipError_t e = pool_hip_alloc((void**)&ord_dev, size_t(n) * sizeof(int32_t));
        if (e == hipSuccess) e = pool_hip_alloc((void**)&p->prior, prior_bytes);
        if (e == hipSuccess) e = hipMemsetAsync(p->prior, 0, prior_bytes, p->stream);
        if (e == hipSuccess) e = hipMemcpy2DAsync(tmp, size_t(n) * 4, opt->apriori, size_t(opt->ld_apriori) * 4, size_t(n) * 4,
                                                  size_t(n), hipMemcpyHostToDevice, p->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(ord_dev, ord.data(), size_t(n) * 4, hipMemcpyHostToDevice, p->stream);
        if (e == hipSuccess) {
            // dst[i][j] = src[ord[i]][ord[j]]
            rc = simrank_permute_layout(tmp, n, 0, p->prior, 32, p->rows_pad, n, n, ord_dev, ord_dev, 4, p->stream);
            e = hipStreamSynchronize(p->stream);
        }
        (void)pool_free(tmp);
        (void)pool_free(ord_dev);
        if (e != hipSuccess) {
            set_error("plan prior upload: %s", hipGetErrorString(e));
            return fail(SIMRANK_ERR_HIP);
        }
        if (rc) return fail(rc);
    }
#undef PLAN_HIP
    lap("live segments, prior");
    rc = simrank_plan_reset(p);
    if (rc) return fail(rc);
    lap("reset queued");
    *out = p;
    return SIMRANK_OK;
}

int simrank_plan_reset(simrank_plan* p) {
    SR_REQUIRE(p, "plan is NULL");
    SR_REQUIRE(p->S[0], "the plan's matrices were released (simrank_plan_trim)");
    p->cur = 0;
    p->updates = 0;
    p->at_identity = 1;
    if (p->half) return simrank_fill_identity_blocked_h16(p->S[0], p->n, p->n, p->rows_pad, 0, kHalfScale, p->stream);
    return simrank_fill_identity_blocked(p->S[0], p->n, p->n, p->rows_pad, 0, p->stream);
}

int simrank_plan_step(simrank_plan* p, double eps, int32_t exact_count, int64_t* n_changed) {
    SR_REQUIRE(p, "plan is NULL");
    SR_REQUIRE(p->S[0], "the plan's matrices were released (simrank_plan_trim)");
    const int rc = leg_pair(p, eps, exact_count, 0);
    if (rc) return rc;
    p->cur ^= 1;
    ++p->updates;
    if (n_changed) {
        unsigned long long c = 0;
        const int rc2 = read_count(p, 0, &c);
        if (rc2) return rc2;
        *n_changed = (int64_t)c;
    }
    return SIMRANK_OK;
}

int simrank_plan_run_cb(simrank_plan* p, int32_t iterations, double eps, simrank_progress_fn progress, void* user,
                        int32_t* updates_done, int32_t* converged_at) {
    SR_REQUIRE(p, "plan is NULL");
    SR_REQUIRE(iterations >= 0, "iterations < 0");
    SR_REQUIRE(p->S[0], "the plan's matrices were released (simrank_plan_trim)");
    int rc = simrank_plan_reset(p);
    if (rc) return rc;
    int32_t conv = -1, done = 0;
    // progress(user, k, 0): loop index k goes on to an update (SimRank.py:135 `update_progress(k / iterations)`);
    // progress(user, k, 1): the test passed at loop index k (:131-133).  A nonzero return value ends the loop there.
    auto tell = [&](int32_t k, int32_t converged) { return progress ? progress(user, k, converged) : 0; };
    bool stop = false;
    if (iterations > 0 && !(1.0 > eps)) {
        conv = 0;           // loop index 0 compares S_0 = I with the zero matrix: "converged" unless 1 > eps
        (void)tell(0, 1);
    } else if (iterations > 0 && !(stop = tell(0, 0) != 0)) {
        rc = leg_pair(p, eps, 0, 1);                 // update 1: reads S[cur], writes S[cur ^ 1]
        if (rc) return rc;
        for (int32_t k = 1;; ++k) {
            // updates 1 .. k are queued, 1 .. k - 1 adopted; the count of update k is on its way
            p->cur ^= 1;                             // S[cur] = result of update k
            done = k;
            if (k == iterations) break;              // the reference makes no test after its last update
            // loop index k tests the count of update k and, if it may go on, runs update k + 1 — which is
            // queued NOW, before the count is known (it reads S[cur], writes the buffer of the iterate before)
            // (small graphs only, common.h kSpeculateBelow: a long update is queued once its predecessor's count is known)
            const bool spec = p->n < kSpeculateBelow;
            if (spec) {
                rc = leg_pair(p, eps, 0, (k + 1) & 1);
                if (rc) return rc;
            }
            unsigned long long c = 0;
            rc = read_count(p, k & 1, &c);           // waits for update k only
            if (rc) return rc;
            if (c == 0) {                            // converged at loop index k: k updates applied; the
                conv = k;                            // speculative one is not adopted
                (void)tell(k, 1);
                break;
            }
            if (tell(k, 0) != 0) {                   // the caller ends the loop: k updates applied
                stop = true;
                break;
            }
            if (!spec) {
                rc = leg_pair(p, eps, 0, (k + 1) & 1);
                if (rc) return rc;
            }
        }
    }
    SR_HIP(hipStreamSynchronize(p->stream));
    p->updates = done;
    if (updates_done) *updates_done = done;
    if (converged_at) *converged_at = conv;
    (void)stop;
    return SIMRANK_OK;
}

int simrank_plan_run(simrank_plan* p, int32_t iterations, double eps, int32_t* updates_done, int32_t* converged_at) {
    return simrank_plan_run_cb(p, iterations, eps, nullptr, nullptr, updates_done, converged_at);
}

int simrank_plan_result(simrank_plan* p, float* dst, int64_t ld) {
    SR_REQUIRE(p && dst && ld >= p->n, "bad result arguments");
    SR_REQUIRE(p->S[0], "the plan's matrices were released (simrank_plan_trim)");
    // dst[i][j] = S[inv[i]][inv[j]]: out of the panel-blocked layout and the solver's node order in one pass
    if (p->half) {
        // through an f32 panel-blocked scratch copy
        float* wide = nullptr;
        const size_t bytes = size_t((p->n + 31) / 32) * size_t(p->rows_pad) * 32 * sizeof(float);
        SR_HIP(pool_hip_alloc((void**)&wide, bytes));
        int rc = simrank_widen_blocked_h16(p->S[p->cur], p->rows_pad, wide, p->rows_pad, p->n, p->n, kHalfScale, p->stream);
        if (!rc) rc = simrank_permute_layout(wide, 32, p->rows_pad, dst, ld, 0, p->n, p->n, p->inv, p->inv, 4, p->stream);
        (void)hipStreamSynchronize(p->stream);
        (void)pool_free(wide);
        return rc;
    }
    return simrank_permute_layout(p->S[p->cur], 32, p->rows_pad, dst, ld, 0, p->n, p->n, p->inv, p->inv, 4, p->stream);
}

int simrank_plan_result_f64(simrank_plan* p, double* dst, int64_t ld) {
    SR_REQUIRE(p && dst && ld >= p->n, "bad result arguments");
    SR_REQUIRE(p->S[0], "the plan's matrices were released (simrank_plan_trim)");
    // Out of the panel-blocked layout and the solver's node order band by band (handback.hip), FULL form (mode 0): every
    // element crosses PCIe.  The symmetric form (upper triangle over PCIe, mirrored by the host threads) is opt-in
    // (SIMRANK_SYM_HANDBACK=1) and checks its premise on the device first — a plan with an asymmetric prior has asymmetric
    // iterates (SimRank.py:453), and even symmetric ones are bitwise symmetric only outside the diagonal tiles.
    const float* src = p->S[p->cur];
    float* wide = nullptr;
    if (p->half) {
        const size_t bytes = size_t((p->n + 31) / 32) * size_t(p->rows_pad) * 32 * sizeof(float);
        SR_HIP(pool_hip_alloc((void**)&wide, bytes));
        const int rc = simrank_widen_blocked_h16(p->S[p->cur], p->rows_pad, wide, p->rows_pad, p->n, p->n, kHalfScale, p->stream);
        if (rc) {
            (void)hipStreamSynchronize(p->stream);
            (void)pool_free(wide);
            return rc;
        }
        src = wide;
    }
    const int rc = simrank_handback_f64(dst, ld, src, 32, p->rows_pad, p->n, p->inv, 0, p->stream);
    (void)hipStreamSynchronize(p->stream);
    (void)pool_free(wide);
    return rc;
}

int simrank_plan_evidence_u8(simrank_plan* p, uint8_t* dst, int64_t ld) {
    SR_REQUIRE(p && dst && ld >= p->n, "bad evidence arguments");
    SR_REQUIRE(p->ev, "the plan was created without evidence");
    // dst[i][j] = counts[inv[i]][inv[j]] (saturated at 255; Evidence = 1 - 0.5 ** count, SimRank.py:316)
    uint8_t* tmp = nullptr;
    SR_HIP(pool_hip_alloc((void**)&tmp, size_t(p->n) * size_t(p->n)));
    int rc = simrank_permute_layout(p->ev, 32, p->rows_pad, tmp, p->n, 0, p->n, p->n, p->inv, p->inv, 1, p->stream);
    hipError_t e = hipSuccess;
    if (!rc) e = hipMemcpy2DAsync(dst, size_t(ld), tmp, size_t(p->n), size_t(p->n), size_t(p->n), hipMemcpyDeviceToHost, p->stream);
    const hipError_t e2 = hipStreamSynchronize(p->stream);
    (void)pool_free(tmp);
    if (e != hipSuccess || e2 != hipSuccess) {
        set_error("simrank_plan_evidence_u8: %s", hipGetErrorString(e != hipSuccess ? e : e2));
        return SIMRANK_ERR_HIP;
    }
    return rc;
}

int simrank_plan_trim(simrank_plan* p) {
    SR_REQUIRE(p, "plan is NULL");
    // what a finished fit no longer needs: the iterates, the transposed product, the prior (the evidence counts and the
    // node orders stay: simrank_plan_evidence_u8 reads them later — the estimators' lazy `Evidence` attribute)
    if (p->stream) SR_HIP(hipStreamSynchronize(p->stream));
    (void)pool_free(p->S[0]); (void)pool_free(p->S[1]); (void)pool_free(p->Tt); (void)pool_free(p->prior);
    p->S[0] = p->S[1] = p->Tt = p->prior = nullptr;
    return SIMRANK_OK;
}

int simrank_plan_rows_f32(simrank_plan* p, const int32_t* rows, int32_t n_rows, float* dst, int64_t ld) {
    SR_REQUIRE(p && rows && dst && n_rows > 0 && ld >= p->n, "bad row arguments");
    SR_REQUIRE(p->S[0], "the plan's matrices were released (simrank_plan_trim)");
    return rows_to_host(p->S[p->cur], p->rows_pad, p->n, p->inv, rows, n_rows, dst, ld, p->half ? 2 : 4, kHalfScale, p->stream);
}

int simrank_plan_topk(simrank_plan* p, int32_t k, int32_t exclude_diag, int32_t* idx_host, float* val_host) {
    SR_REQUIRE(p && idx_host && val_host && k > 0 && k <= 1024, "bad top-k arguments");
    SR_REQUIRE(p->S[0], "the plan's matrices were released (simrank_plan_trim)");
    // The selection runs on the plan's own panel-blocked matrix in the solver's order (one pass, eight rows per wave;
    // fp16-held: on its f32 copy), reporting the caller's ids; the rows go back into the caller's order on the host —
    // 2 x n x k values across PCIe instead of n^2, and no n^2 copy on the device either.
    const int64_t n = p->n;
    const size_t wide_bytes = size_t((n + 31) / 32) * size_t(p->rows_pad) * 32 * sizeof(float);
    float* wide = nullptr;
    int32_t* idx_dev = nullptr;
    float* val_dev = nullptr;
    hipError_t e = hipSuccess;
    int rc = SIMRANK_OK;
    if (p->half) {
        e = pool_hip_alloc((void**)&wide, wide_bytes);
        if (e == hipSuccess)
            rc = simrank_widen_blocked_h16(p->S[p->cur], p->rows_pad, wide, p->rows_pad, n, n, kHalfScale, p->stream);
    }
    if (e == hipSuccess) e = pool_hip_alloc((void**)&idx_dev, size_t(n) * size_t(k) * sizeof(int32_t));
    if (e == hipSuccess) e = pool_hip_alloc((void**)&val_dev, size_t(n) * size_t(k) * sizeof(float));
    std::vector<int32_t> idx_s;
    std::vector<float> val_s;
    if (e == hipSuccess && !rc) {
        // (the diagonal of the solver's order is the diagonal of the caller's: position r against position r)
        rc = simrank_topk_rows_blocked(p->half ? wide : p->S[p->cur], p->rows_pad, n, n, 0, p->ord_dev, k, exclude_diag, idx_dev,
                                       val_dev, p->stream);
        idx_s.resize(size_t(n) * size_t(k));
        val_s.resize(size_t(n) * size_t(k));
    }
    if (e == hipSuccess && !rc)
        e = hipMemcpyAsync(idx_s.data(), idx_dev, size_t(n) * size_t(k) * sizeof(int32_t), hipMemcpyDeviceToHost, p->stream);
    if (e == hipSuccess && !rc)
        e = hipMemcpyAsync(val_s.data(), val_dev, size_t(n) * size_t(k) * sizeof(float), hipMemcpyDeviceToHost, p->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    else (void)hipStreamSynchronize(p->stream);
    (void)pool_free(wide); (void)pool_free(idx_dev); (void)pool_free(val_dev);
    if (e != hipSuccess) {
        set_error("simrank_plan_topk: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        return SIMRANK_ERR_HIP;
    }
    if (rc) return rc;
    for (int64_t r = 0; r < n; ++r) {
        const int64_t a = p->ord[(size_t)r];
        std::memcpy(idx_host + a * k, idx_s.data() + r * k, size_t(k) * sizeof(int32_t));
        std::memcpy(val_host + a * k, val_s.data() + r * k, size_t(k) * sizeof(float));
    }
    return SIMRANK_OK;
}

int simrank_plan_set_timing(simrank_plan* p, int32_t updates) {
    SR_REQUIRE(p && updates >= 0 && updates <= (1 << 20), "bad timing arguments");
    // events for `updates` updates are created NOW (outside whatever the caller times); 0 switches the stamps off
    for (hipEvent_t e : p->ev_used) p->ev_pool.push_back(e);
    p->ev_used.clear();
    while ((int64_t)p->ev_pool.size() < 3 * (int64_t)updates) {
        hipEvent_t e;
        SR_HIP(hipEventCreate(&e));
        p->ev_pool.push_back(e);
    }
    p->timing = updates > 0;
    return SIMRANK_OK;
}

int simrank_plan_leg_times(simrank_plan* p, double* leg1_ms, double* leg2_ms, int32_t* updates) {
    SR_REQUIRE(p, "plan is NULL");
    SR_HIP(hipStreamSynchronize(p->stream));
    double a = 0, b = 0;
    const size_t n = p->ev_used.size() / 3;
    for (size_t u = 0; u < n; ++u) {
        float t1 = 0, t2 = 0;
        SR_HIP(hipEventElapsedTime(&t1, p->ev_used[3 * u], p->ev_used[3 * u + 1]));
        SR_HIP(hipEventElapsedTime(&t2, p->ev_used[3 * u + 1], p->ev_used[3 * u + 2]));
        a += t1;
        b += t2;
    }
    if (leg1_ms) *leg1_ms = n ? a / (double)n : 0.0;
    if (leg2_ms) *leg2_ms = n ? b / (double)n : 0.0;
    if (updates) *updates = (int32_t)n;
    for (hipEvent_t e : p->ev_used) p->ev_pool.push_back(e);
    p->ev_used.clear();
    return SIMRANK_OK;
}

int simrank_plan_info(const simrank_plan* p, int64_t* n, int32_t* updates, const simrank_graph** graph) {
    SR_REQUIRE(p, "plan is NULL");
    if (n) *n = p->n;
    if (updates) *updates = p->updates;
    if (graph) *graph = p->g;
    return SIMRANK_OK;
}

}  // extern "C"
