// The loop of BipartiteSimRank.fit / BipartiteSimRankPP.fit / BipartitleAprioriSimRank.fit behind the C ABI
// (create_plan(N or (n1, n2)) / step / download), the two-matrix twin of plan.hip:
//
//     for k in range(iterations):                               SimRank.py:288-302 (:410-424, :478-492)
//         if converged(S1_old, S1) and converged(S2_old, S2): break
//         S1 = E1 * C1 * W12.S2.W12^T (+ lbd1 A1); diag <- 1       the group-1 update reads S2 of the iteration before,
//         S2 = E2 * C2 * W21.S1.W21^T (+ lbd2 A2); diag <- 1       the group-2 update the NEW S1 (Gauss-Seidel, :300-302)
//
// One edge set describes both patterns: W12 = diag(rowscale1) . A (n1 x n2), W21 = diag(rowscale2) . A^T.
// Each update is two launches (leg 1: fused_trans_kernel on a rectangular pattern, leg 2: upper-triangle gather
// with the fused epilogue and count); iteration k + 1 is queued before the counts of iteration k are read, as in
// plan.hip.  Evidence: by default the corrected form — E1 from the group-1 pattern, E2 from the group-2 pattern.
// options.strict_reference = 1 is the reference's own behaviour (SimRank.py:420-423, :488-491, SURVEY.md quirk
// Q2): BOTH updates are multiplied by Evidence_N1, position by position in the caller's node order — n1 = n2:
// the group-2 update is gated by the counts of the group-1 pattern; n1 = 1: NumPy broadcasts the 1 x 1 array,
// one count gates every element; otherwise NumPy raises "operands could not be broadcast together" when the
// first group-2 update RUNS (iterations = 0 or eps >= 1 still return the identities): so do step / run here.
#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>

#include <chrono>
#include <string>
#include <thread>

#include "common.h"

namespace {

struct side_t {
    int64_t n = 0, k = 0, rows_pad = 0, k_rows_pad = 0;    // n: own group, k: the other group
    size_t mat_bytes = 0, t_bytes = 0;
    simrank_graph* g = nullptr;                // n x k, solver order on both sides
    float* S[2] = {nullptr, nullptr};          // n x n, panel-blocked, ping-pong
    float* Tt = nullptr;                       // k x n: (W . S_other)^T
    uint8_t* ev = nullptr;
    float* prior = nullptr;
    int32_t* inv = nullptr;                    // device: position of caller's node i in the solver's order
    float coef = 0.8f, lbd = 0.f;
    int32_t restrict_support = 0;
    int cur = 0;
};

}  // namespace

struct simrank_biplan {
    side_t s[2];
    unsigned long long* counters = nullptr;                          // device, SIMRANK_CHANGED_SLOTS (zeroed per leg 2)
    unsigned long long* host_counters[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [iteration & 1][side]
    hipEvent_t counted[2] = {nullptr, nullptr};                      // both counts of an iteration have landed
    hipStream_t stream = nullptr;
    int32_t updates = 0;
    int32_t broadcast_error = 0;     // strict_reference with evidence and n1 != n2, n1 != 1 (quirk Q2)
    int32_t asym = 0;                // a prior of either group is not symmetric: both iterates are asymmetric (un-fused epilogue)
    int32_t identity_leg1 = 1;       // group 1's first update reads S2 = I: its leg 1 is W12^T written directly (SIMRANK_IDENTITY_LEG1=0: off)
    int32_t at_identity = 0;         // S2 is the identity (reset), no update queued since
};

namespace simrank {

static int side_update(simrank_biplan* p, int w, double eps, int32_t exact_count, unsigned long long* host_slot) {
    side_t& a = p->s[w];
    const side_t& o = p->s[w ^ 1];
    // leg 1: Tt (k x n) = (W . S_other)^T; S_other is k x k — the identity for the very first update of group 1 (SimRank.py:280-285:
    // group 2's first update already reads the new S1), whose product is W^T written directly: the same bits, no gathers
    const bool from_identity = w == 0 && p->at_identity && p->identity_leg1;
    if (w == 0) p->at_identity = 0;
    int rc = from_identity ? identity_leg1_blocked(a.g, a.Tt, a.k_rows_pad, p->stream)
                           : simrank_spmm_blocked(a.g, o.S[o.cur], o.rows_pad, a.k, a.Tt, a.k_rows_pad, 1, nullptr, p->stream);
    if (rc) return rc;
    simrank_epilogue ep{};
    ep.coef = a.coef;
    ep.lbd = a.lbd;
    ep.evidence = a.ev;
    ep.ld_evidence = 32;
    ep.apriori = a.prior;
    ep.ld_apriori = 32;
    ep.previous = a.S[a.cur];
    ep.ld_previous = 32;
    ep.eps = eps;
    ep.n_changed = p->counters;
    ep.diag_col0 = 0;
    ep.set_diag = 1;
    ep.symmetric = 1;
    ep.restrict_support = a.restrict_support;
    ep.count_any = exact_count ? 0 : 1;
    if (p->asym) {
        // asymmetric iterates (SimRank.py:488, :491 with a prior that is not symmetric): W . Tt is the transpose of
        // W S_other W^T — stored transposed, then the epilogue as a pass of its own (exact count)
        ep.symmetric = 0;
        ep.restrict_support = 0;
        rc = simrank_spmm_blocked(a.g, a.Tt, a.k_rows_pad, a.n, a.S[a.cur ^ 1], a.rows_pad, 1, nullptr, p->stream);
        if (!rc) rc = simrank_epilogue_apply_blocked(a.S[a.cur ^ 1], a.S[a.cur ^ 1], a.n, a.n, a.rows_pad, &ep, p->stream);
    } else {
        rc = simrank_spmm_blocked(a.g, a.Tt, a.k_rows_pad, a.n, a.S[a.cur ^ 1], a.rows_pad, 0, &ep, p->stream);
    }
    if (rc) return rc;
    SR_HIP(hipMemcpyAsync(host_slot, p->counters, sizeof(unsigned long long) * SIMRANK_CHANGED_SLOTS,
                          hipMemcpyDeviceToHost, p->stream));
    a.cur ^= 1;              // (the group-2 update of the same iteration reads the new S1)
    return SIMRANK_OK;
}

// one loop body: both updates, counts into slot `it`
static int iteration(simrank_biplan* p, double eps, int32_t exact_count, int it) {
    // (the reference has updated S1 when NumPy raises at :423 / :491; nobody sees that S1: the exception ends fit)
    SR_REQUIRE(!p->broadcast_error, "operands could not be broadcast together with shapes (%lld,%lld) (%lld,%lld) ",
               (long long)p->s[0].n, (long long)p->s[0].n, (long long)p->s[1].n, (long long)p->s[1].n);
    int rc = side_update(p, 0, eps, exact_count, p->host_counters[it][0]);
    if (!rc) rc = side_update(p, 1, eps, exact_count, p->host_counters[it][1]);
    if (rc) return rc;
    SR_HIP(hipEventRecord(p->counted[it], p->stream));
    return SIMRANK_OK;
}

static int read_counts(simrank_biplan* p, int it, unsigned long long* c1, unsigned long long* c2) {
    SR_HIP(hipEventSynchronize(p->counted[it]));
    unsigned long long t[2] = {0, 0};
    for (int w = 0; w < 2; ++w)
        for (int i = 0; i < SIMRANK_CHANGED_SLOTS; ++i) t[w] += p->host_counters[it][w][i];
    *c1 = t[0];
    *c2 = t[1];
    return SIMRANK_OK;
}

}  // namespace simrank

using namespace simrank;

extern "C" {

int simrank_biplan_destroy(simrank_biplan* p) {
    if (!p) return SIMRANK_OK;
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    for (side_t& a : p->s) {
        (void)pool_free(a.S[0]); (void)pool_free(a.S[1]); (void)pool_free(a.Tt); (void)pool_free(a.ev);
        (void)pool_free(a.prior); (void)pool_free(a.inv);
        simrank_graph_destroy(a.g);
    }
    (void)pool_free(p->counters);
    for (int i = 0; i < 2; ++i) {
        for (int w = 0; w < 2; ++w)
            if (p->host_counters[i][w]) (void)hipHostFree(p->host_counters[i][w]);
        if (p->counted[i]) (void)hipEventDestroy(p->counted[i]);
    }
    delete p;
    return SIMRANK_OK;
}

int simrank_biplan_reset(simrank_biplan* p) {
    SR_REQUIRE(p, "plan is NULL");
    SR_REQUIRE(p->s[0].S[0], "the plan's matrices were released (simrank_biplan_trim)");
    p->updates = 0;
    p->at_identity = 1;
    for (side_t& a : p->s) {
        a.cur = 0;
        const int rc = simrank_fill_identity_blocked(a.S[0], a.n, a.n, a.rows_pad, 0, p->stream);
        if (rc) return rc;
    }
    return SIMRANK_OK;
}

int simrank_biplan_create(int64_t n1, int64_t n2, int64_t nnz, const int32_t* rowptr12, const int32_t* col12,
                          const float* rowscale1, const float* rowscale2, const simrank_biplan_options* opt,
                          void* stream, simrank_biplan** out) {
    SR_REQUIRE(out, "out is NULL");
    *out = nullptr;
    const bool timed = std::getenv("SIMRANK_TIME_BUILD") != nullptr;     // diagnostic: phase durations on stderr
    const auto t_start = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (timed)
            std::fprintf(stderr, "simrank_biplan_create: %6.1f ms  %s\n",
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(), what);
    };
    BiPlanPrep pp;
    {
        const int rc = biplan_prepare(n1, n2, nnz, rowptr12, col12, rowscale1, rowscale2, opt, &pp);   // (planprep.hip)
        if (rc) return rc;
    }
    lap("validated, ordered, both patterns renamed");
    const float* priors[2] = {opt->apriori1, opt->apriori2};
    const int64_t lds[2] = {opt->ld_apriori1, opt->ld_apriori2};
    const int64_t ns[2] = {n1, n2};
    const std::vector<int32_t>* ord = pp.ord;
    const std::vector<int32_t>* inv = pp.inv;
    simrank_biplan* p = new simrank_biplan;
    p->stream = as_stream(stream);
    p->asym = pp.asym ? 1 : 0;
    if (const char* e = std::getenv("SIMRANK_IDENTITY_LEG1")) p->identity_leg1 = (*e == '0') ? 0 : 1;
    auto fail = [&](int code) { simrank_biplan_destroy(p); return code; };
    {
        // The two graph objects are independent of each other: on graphs large enough for threads to pay the second one is
        // created on a thread beside the first (MovieLens-shaped: 9.4 + 7.9 ms one after the other).  As in plan.hip: leg 1 of
        // both groups is the one-launch kernel wherever spmm.hip's conditions hold, so the dense-block plan could only serve
        // the upper-triangle leg 2 — built only if that leg would take it (dense_lazy).
        Tuning tw[2] = {tuning_snapshot(), Tuning()};
        tw[1] = tw[0];
        int rcs[2] = {SIMRANK_OK, SIMRANK_OK};
        std::string errs[2];
        for (int w = 0; w < 2; ++w) {
            side_t& a = p->s[w];
            a.n = ns[w];
            a.k = ns[w ^ 1];
            a.coef = w == 0 ? opt->c1 : opt->c2;
            a.lbd = w == 0 ? opt->lbd1 : opt->lbd2;
            a.rows_pad = (a.n + 7) / 8 * 8 + 8;
            a.k_rows_pad = (a.k + 7) / 8 * 8 + 8;
            a.mat_bytes = size_t((a.n + 31) / 32) * size_t(a.rows_pad) * 32 * sizeof(float);     // n x n
            a.t_bytes = size_t((a.n + 31) / 32) * size_t(a.k_rows_pad) * 32 * sizeof(float);      // k x n
            if (tw[w].fuse == 1 && ((tw[w].triangle && a.n >= 64) || p->asym) && a.k <= tw[w].fuse_max_rows &&
                (a.k_rows_pad + 1) * 128 < (int64_t(1) << 31))
                tw[w].dense_lazy = 1;
        }
        auto make = [&](int w) {
            side_t& a = p->s[w];
            rcs[w] = graph_create_with(tw[w], a.n, a.k, nnz, pp.rp[w].data(), pp.cl[w].data(), pp.rs[w].data(), &a.g);
            if (rcs[w]) errs[w] = simrank_last_error();
        };
        if (nnz >= 20000) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            std::thread second([&]() {
                (void)hipSetDevice(dev);
                make(1);
            });
            make(0);
            second.join();
        } else {
            make(0);
            if (!rcs[0]) make(1);
        }
        for (int w = 0; w < 2; ++w)
            if (rcs[w]) {
                set_error("%s", errs[w].c_str());
                return fail(rcs[w]);
            }
    }
#define BIPLAN_HIP(call)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            (void)hipGetLastError();                                                              \
            return fail(e_ == hipErrorOutOfMemory ? SIMRANK_ERR_ALLOC : SIMRANK_ERR_HIP);         \
        }                                                                                         \
    } while (0)
    lap("graph objects");
    BIPLAN_HIP(pool_hip_alloc((void**)&p->counters, sizeof(unsigned long long) * SIMRANK_CHANGED_SLOTS));
    for (int i = 0; i < 2; ++i) {
        for (int w = 0; w < 2; ++w)
            BIPLAN_HIP(hipHostMalloc((void**)&p->host_counters[i][w], sizeof(unsigned long long) * SIMRANK_CHANGED_SLOTS,
                                     hipHostMallocPortable));
        BIPLAN_HIP(hipEventCreateWithFlags(&p->counted[i], hipEventDisableTiming));
    }
    for (int w = 0; w < 2; ++w) {
        side_t& a = p->s[w];
        // (no memset: as in plan.hip — reset fills S[0], every update writes all of Tt and of the other iterate first)
        for (float** b : {&a.S[0], &a.S[1]}) BIPLAN_HIP(pool_hip_alloc((void**)b, a.mat_bytes));
        BIPLAN_HIP(pool_hip_alloc((void**)&a.Tt, a.t_bytes));
        BIPLAN_HIP(pool_hip_alloc((void**)&a.inv, size_t(a.n) * sizeof(int32_t)));
        BIPLAN_HIP(hipMemcpyAsync(a.inv, inv[w].data(), size_t(a.n) * sizeof(int32_t), hipMemcpyHostToDevice, p->stream));
        BIPLAN_HIP(hipStreamSynchronize(p->stream));
        const bool q2 = opt->evidence && opt->strict_reference && w == 1;      // Evidence_N1 on the group-2 update
        if (q2 && n1 != n2 && n1 != 1) {
            p->broadcast_error = 1;
        } else if (opt->evidence) {
            // common-neighbour counts inside the group (SimRank.py:311-320 on this group's pattern)
            const size_t ev_bytes = size_t((a.n + 31) / 32) * size_t(a.rows_pad) * 32;
            BIPLAN_HIP(pool_hip_alloc((void**)&a.ev, ev_bytes));
            int rc = SIMRANK_OK;
            if (q2 && n1 == 1 && n2 != 1) {
                // the 1 x 1 Evidence_N1 broadcasts: the one group-1 node's count (with itself) gates every element
                const int cnt = rowscale1[0] != 0.f ? (int)std::min<int64_t>(255, nnz) : 0;
                BIPLAN_HIP(hipMemsetAsync(a.ev, cnt, ev_bytes, p->stream));
            } else if (q2) {
                // n1 = n2: element (i, j) of the group-2 update is multiplied by Evidence_N1[i][j], positions in the
                // caller's order: the counts of the group-1 pattern with its rows taken in THIS group's solver order
                std::vector<int32_t> rp((size_t)n1 + 1, 0), cl((size_t)std::max<int64_t>(1, nnz));
                std::vector<float> rs((size_t)n1);
                for (int64_t r = 0; r < n1; ++r) {
                    const int32_t src = ord[1][(size_t)r];
                    const int32_t b = rowptr12[src], e = rowptr12[src + 1];
                    std::copy(col12 + b, col12 + e, cl.data() + rp[(size_t)r]);
                    std::sort(cl.data() + rp[(size_t)r], cl.data() + rp[(size_t)r] + (e - b));
                    rp[(size_t)r + 1] = rp[(size_t)r] + (e - b);
                    rs[(size_t)r] = rowscale1[src];
                }
                simrank_graph* g1 = nullptr;
                rc = simrank_graph_create(n1, n2, nnz, rp.data(), cl.data(), rs.data(), &g1);
                if (rc) return fail(rc);
                hipError_t e = hipMemsetAsync(a.ev, 0, ev_bytes, p->stream);
                rc = e == hipSuccess ? simrank_evidence_counts_blocked(g1, 0, a.n, a.ev, a.rows_pad, p->stream) : SIMRANK_ERR_HIP;
                (void)hipStreamSynchronize(p->stream);
                simrank_graph_destroy(g1);
            } else {
                BIPLAN_HIP(hipMemsetAsync(a.ev, 0, ev_bytes, p->stream));
                rc = simrank_evidence_counts_blocked(a.g, 0, a.n, a.ev, a.rows_pad, p->stream);
            }
            if (rc) return fail(rc);
            int64_t live = 0, total = 1;
            rc = simrank_evidence_live_segments(a.ev, 32, a.rows_pad, a.n, a.n, &live, &total, p->stream);
            if (rc) return fail(rc);
            a.restrict_support = 2 * live < total ? 1 : 0;
        }
        if (priors[w]) {
            float* tmp = nullptr;
            int32_t* ord_dev = nullptr;
            BIPLAN_HIP(pool_hip_alloc((void**)&tmp, size_t(a.n) * size_t(a.n) * sizeof(float)));
            hipError_t e = pool_hip_alloc((void**)&ord_dev, size_t(a.n) * sizeof(int32_t));
            if (e == hipSuccess) e = pool_hip_alloc((void**)&a.prior, a.mat_bytes);
            if (e == hipSuccess) e = hipMemsetAsync(a.prior, 0, a.mat_bytes, p->stream);
            if (e == hipSuccess) e = hipMemcpy2DAsync(tmp, size_t(a.n) * 4, priors[w], size_t(lds[w]) * 4, size_t(a.n) * 4,
                                                      size_t(a.n), hipMemcpyHostToDevice, p->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(ord_dev, ord[w].data(), size_t(a.n) * 4, hipMemcpyHostToDevice, p->stream);
            int rc = SIMRANK_OK;
            if (e == hipSuccess) {
                rc = simrank_permute_layout(tmp, a.n, 0, a.prior, 32, a.rows_pad, a.n, a.n, ord_dev, ord_dev, 4, p->stream);
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
    }
#undef BIPLAN_HIP
    lap("matrices, evidence counts, live segments, priors");
    const int rc = simrank_biplan_reset(p);
    if (rc) return fail(rc);
    lap("reset queued");
    *out = p;
    return SIMRANK_OK;
}

int simrank_biplan_step(simrank_biplan* p, double eps, int32_t exact_count, int64_t* changed1, int64_t* changed2) {
    SR_REQUIRE(p, "plan is NULL");
    SR_REQUIRE(p->s[0].S[0], "the plan's matrices were released (simrank_biplan_trim)");
    const int rc = iteration(p, eps, exact_count, 0);
    if (rc) return rc;
    ++p->updates;
    if (changed1 || changed2) {
        unsigned long long c1 = 0, c2 = 0;
        const int rc2 = read_counts(p, 0, &c1, &c2);
        if (rc2) return rc2;
        if (changed1) *changed1 = (int64_t)c1;
        if (changed2) *changed2 = (int64_t)c2;
    }
    return SIMRANK_OK;
}

int simrank_biplan_run_cb(simrank_biplan* p, int32_t iterations, double eps, simrank_progress_fn progress, void* user,
                          int32_t* updates_done, int32_t* converged_at) {
    SR_REQUIRE(p, "plan is NULL");
    SR_REQUIRE(iterations >= 0, "iterations < 0");
    SR_REQUIRE(p->s[0].S[0], "the plan's matrices were released (simrank_biplan_trim)");
    int rc = simrank_biplan_reset(p);
    if (rc) return rc;
    int32_t conv = -1, done = 0;
    // progress(user, k, converged) as in simrank_plan_run_cb (SimRank.py:289-296): a nonzero return value ends the loop
    auto tell = [&](int32_t k, int32_t converged) { return progress ? progress(user, k, converged) : 0; };
    if (iterations > 0 && !(1.0 > eps)) {
        conv = 0;           // loop index 0 compares the identities with zero matrices: "converged" unless 1 > eps
        (void)tell(0, 1);
    } else if (iterations > 0 && tell(0, 0) == 0) {
        rc = iteration(p, eps, 0, 1);                    // iteration 1
        if (rc) return rc;
        for (int32_t k = 1;; ++k) {
            done = k;
            if (k == iterations) break;                  // the reference makes no test after its last iteration
            // iteration k + 1 is queued before the counts of iteration k are known; if they say "converged" it is
            // not adopted: it wrote the buffers of the iterates before last, the current ones are untouched
            // (small graphs only, common.h kSpeculateBelow: a long iteration is queued once its predecessor's counts are known)
            const bool spec = std::max(p->s[0].n, p->s[1].n) < kSpeculateBelow;
            const int c1_cur = p->s[0].cur, c2_cur = p->s[1].cur;
            if (spec) {
                rc = iteration(p, eps, 0, (k + 1) & 1);
                if (rc) return rc;
            }
            unsigned long long c1 = 0, c2 = 0;
            rc = read_counts(p, k & 1, &c1, &c2);
            if (rc) return rc;
            const bool conv_now = c1 == 0 && c2 == 0;    // SimRank.py:289: both groups
            const bool stop = !conv_now && tell(k, 0) != 0;
            if (conv_now || stop) {
                if (conv_now) {
                    conv = k;
                    (void)tell(k, 1);
                }
                SR_HIP(hipStreamSynchronize(p->stream));  // (the speculative iteration must not outlive its inputs)
                p->s[0].cur = c1_cur;
                p->s[1].cur = c2_cur;
                break;
            }
            if (!spec) {
                rc = iteration(p, eps, 0, (k + 1) & 1);
                if (rc) return rc;
            }
        }
    }
    SR_HIP(hipStreamSynchronize(p->stream));
    p->updates = done;
    if (updates_done) *updates_done = done;
    if (converged_at) *converged_at = conv;
    return SIMRANK_OK;
}

int simrank_biplan_run(simrank_biplan* p, int32_t iterations, double eps, int32_t* updates_done, int32_t* converged_at) {
    return simrank_biplan_run_cb(p, iterations, eps, nullptr, nullptr, updates_done, converged_at);
}

int simrank_biplan_result_f64(simrank_biplan* p, int32_t group, double* dst, int64_t ld) {
    SR_REQUIRE(p && dst && (group == 1 || group == 2), "bad result arguments");
    side_t& a = p->s[group - 1];
    SR_REQUIRE(ld >= a.n, "ld %lld < n", (long long)ld);
    SR_REQUIRE(a.S[0], "the plan's matrices were released (simrank_biplan_trim)");
    // dst[i][j] = S[inv[i]][inv[j]], full form (mode 0: every element crosses PCIe; asymmetric priors give asymmetric iterates)
    const int rc = simrank_handback_f64(dst, ld, a.S[a.cur], 32, a.rows_pad, a.n, a.inv, 0, p->stream);
    (void)hipStreamSynchronize(p->stream);
    return rc;
}

int simrank_biplan_rows_f32(simrank_biplan* p, int32_t group, const int32_t* rows, int32_t n_rows, float* dst, int64_t ld) {
    SR_REQUIRE(p && rows && dst && (group == 1 || group == 2) && n_rows > 0, "bad row arguments");
    side_t& a = p->s[group - 1];
    SR_REQUIRE(ld >= a.n, "ld %lld < n", (long long)ld);
    SR_REQUIRE(a.S[0], "the plan's matrices were released (simrank_biplan_trim)");
    return rows_to_host(a.S[a.cur], a.rows_pad, a.n, a.inv, rows, n_rows, dst, ld, 4, 1.0f, p->stream);
}

int simrank_biplan_topk(simrank_biplan* p, int32_t group, int32_t k, int32_t exclude_diag, int32_t* idx_host, float* val_host) {
    SR_REQUIRE(p && idx_host && val_host && (group == 1 || group == 2) && k > 0 && k <= 1024, "bad top-k arguments");
    side_t& a = p->s[group - 1];
    SR_REQUIRE(a.S[0], "the plan's matrices were released (simrank_biplan_trim)");
    const int64_t n = a.n;
    // as simrank_plan_topk: selected on the panel-blocked matrix in the solver's order, caller's ids reported, the rows
    // put back into the caller's order on the host
    int32_t* idx_dev = nullptr;
    int32_t* ord_dev = nullptr;
    float* val_dev = nullptr;
    std::vector<int32_t> ord((size_t)n), idx_s((size_t)n * (size_t)k);
    std::vector<float> val_s((size_t)n * (size_t)k);
    hipError_t e = pool_hip_alloc((void**)&idx_dev, size_t(n) * size_t(k) * sizeof(int32_t));
    if (e == hipSuccess) e = pool_hip_alloc((void**)&val_dev, size_t(n) * size_t(k) * sizeof(float));
    if (e == hipSuccess) e = pool_hip_alloc((void**)&ord_dev, size_t(n) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemcpyAsync(ord.data(), a.inv, size_t(n) * 4, hipMemcpyDeviceToHost, p->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    int rc = SIMRANK_OK;
    if (e == hipSuccess) {
        std::vector<int32_t> o((size_t)n);
        for (int64_t i = 0; i < n; ++i) o[(size_t)ord[(size_t)i]] = (int32_t)i;      // position r holds caller's node o[r]
        ord.swap(o);
        e = hipMemcpyAsync(ord_dev, ord.data(), size_t(n) * 4, hipMemcpyHostToDevice, p->stream);
    }
    if (e == hipSuccess)
        rc = simrank_topk_rows_blocked(a.S[a.cur], a.rows_pad, n, n, 0, ord_dev, k, exclude_diag, idx_dev, val_dev, p->stream);
    if (e == hipSuccess && !rc)
        e = hipMemcpyAsync(idx_s.data(), idx_dev, size_t(n) * size_t(k) * sizeof(int32_t), hipMemcpyDeviceToHost, p->stream);
    if (e == hipSuccess && !rc)
        e = hipMemcpyAsync(val_s.data(), val_dev, size_t(n) * size_t(k) * sizeof(float), hipMemcpyDeviceToHost, p->stream);
    const hipError_t e2 = hipStreamSynchronize(p->stream);
    (void)pool_free(idx_dev); (void)pool_free(val_dev); (void)pool_free(ord_dev);
    if (e != hipSuccess || e2 != hipSuccess) {
        set_error("simrank_biplan_topk: %s", hipGetErrorString(e != hipSuccess ? e : e2));
        (void)hipGetLastError();
        return SIMRANK_ERR_HIP;
    }
    if (rc) return rc;
    for (int64_t r = 0; r < n; ++r) {
        const int64_t node = ord[(size_t)r];
        std::memcpy(idx_host + node * k, idx_s.data() + r * k, size_t(k) * sizeof(int32_t));
        std::memcpy(val_host + node * k, val_s.data() + r * k, size_t(k) * sizeof(float));
    }
    return SIMRANK_OK;
}

int simrank_biplan_evidence_u8(simrank_biplan* p, int32_t group, uint8_t* dst, int64_t ld) {
    SR_REQUIRE(p && dst && (group == 1 || group == 2), "bad evidence arguments");
    side_t& a = p->s[group - 1];
    SR_REQUIRE(ld >= a.n, "ld %lld < n", (long long)ld);
    SR_REQUIRE(a.ev, "no evidence counts for group %d (created without evidence, or strict_reference with n1 != n2)", group);
    // the counts that GATE this group's update (strict_reference: Evidence_N1's, position by position, for group 2 as well)
    uint8_t* tmp = nullptr;
    SR_HIP(pool_hip_alloc((void**)&tmp, size_t(a.n) * size_t(a.n)));
    const int rc = simrank_permute_layout(a.ev, 32, a.rows_pad, tmp, a.n, 0, a.n, a.n, a.inv, a.inv, 1, p->stream);
    hipError_t e = hipSuccess;
    if (!rc) e = hipMemcpy2DAsync(dst, size_t(ld), tmp, size_t(a.n), size_t(a.n), size_t(a.n), hipMemcpyDeviceToHost, p->stream);
    const hipError_t e2 = hipStreamSynchronize(p->stream);
    (void)pool_free(tmp);
    if (e != hipSuccess || e2 != hipSuccess) {
        set_error("simrank_biplan_evidence_u8: %s", hipGetErrorString(e != hipSuccess ? e : e2));
        return SIMRANK_ERR_HIP;
    }
    return rc;
}

int simrank_biplan_trim(simrank_biplan* p) {
    SR_REQUIRE(p, "plan is NULL");
    if (p->stream) SR_HIP(hipStreamSynchronize(p->stream));
    for (side_t& a : p->s) {
        (void)pool_free(a.S[0]); (void)pool_free(a.S[1]); (void)pool_free(a.Tt); (void)pool_free(a.prior);
        a.S[0] = a.S[1] = a.Tt = a.prior = nullptr;
    }
    return SIMRANK_OK;
}

}  // extern "C"
