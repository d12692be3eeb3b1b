// Host-only half of simrank_plan_create / simrank_biplan_create: argument validation, the solver's node
// order (ascending row length), the renamed CSR patterns, the transposed pattern of the
// bipartite plans.  No HIP call in here: `make asan` compiles this file for the host and tools/host/
// host_fuzz.cpp throws malformed inputs at it (non-monotone rowptr, columns out of range or repeated,
// asymmetric / non-finite priors), expecting SIMRANK_ERR_INVALID and no out-of-bounds read.
// Replaces nothing of the reference by itself: it is what `_create_graph` (SimRank.py:24-52, :168-200) does to
// an edge list, restated for callers that hand over CSR arrays.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <numeric>
#include <thread>

#include "common.h"

namespace simrank {

static int check_csr(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* rowptr, const int32_t* col) {
    SR_REQUIRE(rowptr[0] == 0 && rowptr[n_rows] == nnz, "rowptr does not span [0, nnz]");
    for (int64_t a = 0; a < n_rows; ++a)
        SR_REQUIRE(rowptr[a + 1] >= rowptr[a] && rowptr[a + 1] <= nnz, "rowptr not monotone at row %lld", (long long)a);
    for (int64_t j = 0; j < nnz; ++j) SR_REQUIRE(col[j] >= 0 && col[j] < n_cols, "column index %d out of range", col[j]);
    return SIMRANK_OK;
}

// `asym` (may be NULL): where a prior that is not symmetric is reported instead of refused — the single-GPU f32 plans run
// such a fit with leg 2 stored transposed and the epilogue as a pass of its own; the sharded plans and fp16-held matrices
// refuse it (SIMRANK_ERR_INVALID)
static int check_prior(const float* a, int64_t ld, int64_t n, int which, bool half, bool* asym = nullptr) {
    if (!a) return SIMRANK_OK;
    SR_REQUIRE(ld >= n, "prior %d: ld %lld < n", which, (long long)ld);
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = i; j < n; ++j) {
            const float v = a[i * ld + j];
            // (off the diagonal only, as prior_symmetric below: the reference overwrites the diagonal — fill_diagonal,
            // SimRank.py:454 — so a NaN there is neither an asymmetry nor anything the update reads; one rule for both scans,
            // or a plan could be built in its half form AND as asymmetric)
            if (j > i && !(v == a[j * ld + i])) {
                SR_REQUIRE(asym && !half, "this plan needs symmetric priors (prior %d, element %lld, %lld)", which, (long long)i,
                           (long long)j);
                *asym = true;
            }
            // (fp16-held matrices store value x 2^14: anything from 4 up, or not finite, leaves fp16's range and
            // turns into NaN behind the evidence factor)
            SR_REQUIRE(!half || (std::isfinite(v) && std::fabs(v) < 3.99f),
                       "storage_fp16 needs finite prior values below 4 in magnitude (element %lld, %lld)", (long long)i,
                       (long long)j);
        }
    return SIMRANK_OK;
}

bool prior_symmetric(const float* a, int64_t ld, int64_t n) {
    if (!a || ld < n) return true;                 // (a short ld is check_prior's to refuse)
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = i + 1; j < n; ++j)
            if (!(a[i * ld + j] == a[j * ld + i])) return false;
    return true;
}

// rows of (rowptr, col) taken in the order `ord`, columns renamed by `inv_cols`, sorted; duplicates refused
static int renamed(int64_t n_rows, const int32_t* rowptr, const int32_t* col, const float* scale,
                   const std::vector<int32_t>& ord, const std::vector<int32_t>& inv_cols, int64_t nnz, const char* what,
                   std::vector<int32_t>& rp, std::vector<int32_t>& cl, std::vector<float>& rs) {
    rp.assign((size_t)n_rows + 1, 0);
    cl.assign((size_t)std::max<int64_t>(1, nnz), 0);
    rs.assign((size_t)n_rows, 0.f);
    for (int64_t r = 0; r < n_rows; ++r) {
        const int32_t a = ord[(size_t)r];
        rp[(size_t)r + 1] = rp[(size_t)r] + (rowptr[a + 1] - rowptr[a]);
        rs[(size_t)r] = scale[a];
    }
    // the rows are independent once their places are known: renamed and sorted on a few threads (config 5: 1.6 M
    // entries, 25 of the 35 ms this file took; the first duplicate of the lowest row is the one reported)
    std::atomic<int64_t> dup_row{-1};
    auto work = [&](int64_t r0, int64_t r1) {
        for (int64_t r = r0; r < r1; ++r) {
            const int32_t a = ord[(size_t)r];
            const int32_t s = rowptr[a], e = rowptr[a + 1];
            int32_t* dst = cl.data() + rp[(size_t)r];
            for (int32_t j = s; j < e; ++j) dst[j - s] = inv_cols[(size_t)col[j]];
            std::sort(dst, dst + (e - s));
            for (int32_t j = 1; j < e - s; ++j)
                if (dst[j] == dst[j - 1]) {
                    int64_t seen = dup_row.load();
                    while ((seen < 0 || r < seen) && !dup_row.compare_exchange_weak(seen, r)) {}
                    break;
                }
        }
    };
    // (up to 16: what the bench box grants a process; 8 -> 16 threads: 6.8 -> ~5 ms of a config-4 plan_create)
    const int n_thr = nnz >= 100000 ? (int)std::max(8u, std::min(16u, std::thread::hardware_concurrency())) : 1;
    if (n_thr == 1) {
        work(0, n_rows);
    } else {
        // equal shares of the ENTRIES (the order is ascending in row length: equal shares of the rows would not balance)
        std::vector<std::thread> th;
        int64_t r0 = 0;
        for (int t = 0; t < n_thr; ++t) {
            const int64_t target = nnz * (t + 1) / n_thr;
            int64_t r1 = t + 1 == n_thr ? n_rows
                                        : int64_t(std::upper_bound(rp.begin() + r0, rp.end(), (int32_t)target) - rp.begin()) - 1;
            r1 = std::max(r0, std::min(n_rows, r1));
            th.emplace_back(work, r0, r1);
            r0 = r1;
        }
        for (std::thread& x : th) x.join();
    }
    SR_REQUIRE(dup_row.load() < 0, "duplicate entry in row %d%s", ord[(size_t)dup_row.load()], what);
    return SIMRANK_OK;
}

static void length_order(int64_t n, const int32_t* rowptr, bool reorder, std::vector<int32_t>& ord, std::vector<int32_t>& inv) {
    ord.resize((size_t)n);
    inv.resize((size_t)n);
    std::iota(ord.begin(), ord.end(), 0);
    if (reorder)
        std::stable_sort(ord.begin(), ord.end(),
                         [rowptr](int32_t x, int32_t y) { return rowptr[x + 1] - rowptr[x] < rowptr[y + 1] - rowptr[y]; });
    for (int64_t r = 0; r < n; ++r) inv[(size_t)ord[(size_t)r]] = (int32_t)r;
}

// the ascending order `ord` dealt to `deal` shards in runs of 128 (32) nodes when n divides evenly (driver.dealt_order)
static void deal_order(int64_t n, int32_t deal, std::vector<int32_t>& ord, std::vector<int32_t>& inv) {
    if (deal <= 1 || n % (32 * int64_t(deal))) return;
    const int64_t unit = n % (128 * int64_t(deal)) == 0 ? 128 : 32, per = n / (unit * deal);
    std::vector<int32_t> dealt((size_t)n);
    for (int64_t b = 0; b < per; ++b)
        for (int64_t w = 0; w < deal; ++w)
            for (int64_t i = 0; i < unit; ++i)
                dealt[(size_t)((w * per + b) * unit + i)] = ord[(size_t)((b * deal + w) * unit + i)];
    ord.swap(dealt);
    for (int64_t r = 0; r < n; ++r) inv[(size_t)ord[(size_t)r]] = (int32_t)r;
}

int plan_prepare(int64_t n, int64_t nnz, const int32_t* rowptr, const int32_t* col, const float* rowscale,
                 const simrank_plan_options* opt, PlanPrep* out) {
    SR_REQUIRE(opt && rowptr && rowscale && (col || nnz == 0) && n > 0 && nnz >= 0 && out, "bad plan arguments");
    SR_REQUIRE(n < (int64_t(1) << 24) - 16, "a plan holds at most 2^24 nodes");
    SR_REQUIRE(opt->dense_terms == 0 || opt->dense_terms == 1 || opt->dense_terms == 3, "dense_terms must be 0, 1 or 3");
    int rc = check_csr(n, n, nnz, rowptr, col);
    out->asym = false;
    if (!rc) rc = check_prior(opt->apriori, opt->ld_apriori, n, 1, opt->storage_fp16 != 0, &out->asym);
    if (rc) return rc;
    length_order(n, rowptr, opt->reorder != 0, out->ord, out->inv);
    return renamed(n, rowptr, col, rowscale, out->ord, out->inv, nnz, "", out->rp, out->cl, out->rs);
}

int shard_prepare(int64_t n, int64_t nnz, const int32_t* rowptr, const int32_t* col, const float* rowscale,
                  const float* apriori, int64_t ld_apriori, bool reorder, int32_t deal, PlanPrep* out, bool allow_asym) {
    SR_REQUIRE(rowptr && rowscale && (col || nnz == 0) && n > 0 && nnz >= 0 && out, "bad plan arguments");
    SR_REQUIRE(n < (int64_t(1) << 24) - 16, "a plan holds at most 2^24 nodes");
    int rc = check_csr(n, n, nnz, rowptr, col);
    out->asym = false;
    if (!rc) rc = check_prior(apriori, ld_apriori, n, 1, false, allow_asym ? &out->asym : nullptr);
    if (rc) return rc;
    length_order(n, rowptr, reorder, out->ord, out->inv);
    // tile t of the ascending order goes to shard t mod deal: every shard the same mix of short and long rows,
    // ascending inside (what halves every rank's gathers in the half-form leg 2)
    if (reorder) deal_order(n, deal, out->ord, out->inv);
    return renamed(n, rowptr, col, rowscale, out->ord, out->inv, nnz, "", out->rp, out->cl, out->rs);
}

// the two-matrix plan's host half for SHARDS: as biplan_prepare, each group's ascending order dealt to deal1 / deal2 shards
// (what the half-form leg 2 of that group wants; <= 1: plain ascending order)
int shard_biplan_prepare(int64_t n1, int64_t n2, int64_t nnz, const int32_t* rowptr12, const int32_t* col12,
                         const float* rowscale1, const float* rowscale2, const simrank_biplan_options* opt, int32_t deal1,
                         int32_t deal2, BiPlanPrep* out, bool allow_asym) {
    SR_REQUIRE(opt && rowptr12 && rowscale1 && rowscale2 && (col12 || nnz == 0) && n1 > 0 && n2 > 0 && nnz >= 0 && out,
               "bad plan arguments");
    SR_REQUIRE(n1 < (int64_t(1) << 24) - 16 && n2 < (int64_t(1) << 24) - 16, "a plan holds at most 2^24 nodes per group");
    int rc = check_csr(n1, n2, nnz, rowptr12, col12);
    out->asym = false;
    if (!rc) rc = check_prior(opt->apriori1, opt->ld_apriori1, n1, 1, false, allow_asym ? &out->asym : nullptr);
    if (!rc) rc = check_prior(opt->apriori2, opt->ld_apriori2, n2, 2, false, allow_asym ? &out->asym : nullptr);
    if (rc) return rc;
    std::vector<int32_t>& rowptr21 = out->rowptr21;
    std::vector<int32_t>& col21 = out->col21;
    rowptr21.assign((size_t)n2 + 1, 0);
    col21.assign((size_t)std::max<int64_t>(1, nnz), 0);
    for (int64_t j = 0; j < nnz; ++j) ++rowptr21[(size_t)col12[j] + 1];
    for (int64_t i = 0; i < n2; ++i) rowptr21[(size_t)i + 1] += rowptr21[(size_t)i];
    {
        std::vector<int32_t> fill(rowptr21.begin(), rowptr21.end() - 1);
        for (int64_t a = 0; a < n1; ++a)
            for (int32_t j = rowptr12[a]; j < rowptr12[a + 1]; ++j) col21[(size_t)fill[(size_t)col12[j]]++] = (int32_t)a;
    }
    const int64_t ns[2] = {n1, n2};
    const int32_t deals[2] = {deal1, deal2};
    const int32_t* rps[2] = {rowptr12, rowptr21.data()};
    const int32_t* cls[2] = {col12, col21.data()};
    const float* scales[2] = {rowscale1, rowscale2};
    for (int w = 0; w < 2; ++w) {
        length_order(ns[w], rps[w], opt->reorder != 0, out->ord[w], out->inv[w]);
        if (opt->reorder) deal_order(ns[w], deals[w], out->ord[w], out->inv[w]);
    }
    for (int w = 0; w < 2; ++w) {
        rc = renamed(ns[w], rps[w], cls[w], scales[w], out->ord[w], out->inv[w ^ 1], nnz, w ? " of group 2" : " of group 1",
                     out->rp[w], out->cl[w], out->rs[w]);
        if (rc) return rc;
    }
    return SIMRANK_OK;
}

int biplan_prepare(int64_t n1, int64_t n2, int64_t nnz, const int32_t* rowptr12, const int32_t* col12,
                   const float* rowscale1, const float* rowscale2, const simrank_biplan_options* opt, BiPlanPrep* out) {
    SR_REQUIRE(opt && rowptr12 && rowscale1 && rowscale2 && (col12 || nnz == 0) && n1 > 0 && n2 > 0 && nnz >= 0 && out,
               "bad plan arguments");
    SR_REQUIRE(n1 < (int64_t(1) << 24) - 16 && n2 < (int64_t(1) << 24) - 16, "a plan holds at most 2^24 nodes per group");
    int rc = check_csr(n1, n2, nnz, rowptr12, col12);
    out->asym = false;
    if (!rc) rc = check_prior(opt->apriori1, opt->ld_apriori1, n1, 1, false, &out->asym);
    if (!rc) rc = check_prior(opt->apriori2, opt->ld_apriori2, n2, 2, false, &out->asym);
    if (rc) return rc;
    // the group-2 pattern: the transpose
    std::vector<int32_t>& rowptr21 = out->rowptr21;
    std::vector<int32_t>& col21 = out->col21;
    rowptr21.assign((size_t)n2 + 1, 0);
    col21.assign((size_t)std::max<int64_t>(1, nnz), 0);
    for (int64_t j = 0; j < nnz; ++j) ++rowptr21[(size_t)col12[j] + 1];
    for (int64_t i = 0; i < n2; ++i) rowptr21[(size_t)i + 1] += rowptr21[(size_t)i];
    {
        std::vector<int32_t> fill(rowptr21.begin(), rowptr21.end() - 1);
        for (int64_t a = 0; a < n1; ++a)
            for (int32_t j = rowptr12[a]; j < rowptr12[a + 1]; ++j) col21[(size_t)fill[(size_t)col12[j]]++] = (int32_t)a;
    }
    const int64_t ns[2] = {n1, n2};
    const int32_t* rps[2] = {rowptr12, rowptr21.data()};
    const int32_t* cls[2] = {col12, col21.data()};
    const float* scales[2] = {rowscale1, rowscale2};
    for (int w = 0; w < 2; ++w) length_order(ns[w], rps[w], opt->reorder != 0, out->ord[w], out->inv[w]);
    for (int w = 0; w < 2; ++w) {
        rc = renamed(ns[w], rps[w], cls[w], scales[w], out->ord[w], out->inv[w ^ 1], nnz, w ? " of group 2" : " of group 1",
                     out->rp[w], out->cl[w], out->rs[w]);
        if (rc) return rc;
    }
    return SIMRANK_OK;
}

}  // namespace simrank
