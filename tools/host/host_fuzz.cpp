// Sanitizer run of the HOST side of libsimrank_hip (make -C simrank_amd/csrc asan): the library's
// api.hip / blockdense.hip / fused.hip / planprep.hip compiled for the host only with -DSIMRANK_HOST_ONLY
// -fsanitize=address,undefined, so that simrank_graph_create — argument validation, transposed pattern,
// balanced tiles and launch lists, dense sets (blockdense.hip), the one-launch plan (fused.hip) — runs on
// random graphs without a GPU, and the plans it builds are checked entry by entry against the CSR:
//   * tiles: consecutive, never across a multiple of 32, every row once; launch order a permutation;
//   * dense plan: covered + remainder = nnz, remainder rows ascending and a subset of the row;
//   * fused plan: every entry of every block is EITHER a pattern bit of the block's dense set OR an id in
//     exactly one lane group's stream (in ascending order inside its row), row ends and scales as recorded;
//   * (experiment build, -DSIMRANK_EXPERIMENT_FUSED2) the persistent plan of tools/experiments/fused2.hip likewise;
//   * plan inputs (planprep.hip): the node order is a permutation, the renamed patterns are the pattern, the
//     transpose is the transpose; malformed CSR arrays and priors are refused.
// Exit code 0 = all graphs passed.  No kernel is launched.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <random>
#include <set>
#include <vector>

#include "common.h"

#define CHECK(c, ...)                                         \
    do {                                                      \
        if (!(c)) {                                           \
            fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                     \
            fprintf(stderr, "\n");                            \
            exit(1);                                          \
        }                                                     \
    } while (0)

struct Csr {
    int64_t M, K;
    std::vector<int32_t> rowptr, col;
    std::vector<float> scale;
};

static Csr random_graph(std::mt19937& rng, int64_t M, int64_t K, double avg, int hubs, double p_hub, bool sorted_rows) {
    Csr g{M, K, {0}, {}, {}};
    std::vector<std::set<int32_t>> rows((size_t)M);
    std::poisson_distribution<int> deg(avg);
    std::uniform_int_distribution<int32_t> any(0, (int32_t)K - 1);
    std::uniform_real_distribution<double> u(0, 1);
    for (int64_t a = 0; a < M; ++a) {
        if (u(rng) < 0.08) continue;                          // empty row
        const int d = std::min<int64_t>(K, deg(rng));
        while ((int)rows[a].size() < d) rows[a].insert(any(rng));
        if (a >= M / 2)
            for (int h = 0; h < std::min<int64_t>(hubs, K); ++h)
                if (u(rng) < p_hub * double(a) / double(M)) rows[a].insert(h);
    }
    if (M > 3 && u(rng) < 0.3)                                // one very long row
        for (int32_t c = 0; c < K; c += 1 + int(u(rng) * 3)) rows[M / 3].insert(c);
    if (sorted_rows) std::stable_sort(rows.begin(), rows.end(), [](auto& x, auto& y) { return x.size() < y.size(); });
    for (int64_t a = 0; a < M; ++a) {
        for (int32_t c : rows[a]) g.col.push_back(c);
        g.rowptr.push_back((int32_t)g.col.size());
        g.scale.push_back(u(rng) < 0.05 ? 0.f : float(0.1 + u(rng)));
    }
    return g;
}

static void check_tiles(const simrank_graph* g, const Csr& c) {
    if (!g->n_tiles) return;
    const int32_t* t = g->tile_row0;
    CHECK(t[0] == 0 && t[g->n_tiles] == c.M, "tiles do not span the rows");
    for (int i = 0; i < g->n_tiles; ++i) {
        CHECK(t[i + 1] > t[i] && t[i + 1] - t[i] <= 32, "tile %d has %d rows", i, t[i + 1] - t[i]);
        CHECK((t[i] >> 5) == ((t[i + 1] - 1) >> 5), "tile %d crosses a multiple of 32", i);
    }
    const int groups = (g->n_tiles + 3) / 4;
    std::vector<int> seen((size_t)groups, 0);
    for (int i = 0; i < groups; ++i) {
        const int gi = t[g->n_tiles + 1 + i];
        CHECK(gi >= 0 && gi < groups && !seen[(size_t)gi]++, "launch order is not a permutation");
    }
    for (int i = 0; i < g->sym_blocks; ++i) {
        const int p = g->sym_map[2 * i], w = g->sym_map[2 * i + 1];
        CHECK(p == -1 || (p >= 0 && p < (c.M + 31) / 32 && w >= 0 && w < groups), "bad triangle launch entry");
    }
}

static void check_dense(const simrank_graph* g, const Csr& c) {
    const simrank_dense_plan* pl = g->dense;
    if (!pl) return;
    CHECK(pl->nnz_covered + pl->r_nnz == (int64_t)c.col.size(), "dense plan loses entries");
    for (int64_t a = 0; a < c.M; ++a) {
        const int32_t* r = pl->r_col + pl->r_rowptr[a];
        const int n = pl->r_rowptr[a + 1] - pl->r_rowptr[a];
        CHECK(n >= 0 && n <= c.rowptr[a + 1] - c.rowptr[a], "remainder row longer than the row");
        for (int j = 0; j < n; ++j) {
            CHECK(j == 0 || r[j] > r[j - 1], "remainder row not ascending");
            CHECK(std::binary_search(c.col.begin() + c.rowptr[a], c.col.begin() + c.rowptr[a + 1], r[j]),
                  "remainder entry not in the row");
        }
    }
}

static void check_fused(const simrank_graph* g, const Csr& c) {
    const simrank_fused_plan* pl = g->fused;
    if (!pl) return;
    const int64_t nblk = (c.M + 127) / 128;
    CHECK(pl->nnz_covered + pl->r_nnz == (int64_t)c.col.size(), "fused plan loses entries");
    std::vector<std::multiset<int32_t>> got((size_t)c.M);
    std::vector<int> seen_block((size_t)nblk, 0), owners((size_t)c.M, 0);
    const uint32_t* ab = reinterpret_cast<const uint32_t*>(pl->abits);
    const int32_t* gm = reinterpret_cast<const int32_t*>(pl->gmeta);
    std::map<int, int> units_of, quads_of, cslot_of, pbase_of;
    std::set<int> pslots_seen;
    for (int u = 0; u < pl->n_units; ++u) {
        const int32_t* un = pl->units + size_t(u) * 32;
        const int b0 = un[0], q0 = un[1], nq = un[2], k = un[3], nb = un[4], nsub = un[8];
        CHECK(b0 >= 0 && b0 + nsub <= nblk && nsub >= 1 && nsub <= 4 && k >= 0 && k < nb, "bad unit record");
        CHECK((nq > 0) == (un[7] != 0) && (nsub == 1 || nq == 0) && (nsub == 1 || nb == 1), "unit kinds mixed up");
        CHECK((nb > 1) == (un[5] >= 0) && (nb > 1) == (un[6] >= 0), "partial slots");
        if (nb > 1) {
            // every unit of a split block has a partial-sum slot of its own inside the plan's range, all of them share
            // the block's ticket — whatever the launch order put between them
            CHECK(un[5] < pl->n_pslots && un[6] < pl->n_cslots, "partial slot out of range");
            CHECK(pslots_seen.insert(un[5]).second, "two units share a partial slot");
            auto it = cslot_of.find(b0);
            if (it == cslot_of.end()) { cslot_of[b0] = un[6]; pbase_of[b0] = un[5] - k; }
            else CHECK(it->second == un[6] && pbase_of[b0] == un[5] - k, "units of a block disagree about their slots");
        }
        units_of[b0] += 1;
        // dense part: pattern bits -> entries
        for (int qd = q0; qd < q0 + nq; ++qd)
            for (int lane = 0; lane < 64; ++lane)
                for (int s = 0; s < 4; ++s) {
                    const uint32_t w = ab[(size_t(qd) * 64 + lane) * 4 + s];
                    for (int bit = 0; bit < 32; ++bit)
                        if (w >> bit & 1) {
                            const int t = bit >> 3, j = bit & 7, h = lane >> 5, m = lane & 31;
                            const int row = b0 * 128 + 32 * t + m, kk = 16 * s + 8 * h + j;
                            CHECK(row < c.M, "pattern bit past the last row");
                            const int32_t id = pl->ids16 ? (int32_t)pl->dcols16[size_t(qd) * 64 + kk]
                                                         : pl->dcols32[size_t(qd) * 64 + kk];
                            got[(size_t)row].insert(id);
                        }
                }
        // gather streams: an unsplit block's unit holds them all; of a split block (round 4) every unit holds the rows
        // recorded in ITS slot of the row records (matrix-core units: none), every row in exactly one unit
        const int gslot = un[29], split = un[30];
        CHECK((split != 0) == (nb > 1), "split flag of unit %d", u);
        CHECK(gslot >= 0, "row-record slot of unit %d", u);
        for (int sb = 0; sb < nsub; ++sb) {
            const int b = b0 + sb;
            if (!split) CHECK(!seen_block[(size_t)b]++, "block %d gathered twice", b);
            else if (k == nb - 1) ++seen_block[(size_t)b];
            for (int w = 0; w < 4; ++w) {
                const int32_t* wm = un + 9 + w * 5;
                const int r_lo = sb ? wm[sb] : 0, r_hi = wm[1 + sb];
                CHECK(r_hi >= r_lo, "round counts not monotone");
                for (int gg = 0; gg < 8; ++gg) {
                    const int32_t* m = gm + (((size_t(gslot + sb) * 4 + w) * 8) + gg) * 8;
                    int f = 0;
                    bool empty_seen = false;
                    for (int kr = 0; kr < 4; ++kr) {
                        const uint32_t packed = uint32_t(m[2 * kr]);
                        const int row = int(packed & 255u) == 255 ? -1 : int(packed & 255u);
                        const int end = (packed >> 8) == 0xFFFFFFu ? -1 : int(packed >> 8);
                        if (row < 0) continue;
                        CHECK(nq == 0 || !split, "a matrix-core unit of a split block owns a row");
                        CHECK(row < 128 && b * 128 + row < c.M, "row of a lane group out of range");
                        ++owners[size_t(b) * 128 + row];
                        float sc;
                        memcpy(&sc, &m[2 * kr + 1], 4);
                        CHECK(sc == c.scale[size_t(b) * 128 + row], "row scale");
                        if (end < 0) { empty_seen = true; continue; }
                        CHECK(!empty_seen && end > f, "rows without a remainder must come last");
                        int32_t prev = -1;
                        for (; f < end; ++f) {
                            CHECK(f / 8 < r_hi - r_lo, "row runs past the wave's rounds");
                            const size_t at = (size_t(wm[0]) + r_lo + f / 8) * 64 + gg * 8 + f % 8;
                            const int32_t id = pl->ids16 ? (pl->sids16[at] == 0xFFFF ? -1 : (int32_t)pl->sids16[at])
                                                         : pl->sids32[at];
                            CHECK(id > prev, "stream ids of a row not ascending (or a marker inside a row)");
                            prev = id;
                            got[size_t(b) * 128 + row].insert(id);
                        }
                    }
                    for (; f < 8 * (r_hi - r_lo); ++f) {      // past the lane group's stream: markers only
                        const size_t at = (size_t(wm[0]) + r_lo + f / 8) * 64 + gg * 8 + f % 8;
                        CHECK(pl->ids16 ? pl->sids16[at] == 0xFFFF : pl->sids32[at] < 0, "id past the end of a stream");
                    }
                }
            }
        }
    }
    for (int64_t a = 0; a < c.M; ++a) CHECK(owners[(size_t)a] == 1, "row %lld owned by %d lane groups", (long long)a, owners[(size_t)a]);
    for (int64_t b = 0; b < nblk; ++b) CHECK(seen_block[(size_t)b] == 1, "block %lld not gathered", (long long)b);
    for (int64_t a = 0; a < c.M; ++a) {
        std::multiset<int32_t> want(c.col.begin() + c.rowptr[a], c.col.begin() + c.rowptr[a + 1]);
        CHECK(got[(size_t)a] == want, "row %lld: plan entries differ from the CSR (%zu vs %zu)", (long long)a,
              got[(size_t)a].size(), want.size());
    }
}

#ifdef SIMRANK_EXPERIMENT_FUSED2
// the persistent one-launch plan (fused2.hip): every entry of every block is EITHER a pattern bit of exactly one
// piece's share of the block's dense set OR an id in exactly one lane group's stream of exactly one piece
static void check_fused2(const simrank_graph* g, const Csr& c) {
    const simrank_fused2_plan* pl = g->fused2;
    if (!pl) return;
    const int64_t nblk = (c.M + 127) / 128;
    CHECK(pl->nnz_covered + pl->r_nnz == (int64_t)c.col.size(), "fused2 plan loses entries");
    std::vector<std::multiset<int32_t>> got((size_t)c.M);
    std::vector<std::vector<int>> owner((size_t)c.M);
    const uint32_t* ab = reinterpret_cast<const uint32_t*>(pl->abits);
    const int32_t* gm = reinterpret_cast<const int32_t*>(pl->gmeta);
    std::map<int, int> pieces_seen;
    int expect_pslot = 0, expect_cslot = 0;
    for (int u = 0; u < pl->n_items; ++u) {
        const int32_t* it = pl->items + size_t(u) * 16;
        const int b0 = it[0], q0 = it[1], nq = it[2], gmi = it[3], np = it[4], ps = it[5], cs = it[6], k = it[7];
        CHECK(b0 >= 0 && b0 < nblk && np >= 1 && np <= 64 && k >= 0 && k < np && nq >= 0 && q0 >= 0 && q0 + nq <= pl->n_quads,
              "bad item record %d", u);
        CHECK(pieces_seen[b0]++ == k, "pieces of block %d out of order", b0);
        if (np > 1) {
            CHECK(ps == expect_pslot && cs == expect_cslot, "partial slot of item %d: %d (want %d), ticket %d (want %d)", u, ps,
                  expect_pslot, cs, expect_cslot);
            ++expect_pslot;
            if (k == np - 1) ++expect_cslot;
            if (k > 0) CHECK(pl->items[size_t(u - 1) * 16] == b0, "pieces of a block must be neighbours in the order");
        } else {
            CHECK(ps == -1 && cs == -1, "slots on an unsplit block");
        }
        for (int qd = q0; qd < q0 + nq; ++qd)
            for (int lane = 0; lane < 64; ++lane)
                for (int s = 0; s < 4; ++s) {
                    const uint32_t w = ab[(size_t(qd) * 64 + lane) * 4 + s];
                    for (int bit = 0; bit < 32; ++bit)
                        if (w >> bit & 1) {
                            const int t = bit >> 3, j = bit & 7, h = lane >> 5, m = lane & 31;
                            const int row = b0 * 128 + 32 * t + m, kk = 16 * s + 8 * h + j;
                            CHECK(row < c.M, "pattern bit past the last row");
                            const int32_t id = pl->ids16 ? (int32_t)pl->dcols16[size_t(qd) * 64 + kk]
                                                         : pl->dcols32[size_t(qd) * 64 + kk];
                            got[(size_t)row].insert(id);
                        }
                }
        for (int w = 0; w < 4; ++w) {
            const int round0 = it[8 + w], rounds = it[12 + w];
            CHECK(round0 >= 0 && rounds >= 0 && (rounds & 1) == 0, "rounds of a wave must be even");
            for (int gg = 0; gg < 8; ++gg) {
                const int32_t* m = gm + (((size_t(gmi) * 4 + w) * 8) + gg) * 8;
                int f = 0;
                bool empty_seen = false;
                for (int kr = 0; kr < 4; ++kr) {
                    const uint32_t packed = uint32_t(m[2 * kr]);
                    const int row = int(packed & 255u) == 255 ? -1 : int(packed & 255u);
                    const int end = (packed >> 8) == 0xFFFFFFu ? -1 : int(packed >> 8);
                    if (row < 0) continue;
                    CHECK(row < 128 && b0 * 128 + row < c.M, "row of a lane group out of range");
                    owner[size_t(b0) * 128 + row].push_back(u);
                    float sc;
                    memcpy(&sc, &m[2 * kr + 1], 4);
                    CHECK(sc == c.scale[size_t(b0) * 128 + row], "row scale");
                    if (end < 0) { empty_seen = true; continue; }
                    CHECK(!empty_seen && end > f, "rows without a remainder must come last");
                    int32_t prev = -1;
                    for (; f < end; ++f) {
                        CHECK(f / 8 < rounds, "row runs past the wave's rounds");
                        const size_t at = (size_t(round0) + f / 8) * 64 + gg * 8 + f % 8;
                        const int32_t id = pl->ids16 ? (pl->sids16[at] == 0xFFFF ? -1 : (int32_t)pl->sids16[at]) : pl->sids32[at];
                        CHECK(id > prev, "stream ids of a row not ascending (or a marker inside a row)");
                        prev = id;
                        got[size_t(b0) * 128 + row].insert(id);
                    }
                }
                for (; f < 8 * rounds; ++f) {
                    const size_t at = (size_t(round0) + f / 8) * 64 + gg * 8 + f % 8;
                    CHECK(pl->ids16 ? pl->sids16[at] == 0xFFFF : pl->sids32[at] < 0, "id past the end of a stream");
                }
            }
        }
    }
    CHECK(expect_pslot == pl->n_pslots && expect_cslot == pl->n_cslots, "slot counts");
    for (int64_t b = 0; b < nblk; ++b) {
        const int32_t np = pieces_seen[(int)b];
        CHECK(np >= 1, "block %lld has no item", (long long)b);
    }
    for (int64_t a = 0; a < c.M; ++a) {
        CHECK(owner[(size_t)a].size() == 1, "row %lld owned by %zu lane groups", (long long)a, owner[(size_t)a].size());
        std::multiset<int32_t> want(c.col.begin() + c.rowptr[a], c.col.begin() + c.rowptr[a + 1]);
        CHECK(got[(size_t)a] == want, "fused2 row %lld: plan entries differ from the CSR (%zu vs %zu)", (long long)a,
              got[(size_t)a].size(), want.size());
    }
}
#endif

// the host half of the plans (planprep.hip): well-formed input gives a permutation and the same pattern under new
// names; malformed input — rowptr not monotone or not spanning, columns out of range or repeated, a prior that is
// not symmetric (or not finite / too large for fp16-held matrices) — is SIMRANK_ERR_INVALID, never a wild read
static void fuzz_plan_inputs(std::mt19937& rng, const Csr& sq, const Csr& rect) {
    std::uniform_real_distribution<double> u(0, 1);
    simrank_plan_options opt{};
    opt.coef = 0.8f;
    opt.reorder = 1;
    simrank::PlanPrep pp;
    const int64_t n = sq.M, nnz = (int64_t)sq.col.size();
    CHECK(simrank::plan_prepare(n, nnz, sq.rowptr.data(), sq.col.data(), sq.scale.data(), &opt, &pp) == SIMRANK_OK,
          "plan_prepare: %s", simrank_last_error());
    {
        std::vector<int> seen((size_t)n, 0);
        for (int64_t r = 0; r < n; ++r) {
            CHECK(pp.ord[r] >= 0 && pp.ord[r] < n && !seen[pp.ord[r]]++ && pp.inv[pp.ord[r]] == r, "node order is not a permutation");
            CHECK(r == 0 || pp.rp[r] - pp.rp[r - 1] <= pp.rp[r + 1] - pp.rp[r], "rows not in ascending length");
            std::multiset<int32_t> want, have;
            for (int32_t j = sq.rowptr[pp.ord[r]]; j < sq.rowptr[pp.ord[r] + 1]; ++j) want.insert(pp.inv[sq.col[j]]);
            for (int32_t j = pp.rp[r]; j < pp.rp[r + 1]; ++j) have.insert(pp.cl[j]);
            CHECK(want == have && pp.rs[r] == sq.scale[pp.ord[r]], "renamed row %lld differs", (long long)r);
        }
    }
    if (nnz > 2) {
        std::vector<int32_t> rp = sq.rowptr, cl = sq.col;
        cl[nnz / 2] = u(rng) < 0.5 ? (int32_t)n : -1;
        CHECK(simrank::plan_prepare(n, nnz, rp.data(), cl.data(), sq.scale.data(), &opt, &pp) == SIMRANK_ERR_INVALID, "bad column accepted");
        cl = sq.col;
        int64_t a = 0;
        while (a < n && rp[a + 1] - rp[a] < 2) ++a;
        if (a < n) {
            cl[rp[a] + 1] = cl[rp[a]];                           // a repeated column
            CHECK(simrank::plan_prepare(n, nnz, rp.data(), cl.data(), sq.scale.data(), &opt, &pp) == SIMRANK_ERR_INVALID, "duplicate accepted");
            cl = sq.col;
            rp[a + 1] = rp[a] - 1 < 0 ? (int32_t)nnz + 5 : rp[a] - 1;     // not monotone / past the end
            CHECK(simrank::plan_prepare(n, nnz, rp.data(), cl.data(), sq.scale.data(), &opt, &pp) == SIMRANK_ERR_INVALID, "bad rowptr accepted");
            rp = sq.rowptr;
        }
        rp[n] = (int32_t)nnz - 1;
        CHECK(simrank::plan_prepare(n, nnz, rp.data(), cl.data(), sq.scale.data(), &opt, &pp) == SIMRANK_ERR_INVALID, "short rowptr accepted");
    }
    if (n <= 400) {
        std::vector<float> prior((size_t)n * n);
        for (int64_t i = 0; i < n; ++i)
            for (int64_t j = i; j < n; ++j) prior[i * n + j] = prior[j * n + i] = float(u(rng));
        opt.apriori = prior.data();
        opt.ld_apriori = n;
        opt.lbd = 0.3f;
        CHECK(simrank::plan_prepare(n, nnz, sq.rowptr.data(), sq.col.data(), sq.scale.data(), &opt, &pp) == SIMRANK_OK, "symmetric prior refused");
        opt.storage_fp16 = 1;
        CHECK(simrank::plan_prepare(n, nnz, sq.rowptr.data(), sq.col.data(), sq.scale.data(), &opt, &pp) == SIMRANK_OK, "small prior refused for fp16");
        prior[0] = 4.5f;
        CHECK(simrank::plan_prepare(n, nnz, sq.rowptr.data(), sq.col.data(), sq.scale.data(), &opt, &pp) == SIMRANK_ERR_INVALID, "prior of 4.5 accepted for fp16");
        prior[0] = std::numeric_limits<float>::infinity();
        CHECK(simrank::plan_prepare(n, nnz, sq.rowptr.data(), sq.col.data(), sq.scale.data(), &opt, &pp) == SIMRANK_ERR_INVALID, "infinite prior accepted for fp16");
        prior[0] = 0.5f;
        opt.storage_fp16 = 0;
        if (n > 1) {
            const float keep = prior[1];
            prior[1] = keep + 0.25f;
            // (an f32 plan takes it and says so: un-fused epilogue; fp16-held matrices and the sharded plans refuse it)
            CHECK(simrank::plan_prepare(n, nnz, sq.rowptr.data(), sq.col.data(), sq.scale.data(), &opt, &pp) == SIMRANK_OK && pp.asym,
                  "asymmetric prior not reported");
            opt.storage_fp16 = 1;
            CHECK(simrank::plan_prepare(n, nnz, sq.rowptr.data(), sq.col.data(), sq.scale.data(), &opt, &pp) == SIMRANK_ERR_INVALID,
                  "asymmetric prior accepted for fp16");
            opt.storage_fp16 = 0;
            simrank::PlanPrep sp2;
            CHECK(simrank::shard_prepare(n, nnz, sq.rowptr.data(), sq.col.data(), sq.scale.data(), prior.data(), n, true, 2, &sp2) == SIMRANK_ERR_INVALID,
                  "asymmetric prior accepted by the sharded plan");
            prior[1] = keep;
            CHECK(simrank::plan_prepare(n, nnz, sq.rowptr.data(), sq.col.data(), sq.scale.data(), &opt, &pp) == SIMRANK_OK && !pp.asym,
                  "symmetric prior reported as asymmetric");
            prior[1] = keep + 0.25f;
        }
        opt.ld_apriori = n - 1;
        CHECK(simrank::plan_prepare(n, nnz, sq.rowptr.data(), sq.col.data(), sq.scale.data(), &opt, &pp) == SIMRANK_ERR_INVALID, "short ld accepted");
    }
    // the sharded plan's node order: a permutation whatever the shard count; dealt in runs when n divides evenly, each
    // shard's rows ascending; malformed input refused
    for (int32_t deal : {1, 2, 3, 8}) {
        simrank::PlanPrep sp;
        CHECK(simrank::shard_prepare(n, nnz, sq.rowptr.data(), sq.col.data(), sq.scale.data(), nullptr, 0, true, deal, &sp) == SIMRANK_OK,
              "shard_prepare: %s", simrank_last_error());
        std::vector<int> seen((size_t)n, 0);
        for (int64_t r = 0; r < n; ++r)
            CHECK(sp.ord[r] >= 0 && sp.ord[r] < n && !seen[sp.ord[r]]++ && sp.inv[sp.ord[r]] == r, "dealt order is not a permutation");
        CHECK(sp.rp[n] == nnz, "dealt pattern loses entries");
        const bool dealt = deal > 1 && n % (32 * int64_t(deal)) == 0;
        const int64_t per = dealt ? n / deal : n;
        for (int64_t r = 1; r < n; ++r)
            CHECK(r % per == 0 || sp.rp[r] - sp.rp[r - 1] <= sp.rp[r + 1] - sp.rp[r], "rows of a shard not in ascending length");
        if (dealt) {                               // every shard holds the same mix: shard totals within one run of rows
            std::vector<int64_t> tot((size_t)deal, 0);
            for (int64_t r = 0; r < n; ++r) tot[(size_t)(r / per)] += sp.rp[r + 1] - sp.rp[r];
            for (int32_t w = 1; w < deal; ++w) CHECK(tot[w] >= tot[w - 1], "later shards must not be lighter (dealt ascending)");
        }
        if (nnz > 2) {
            std::vector<int32_t> cl = sq.col;
            cl[nnz / 2] = (int32_t)n;
            CHECK(simrank::shard_prepare(n, nnz, sq.rowptr.data(), cl.data(), sq.scale.data(), nullptr, 0, true, deal, &sp) == SIMRANK_ERR_INVALID,
                  "shard_prepare accepted a bad column");
        }
    }
    // bipartite: the transpose and both renamed patterns
    simrank_biplan_options bo{};
    bo.c1 = bo.c2 = 0.8f;
    bo.reorder = 1;
    simrank::BiPlanPrep bp;
    const int64_t n1 = rect.M, n2 = rect.K, z = (int64_t)rect.col.size();
    std::vector<float> s2((size_t)n2, 0.5f);
    CHECK(simrank::biplan_prepare(n1, n2, z, rect.rowptr.data(), rect.col.data(), rect.scale.data(), s2.data(), &bo, &bp) == SIMRANK_OK,
          "biplan_prepare: %s", simrank_last_error());
    CHECK(bp.rowptr21[n2] == z, "transpose loses entries");
    for (int64_t i = 0; i < n2; ++i)
        for (int32_t j = bp.rowptr21[i]; j < bp.rowptr21[i + 1]; ++j) {
            const int32_t a = bp.col21[j];
            CHECK(a >= 0 && a < n1 && std::binary_search(rect.col.begin() + rect.rowptr[a], rect.col.begin() + rect.rowptr[a + 1], (int32_t)i),
                  "transpose entry (%lld, %d) not in the pattern", (long long)i, a);
        }
    for (int w = 0; w < 2; ++w) {
        const int64_t nw = w ? n2 : n1, kw = w ? n1 : n2;
        CHECK((int64_t)bp.rp[w].size() == nw + 1 && bp.rp[w][nw] == z, "renamed pattern of group %d", w + 1);
        for (int64_t j = 0; j < z; ++j) CHECK(bp.cl[w][j] >= 0 && bp.cl[w][j] < kw, "renamed column out of range");
    }
    if (z > 2) {
        std::vector<int32_t> cl = rect.col;
        cl[z / 3] = (int32_t)n2 + 3;
        CHECK(simrank::biplan_prepare(n1, n2, z, rect.rowptr.data(), cl.data(), rect.scale.data(), s2.data(), &bo, &bp) == SIMRANK_ERR_INVALID,
              "bipartite: bad column accepted");
        std::vector<int32_t> rp = rect.rowptr;
        rp[n1 / 2 + 1] = (int32_t)z + 7;
        CHECK(simrank::biplan_prepare(n1, n2, z, rp.data(), rect.col.data(), rect.scale.data(), s2.data(), &bo, &bp) == SIMRANK_ERR_INVALID,
              "bipartite: bad rowptr accepted");
    }
}

int main(int argc, char** argv) {
    const int n_graphs = argc > 1 ? atoi(argv[1]) : 60;
    std::mt19937 rng(12345);
    std::uniform_real_distribution<double> u(0, 1);
    for (int it = 0; it < n_graphs; ++it) {
        const int64_t M = 1 + int64_t(u(rng) * (it % 7 == 0 ? 3000 : 700));
        const int64_t K = it % 3 == 0 ? M : 1 + int64_t(u(rng) * (it % 11 == 0 ? 70000 : 900));
        simrank_set_tuning("balance", it % 5 == 0 ? 0 : 1 + it % 4);
        simrank_set_tuning("dense_min", 2 + it % 4);
        simrank_set_tuning("dense_cols", it % 2 ? 16 : 64);
        simrank_set_tuning("fuse_min", it % 4 == 3 ? 0 : 2 + it % 3);          // (0: the set by quads that pay)
        simrank_set_tuning("fuse_pays", it % 8 == 3 ? -1 : 64 + 64 * (it % 5));
        simrank_set_tuning("fuse_steps", (it % 4) * 3);
        simrank_set_tuning("fuse_unit", it % 3 == 0 ? 4 : (it % 3 == 1 ? 64 : 1 << 20));
        simrank_set_tuning("fuse_rows", it % 4 == 0 ? 64 : (it % 4 == 1 ? 700 : 8192));
        simrank_set_tuning("fuse_group", 1 + it % 4);
        simrank_set_tuning("fuse_order", it % 5 == 1 ? 2 : (it % 5 == 3 ? 1 : 0));
#ifdef SIMRANK_EXPERIMENT_FUSED2
        simrank_set_tuning("fuse", it % 2 ? 2 : 1);
#endif
        simrank_set_tuning("fuse_cap", it % 3 == 0 ? 1000 : (it % 3 == 1 ? 3000 : 1 << 30));
        // (every 17th graph has no entries at all: plans of nothing but empty rows)
        Csr c = random_graph(rng, M, K, it % 17 == 16 ? 0.0 : 1 + u(rng) * 12, it % 17 == 16 ? 0 : int(u(rng) * 200),
                             u(rng), it % 2 == 0);
        if (it % 17 == 16) {
            c.col.clear();
            std::fill(c.rowptr.begin(), c.rowptr.end(), 0);
        }
        simrank_graph* g = nullptr;
        const int rc = simrank_graph_create(c.M, c.K, (int64_t)c.col.size(), c.rowptr.data(), c.col.data(),
                                            c.scale.data(), &g);
        CHECK(rc == SIMRANK_OK && g, "simrank_graph_create: %s", simrank_last_error());
        check_tiles(g, c);
        check_dense(g, c);
        check_fused(g, c);
#ifdef SIMRANK_EXPERIMENT_FUSED2
        check_fused2(g, c);
#endif
        if (c.M == c.K && it % 4 == 1) {
            Csr rect = random_graph(rng, 1 + int64_t(u(rng) * 300), 1 + int64_t(u(rng) * 300), 1 + u(rng) * 6, 20, u(rng), false);
            fuzz_plan_inputs(rng, c, rect);
        }
        int64_t steps = 0, cov = 0, rem = 0;
        simrank_graph_fused_stats(g, &steps, &cov, &rem);
        CHECK(cov + rem == (int64_t)c.col.size(), "fused stats");
        simrank_graph_destroy(g);
        // malformed input must be refused, not read out of bounds
        if (c.col.size() > 2) {
            std::vector<int32_t> bad = c.col;
            bad[bad.size() / 2] = (int32_t)c.K;               // out of range
            CHECK(simrank_graph_create(c.M, c.K, (int64_t)bad.size(), c.rowptr.data(), bad.data(), c.scale.data(), &g)
                      == SIMRANK_ERR_INVALID && !g, "out-of-range column accepted");
        }
    }
    // launch orders that put other units between the units of a split block (fuse_order > 0): graphs with many blocks, some
    // split into matrix-core and gather units, many not
    for (int order : {1, 2, 3}) {
        simrank_set_tuning("fuse", 1);
        simrank_set_tuning("fuse_min", 3);
        simrank_set_tuning("fuse_steps", 2);
        simrank_set_tuning("fuse_unit", 4);
        simrank_set_tuning("fuse_rows", 400);
        simrank_set_tuning("fuse_group", 3);
        simrank_set_tuning("fuse_order", order);
        Csr c = random_graph(rng, 4000 + 700 * order, 5000, 6.0, 300, 0.08, true);
        simrank_graph* g = nullptr;
        CHECK(simrank_graph_create(c.M, c.K, (int64_t)c.col.size(), c.rowptr.data(), c.col.data(), c.scale.data(), &g) == SIMRANK_OK && g,
              "simrank_graph_create: %s", simrank_last_error());
        check_fused(g, c);
        {
            // the same plan from ONE builder thread and from several: the arrays must be the same, entry for entry
            simrank_graph* g1 = nullptr;
            setenv("SIMRANK_BUILD_THREADS", order == 2 ? "5" : "1", 1);
            CHECK(simrank_graph_create(c.M, c.K, (int64_t)c.col.size(), c.rowptr.data(), c.col.data(), c.scale.data(), &g1) == SIMRANK_OK && g1,
                  "simrank_graph_create: %s", simrank_last_error());
            unsetenv("SIMRANK_BUILD_THREADS");
            const simrank_fused_plan* a = g->fused;
            const simrank_fused_plan* b = g1->fused;
            CHECK(a && b && a->n_units == b->n_units && a->n_quads == b->n_quads && a->n_steps == b->n_steps &&
                      a->nnz_covered == b->nnz_covered && a->r_nnz == b->r_nnz && a->n_pslots == b->n_pslots &&
                      a->n_cslots == b->n_cslots && a->ids16 == b->ids16, "threaded builder: other totals");
            CHECK(!memcmp(a->units, b->units, size_t(a->n_units) * 32 * 4), "threaded builder: other unit records");
            CHECK(!memcmp(a->abits, b->abits, size_t(a->n_quads) * 64 * 16), "threaded builder: other pattern bits");
            CHECK(a->ids16 ? !memcmp(a->dcols16, b->dcols16, size_t(a->n_quads) * 64 * 2)
                           : !memcmp(a->dcols32, b->dcols32, size_t(a->n_quads) * 64 * 4), "threaded builder: other set columns");
            // the streams: as many rounds as the last unit record says
            int64_t rounds = 0;
            for (int u = 0; u < a->n_units; ++u)
                for (int w = 0; w < 4; ++w) {
                    const int32_t* wm = a->units + size_t(u) * 32 + 9 + w * 5;
                    rounds = std::max<int64_t>(rounds, int64_t(wm[0]) + std::max({wm[1], wm[2], wm[3], wm[4]}));
                }
            CHECK(a->ids16 ? !memcmp(a->sids16, b->sids16, size_t(rounds) * 64 * 2)
                           : !memcmp(a->sids32, b->sids32, size_t(rounds) * 64 * 4), "threaded builder: other id streams");
            simrank_graph_destroy(g1);
        }
        simrank_graph_destroy(g);
        if (order == 1) {
            // a square graph large enough for the threaded half of plan_prepare (renamed rows on eight threads): the same
            // checks as the small ones, and a duplicate in a late row is found and named
            Csr sq = random_graph(rng, 5000, 5000, 20.0, 300, 0.08, true);
            Csr rect = random_graph(rng, 50, 40, 3.0, 5, 0.1, false);
            CHECK((int64_t)sq.col.size() >= 100000, "the large plan graph has only %lld entries", (long long)sq.col.size());
            fuzz_plan_inputs(rng, sq, rect);
        }
    }
    {
        // exactly 65536 operand rows, the last column referenced: the id 0xFFFF must not be mistaken for the empty-slot marker
        // of 16-bit streams (such a graph takes 32-bit ids)
        simrank_set_tuning("fuse", 1);
        simrank_set_tuning("fuse_min", 3);
        simrank_set_tuning("fuse_steps", 8);
        simrank_set_tuning("fuse_unit", 48);
        simrank_set_tuning("fuse_rows", 8192);
        simrank_set_tuning("fuse_order", 0);
        Csr c = random_graph(rng, 700, 65536, 5.0, 40, 0.2, true);
        // (rows are ascending: the last column can simply be appended)
        Csr d{c.M, c.K, {0}, {}, c.scale};
        for (int64_t a = 0; a < c.M; ++a) {
            d.col.insert(d.col.end(), c.col.begin() + c.rowptr[a], c.col.begin() + c.rowptr[a + 1]);
            if (a % 37 == 5 && (d.col.size() == (size_t)d.rowptr.back() || d.col.back() != 65535)) d.col.push_back(65535);
            d.rowptr.push_back((int32_t)d.col.size());
        }
        simrank_graph* g = nullptr;
        CHECK(simrank_graph_create(d.M, d.K, (int64_t)d.col.size(), d.rowptr.data(), d.col.data(), d.scale.data(), &g) == SIMRANK_OK && g,
              "simrank_graph_create: %s", simrank_last_error());
        check_fused(g, d);
        simrank_graph_destroy(g);
    }
    printf("host_fuzz: %d graphs passed\n", n_graphs);
    return 0;
}
