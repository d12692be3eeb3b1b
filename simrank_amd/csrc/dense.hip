// Dense leg for graphs whose normalised adjacency really is dense (BTS flights 84/90
// pairs, K(10,10), MovieLens-like 4.5 %): W densified once, then
//     C[M x N] = epilogue( A[M x K] . B[N x K]^T )
// on the exact-f32 matrix instruction v_mfma_f32_32x32x2_f32 (gfx950 has no xf32/TF32;
// this instruction is bit-for-bit an ordered fmaf chain).  Both products of an update are
// of this NT shape because S is symmetric:  T = Wd . S = Wd . S^T,  S' = T . Wd^T.
//
// Tiling: 128 x 128 x 32 per 256-thread workgroup, 4 waves as 2 x 2, each wave 64 x 64 =
// 2 x 2 MFMA tiles (64 accumulator registers).  Operand tiles are register-staged
// (float4 global loads issued one K-tile ahead) into LDS rows padded to 33 floats, which
// makes both the staging stores and the per-lane fragment reads (lane = row, two K
// columns per wave) bank-conflict free.  Workgroups are dealt to XCDs in 8 x 8 super-tiles
// so the 64 resident workgroups of an XCD re-use 8 A panels and 8 B panels from its L2.
// The epilogue is the same fused one as the sparse leg (coef, evidence, prior, diagonal,
// convergence count).
#include <algorithm>

#include "common.h"

namespace simrank {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    int64_t lda, ldb, ldc;
    int M, N, K;
    int nbm, nbn;
    int has_ep;
    float coef, lbd;
    const uint8_t* ev;
    int64_t ld_ev;
    const float* ap;
    int64_t ld_ap;
    const float* prev;
    int64_t ld_prev;
    double eps;
    unsigned long long* n_changed;
    int64_t diag_col0;
    int set_diag;
};

constexpr int BM = 128, BN = 128, BK = 32, LDT = BK + 1;

__device__ __forceinline__ float4 load_k4(const float* base, int64_t ld, int row, int nrows,
                                           int k, int K) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < nrows) {
        const float* p = base + int64_t(row) * ld + k;
        if (k + 3 < K) {
            v = *reinterpret_cast<const float4*>(p);
        } else {
            if (k < K) v.x = p[0];
            if (k + 1 < K) v.y = p[1];
            if (k + 2 < K) v.z = p[2];
        }
    }
    return v;
}

__global__ __launch_bounds__(256) void gemm_nt_mfma_kernel(const GemmArgs p) {
    __shared__ float lds[2][2][BM * LDT];  // [buffer][A|B][row * LDT + k]

    // XCD-aware tile order (speed only): blocks equal mod 8 share an L2
    int bm, bn;
    {
        const int nwg = p.nbm * p.nbn;
        const int bid = blockIdx.x;
        int t = bid;
        if (nwg % 8 == 0) t = (bid % 8) * (nwg / 8) + bid / 8;
        if (p.nbm % 8 == 0 && p.nbn % 8 == 0) {
            const int sup = t / 64, in = t % 64;
            const int sup_n = p.nbn / 8;
            bm = (sup / sup_n) * 8 + in / 8;
            bn = (sup % sup_n) * 8 + in % 8;
        } else {
            bm = t / p.nbn;
            bn = t % p.nbn;
        }
    }
    const int m0 = bm * BM, n0 = bn * BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // staging map: thread -> (row = tid/8 + 32*i, k = (tid%8)*4), i = 0..3
    const int srow = tid >> 3;
    const int sk = (tid & 7) * 4;
    const float* Ab = p.A + int64_t(m0) * p.lda;
    const float* Bb = p.B + int64_t(n0) * p.ldb;
    const int mrows = p.M - m0, nrows = p.N - n0;

    float4 ra[4], rb[4];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = load_k4(Ab, p.lda, srow + 32 * i, mrows, kt * BK + sk, p.K);
            rb[i] = load_k4(Bb, p.ldb, srow + 32 * i, nrows, kt * BK + sk, p.K);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float* a = &lds[buf][0][(srow + 32 * i) * LDT + sk];
            float* b = &lds[buf][1][(srow + 32 * i) * LDT + sk];
            a[0] = ra[i].x; a[1] = ra[i].y; a[2] = ra[i].z; a[3] = ra[i].w;
            b[0] = rb[i].x; b[1] = rb[i].y; b[2] = rb[i].z; b[3] = rb[i].w;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = (p.K + BK - 1) / BK;
    fetch(0);
    stash(0);
    __syncthreads();
    const int fr = lane & 31;  // fragment row (A: m, B: n)
    const int fk = lane >> 5;  // fragment k within a 2-deep step
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) fetch(kt + 1);
        const float* as = &lds[buf][0][(wm * 64 + fr) * LDT + fk];
        const float* bs = &lds[buf][1][(wn * 64 + fr) * LDT + fk];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a0 = as[kk], a1 = as[32 * LDT + kk];
            const float b0 = bs[kk], b1 = bs[32 * LDT + kk];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nkt) stash(buf ^ 1);
        __syncthreads();
    }

    // epilogue.  C/D map of the 32x32 tile: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    unsigned changed = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t n = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M && n < p.N) {
                    float v = acc[i][j][r];
                    if (p.has_ep) {
                        v *= p.coef;
                        if (p.ev)
                            v *= 1.0f - __builtin_ldexpf(1.0f, -int(p.ev[m * p.ld_ev + n]));
                        if (p.ap) v = (1.0f - p.lbd) * v + p.lbd * p.ap[m * p.ld_ap + n];
                        if (p.set_diag && m == p.diag_col0 + n) v = 1.0f;
                        if (p.prev)
                            changed += fabs(double(v) - double(p.prev[m * p.ld_prev + n])) > p.eps
                                           ? 1u : 0u;
                    }
                    p.C[m * p.ldc + n] = v;
                }
            }
        }
    }
    if (p.has_ep && p.prev) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) changed += __shfl_down(changed, off);
        if (lane == 0 && changed)
                atomicAdd(p.n_changed + ((blockIdx.x * 4u + (threadIdx.x >> 6)) * 7u) % SIMRANK_CHANGED_SLOTS,
                          (unsigned long long)changed);
    }
}

__global__ __launch_bounds__(256) void densify_kernel(const int32_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ col,
                                                      const float* __restrict__ rowscale,
                                                      int64_t M, float* Wd, int64_t ld) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * int64_t(blockDim.x) + threadIdx.x) >> 6;
    const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
    for (int64_t a = wave; a < M; a += nwaves) {
        const float v = rowscale[a];
        for (int j = rowptr[a] + lane; j < rowptr[a + 1]; j += 64) Wd[a * ld + col[j]] = v;
    }
}

}  // namespace simrank

using namespace simrank;

extern "C" {

int simrank_graph_densify(const simrank_graph* g, float* Wd, int64_t ld, void* stream) {
    SR_REQUIRE(g && Wd && ld >= g->n_cols, "bad densify arguments");
    hipStream_t st = as_stream(stream);
    SR_HIP(hipMemset2DAsync(Wd, size_t(ld) * 4, 0, size_t(g->n_cols) * 4, size_t(g->n_rows), st));
    const int grid = (int)std::min<int64_t>((g->n_rows + 3) / 4, 256 * 8);
    hipLaunchKernelGGL(densify_kernel, dim3(grid), dim3(256), 0, st, g->rowptr, g->col,
                       g->rowscale, g->n_rows, Wd, ld);
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

int simrank_gemm_nt(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                    int64_t ldb, float* C, int64_t ldc, const simrank_epilogue* ep, void* stream) {
    SR_REQUIRE(A && B && C, "NULL matrix");
    SR_REQUIRE(M > 0 && N > 0 && K > 0 && M < (1 << 30) && N < (1 << 30) && K < (1 << 30),
               "bad GEMM shape");
    SR_REQUIRE(lda >= K && ldb >= K && ldc >= N, "leading dimension too small");
    SR_REQUIRE(aligned16(A) && aligned16(B) && lda % 4 == 0 && ldb % 4 == 0,
               "simrank_gemm_nt needs 16-byte aligned A, B and lda, ldb multiples of 4");
    GemmArgs a{};
    a.A = A; a.B = B; a.C = C;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.nbm = int((M + BM - 1) / BM);
    a.nbn = int((N + BN - 1) / BN);
    hipStream_t st = as_stream(stream);
    if (ep) {
        a.has_ep = 1;
        a.coef = ep->coef; a.lbd = ep->lbd;
        a.ev = ep->evidence; a.ld_ev = ep->ld_evidence;
        a.ap = ep->apriori; a.ld_ap = ep->ld_apriori;
        a.prev = ep->previous; a.ld_prev = ep->ld_previous;
        a.eps = ep->eps; a.n_changed = ep->n_changed;
        a.diag_col0 = ep->diag_col0; a.set_diag = ep->set_diag;
        SR_REQUIRE(!a.prev || a.n_changed, "previous needs a counter");
        if (a.prev) SR_HIP(hipMemsetAsync(a.n_changed, 0, sizeof(unsigned long long) * SIMRANK_CHANGED_SLOTS, st));
    }
    const int64_t grid = int64_t(a.nbm) * a.nbn;
    SR_REQUIRE(grid < (int64_t(1) << 31), "grid too large");
    hipLaunchKernelGGL(gemm_nt_mfma_kernel, dim3((unsigned)grid), dim3(256), 0, st, a);
    SR_HIP(hipGetLastError());
    return SIMRANK_OK;
}

}  // extern "C"
