// Hand-back of a similarity matrix (SimRank.py:141, :303: `pd.DataFrame(new_S, ...)` — the float64 N x N frame the
// reference returns) out of the solver's layout and node order: dst[i][j] = (double)src[idx[i]][idx[j]].
//
// PIPELINED (both forms): the result is cut into row bands; band b is brought into the caller's order and packed on the
// device (one simrank_permute_layout per band), copied into a pinned slab, and widened to float64 by a crew of host
// threads while band b + 1 travels — three device and three pinned slabs.  Round 4 permuted the WHOLE matrix before the
// first byte moved.  The crew is sized by the CPUs the process may really use (cgroup quota: 16 on the bench box — 64
// threads took 3 x as long as 16 for the same work); its threads meet on atomics, a condition variable only when one of
// them really has to wait.
//
// FULL form (default): band = rows [r0, r1) x all columns; every thread widens whole rows.  PCIe-bound on the bench box:
// 4.3 GB at 57 GB/s = 75 ms of a ~85 ms hand-back at N = 32768; the host side needs 59 ms on 16 threads.
//
// SYMMETRIC form (mode 1 / SIMRANK_SYM_HANDBACK=1): only the elements on or above the diagonal
// cross PCIe.  A symmetric update in its upper-triangle form (spmm.hip kSym, half.hip SYM) computes the 32 x 32 tiles on
// or above the diagonal and stores each strictly-upper tile a second time, transposed: outside the 32 x 32 blocks ON the
// diagonal of the solver's order the iterate is BITWISE symmetric (inside them both triangles are computed, each with
// its own summation order), and a renaming applied to rows and columns alike keeps it so.
//   1. a kernel CHECKS the premise — every tile pair (I, J), I < J, against its mirror image (one pass at memory rate,
//      1.5 ms); a matrix that fails (full-form leg 2, fewer than 64 nodes, asymmetric iterates) takes the full form: the
//      result never depends on the premise;
//   2. band b is the trapezoid [r0, r1) x [r0, N); the crew widens it twice: dst[r][c] and, for c > r, dst[c][r].  A
//      thread owns GROUPS OF RESULT ROWS, the same ones in every band (widen_band);
//   3. the diagonal blocks of the SOURCE order (N x 32 floats) come over by themselves and are written where the
//      caller's order puts them, both triangles.
// Bit for bit the full form's result (tests/test_gpu_product_path.py, every element at N = 32768) — and on the bench box
// SLOWER: half the PCIe bytes (38 ms), but the mirrored writes (64-double runs into rows 256 KiB apart, read down the
// columns of the slab) cost the 16 host threads 80 - 120 ms against 59 for whole rows (tools/micro/widen_bench.cpp,
// profiles/r05_widen_bench*.log): 126 ms against 85.  It pays where the host side is not the bottleneck (a CPU share of
// several times 16); hence opt-in.
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include <hip/hip_fp16.h>

#include "common.h"

namespace simrank {
namespace {

int64_t slab_bytes() {                                // per pinned / device slab
    static const int64_t v = [] {
        const char* e = std::getenv("SIMRANK_HANDBACK_SLAB_MB");
        return (e && std::atoll(e) > 0 ? std::atoll(e) : 128) << 20;
    }();
    return v;
}
constexpr int64_t kTile = 64;                         // the host transposes 64 x 64 blocks (16 KiB of floats: L1)
constexpr int64_t kGroup = 16;                        // result rows a host thread owns together (one 64-byte line of every slab row)
constexpr int64_t kDiag = 32;                         // diagonal blocks of the source that are handed over whole

inline void store_nt(double* p, double v) { __builtin_nontemporal_store(v, p); }

__device__ __forceinline__ int64_t at(int64_t r, int64_t c, int64_t ld, int64_t rows_pad) {
    return rows_pad ? ((c >> 5) * rows_pad + r) * 32 + (c & 31) : r * ld + c;
}

// flag |= 1 when some src[a][b] != src[b][a] (compared as bits) with a, b in different 32-blocks.  One workgroup per
// tile pair I < J, through LDS so that both tiles are read along their rows.
__global__ __launch_bounds__(256) void mirror_check_kernel(const uint32_t* __restrict__ src, int64_t ld, int64_t rows_pad,
                                                           int64_t n, int64_t n_tiles, int32_t* __restrict__ flag) {
    __shared__ uint32_t tile[32][33];
    // pair index -> (I, J), I < J: row J of the strict lower triangle holds J entries
    const int64_t p = blockIdx.x;
    int64_t J = (int64_t)((1.0 + sqrt(1.0 + 8.0 * (double)p)) * 0.5);
    while (J * (J - 1) / 2 > p) --J;
    while ((J + 1) * J / 2 <= p) ++J;
    const int64_t I = p - J * (J - 1) / 2;
    if (J >= n_tiles) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 8 rows per pass
    for (int r = ty; r < 32; r += 8) {
        const int64_t a = I * 32 + r, b = J * 32 + tx;
        tile[r][tx] = (a < n && b < n) ? src[at(a, b, ld, rows_pad)] : 0u;
    }
    __syncthreads();
    bool bad = false;
    for (int r = ty; r < 32; r += 8) {
        const int64_t b = J * 32 + r, a = I * 32 + tx;               // element (b, a) against (a, b) = tile[tx][r]
        if (a < n && b < n) bad |= src[at(b, a, ld, rows_pad)] != tile[tx][r];
    }
    if (bad) atomicOr(flag, 1);
}

// out[a][t] = src[a][32 (a / 32) + t]: the diagonal blocks, compact
__global__ __launch_bounds__(256) void diag_blocks_kernel(const float* __restrict__ src, int64_t ld, int64_t rows_pad, int64_t n,
                                                          float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t a = i >> 5, t = i & 31;
    if (a >= n) return;
    const int64_t b = (a >> 5) * 32 + t;
    out[i] = b < n ? src[at(a, b, ld, rows_pad)] : 0.f;
}

struct Band {
    int64_t r0 = 0, nr = 0;          // rows [r0, r0 + nr) of the result
    int64_t pitch = 0;               // floats per packed row: the width rounded up to an ODD number of 64-byte lines — a
                                     // power-of-two pitch sends the 64 rows of a block to ONE set of the L1 and L2
    int64_t c0 = 0;                  // first column: r0 (symmetric form: the trapezoid) or 0 (full form)
};

int64_t odd_lines(int64_t w) {
    int64_t lines = (w + 15) / 16;
    if (!(lines & 1)) ++lines;
    return lines * 16;
}

std::vector<Band> cut_bands(int64_t n, bool sym = true) {
    std::vector<Band> bands;
    for (int64_t r0 = 0; r0 < n;) {
        const int64_t c0 = sym ? r0 : 0;
        const int64_t pitch = odd_lines(n - c0);
        int64_t nr = std::max<int64_t>(kTile, (slab_bytes() / 4 / pitch) / kTile * kTile);
        nr = std::min(nr, n - r0);
        bands.push_back({r0, nr, pitch, c0});
        r0 += nr;
    }
    return bands;
}

// the full form: whole rows of the band, a contiguous share per thread
template <bool NT>
void widen_rows(const float* slab, const Band& b, int64_t n, double* dst, int64_t ld, int64_t t, int64_t nt) {
    for (int64_t j = b.nr * t / nt; j < b.nr * (t + 1) / nt; ++j) {
        const float* s = slab + j * b.pitch;
        double* d = dst + (b.r0 + j) * ld;
        if (NT)
            for (int64_t c = 0; c < n; ++c) store_nt(d + c, (double)s[c]);
        else
            for (int64_t c = 0; c < n; ++c) d[c] = (double)s[c];
    }
}

constexpr int kSlabs = 3;            // bands in flight: one being widened, one on the wire, one queued behind it (with two the copy
                                     // engine idled between a band's last host thread and the next band's packing kernel: ~0.4 ms
                                     // x 35 bands at N = 32768)
struct Slabs {                       // kept per device for the life of the process (pinning memory is slow)
    float* pin[kSlabs] = {};
    hipEvent_t done[kSlabs] = {};                  // band in its pinned slab
    hipEvent_t packed[kSlabs] = {};                // band packed in its device slab
    hipEvent_t begin = nullptr;                    // the caller's stream has produced the source
    hipStream_t side = nullptr;                    // the packing kernels run here, beside the copies on the caller's stream
    size_t cap = 0;
};
std::mutex g_slab_mutex;
Slabs g_slabs[16];

// one thread's share of a band.  MIRRORED part: the GROUPS of kGroup consecutive result rows g = t, t + nt, ... (the same
// rows in every band: a thread always writes the same pages of the frame); a row R below the band's first row receives
// dst[R][r0 + i] = slab[i][R - r0] for the band rows i above it — read down a column of the slab, kTile band rows at a time
// (16 x 64 floats: whole 64-byte lines, which stay in the L1), the tiles of different threads started at different columns
// (all of them walking the same columns of rows 256 KiB apart would queue on the same memory channels).  DIRECT part: the
// band's own rows, dealt one by one: dst[R][R .. n) = slab[R - r0][R - r0 .. w), one long run each.
// (tools/micro/widen_bench.cpp on the bench box: 76 - 81 ms for N = 32768 on 16 threads in this form, 94 - 100 in the
// first cut — groups of 8, direct rows with their group; the box gives a process 16 CPUs' worth of time: 64 threads take
// 200 - 270 ms for the same work, which is what the first version of this file measured.)
void widen_band(const float* slab, const Band& b, int64_t n, double* dst, int64_t ld, int64_t t, int64_t nt) {
    const int64_t w = n - b.r0;
    const int64_t g0 = b.r0 / kGroup, g_end = (n + kGroup - 1) / kGroup;       // (r0 is a multiple of kTile)
    for (int64_t g = g0 + ((t - g0 % nt) % nt + nt) % nt; g < g_end; g += nt) {
        const int64_t j0 = g * kGroup - b.r0, j1 = std::min(n, (g + 1) * kGroup) - b.r0;
        const int64_t imax = std::min(b.nr, j1 - 1);                              // band rows above the group's last row
        const int64_t chunks = (imax + kTile - 1) / kTile;
        const int64_t first = chunks > 0 ? (t * 3 + g) % chunks : 0;
        for (int64_t k = 0; k < chunks; ++k) {
            const int64_t ic = ((first + k) % chunks) * kTile, ie = std::min(imax, ic + kTile);
            for (int64_t j = j0; j < j1; ++j) {
                double* d = dst + (b.r0 + j) * ld + b.r0;
                const float* s = slab + j;
                const int64_t iend = std::min(ie, j);
                for (int64_t i = ic; i < iend; ++i) store_nt(d + i, (double)s[i * b.pitch]);
            }
        }
    }
    for (int64_t j = t; j < b.nr; j += nt) {
        const float* s = slab + j * b.pitch;
        double* d = dst + (b.r0 + j) * ld + b.r0;
        for (int64_t c = j; c < w; ++c) store_nt(d + c, (double)s[c]);
    }
}

// CPUs this process may actually use: the affinity mask, cut by the cgroup's CPU quota (a container with 16 CPUs' worth
// of time on a 256-thread host is what the bench box is: threads beyond the quota are throttled, not run)
int64_t cpu_share() {
    int64_t n = std::max<int64_t>(1, (int64_t)std::thread::hardware_concurrency());
    for (const char* path : {"/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"}) {
        FILE* f = std::fopen(path, "r");
        if (!f) continue;
        char a[64] = "", bq[64] = "";
        const int got = std::fscanf(f, "%63s %63s", a, bq);
        std::fclose(f);
        if (got < 1 || !std::strcmp(a, "max")) continue;
        double quota = std::atof(a), period = got == 2 ? std::atof(bq) : 0;
        if (period <= 0) {                                   // cgroup v1: the period is in its own file
            period = 100000;
            if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (std::fscanf(g, "%63s", bq) == 1) period = std::atof(bq);
                std::fclose(g);
            }
        }
        if (quota > 0 && period > 0) n = std::min<int64_t>(n, std::max<int64_t>(1, (int64_t)(quota / period + 0.5)));
        break;
    }
    if (const char* e = std::getenv("SIMRANK_HOST_THREADS")) n = std::max<int64_t>(1, std::atoll(e));
    return n;
}

}  // namespace

#ifndef SIMRANK_HOST_ONLY
// dst[i][j] = src[idx[i]][col_idx[j]] out of fp16-held 64-column panels, widened
__global__ __launch_bounds__(256) void rows_h16_kernel(const uint16_t* __restrict__ src, int64_t rows_pad, float* __restrict__ dst,
                                                       int64_t ld, int64_t n_rows, int64_t n_cols, const int32_t* __restrict__ row_idx,
                                                       const int32_t* __restrict__ col_idx, float inv_scale) {
    for (int64_t i = blockIdx.x; i < n_rows; i += gridDim.x) {
        const int64_t r = row_idx[i];
        for (int64_t j = threadIdx.x; j < n_cols; j += blockDim.x) {
            const int64_t c = col_idx[j];
            const uint16_t h = src[((c >> 6) * rows_pad + r) * 64 + (c & 63)];
            dst[i * ld + j] = __half2float(__ushort_as_half(h)) * inv_scale;
        }
    }
}
#endif

int rows_to_host(const void* S, int64_t rows_pad, int64_t n, const int32_t* inv_dev, const int32_t* rows, int32_t n_rows,
                 float* dst, int64_t ld, int elem, float scale, hipStream_t stream) {
#ifdef SIMRANK_HOST_ONLY
    (void)S; (void)rows_pad; (void)n; (void)inv_dev; (void)rows; (void)n_rows; (void)dst; (void)ld; (void)elem; (void)scale; (void)stream;
    SR_REQUIRE(false, "host-only build: no device");
#else
    SR_REQUIRE(S && inv_dev && rows && dst && n_rows > 0 && ld >= n, "bad row arguments");
    for (int32_t i = 0; i < n_rows; ++i) SR_REQUIRE(rows[i] >= 0 && rows[i] < n, "row %d out of range", rows[i]);
    std::vector<int32_t> inv((size_t)n), pos((size_t)n_rows);
    SR_HIP(hipMemcpyAsync(inv.data(), inv_dev, size_t(n) * 4, hipMemcpyDeviceToHost, stream));
    SR_HIP(hipStreamSynchronize(stream));
    for (int32_t i = 0; i < n_rows; ++i) pos[(size_t)i] = inv[(size_t)rows[i]];
    const int64_t lds = (n + 3) / 4 * 4;
    int32_t* pos_dev = nullptr;
    float* out_dev = nullptr;
    hipError_t e = pool_hip_alloc((void**)&pos_dev, size_t(n_rows) * 4);
    if (e == hipSuccess) e = pool_hip_alloc((void**)&out_dev, size_t(n_rows) * size_t(lds) * 4);
    int rc = SIMRANK_OK;
    if (e == hipSuccess) e = hipMemcpyAsync(pos_dev, pos.data(), size_t(n_rows) * 4, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) {
        if (elem == 4) {
            rc = simrank_permute_layout(S, 32, rows_pad, out_dev, lds, 0, n_rows, n, pos_dev, inv_dev, 4, stream);
        } else {
            hipLaunchKernelGGL(rows_h16_kernel, dim3((unsigned)std::min<int64_t>(n_rows, 4096)), dim3(256), 0, stream,
                               (const uint16_t*)S, rows_pad, out_dev, lds, (int64_t)n_rows, n, pos_dev, inv_dev, 1.0f / scale);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess && !rc)
        e = hipMemcpy2DAsync(dst, size_t(ld) * 4, out_dev, size_t(lds) * 4, size_t(n) * 4, size_t(n_rows), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    else (void)hipStreamSynchronize(stream);
    (void)pool_free(pos_dev); (void)pool_free(out_dev);
    if (e != hipSuccess) {
        set_error("rows hand-back: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        return SIMRANK_ERR_HIP;
    }
    return rc;
#endif
}

// The pinned slabs (3 x 128 MiB and more per device) are kept for the life of the process because pinning is slow;
// simrank_pool_trim() — "give back what the library holds at rest" — releases them too (advisor, round 5).
void handback_release_slabs(int device) {
#ifndef SIMRANK_HOST_ONLY
    std::lock_guard<std::mutex> lock(g_slab_mutex);      // (never while a hand-back is using them)
    for (int d = 0; d < 16; ++d) {
        if (device >= 0 && d != (device & 15)) continue;
        Slabs& sl = g_slabs[d];
        for (int i = 0; i < kSlabs; ++i) {
            if (sl.pin[i]) (void)hipHostFree(sl.pin[i]);
            sl.pin[i] = nullptr;
        }
        sl.cap = 0;
    }
#else
    (void)device;
#endif
}
}  // namespace simrank

using namespace simrank;

extern "C" {

int simrank_handback_f64(double* dst, int64_t ld_dst, const float* src, int64_t ld_src, int64_t src_rows_pad, int64_t n,
                         const int32_t* idx, int32_t mode, void* stream) {
    SR_REQUIRE(n >= 0 && ld_dst >= n && (src_rows_pad > 0 ? src_rows_pad >= n : ld_src >= n), "bad shape");
    SR_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (full form) or 1 (symmetric form)");
    if (n == 0) return SIMRANK_OK;
    SR_REQUIRE(dst && src, "NULL pointer");
    const bool timed = std::getenv("SIMRANK_TIME_HANDBACK") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    hipStream_t st = as_stream(stream);
    if (const char* e = std::getenv("SIMRANK_SYM_HANDBACK")) mode = (*e && *e != '0') ? 1 : 0;     // (the benches' A/B switch)
    bool sym = mode == 1 && n > 2 * kDiag && n <= (int64_t(1) << 20);

    int32_t* flag = nullptr;
    float* diag_dev = nullptr;
    auto cleanup_small = [&] { (void)pool_free(flag); (void)pool_free(diag_dev); flag = nullptr; diag_dev = nullptr; };
    double t_checked = 0;
    std::vector<float> diag;
    std::vector<int32_t> pos;                     // pos[i] = source position of caller's node i
    if (sym) {
        // ---- 1. the premise, checked: mirror-equal outside the diagonal blocks?
        const int64_t n_tiles = (n + 31) / 32, n_pairs = n_tiles * (n_tiles - 1) / 2;
        const int rc0 = pool_alloc((void**)&flag, 64);
        if (rc0) return rc0;
        hipError_t e = hipMemsetAsync(flag, 0, 4, st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(mirror_check_kernel, dim3((unsigned)n_pairs), dim3(256), 0, st, (const uint32_t*)src, ld_src,
                               src_rows_pad, n, n_tiles, flag);
            e = hipGetLastError();
        }
        int32_t bad = 1;
        if (e == hipSuccess) e = hipMemcpyAsync(&bad, flag, 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            cleanup_small();
            set_error("simrank_handback_f64 (symmetry check): %s", hipGetErrorString(e));
            return SIMRANK_ERR_HIP;
        }
        if (bad) {
            sym = false;
            if (timed) std::fprintf(stderr, "simrank_handback_f64: not mirror-equal: full form\n");
        }
        t_checked = since();
    }
    if (sym) {
        // ---- 3 (queued first: small). the diagonal blocks of the source order and the order itself
        diag.resize((size_t)n * kDiag);
        int rc = pool_alloc((void**)&diag_dev, size_t(n) * kDiag * sizeof(float));
        if (rc) { cleanup_small(); return rc; }
        hipLaunchKernelGGL(diag_blocks_kernel, dim3((unsigned)((n * kDiag + 255) / 256)), dim3(256), 0, st, src, ld_src,
                           src_rows_pad, n, diag_dev);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(diag.data(), diag_dev, size_t(n) * kDiag * sizeof(float), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && idx) {
            pos.resize((size_t)n);
            e = hipMemcpyAsync(pos.data(), idx, size_t(n) * sizeof(int32_t), hipMemcpyDeviceToHost, st);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            cleanup_small();
            set_error("simrank_handback_f64 (diagonal blocks): %s", hipGetErrorString(e));
            return SIMRANK_ERR_HIP;
        }
    }
    cleanup_small();

    // ---- 2. the bands
    const std::vector<Band> bands = cut_bands(n, sym);
    const int64_t nb = (int64_t)bands.size();
    size_t need = 0;
    for (const Band& b : bands) need = std::max(need, size_t(b.nr) * size_t(b.pitch) * 4);
    int dev = 0;
    SR_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_slab_mutex);      // (one dense hand-back per process at a time: it uses every core)
    Slabs& sl = g_slabs[dev & 15];
    for (int i = 0; i < kSlabs; ++i) {
        if (!sl.done[i]) SR_HIP(hipEventCreateWithFlags(&sl.done[i], hipEventDisableTiming));
        if (!sl.packed[i]) SR_HIP(hipEventCreateWithFlags(&sl.packed[i], hipEventDisableTiming));
    }
    if (!sl.begin) SR_HIP(hipEventCreateWithFlags(&sl.begin, hipEventDisableTiming));
    if (!sl.side) SR_HIP(hipStreamCreateWithFlags(&sl.side, hipStreamNonBlocking));
    SR_HIP(hipEventRecord(sl.begin, st));
    SR_HIP(hipStreamWaitEvent(sl.side, sl.begin, 0));
    if (need > sl.cap) {
        for (int i = 0; i < kSlabs; ++i) {
            if (sl.pin[i]) (void)hipHostFree(sl.pin[i]);
            sl.pin[i] = nullptr;
        }
        sl.cap = 0;
        for (int i = 0; i < kSlabs; ++i) SR_HIP(hipHostMalloc((void**)&sl.pin[i], need, hipHostMallocPortable));
        sl.cap = need;
    }
    // (SIMRANK_HANDBACK_SLABS=2: the two-slab pipeline of the first version, for A/B runs)
    static const int n_slabs = [] { const char* e = std::getenv("SIMRANK_HANDBACK_SLABS"); return e && std::atoi(e) == 2 ? 2 : kSlabs; }();
    float* dev_slab[kSlabs] = {};
    for (int i = 0; i < n_slabs; ++i) {
        const int rc = pool_alloc((void**)&dev_slab[i], need);
        if (rc) {
            for (int k = 0; k < i; ++k) (void)pool_free(dev_slab[k]);
            return rc;
        }
    }
    // the frame is written once, front to back, by many threads: ask for huge pages where the system leaves it to the
    // caller (transparent_hugepage = madvise): 4 k first-touch faults instead of 2 M at N = 32768.  A hint, nothing more.
    if (ld_dst == n && n >= 4096) {
        const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
        const uintptr_t lo = ((uintptr_t)dst + page - 1) / page * page, hi = ((uintptr_t)(dst + n * n)) / page * page;
        if (hi > lo) (void)madvise((void*)lo, hi - lo, MADV_HUGEPAGE);
    }
    // band b: packed in the caller's order on the device (rows r0 .., columns c0 ..), then into its pinned slab
    auto issue = [&](int64_t b) -> int {
        const Band& bd = bands[(size_t)b];
        const int64_t w = n - bd.c0;
        const float* s = src;
        const int32_t* ri = idx ? idx + bd.r0 : nullptr;
        const int32_t* ci = idx ? idx + bd.c0 : nullptr;
        if (!idx)       // identity order: the band starts at (r0, c0) of the source (both multiples of 64, or c0 = 0)
            s = src_rows_pad > 0 ? src + ((bd.c0 >> 5) * src_rows_pad + bd.r0) * 32 : src + bd.r0 * ld_src + bd.c0;
        // (the packing kernel of band b + 2 runs on the side stream while band b + 1 is still on the wire: on one stream
        // every band's copy waited for the next band's kernel — 35 x 0.2 ms at N = 32768)
        const int k = int(b % n_slabs);
        int rc = simrank_permute_layout(s, ld_src, src_rows_pad, dev_slab[k], bd.pitch, 0, bd.nr, w, ri, ci, 4, sl.side);
        if (rc) return rc;
        SR_HIP(hipEventRecord(sl.packed[k], sl.side));
        SR_HIP(hipStreamWaitEvent(st, sl.packed[k], 0));
        SR_HIP(hipMemcpyAsync(sl.pin[k], dev_slab[k], size_t(bd.nr) * size_t(bd.pitch) * 4, hipMemcpyDeviceToHost, st));
        SR_HIP(hipEventRecord(sl.done[k], st));
        return SIMRANK_OK;
    };
    const char* nt_env = std::getenv("SIMRANK_HANDBACK_NT");
    const bool nt_stores = nt_env ? (*nt_env && *nt_env != '0') : true;
    auto widen = [&](int64_t b, int64_t t, int64_t nt) {
        const float* slab = sl.pin[b % n_slabs];
        if (sym) widen_band(slab, bands[(size_t)b], n, dst, ld_dst, t, nt);
        else if (nt_stores) widen_rows<true>(slab, bands[(size_t)b], n, dst, ld_dst, t, nt);
        else widen_rows<false>(slab, bands[(size_t)b], n, dst, ld_dst, t, nt);
    };
    // (16 threads: 93 ms for N = 32768 on the bench box, 32: 111 - 116 — its quota is 16 CPUs — 8: 104; profiles/r05_handback_knobs.log)
    int64_t nt = std::max<int64_t>(1, std::min<int64_t>({16, cpu_share(), (n + 4 * kGroup - 1) / (4 * kGroup),
                                                         std::max<int64_t>(1, (n * n) >> 18)}));
    std::mutex m;
    std::condition_variable cv;
    std::atomic<int64_t> ready{0};           // bands 0 .. ready - 1 are in their pinned slabs
    std::vector<std::atomic<int64_t>> finished((size_t)nb);
    for (auto& f : finished) f.store(0);
    std::atomic<bool> abort{false};
    // (the host is often the slower side: the next band is then there when a thread asks for it — no system call; only
    // the last thread through a band wakes the publisher)
    auto crew = [&](int64_t t) {
        for (int64_t b = 0; b < nb; ++b) {
            if (ready.load(std::memory_order_acquire) <= b && !abort.load()) {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return ready.load(std::memory_order_acquire) > b || abort.load(); });
            }
            if (abort.load()) return;
            widen(b, t, nt);
            if (finished[(size_t)b].fetch_add(1, std::memory_order_acq_rel) + 1 == nt) {
                std::lock_guard<std::mutex> lk(m);
                cv.notify_all();
            }
        }
    };
    std::vector<std::thread> threads;
    if (nt > 1) {
        try {
            for (int64_t t = 0; t < nt; ++t) threads.emplace_back(crew, t);
        } catch (const std::exception&) {
            // (no more threads to be had: the ones that started are still waiting for band 0 — send them home, widen here)
            {
                std::lock_guard<std::mutex> lk(m);
                abort.store(true);
            }
            cv.notify_all();
            for (std::thread& th : threads) th.join();
            threads.clear();
            abort.store(false);
            nt = 1;
        }
    }
    int rc = SIMRANK_OK;
    for (int64_t b = 0; b < std::min<int64_t>(nb, n_slabs) && !rc; ++b) rc = issue(b);
    double t_wait_dev = 0, t_wait_host = 0;
    for (int64_t b = 0; b < nb && !rc; ++b) {
        const double a = since();
        const hipError_t e = hipEventSynchronize(sl.done[b % n_slabs]);
        t_wait_dev += since() - a;
        if (e != hipSuccess) {
            set_error("simrank_handback_f64: %s", hipGetErrorString(e));
            rc = SIMRANK_ERR_HIP;
            break;
        }
        const double c = since();
        if (nt == 1) {
            widen(b, 0, 1);
        } else {
            {
                std::lock_guard<std::mutex> lk(m);
                ready.store(b + 1, std::memory_order_release);
            }
            cv.notify_all();
            // slab b % kSlabs is free for band b + kSlabs once every thread is through with band b
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return finished[(size_t)b].load(std::memory_order_acquire) == nt; });
        }
        t_wait_host += since() - c;
        if (b + n_slabs < nb) rc = issue(b + n_slabs);
    }
    if (nt > 1) {
        {
            std::lock_guard<std::mutex> lk(m);
            if (rc) abort.store(true);
        }
        cv.notify_all();
        for (std::thread& th : threads) th.join();
    }
    (void)hipStreamSynchronize(sl.side);
    (void)hipStreamSynchronize(st);
    for (int i = 0; i < kSlabs; ++i) (void)pool_free(dev_slab[i]);
    const double t_bands = since();
    // ---- 3. the diagonal blocks of the source order, both triangles, where the caller's order puts them
    if (!rc && sym) {
        std::vector<int32_t> node;               // node[a] = caller's node at source position a
        if (idx) {
            node.resize((size_t)n);
            for (int64_t i = 0; i < n; ++i) node[(size_t)pos[(size_t)i]] = (int32_t)i;
        }
        auto patch = [&](int64_t a0, int64_t a1) {
            for (int64_t a = a0; a < a1; ++a) {
                const int64_t base = (a / kDiag) * kDiag, i = idx ? node[(size_t)a] : a;
                double* d = dst + i * ld_dst;
                const float* s = diag.data() + a * kDiag;
                for (int64_t t = 0; t < kDiag && base + t < n; ++t) d[idx ? node[(size_t)(base + t)] : base + t] = (double)s[t];
            }
        };
        const int64_t pt = std::max<int64_t>(1, std::min<int64_t>(nt, n / 1024));
        if (pt == 1) {
            patch(0, n);
        } else {
            // (a thread that cannot be started must not take the process down — no exception may cross the C ABI: the shares
            // that got no thread are patched here; advisor, round 5)
            std::vector<std::thread> ts;
            int64_t started = 0;
            try {
                for (; started < pt; ++started) ts.emplace_back(patch, n * started / pt, n * (started + 1) / pt);
            } catch (...) {
            }
            if (started < pt) patch(n * started / pt, n);
            for (std::thread& th : ts) th.join();
        }
    }
    if (timed)
        std::fprintf(stderr, "simrank_handback_f64: n %lld, %s form, %lld bands, %lld threads: %.1f ms (symmetry check %.1f, bands %.1f — "
                             "waiting for the device %.1f, for the host crew %.1f —, diagonal blocks %.1f)\n", (long long)n,
                     sym ? "symmetric" : "full", (long long)nb, (long long)nt, since(), t_checked, t_bands - t_checked, t_wait_dev,
                     t_wait_host, since() - t_bands);
    return rc;
}

}  // extern "C"
