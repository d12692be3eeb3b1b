// Hand-back of a SYMMETRIC similarity matrix (SimRank.py:141, :303: `pd.DataFrame(new_S, ...)` — the float64 N x N
// frame the reference returns): only the elements on or above the diagonal cross PCIe.
//
// A symmetric update leaves a BITWISE symmetric matrix behind (leg 2 computes the 32 x 32 tiles on or above the
// diagonal and stores each strictly-upper tile a second time, transposed: spmm.hip kSym, half.hip SYM), and a renaming
// of the nodes applied to rows and columns alike keeps it so.  The full hand-back moved both triangles: N^2 floats over
// PCIe, a third of a config-4 fit (0.089 of 0.265 s).  Here the result is cut into row bands; band b
// (rows r0 .. r1) is brought into the caller's order and packed on the device as the trapezoid [r0, r1) x [r0, N) —
// one simrank_permute_layout per band —, copied into a pinned slab, and a crew of host threads widens it to float64
// twice: dst[r][c] and, for c > r, dst[c][r].  Two device and two pinned slabs: band b + 1 travels while band b is
// widened.  The threads split a band by COLUMN blocks, so the direct writes (rows of the band, their columns) and the
// mirrored writes (their columns as rows) of two threads never meet; stores are non-temporal (the frame is written once
// and is far larger than any cache).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"

namespace simrank {
namespace {

constexpr int64_t kSlabBytes = int64_t(64) << 20;     // per pinned / device slab
constexpr int64_t kTile = 64;                         // host threads work in 64 x 64 blocks

struct Band {
    int64_t r0 = 0, nr = 0;          // rows [r0, r0 + nr) of the result; columns [r0, n): width n - r0
};

// rows of a band: as many as fit a slab, a multiple of kTile (the last band takes what is left)
std::vector<Band> cut_bands(int64_t n) {
    std::vector<Band> bands;
    for (int64_t r0 = 0; r0 < n;) {
        const int64_t w = n - r0;
        int64_t nr = std::max<int64_t>(kTile, (kSlabBytes / 4 / w) / kTile * kTile);
        // (a band may not be wider than tall beyond the slab: with nr rows it holds nr * w floats)
        nr = std::min(nr, n - r0);
        bands.push_back({r0, nr});
        r0 += nr;
    }
    return bands;
}

struct Slabs {                       // kept per device for the life of the process (pinning memory is slow)
    float* pin[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    size_t cap = 0;
};
std::mutex g_slab_mutex;
Slabs g_slabs[16];

inline void store_nt(double* p, double v) { __builtin_nontemporal_store(v, p); }

// one thread's share of a band: column blocks jb = t, t + nt, ...
void widen_band(const float* slab, const Band& b, int64_t n, double* dst, int64_t ld, int64_t t, int64_t nt) {
    const int64_t w = n - b.r0;
    const int64_t col_blocks = (w + kTile - 1) / kTile, row_blocks = (b.nr + kTile - 1) / kTile;
    for (int64_t jb = t; jb < col_blocks; jb += nt) {
        const int64_t j0 = jb * kTile, j1 = std::min(w, j0 + kTile);
        for (int64_t ib = 0; ib < row_blocks && ib <= jb; ++ib) {       // (blocks left of the diagonal: nothing to do)
            const int64_t i0 = ib * kTile, i1 = std::min(b.nr, i0 + kTile);
            const bool diag = ib == jb;
            // direct: rows of the band, this thread's columns
            for (int64_t i = i0; i < i1; ++i) {
                const float* s = slab + i * w;
                double* d = dst + (b.r0 + i) * ld + b.r0;
                for (int64_t j = diag ? std::max(j0, i) : j0; j < j1; ++j) store_nt(d + j, (double)s[j]);
            }
            // mirrored: this thread's columns as rows (strictly below the diagonal of the result)
            for (int64_t j = j0; j < j1; ++j) {
                double* d = dst + (b.r0 + j) * ld + b.r0;
                const float* s = slab + j;
                for (int64_t i = i0; i < (diag ? std::min(i1, j) : i1); ++i) store_nt(d + i, (double)s[i * w]);
            }
        }
    }
}

}  // namespace
}  // namespace simrank

using namespace simrank;

extern "C" {

int simrank_download_f64_sym(double* dst, int64_t ld_dst, const float* src, int64_t ld_src, int64_t src_rows_pad, int64_t n,
                             const int32_t* idx, void* stream) {
    SR_REQUIRE(n >= 0 && ld_dst >= n && (src_rows_pad > 0 ? src_rows_pad >= n : ld_src >= n), "bad shape");
    if (n == 0) return SIMRANK_OK;
    SR_REQUIRE(dst && src, "NULL pointer");
    const bool timed = std::getenv("SIMRANK_TIME_HANDBACK") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    const std::vector<Band> bands = cut_bands(n);
    const int64_t nb = (int64_t)bands.size();
    size_t need = 0;
    for (const Band& b : bands) need = std::max(need, size_t(b.nr) * size_t(n - b.r0) * 4);
    int dev = 0;
    SR_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_slab_mutex);      // (one symmetric hand-back per process at a time: it uses every core)
    Slabs& sl = g_slabs[dev & 15];
    for (int i = 0; i < 2; ++i)
        if (!sl.done[i]) SR_HIP(hipEventCreateWithFlags(&sl.done[i], hipEventDisableTiming));
    if (need > sl.cap) {
        for (int i = 0; i < 2; ++i) {
            if (sl.pin[i]) (void)hipHostFree(sl.pin[i]);
            sl.pin[i] = nullptr;
        }
        sl.cap = 0;
        for (int i = 0; i < 2; ++i) SR_HIP(hipHostMalloc((void**)&sl.pin[i], need, hipHostMallocPortable));
        sl.cap = need;
    }
    float* dev_slab[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; ++i) {
        const int rc = pool_alloc((void**)&dev_slab[i], need);
        if (rc) {
            (void)pool_free(dev_slab[0]);
            return rc;
        }
    }
    hipStream_t st = as_stream(stream);
    // band b: packed trapezoid in the caller's order on the device, then into its pinned slab
    auto issue = [&](int64_t b) -> int {
        const Band& bd = bands[(size_t)b];
        const int64_t w = n - bd.r0;
        const float* s = src;
        const int32_t* ri = idx ? idx + bd.r0 : nullptr;
        if (!idx) {
            // identity order: the band starts at (r0, r0) of the source.  Panel-blocked: r0 is a multiple of 32 (kTile)
            s = src_rows_pad > 0 ? src + ((bd.r0 >> 5) * src_rows_pad + bd.r0) * 32 : src + bd.r0 * ld_src + bd.r0;
        }
        int rc = simrank_permute_layout(s, ld_src, src_rows_pad, dev_slab[b & 1], w, 0, bd.nr, w, ri, ri, 4, st);
        if (rc) return rc;
        SR_HIP(hipMemcpyAsync(sl.pin[b & 1], dev_slab[b & 1], size_t(bd.nr) * size_t(w) * 4, hipMemcpyDeviceToHost, st));
        SR_HIP(hipEventRecord(sl.done[b & 1], st));
        return SIMRANK_OK;
    };
    // the crew: every thread takes its column blocks of every band, in band order
    const int64_t nt = std::max<int64_t>(1, std::min<int64_t>({32, (int64_t)std::thread::hardware_concurrency(), (n + kTile - 1) / kTile,
                                                               std::max<int64_t>(1, (n * n) >> 18)}));
    std::mutex m;
    std::condition_variable cv;
    int64_t ready = 0;                       // bands 0 .. ready - 1 are in their pinned slabs
    std::vector<int64_t> finished((size_t)nb, 0);
    bool abort = false;
    auto crew = [&](int64_t t) {
        for (int64_t b = 0; b < nb; ++b) {
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return ready > b || abort; });
                if (abort) return;
            }
            widen_band(sl.pin[b & 1], bands[(size_t)b], n, dst, ld_dst, t, nt);
            {
                std::lock_guard<std::mutex> lk(m);
                ++finished[(size_t)b];
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> threads;
    if (nt > 1)
        for (int64_t t = 0; t < nt; ++t) threads.emplace_back(crew, t);
    int rc = issue(0);
    if (!rc && nb > 1) rc = issue(1);
    double t_wait_dev = 0, t_wait_host = 0;
    for (int64_t b = 0; b < nb && !rc; ++b) {
        const double a = since();
        const hipError_t e = hipEventSynchronize(sl.done[b & 1]);
        t_wait_dev += since() - a;
        if (e != hipSuccess) {
            set_error("simrank_download_f64_sym: %s", hipGetErrorString(e));
            rc = SIMRANK_ERR_HIP;
            break;
        }
        const double c = since();
        if (nt == 1) {
            widen_band(sl.pin[b & 1], bands[(size_t)b], n, dst, ld_dst, 0, 1);
        } else {
            {
                std::lock_guard<std::mutex> lk(m);
                ready = b + 1;
            }
            cv.notify_all();
            // slab b & 1 is free for band b + 2 once every thread is through with band b
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return finished[(size_t)b] == nt; });
        }
        t_wait_host += since() - c;
        if (b + 2 < nb) rc = issue(b + 2);
    }
    if (nt > 1) {
        {
            std::lock_guard<std::mutex> lk(m);
            if (rc) abort = true;
        }
        cv.notify_all();
        for (std::thread& th : threads) th.join();
    }
    (void)hipStreamSynchronize(st);
    (void)pool_free(dev_slab[0]);
    (void)pool_free(dev_slab[1]);
    if (timed)
        std::fprintf(stderr, "simrank_download_f64_sym: n %lld, %lld bands, %lld threads: %.1f ms (waiting for the device %.1f, "
                             "for the host crew %.1f)\n", (long long)n, (long long)nb, (long long)nt, since(), t_wait_dev, t_wait_host);
    return rc;
}

}  // extern "C"
