// Host half of the symmetric hand-back (csrc/handback.hip: widen_band) by itself, on the GPU box's CPUs: what do the
// threads reach when the slab is already in host memory?  Variants of the access pattern, thread counts, store kinds.
//   /opt/rocm/lib/llvm/bin/clang++ -O3 -std=c++17 -pthread tools/micro/widen_bench.cpp -o build/widen_bench && build/widen_bench [n = 32768]
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Band { int64_t r0, nr, pitch; };
static int64_t odd_lines(int64_t w) { int64_t l = (w + 15) / 16; if (!(l & 1)) ++l; return l * 16; }
static std::vector<Band> cut_bands(int64_t n, int64_t slab_bytes) {
    std::vector<Band> b;
    for (int64_t r0 = 0; r0 < n;) {
        const int64_t pitch = odd_lines(n - r0);
        int64_t nr = std::max<int64_t>(64, (slab_bytes / 4 / pitch) / 64 * 64);
        nr = std::min(nr, n - r0);
        b.push_back({r0, nr, pitch});
        r0 += nr;
    }
    return b;
}

struct Variant {
    const char* name;
    int64_t group;      // result rows a thread owns together
    int64_t chunk;      // band rows per transposed tile (run length of the mirrored writes, in doubles)
    bool nt;            // non-temporal stores
    bool rotate;        // threads start their chunks at different columns
    bool direct_rr;     // direct rows dealt one by one (row % nt) instead of with their group
    int mode = 0;       // 0: as above; 1: + software prefetch of the next tile's lines; 2: big tiles through a local buffer
};

template <bool NT>
static inline void st(double* p, double v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <bool NT>
static void widen(const float* slab, const Band& b, int64_t n, double* dst, int64_t ld, int64_t t, int64_t nt, const Variant& v) {
    const int64_t w = n - b.r0, G = v.group, C = v.chunk;
    const int64_t g0 = b.r0 / G, g_end = (n + G - 1) / G;
    for (int64_t g = g0 + ((t - g0 % nt) % nt + nt) % nt; g < g_end; g += nt) {
        const int64_t j0 = g * G - b.r0, j1 = std::min(n, (g + 1) * G) - b.r0;
        const int64_t imax = std::min(b.nr, j1 - 1);
        const int64_t chunks = (imax + C - 1) / C;
        const int64_t first = v.rotate && chunks > 0 ? (t * 3 + g) % chunks : 0;
        for (int64_t k = 0; k < chunks; ++k) {
            const int64_t ic = ((first + k) % chunks) * C, ie = std::min(imax, ic + C);
            for (int64_t j = j0; j < j1; ++j) {
                double* d = dst + (b.r0 + j) * ld + b.r0;
                const float* s = slab + j;
                const int64_t iend = std::min(ie, j);
                for (int64_t i = ic; i < iend; ++i) st<NT>(d + i, (double)s[i * b.pitch]);
            }
        }
        if (!v.direct_rr)
            for (int64_t j = j0; j < std::min(j1, b.nr); ++j) {
                const float* s = slab + j * b.pitch;
                double* d = dst + (b.r0 + j) * ld + b.r0;
                for (int64_t c = j; c < w; ++c) st<NT>(d + c, (double)s[c]);
            }
    }
    if (v.direct_rr)
        for (int64_t j = t; j < b.nr; j += nt) {
            const float* s = slab + j * b.pitch;
            double* d = dst + (b.r0 + j) * ld + b.r0;
            for (int64_t c = j; c < w; ++c) st<NT>(d + c, (double)s[c]);
        }
}

template <bool NT>
static void widen_pf(const float* slab, const Band& b, int64_t n, double* dst, int64_t ld, int64_t t, int64_t nt, const Variant& v) {
    const int64_t w = n - b.r0, G = v.group, C = v.chunk;
    const int64_t g0 = b.r0 / G, g_end = (n + G - 1) / G;
    for (int64_t g = g0 + ((t - g0 % nt) % nt + nt) % nt; g < g_end; g += nt) {
        const int64_t j0 = g * G - b.r0, j1 = std::min(n, (g + 1) * G) - b.r0;
        const int64_t imax = std::min(b.nr, j1 - 1);
        const int64_t chunks = (imax + C - 1) / C;
        for (int64_t k = 0; k < chunks; ++k) {
            const int64_t ic = k * C, ie = std::min(imax, ic + C);
            // the lines of the next tile (or of the next group's first tile)
            const int64_t pc = k + 1 < chunks ? ic + C : 0, pj = k + 1 < chunks ? j0 : j0 + nt * G;
            if (pj < w)
                for (int64_t i = pc; i < std::min(b.nr, pc + C); ++i) __builtin_prefetch(slab + i * b.pitch + pj, 0, 0);
            for (int64_t j = j0; j < j1; ++j) {
                double* d = dst + (b.r0 + j) * ld + b.r0;
                const float* s = slab + j;
                const int64_t iend = std::min(ie, j);
                for (int64_t i = ic; i < iend; ++i) st<NT>(d + i, (double)s[i * b.pitch]);
            }
        }
    }
    for (int64_t j = t; j < b.nr; j += nt) {
        const float* s = slab + j * b.pitch;
        double* d = dst + (b.r0 + j) * ld + b.r0;
        for (int64_t c = j; c < w; ++c) st<NT>(d + c, (double)s[c]);
    }
}

// big tiles: a thread owns groups of G rows (G = 256 .. 1024); per 64 band rows it copies the 64 pieces of G floats
// (sequential 1 - 4 KiB reads) into a local buffer and transposes out of that
template <bool NT>
static void widen_big(const float* slab, const Band& b, int64_t n, double* dst, int64_t ld, int64_t t, int64_t nt, const Variant& v) {
    const int64_t w = n - b.r0, G = v.group, C = 64;
    std::vector<float> T((size_t)(C * G));
    const int64_t g0 = b.r0 / G, g_end = (n + G - 1) / G;
    for (int64_t g = g0 + ((t - g0 % nt) % nt + nt) % nt; g < g_end; g += nt) {
        const int64_t j0 = std::max<int64_t>(0, g * G - b.r0), j1 = std::min(n, (g + 1) * G) - b.r0;
        const int64_t imax = std::min(b.nr, j1 - 1);
        for (int64_t ic = 0; ic < imax; ic += C) {
            const int64_t ie = std::min(imax, ic + C);
            for (int64_t i = ic; i < ie; ++i) std::memcpy(T.data() + (i - ic) * G, slab + i * b.pitch + j0, (size_t)(j1 - j0) * 4);
            for (int64_t j = j0; j < j1; ++j) {
                double* d = dst + (b.r0 + j) * ld + b.r0;
                const float* s = T.data() + (j - j0);
                const int64_t iend = std::min(ie, j);
                for (int64_t i = ic; i < iend; ++i) st<NT>(d + i, (double)s[(i - ic) * G]);
            }
        }
    }
    for (int64_t j = t; j < b.nr; j += nt) {
        const float* s = slab + j * b.pitch;
        double* d = dst + (b.r0 + j) * ld + b.r0;
        for (int64_t c = j; c < w; ++c) st<NT>(d + c, (double)s[c]);
    }
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 32768;
    const int64_t slab_bytes = int64_t(64) << 20;
    double* dst = (double*)mmap(nullptr, (size_t)n * n * 8, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    madvise(dst, (size_t)n * n * 8, MADV_HUGEPAGE);
    const auto bands_for_size = cut_bands(n, slab_bytes);
    std::vector<size_t> band_off;
    size_t total = 0;
    for (const Band& b : bands_for_size) { band_off.push_back(total); total += (size_t)b.nr * (size_t)b.pitch; }
    std::vector<float> slab_all(total + 4096, 0.25f);                  // the whole upper triangle: every band reads DRAM
    struct { std::vector<float>& all; std::vector<size_t>& off; const std::vector<Band>* bands;
             const float* data_of(const Band& b) const { return all.data() + off[(size_t)(&b - bands->data())]; } } slabs{slab_all, band_off, nullptr};
    std::vector<float>& slab = slab_all;
    const Variant variants[] = {
        {"group 16, chunk 64, nt, rotated, direct rr", 16, 64, true, true, true, 0},
        {"group 16, chunk 64, nt, direct rr, prefetch next tile", 16, 64, true, false, true, 1},
        {"group 32, chunk 64, nt, direct rr, prefetch next tile", 32, 64, true, false, true, 1},
        {"big tiles: group 256 through a local buffer", 256, 64, true, false, true, 2},
        {"big tiles: group 1024 through a local buffer", 1024, 64, true, false, true, 2},
    };
    const auto bands = bands_for_size;
    slabs.bands = &bands;
    std::printf("n %lld, %zu bands, hardware threads %u\n", (long long)n, bands.size(), std::thread::hardware_concurrency());
    // first touch of the whole frame by one pattern (not timed separately below)
    for (const Variant& v : variants) {
        for (int64_t nt : {8, 16, 32}) {
            double best = 1e9;
            for (int rep = 0; rep < 2; ++rep) {
                const double t0 = now();
                for (const Band& b : bands) {
                    std::vector<std::thread> th;
                    for (int64_t t = 0; t < nt; ++t)
                        th.emplace_back([&, t] {
                            const float* sb = slabs.data_of(b);
                            if (v.mode == 1) widen_pf<true>(sb, b, n, dst, n, t, nt, v);
                            else if (v.mode == 2) widen_big<true>(sb, b, n, dst, n, t, nt, v);
                            else if (v.nt) widen<true>(sb, b, n, dst, n, t, nt, v);
                            else widen<false>(sb, b, n, dst, n, t, nt, v);
                        });
                    for (auto& x : th) x.join();
                }
                best = std::min(best, now() - t0);
            }
            std::printf("%-52s %3lld threads  %7.1f ms  %6.1f GB/s written\n", v.name, (long long)nt, best * 1e3, n * n * 8.0 / best / 1e9);
            std::fflush(stdout);
        }
    }
    // the same work by a PERSISTENT crew (what handback.hip runs): band b is published, every thread does its share,
    // the publisher waits for all of them — with condition variables (a thundering herd per band) or spinning on atomics
    {
        const Variant v = {"group 16, chunk 64, nt, rotated, direct rr", 16, 64, true, true, true};
        for (int spin = 0; spin < 2; ++spin)
            for (int64_t nt : {8, 16, 32}) {
                double best = 1e9;
                for (int rep = 0; rep < 2; ++rep) {
                    const int64_t nb = (int64_t)bands.size();
                    std::mutex m;
                    std::condition_variable cv;
                    int64_t ready = 0;
                    std::vector<int64_t> finished((size_t)nb, 0);
                    std::atomic<int64_t> a_ready{0};
                    std::vector<std::atomic<int64_t>> a_fin((size_t)nb);
                    for (auto& x : a_fin) x = 0;
                    const double t0 = now();
                    std::vector<std::thread> th;
                    for (int64_t t = 0; t < nt; ++t)
                        th.emplace_back([&, t] {
                            for (int64_t b = 0; b < nb; ++b) {
                                if (spin) {
                                    while (a_ready.load(std::memory_order_acquire) <= b) std::this_thread::yield();
                                } else {
                                    std::unique_lock<std::mutex> lk(m);
                                    cv.wait(lk, [&] { return ready > b; });
                                }
                                widen<true>(slabs.data_of(bands[(size_t)b]), bands[(size_t)b], n, dst, n, t, nt, v);
                                if (spin) {
                                    a_fin[(size_t)b].fetch_add(1, std::memory_order_release);
                                } else {
                                    { std::lock_guard<std::mutex> lk(m); ++finished[(size_t)b]; }
                                    cv.notify_all();
                                }
                            }
                        });
                    for (int64_t b = 0; b < nb; ++b) {
                        if (spin) {
                            a_ready.store(b + 1, std::memory_order_release);
                            while (a_fin[(size_t)b].load(std::memory_order_acquire) < nt) std::this_thread::yield();
                        } else {
                            { std::lock_guard<std::mutex> lk(m); ready = b + 1; }
                            cv.notify_all();
                            std::unique_lock<std::mutex> lk(m);
                            cv.wait(lk, [&] { return finished[(size_t)b] == nt; });
                        }
                    }
                    for (auto& x : th) x.join();
                    best = std::min(best, now() - t0);
                }
                std::printf("%-52s %3lld threads  %7.1f ms  %6.1f GB/s written\n", spin ? "persistent crew, spinning on atomics" : "persistent crew, condition variables",
                            (long long)nt, best * 1e3, n * n * 8.0 / best / 1e9);
                std::fflush(stdout);
            }
    }
    // the full hand-back's pattern for comparison: every thread widens whole rows of a 128 MiB slab
    for (int64_t nt : {8, 16, 32}) {
        double best = 1e9;
        std::vector<float> big((size_t)(128 << 20) / 4, 0.25f);
        const int64_t slab_rows = (int64_t(128) << 20) / (n * 4);
        for (int rep = 0; rep < 2; ++rep) {
            const double t0 = now();
            for (int64_t r0 = 0; r0 < n; r0 += slab_rows) {
                const int64_t nr = std::min(slab_rows, n - r0);
                std::vector<std::thread> th;
                for (int64_t t = 0; t < nt; ++t)
                    th.emplace_back([&, t] {
                        for (int64_t r = nr * t / nt; r < nr * (t + 1) / nt; ++r) {
                            const float* s = big.data() + r * n;
                            double* d = dst + (r0 + r) * n;
                            for (int64_t c = 0; c < n; ++c) d[c] = (double)s[c];
                        }
                    });
                for (auto& x : th) x.join();
            }
            best = std::min(best, now() - t0);
        }
        std::printf("%-52s %3lld threads  %7.1f ms  %6.1f GB/s written\n", "full hand-back: whole rows, plain stores", (long long)nt, best * 1e3,
                    n * n * 8.0 / best / 1e9);
    }
    return 0;
}
