// How fast do HOST threads read the pinned slabs a device-to-host hand-back lands in?  (round 5: the widening crew of
// simrank_download_f64(_sym) ran at 0.4 - 1.5 GB/s per thread.)  Buffers of 256 MiB: malloc, hipHostMalloc with the default /
// portable / non-coherent / write-combined flags; each filled by a device-to-host copy, then summed by 1, 8, 32, 64
// threads with plain loads (and, for the slow ones, with streaming loads: movntdqa).
//   hipcc -O3 -std=c++17 tools/micro/pinned_read.hip -o /tmp/pinned_read -lpthread && /tmp/pinned_read
#include <hip/hip_runtime.h>
#include <immintrin.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static float sum_plain(const float* p, size_t n) {
    float a = 0, b = 0, c = 0, d = 0;
    for (size_t i = 0; i + 3 < n; i += 4) { a += p[i]; b += p[i + 1]; c += p[i + 2]; d += p[i + 3]; }
    return a + b + c + d;
}

__attribute__((target("sse4.1"))) static float sum_stream(const float* p, size_t n) {
    __m128 acc = _mm_setzero_ps();
    for (size_t i = 0; i + 3 < n; i += 4) acc = _mm_add_ps(acc, _mm_castsi128_ps(_mm_stream_load_si128((__m128i*)(p + i))));
    float t[4];
    _mm_storeu_ps(t, acc);
    return t[0] + t[1] + t[2] + t[3];
}

static void bench(const char* name, float* host, size_t n, const float* dev, bool stream) {
    hipMemcpy(host, dev, n * 4, hipMemcpyDeviceToHost);
    for (int nt : {1, 8, 32, 64}) {
        std::vector<float> out((size_t)nt);
        double best = 1e9;
        for (int rep = 0; rep < 2; ++rep) {
            hipMemcpy(host, dev, n * 4, hipMemcpyDeviceToHost);        // (fresh from the device: nothing of it in a CPU cache)
            const double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t)
                th.emplace_back([&, t] {
                    const size_t lo = n * (size_t)t / (size_t)nt / 16 * 16, hi = n * (size_t)(t + 1) / (size_t)nt / 16 * 16;
                    out[(size_t)t] = stream ? sum_stream(host + lo, hi - lo) : sum_plain(host + lo, hi - lo);
                });
            for (auto& x : th) x.join();
            best = std::min(best, now() - t0);
        }
        std::printf("%-34s %2d threads  %7.1f GB/s  (%.2f per thread)%s\n", name, nt, n * 4 / best / 1e9, n * 4 / best / 1e9 / nt,
                    stream ? "  [movntdqa]" : "");
    }
}

int main() {
    const size_t n = (size_t)64 << 20;          // floats: 256 MiB
    float* dev = nullptr;
    hipMalloc(&dev, n * 4);
    hipMemset(dev, 0, n * 4);
    float* m = (float*)aligned_alloc(4096, n * 4);
    bench("malloc (pageable)", m, n, dev, false);
    struct { const char* name; unsigned flags; } kinds[] = {
        {"hipHostMalloc default", hipHostMallocDefault},
        {"hipHostMalloc portable", hipHostMallocPortable},
        {"hipHostMalloc non-coherent", hipHostMallocNonCoherent},
        {"hipHostMalloc coherent", hipHostMallocCoherent},
        {"hipHostMalloc write-combined", hipHostMallocWriteCombined},
        {"hipHostMalloc numa-user", hipHostMallocNumaUser},
    };
    for (auto& k : kinds) {
        float* p = nullptr;
        if (hipHostMalloc((void**)&p, n * 4, k.flags) != hipSuccess) {
            std::printf("%-34s refused\n", k.name);
            (void)hipGetLastError();
            continue;
        }
        bench(k.name, p, n, dev, false);
        bench(k.name, p, n, dev, true);
        hipHostFree(p);
    }
    // registered pageable memory
    if (hipHostRegister(m, n * 4, hipHostRegisterDefault) == hipSuccess) {
        bench("malloc + hipHostRegister", m, n, dev, false);
        hipHostUnregister(m);
    }
    // device-to-host rates into each kind (one 256 MiB copy)
    for (auto& k : kinds) {
        float* p = nullptr;
        if (hipHostMalloc((void**)&p, n * 4, k.flags) != hipSuccess) { (void)hipGetLastError(); continue; }
        hipMemcpy(p, dev, n * 4, hipMemcpyDeviceToHost);
        const double t0 = now();
        for (int i = 0; i < 4; ++i) hipMemcpy(p, dev, n * 4, hipMemcpyDeviceToHost);
        std::printf("D2H into %-28s %6.1f GB/s\n", k.name, 4.0 * n * 4 / (now() - t0) / 1e9);
        hipHostFree(p);
    }
    return 0;
}
