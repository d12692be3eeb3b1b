// Measurement aid (not part of the library): what does gfx950 return per CU for the access shape of the
// SimRank gather legs — a wave instruction of 64 x 16 B in which each group of 8 lanes reads one whole,
// randomly chosen 128-byte line of a slice that fits an XCD's L2?  Row ids are computed (no id loads), the
// sums are kept alive and written once at the end.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/gather_ceiling.hip -o build/gather_ceiling && build/gather_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int INFLIGHT>
__global__ __launch_bounds__(256) void gather_kernel(const float4* __restrict__ base, int rows_per_slice,
                                                     int iters, float* out) {
    const int lane = threadIdx.x & 63, q = lane & 7;
    const int slice = blockIdx.x & 7;                               // blocks equal mod 8 share an XCD
    const float4* s = base + size_t(slice) * rows_per_slice * 8;   // a row = 8 float4 = 128 B
    unsigned h = (blockIdx.x * 256u + threadIdx.x / 8u) * 2654435761u + 12345u;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < iters; ++it) {
        float4 v[INFLIGHT];
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) {
            h = h * 1664525u + 1013904223u;
            const unsigned row = (h >> 8) % unsigned(rows_per_slice);
            v[j] = s[size_t(row) * 8 + q];
        }
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;
    printf("%s: %d CUs, %.2f GHz\n", prop.name, cus, clk / 1e9);
    float* out;
    CHECK(hipMalloc(&out, 64));
    for (int rows : {256, 8192, 32768, 131072}) {                     // 32 KiB (L1), 1 MiB, 4 MiB (= L2), 16 MiB per XCD
        float4* buf;
        const size_t bytes = size_t(8) * rows * 128;
        CHECK(hipMalloc(&buf, bytes));
        CHECK(hipMemset(buf, 0, bytes));
        for (int wgs_per_cu : {2, 4, 6, 8}) {
            for (int inflight : {4, 8}) {
                const int grid = cus * wgs_per_cu, iters = 2048 / inflight * 4;
                hipEvent_t a, b;
                CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
                auto launch = [&] {
                    if (inflight == 4) hipLaunchKernelGGL(gather_kernel<4>, dim3(grid), dim3(256), 0, 0, buf, rows, iters, out);
                    else hipLaunchKernelGGL(gather_kernel<8>, dim3(grid), dim3(256), 0, 0, buf, rows, iters, out);
                };
                launch();
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(a));
                for (int r = 0; r < 3; ++r) launch();
                CHECK(hipEventRecord(b));
                CHECK(hipEventSynchronize(b));
                float ms;
                CHECK(hipEventElapsedTime(&ms, a, b));
                ms /= 3;
                const double total = double(grid) * 256 * 16.0 * iters * inflight;
                const double tbs = total / (ms * 1e-3) / 1e12;
                printf("slice %6.2f MiB/XCD  waves/SIMD %d  in flight %d: %6.2f TB/s = %5.1f GB/s per CU = %4.1f B/clk/CU\n",
                       rows * 128.0 / (1 << 20), wgs_per_cu, inflight, tbs, tbs * 1e3 / cus, tbs * 1e12 / cus / clk);
            }
        }
        CHECK(hipFree(buf));
    }
    return 0;
}
